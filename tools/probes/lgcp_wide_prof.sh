# per-launch durations of the wide-batch lgcp forward by launch type (A / B / C by position in the repeating sequence):
#   bash tools/probes/lgcp_wide_prof.sh <n> [tag]
N=${1:-600}
T=${2:-lgcp_wide_prof}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/probes/lgcp_time.py $N > $O/time_under_rocprof_n$N.txt 2>&1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_n$N.csv
python3 - $O $N <<'PY'
import csv, glob, sys, collections
O, N = sys.argv[1], sys.argv[2]
out = open('%s/per_launch_n%s.txt' % (O, N), 'w')
for f in glob.glob(O + '/prof/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    seq = [r for r in rows if 'lgcp_wide_gemm' in r['Kernel_Name']]
    acc = collections.defaultdict(list)
    gaps = []
    for i, r in enumerate(seq):
        acc["ABC"[i % 3] + ' grid ' + r.get('Grid_Size_X', r.get('Grid_Size', '?'))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        if i:
            gaps.append(int(r['Start_Timestamp']) - int(seq[i - 1]['End_Timestamp']))
    for k, v in sorted(acc.items()):
        v.sort()
        print(k, 'launches', len(v), 'median %.2f us' % (v[len(v) // 2] / 1e3), 'p10 %.2f' % (v[len(v) // 10] / 1e3), 'p90 %.2f' % (v[9 * len(v) // 10] / 1e3), file=out)
    gaps.sort()
    print('gap between consecutive GEMM launches: median %.2f us p90 %.2f' % (gaps[len(gaps) // 2] / 1e3, gaps[9 * len(gaps) // 10] / 1e3), file=out)
    other = collections.defaultdict(list)
    for r in rows:
        if 'lgcp_wide_gemm' not in r['Kernel_Name']:
            other[r['Kernel_Name'][:50]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in sorted(other.items(), key=lambda kv: -sum(kv[1]))[:8]:
        print('%-50s launches %5d total %.3f ms' % (k, len(v), sum(v) / 1e6), file=out)
out.close()
print(open('%s/per_launch_n%s.txt' % (O, N)).read())
PY
rm -rf $O/prof
cat $O/time_under_rocprof_n$N.txt
