# durations and gaps of the headline's launch sequence: bash tools/probes/headline_gaps.sh [tag]
T=${1:-headline_gaps}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
for mode in default prepared; do
rm -rf $O/prof
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 tools/probes/headline_gaps.py $mode > /dev/null 2>&1
python3 - $O $mode <<'PY'
import csv, glob, sys, collections
O, mode = sys.argv[1], sys.argv[2]
for f in glob.glob(O + '/prof/**/*kernel_trace.csv', recursive=True):
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
    rows = rows[-1200:] if mode == 'default' else rows[-800:]      # the last 400 calls
    def short(n): return 'prep' if 'prep_fused' in n else ('coop' if 'coop_kernel' in n else ('finalize' if 'finalize' in n else n[:20]))
    dur = collections.defaultdict(list); gap = collections.defaultdict(list)
    for i, r in enumerate(rows):
        k = short(r['Kernel_Name']); dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
        if i:
            p = rows[i - 1]
            gap[short(p['Kernel_Name']) + ' -> ' + k].append(int(r['Start_Timestamp']) - int(p['End_Timestamp']))
    med = lambda v: sorted(v)[len(v) // 2] / 1e3
    print('== %s path' % mode)
    for k, v in dur.items(): print('  %-10s duration median %.2f us (%d)' % (k, med(v), len(v)))
    for k, v in gap.items(): print('  gap %-22s median %.2f us  p10 %.2f  p90 %.2f' % (k, med(v), sorted(v)[len(v)//10]/1e3, sorted(v)[9*len(v)//10]/1e3))
    span = (int(rows[-1]['End_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3 / 400
    print('  per call (span of the last 400 calls / 400): %.2f us' % span)
PY
done
rm -rf $O/prof
