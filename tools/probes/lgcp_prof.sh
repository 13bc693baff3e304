# per-kernel durations of the lgcp forward / gradient: bash tools/probes/lgcp_prof.sh [lgcp_time.py|lgcp_grad_time.py]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
rm -rf gpurun_out/lgcp_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lgcp_prof -- python3 tools/probes/${1:-lgcp_time.py} 2>&1 | tail -3
python3 - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/lgcp_prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:16]:
        print("%-80s calls %6s avg %8.2f us total %8.2f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/lgcp_prof/**/*kernel_trace.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'lgcp' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:60], r.get('Grid_Size_X', r.get('Grid_Size', '?')), r.get('Grid_Size_Y', ''))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, v in sorted(acc.items()):
        v.sort()
        print(k, 'n', len(v), 'median %.2f us' % (v[len(v) // 2] / 1e3), 'p10 %.2f' % (v[len(v) // 10] / 1e3), 'p90 %.2f' % (v[9 * len(v) // 10] / 1e3))
PY
find gpurun_out/lgcp_prof -name "*trace*.csv" -delete
