# Round-end artefacts of the north-star bench on the GPU box: bash tools/probes/round_end.sh <tag>
# -> gpurun_out/<tag>/{bench.json, kernel_stats.csv, pmc.json, all_configs.jsonl}
T=${1:-r01_o}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$T
python3 bench.py > gpurun_out/$T/bench.json 2> gpurun_out/$T/bench.err || exit 1
rm -rf gpurun_out/$T/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$T/prof -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --saturated 0 --forward-only > gpurun_out/$T/prof_bench.json 2> /dev/null || exit 1
cp $(find gpurun_out/$T/prof -name "*kernel_stats.csv" | head -1) gpurun_out/$T/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/$T/pmc/$c -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --saturated 0 --forward-only > /dev/null 2>&1 || exit 1
done
python3 - $T <<'PY'
import glob, csv, collections, json, sys
T = sys.argv[1]
out = {}
for f in sorted(glob.glob('gpurun_out/%s/pmc/*/*/*counter_collection.csv' % T)):
    rows = [r for r in csv.DictReader(open(f)) if 'coop_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    rows = rows[:7]   # the 2 warm-up + 5 timed forward launches; later coop launches keep the trajectory (training step)
    out[rows[0]['Counter_Name']] = sum(float(r['Counter_Value']) for r in rows) / len(rows)
    out['kernel'] = rows[0]['Kernel_Name'][:80]
out['hbm_bytes_per_launch'] = (2 * out['FETCH_SIZE'] + out['WRITE_SIZE']) * 1024
json.dump(out, open('gpurun_out/%s/pmc.json' % T, 'w'), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/$T/prof gpurun_out/$T/pmc
for c in gmm_n300_k8 funnel_n300_k64 many_gmm_n2000_k256_dds many_gmm_var_n16000_k256 lgcp_n20_k128; do
  python3 bench.py --config $c --no-cpu-baseline --saturated 0 2>/dev/null | tail -1
done > gpurun_out/$T/all_configs.jsonl
head -c 1200 gpurun_out/$T/bench.json; echo; head -8 gpurun_out/$T/kernel_stats.csv
