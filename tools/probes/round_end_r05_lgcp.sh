# The part of round_end_r05.sh that depends on the lgcp sources (cmcd_lgcp.hip, cmcd_lgcp_wide.hip): bash tools/probes/round_end_r05_lgcp.sh <tag>
T=${1:-r05_z}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
rm -rf $O/pmc_lgcp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_lgcp/$c -- python3 bench.py --config lgcp_n20_k128 --steps 3 --warmup 1 --spinup 0 --no-cpu-baseline --saturated 0 --no-legs > /dev/null 2>&1 || echo "lgcp pass failed: $c"
done
python3 - $T <<'PY'
import glob, csv, collections, json, sys, os
sys.path.insert(0, os.getcwd())
import bench
T = sys.argv[1]
acc = collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/%s/pmc_lgcp/*/*/*counter_collection.csv' % T)):
    for r in csv.DictReader(open(f)):
        if 'lgcp_nsk_kernel' in r['Kernel_Name'] or 'lgcp_gemm_kernel' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:52], r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Counter_Name'])].append(float(r['Counter_Value']))
out = {}
for (k, g, c), v in sorted(acc.items()):
    out.setdefault(k + ' grid ' + str(g), {})[c] = sum(v) / len(v)
for k, d in out.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        d['hbm_bytes_per_launch'] = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
out['kernel_sources_sha'] = bench.kernel_sources_sha('lgcp')
json.dump(out, open('gpurun_out/%s/lgcp_pmc_summary.json' % T, 'w'), indent=1)
print(json.dumps(out)[:400])
PY
rm -rf $O/pmc_lgcp
python3 tools/probes/lgcp_time.py 20 32 64 128 600 2048 15000 > $O/lgcp_sizes.txt 2>/dev/null
bash tools/probes/lgcp_nsk_prof.sh $T > /dev/null 2>&1
bash tools/probes/lgcp_wide_prof.sh 600 $T > /dev/null 2>&1
python3 bench.py --config lgcp_n20_k128 --no-cpu-baseline --saturated 0 --no-legs --train-step 2>/dev/null | tail -1 > $O/lgcp_config_line.json
echo "lgcp done"
