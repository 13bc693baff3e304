# HBM traffic of the lgcp launch sequence: FETCH_SIZE / WRITE_SIZE per GEMM launch (one rocprofv3 --pmc pass per counter, as
# MI355X_MICROARCH.md prescribes; FETCH_SIZE doubled: gfx950 counts 128-B requests as 64 B), against the 10.2 - 20.6 MB of
# weights a launch streams algorithmically.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_hbm_lgcp/$c -- python3 bench.py --config lgcp_n20_k128 --steps 3 --warmup 1 --no-cpu-baseline --saturated 0 > /dev/null 2>&1
done
python3 - <<'PY'
import glob,csv,collections,json
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmc_hbm_lgcp/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'lgcp_gemm_kernel' in r['Kernel_Name']:
            acc[(r['Kernel_Name'][:40], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X','?'), r['Counter_Name'])].append(float(r['Counter_Value']))
out={}
for (k,g,c),v in sorted(acc.items()):
    out.setdefault(k+' grid '+str(g), {})[c]=sum(v)/len(v)
for k,d in out.items():
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        d['hbm_bytes_per_launch']=(2*d['FETCH_SIZE']+d['WRITE_SIZE'])*1024
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmc_hbm_lgcp/summary.json','w'), indent=1)
PY
find gpurun_out/pmc_hbm_lgcp -name "*counter_collection.csv" -delete; find gpurun_out/pmc_hbm_lgcp -name "*kernel_trace.csv" -delete
