# HBM traffic of the trajectory kernel per launch: one rocprofv3 --pmc pass per TCC counter (MI355X_MICROARCH.md:
# FETCH_SIZE / WRITE_SIZE in KB; FETCH_SIZE counts 128-B requests as 64 B on gfx950 -> doubled).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_hbm/$c -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --saturated 0 > /dev/null 2>&1
done
python3 - <<'PY'
import glob,csv,collections,json
acc=collections.defaultdict(list)
for f in sorted(glob.glob('gpurun_out/pmc_hbm/*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'coop_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
out={k: sum(v)/len(v) for k,v in acc.items()}
out['hbm_bytes_per_launch']=(2*out['FETCH_SIZE']+out['WRITE_SIZE'])*1024
print(json.dumps(out))
PY
