# SQ counters of the cooperative kernel on the north-star batch (forward launches only), one rocprofv3 --pmc pass per group
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc3
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc3/$tag -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --saturated 0 --forward-only > /dev/null 2>&1 || echo "pass failed: $set"
done
python3 - <<'PY'
import glob, csv, collections, json
out = {}
for f in sorted(glob.glob('gpurun_out/pmc3/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'coop_kernel' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        out[k] = sum(v) / len(v)
json.dump(out, open('gpurun_out/pmc_sq_coop.json', 'w'), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/pmc3
