cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc_item/$tag -- python3 tools/probes/grad_prof.py > /dev/null 2>&1
done
python3 - <<'PY'
import glob,csv,collections
for f in sorted(glob.glob('gpurun_out/pmc_item/*/*/*counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        for key in ('grad_kernel<2, 1, 2, 4, 4, false, true, true>', 'bptt_jac_kernel'):
            if key in r['Kernel_Name']:
                acc[(key[:14], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
