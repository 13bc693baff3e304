# SQ counters of the 2nd-order mode's kernels on the named batch's shape, one rocprofv3 --pmc pass per group
# usage: bash tools/probes/pmc_uha.sh <tag>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1
mkdir -p $O
rm -rf $O/pmc
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/$tag -- python3 tools/probes/uha_run.py 2000 manyonly > /dev/null 2>&1 || echo "pass failed: $set"
done
python3 - $O <<'PY'
import glob, csv, collections, json, sys
O = sys.argv[1]
out = collections.defaultdict(dict)
for f in sorted(glob.glob(O + '/pmc/*/*/*counter_collection.csv')):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'uha_' in n:
            acc[(n.split('(')[0][-48:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (n, c), v in acc.items():
        out[n][c] = sum(v) / len(v)
json.dump(out, open(O + '/pmc_sq_uha.json', 'w'), indent=1)
for n, d in out.items():
    print(n, json.dumps(d))
PY
rm -rf $O/pmc
