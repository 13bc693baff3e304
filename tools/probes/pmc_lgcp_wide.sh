# SQ counters of the wide-batch lgcp GEMM launches (cmcd_lgcp_wide.hip) at N particles: bash tools/probes/pmc_lgcp_wide.sh [n] [tag]
# separate rocprofv3 --pmc passes (kernel trace only), per-launch averages by launch type (A / B / C by grid size)
N=${1:-600}
T=${2:-pmc_lgcp_wide}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p_$tag -- python3 bench.py --config lgcp_n20_k128 --particles $N --steps 2 --warmup 1 --spinup 0 --no-cpu-baseline --saturated 0 --no-legs > /dev/null 2>&1 || echo "pass failed: $set"
done
python3 - $O <<'PY'
import glob, csv, collections, json, sys
O = sys.argv[1]
acc = collections.defaultdict(list)
for f in sorted(glob.glob(O + '/p_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'lgcp_wide_gemm' in r['Kernel_Name']:
            acc[(r.get('Grid_Size', r.get('Grid_Size_X', '?')), r['Counter_Name'])].append(float(r['Counter_Value']))
out = {}
for (g, c), v in sorted(acc.items()):
    out.setdefault('grid ' + str(g), {})[c] = sum(v) / len(v)
json.dump(out, open(O + '/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -delete
