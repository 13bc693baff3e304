# SQ counters of a trajectory kernel through bench.py (separate rocprofv3 --pmc passes, kernel trace only):
#   bash tools/probes/pmc_traj.sh <tag> <kernel-name substring> <bench.py arguments ...>
# -> gpurun_out/<tag>/summary.json: per-launch averages + the issue-slot shares of a SIMD:
#      mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / SIMD cycles, valu_active = 4 (SQ_ACTIVE_INST_VALU - SQ_INSTS_MFMA) / SIMD cycles,
#      SIMD cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the XCDs) x 256 CUs x 4
T=$1; shift
PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p_$tag -- python3 bench.py "$@" > /dev/null 2>&1 || echo "pass failed: $set"
done
python3 - $O "$PAT" "$@" <<'PY'
import glob, csv, collections, json, sys, os
sys.path.insert(0, os.getcwd())
import bench
O, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
name = None
for f in sorted(glob.glob(O + '/p_*/*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
            name = r['Kernel_Name'][:120]
d = {k: sum(v) / len(v) for k, v in acc.items()}
out = {"kernel": name, "bench_args": sys.argv[3:], "kernel_sources_sha": bench.kernel_sources_sha(), "counters_per_launch": d}
try:
    simd = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    mfma = d["SQ_VALU_MFMA_BUSY_CYCLES"]
    valu = 4.0 * (d["SQ_ACTIVE_INST_VALU"] - d["SQ_INSTS_MFMA"])
    out["issue_occupancy"] = {"simd_cycles": simd, "mfma_busy_frac": mfma / simd, "valu_active_frac": valu / simd,
                              "sum_frac": (mfma + valu) / simd, "wait_any_frac_of_wave_cycles": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                              "wait_inst_any_frac_of_wave_cycles": d["SQ_WAIT_INST_ANY"] / d["SQ_WAVE_CYCLES"],
                              "valu_insts_per_wave": (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) / d["SQ_WAVES"],
                              "mfma_insts_per_wave": d["SQ_INSTS_MFMA"] / d["SQ_WAVES"],
                              "trans_insts_per_wave": d.get("SQ_INSTS_VALU_TRANS", 0) / d["SQ_WAVES"]}
except Exception as e:
    out["issue_occupancy_error"] = repr(e)
json.dump(out, open(O + '/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
find $O -name "*.csv" -delete
