# Round-5 evidence of the trajectory kernels on the GPU box: bash tools/probes/round_end_r05.sh <tag>
#   gpurun_out/<tag>/bench_driver_flags_{1..3}.json   the DRIVER's command (--gpus 1 --steps 20 --warmup 5), three times in this lease
#   gpurun_out/<tag>/uha_times.json + kernel_stats_uha.csv   the 2nd-order mode: forward and value + gradient
#   gpurun_out/<tag>/bench.json                 the default bench line (N = 1)
#   gpurun_out/<tag>/kernel_stats.csv           rocprofv3 --kernel-trace --stats of the forward-only bench command
#   gpurun_out/<tag>/kernel_stats_cfg4_shard.csv  the same for config 4's 2000-particle shard (forward + VarGrad step)
#   gpurun_out/<tag>/pmc_summary.json           HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, separate --pmc passes) and SQ counters
#                                               per launch of the two cooperative kernels + the kernel-source sha
#   gpurun_out/<tag>/all_configs.jsonl          one bench line per BASELINE configuration
T=${1:-r05_z}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
for k in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags_$k.json 2> $O/bench.err || exit 1; done
python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
# the plain N > 1 command (no launcher on the command line), two ranks sharing this box's one GPU over gloo (test hook): the
# SHAPE of the N > 1 line (roofline + cpu_baseline + collective + strong_scaling); its rates mean nothing on one GPU
CMCD_BENCH_SHARED_GPU=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench_plain_gpus2_shared_gpu.json 2>> $O/bench.err || echo "plain --gpus 2 failed"
echo "bench done"
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 300 --warmup 50 --no-cpu-baseline --saturated 0 --forward-only --no-legs > $O/prof_bench.json 2> /dev/null || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/probes/t9_grad_run.py 2000 > /dev/null 2>&1 || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_cfg4_shard.csv
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/probes/uha_run.py 2000 > $O/uha_times.json 2>/dev/null || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_uha.csv
rm -rf $O/prof
echo "stats done"
# --pmc passes (own runs, --kernel-trace only): headline batch through bench.py, config 4's shard through the probe
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/a_$tag -- python3 bench.py --steps 5 --warmup 2 --spinup 0 --no-cpu-baseline --saturated 0 --forward-only --no-legs > /dev/null 2>&1 || echo "pass failed: $set"
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc/b_$tag -- python3 tools/probes/t9_grad_run.py 2000 > /dev/null 2>&1 || echo "pass failed (t9): $set"
  echo "pmc $tag done"
done
python3 - $T <<'PY'
import glob, csv, collections, json, sys, os
sys.path.insert(0, os.getcwd())
import bench
T = sys.argv[1]
out = {"kernel_sources_sha": bench.kernel_sources_sha(),
       "_note": "per launch averages; FETCH_SIZE / WRITE_SIZE in KB, hbm_bytes_per_launch = (2 FETCH_SIZE + WRITE_SIZE) KB "
                "(MI355X_MICROARCH.md: gfx950 counts 128-B requests as 64 B); coop_kernel = north-star batch "
                "(2000 particles, K = 256, dds, 8-particle tiles), coop_kernel_t9_half = config 4's 2000-particle shard "
                "(132-wide net, 8-particle tiles, 12 waves); collected by tools/probes/round_end_r05.sh"}
for key, pat, want in (("coop_kernel", "a_", "Li4ELb1ELb0E"), ("coop_kernel_t9_half", "b_", "Li9ELb1ELb1E")):
    acc = collections.defaultdict(list)
    name = None
    for f in sorted(glob.glob('gpurun_out/%s/pmc/%s*/*/*counter_collection.csv' % (T, pat))):
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name']
            if 'coop_kernel' in kn and (("4, true, false" in kn or "4, true>" in kn) if key == "coop_kernel" else "9, true, true" in kn):
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
                name = kn[:90]
    d = {k: sum(v) / len(v) for k, v in acc.items()}
    if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
        d['hbm_bytes_per_launch'] = (2 * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024
    d['kernel'] = name
    out[key] = d
json.dump(out, open('gpurun_out/%s/pmc_summary.json' % T, 'w'), indent=1)
print(json.dumps(out)[:1500])
PY
rm -rf $O/pmc
# lgcp launch sequence: FETCH_SIZE / WRITE_SIZE per GEMM launch type (separate --pmc passes), with the kernel-source sha
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
print(json.dumps(out)[:800])
PY
rm -rf $O/pmc_lgcp
python3 tools/probes/lgcp_time.py 20 32 64 128 600 2048 15000 > $O/lgcp_sizes.txt 2>/dev/null
python3 tools/probes/funnel_ab.py > $O/funnel_ab_wide8_vs_narrow.txt 2>/dev/null
python3 tools/probes/cfg4_sizes.py > $O/cfg4_sizes.txt 2>/dev/null
python3 tools/probes/traj_sizes.py > $O/traj_sizes.txt 2>/dev/null
bash tools/probes/lgcp_nsk_prof.sh $T > /dev/null 2>&1
bash tools/probes/lgcp_wide_prof.sh 600 $T > /dev/null 2>&1
for c in gmm_n300_k8 funnel_n300_k64 many_gmm_n2000_k256_dds many_gmm_var_n16000_k256 lgcp_n20_k128; do
  python3 bench.py --config $c --no-cpu-baseline --saturated 0 --no-legs --train-step 2>/dev/null | tail -1
done > $O/all_configs.jsonl
head -c 600 $O/bench.json; echo; head -6 $O/kernel_stats.csv | cut -c1-160; head -5 $O/kernel_stats_cfg4_shard.csv | cut -c1-160
