# The part of round_end_r05.sh that depends on cmcd_uha.hip (changed after the round-end collection), plus the driver's command
# again on the same build: bash tools/probes/round_end_r05_uha.sh <tag>
T=${1:-r05_y}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$T
mkdir -p $O
for k in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags_$k.json 2> $O/bench.err || exit 1; done
python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/probes/uha_run.py 2000 > $O/uha_times.json 2>/dev/null || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_uha.csv
rm -rf $O/prof
python3 tools/probes/uha_fwd_time.py funnel_n300_k64 > $O/uha_funnel_fwd.txt 2>&1
python3 tools/probes/uha_grad_run.py funnel_n300_k64 300 64 >> $O/uha_funnel_fwd.txt 2>&1
echo "uha done"
