# per-kernel split of config 4's VarGrad step at shard size: bash tools/probes/t9_grad_prof.sh [N]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/t9prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t9prof -- python3 tools/probes/t9_grad_run.py ${1:-2000} > /dev/null 2>&1
cp $(find gpurun_out/t9prof -name "*kernel_stats.csv" | head -1) gpurun_out/t9_grad_kernel_stats.csv
rm -rf gpurun_out/t9prof
cut -c1-200 gpurun_out/t9_grad_kernel_stats.csv | head -14
