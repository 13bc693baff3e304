# per-kernel split of the north-star training steps: bash tools/probes/train_step_prof.sh
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for w in bptt var; do
  rm -rf gpurun_out/tsprof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/tsprof -- python3 tools/probes/train_step_run.py $w > /dev/null 2>&1
  cp $(find gpurun_out/tsprof -name "*kernel_stats.csv" | head -1) gpurun_out/train_step_${w}_kernel_stats.csv
  rm -rf gpurun_out/tsprof
  echo "== $w"; cut -c1-150 gpurun_out/train_step_${w}_kernel_stats.csv | head -16
done
