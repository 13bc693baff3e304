# config 4's 2000-particle shard VarGrad step under rocprofv3, two library builds interleaved: bash tools/probes/t9_vargrad_ab.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t9ab
for r in 1 2; do
  for l in $1 $2; do
    rm -rf gpurun_out/t9ab/prof
    CMCD_LIB_PATH=$PWD/$l rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t9ab/prof -- python3 tools/probes/t9_grad_run.py 2000 > /dev/null 2>&1 || exit 1
    python3 -c "
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'coop_kernel' in r['Name'] or 'grad_kernel' in r['Name']:
        print(sys.argv[2], 'round', sys.argv[3], r['Name'][:48], r['Calls'], '%.1f us' % (float(r['AverageNs']) / 1e3))
" $(find gpurun_out/t9ab/prof -name '*kernel_stats.csv' | head -1) $l $r
  done
done
rm -rf gpurun_out/t9ab/prof
