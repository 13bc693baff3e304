#!/bin/bash
# sweep / Jacobian kernel time of the 2nd-order gradient for several library variants (tools/probes/build_variant.py)
# usage: bash tools/probes/uha_variants.sh <tag> <variant> [...]   ("product" = cmcd_amd/libcmcd_hip.so)
O=gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  if [ "$v" = product ]; then unset CMCD_LIB_PATH; else export CMCD_LIB_PATH=$PWD/cmcd_amd/libcmcd_hip_$v.so; fi
  rm -rf $O/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 tools/probes/uha_run.py ${N:-2000} manyonly > $O/times_$v.json 2>/dev/null || { echo "$v failed"; continue; }
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv, sys
out = []
for r in csv.reader(open(sys.argv[1])):
    if 'uha_grad_kernel' in r[0] or 'uha_coop' in r[0] or 'uha_reduce' in r[0] or 'compose' in r[0]:
        out.append('%s %.1f us' % (r[0].split('(')[0][-34:], float(r[3]) / 1000))
print(sys.argv[2] + ': ' + ' | '.join(out))" $f $v
  rm -rf $O/prof_$v
done
