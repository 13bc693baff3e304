#!/bin/bash
# gradient kernel times of the overdamped modes for several library variants (tools/probes/build_variant.py)
# usage: bash tools/probes/grad_variants.sh <tag> <variant> [...]   ("product" = cmcd_amd/libcmcd_hip.so)
O=gpurun_out/$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
for v in "$@"; do
  if [ "$v" = product ]; then unset CMCD_LIB_PATH; else export CMCD_LIB_PATH=$PWD/cmcd_amd/libcmcd_hip_$v.so; fi
  rm -rf $O/prof_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 tools/probes/grad_run.py ${N:-2000} > $O/times_$v.json 2>/dev/null || { echo "$v failed"; continue; }
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv, sys
out = []
for r in csv.reader(open(sys.argv[1])):
    if 'grad_kernel' in r[0] or 'bptt' in r[0] or 'coop_kernel' in r[0] or 'tails_fused' in r[0] or 'scan' in r[0]:
        out.append('%s %.1f us x%s' % (r[0].split('(')[0][-40:], float(r[3]) / 1000, r[1]))
print(sys.argv[2] + ': ' + ' | '.join(out))" $f $v
  grep GRAD_TIMES $O/times_$v.json
  rm -rf $O/prof_$v
done
