#!/bin/bash
# the 2nd-order mode's value + gradient on the named batch's shape: wall time and the per-kernel table
# usage: bash tools/probes/uha_grad_prof.sh <tag>
O=gpurun_out/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/probes/uha_run.py 2000 nolgcp > $O/uha_times.json 2>$O/uha_times.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 tools/probes/uha_run.py 2000 nolgcp > /dev/null 2>&1 || exit 1
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/kernel_stats_uha.csv
rm -rf $O/prof
cut -d, -f1-4 $O/kernel_stats_uha.csv | head -24
cat $O/uha_times.json
