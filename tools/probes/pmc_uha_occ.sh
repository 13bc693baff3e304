# mean wave residency of the 2nd-order gradient's kernels: SQ_WAVE_CYCLES (wave-cycles / 4) against the launch's duration
# usage: bash tools/probes/pmc_uha_occ.sh <tag> <variant or "product">
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1
mkdir -p $O
if [ "$2" != product ]; then export CMCD_LIB_PATH=$PWD/cmcd_amd/libcmcd_hip_$2.so; fi
rm -rf $O/pmc_occ
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_occ -- python3 tools/probes/uha_run.py 2000 manyonly > /dev/null 2>&1 || echo "pass failed"
python3 - $O <<'PY'
import glob, csv, collections, sys
O = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(O + '/pmc_occ/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'uha_grad' in r['Kernel_Name']:
            acc[(r['Kernel_Name'].split('(')[0][-30:], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(acc):
    print(k, sum(acc[k]) / len(acc[k]))
PY
rm -rf $O/pmc_occ
