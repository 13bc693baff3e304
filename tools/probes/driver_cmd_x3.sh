# the driver's N = 1 command three times in this lease: bash tools/probes/driver_cmd_x3.sh <tag>
T=${1:-lease}
cd $GRAFT_REPO_ROOT
for k in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/${T}_$k.json; done
python3 - $T <<'PY'
import json, sys
T = sys.argv[1]
for k in (1, 2, 3):
    d = json.loads(open('gpurun_out/%s_%d.json' % (T, k)).read())
    print(T, 'run', k, 'ms_per_step %.5f' % d['ms_per_step'], 'value %.4g' % d['value'], 'kernel_ms %.5f' % d['roofline']['kernel_ms'],
          'frac %.4f' % d['roofline']['frac'], 'prepared %.5f' % d['legs']['weak_prepared']['ms_per_step'])
PY
