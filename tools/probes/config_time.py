"""Kernel time of one BASELINE configuration's forward (HIP events around the trajectory kernel), optionally for two
library builds in turn:  python tools/probes/config_time.py funnel_n300_k64 [libA.so libB.so]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 3:
    for r in range(3):
        for l in sys.argv[2:4]:
            o = subprocess.run([sys.executable, os.path.abspath(__file__), sys.argv[1]], env=dict(os.environ, CMCD_LIB_PATH=os.path.abspath(l)),
                               capture_output=True, text=True)
            print(os.path.basename(l), o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:], flush=True)
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
name = sys.argv[1]
b = synthetic.build(name, device="cuda")
n = b["cfg"]["N"]
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
for _ in range(300): f()
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(300): out = f()
torch.cuda.synchronize()
ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
print("%s: kernel %.4f ms, mean loss %.6f" % (name, ms / cnt, float(out[0][torch.isfinite(out[0])].mean())))
