"""The wave-per-tile kernel on the saturating dds batch (2^18 particles of the named shape): product library against a variant
(argv[1]), alternating, each in its own process.  Kernel time by HIP events."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys
sys.path.insert(0, %r)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build(synthetic.NORTH_STAR, device="cuda")
n = 1 << 18
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
mcdbm.KERNEL_VARIANT = 1
f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
f(); f(); torch.cuda.synchronize()
out = []
for rnd in range(3):
    _lib.profile_enable(True)
    for _ in range(6): f()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
    out.append(ms / cnt)
print(" ".join("%%.3f" %% v for v in out))
''' % ROOT
variant = sys.argv[1]
for rnd in range(3):
    for tag, lib in (("product", None), ("variant", variant)):
        env = dict(os.environ)
        if lib: env["CMCD_LIB_PATH"] = os.path.join(ROOT, lib)
        else: env.pop("CMCD_LIB_PATH", None)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(tag, r.stdout.strip() or r.stderr[-300:], flush=True)
