"""Headline batch (many_gmm, N = 2000, K = 256, dds): kernel time by HIP events, interleaved rounds of the product library and
a variant (CMCD_VARIANT_LIB), each in its own process."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build(synthetic.NORTH_STAR, device="cuda")
seeds = torch.from_numpy(synthetic.throughput_seeds(2000)).cuda()
f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
with mcdbm.fixed_parameters():
    for _ in range(1500): f()
    torch.cuda.synchronize()
    out = []
    for rnd in range(4):
        _lib.profile_enable(True)
        for _ in range(400): f()
        torch.cuda.synchronize()
        ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
        out.append(ms / cnt * 1e3)
print(" ".join("%%.2f" %% v for v in out))
''' % ROOT
variant = sys.argv[1]
for rnd in range(3):
    for tag, lib in (("product", None), ("variant", variant)):
        env = dict(os.environ)
        if lib: env["CMCD_LIB_PATH"] = os.path.join(ROOT, lib)
        else: env.pop("CMCD_LIB_PATH", None)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        print(tag, r.stdout.strip() or r.stderr[-300:], flush=True)
