"""A / B of the funnel (BASELINE configs[1]: d = 10, N = 300, K = 64) on 8-particle tiles: kernel variant 4 (coop_wide8_kernel)
against variant 5 (coop_kernel's 8-particle instance, the r04 form).  Interleaved rounds, HIP-event kernel time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm

name = sys.argv[1] if len(sys.argv) > 1 else "funnel_n300_k64"
b = synthetic.build(name, device="cuda")
n = int(sys.argv[2]) if len(sys.argv) > 2 else b["cfg"]["N"]
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
res = {4: [], 5: []}
for rnd in range(5):
    for v in (4, 5):
        mcdbm.KERNEL_VARIANT = v
        for _ in range(200):
            mcdbm.bound_forward(*args, **kw)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(400):
            mcdbm.bound_forward(*args, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 400
        ms, cnt = _lib.profile_collect()
        _lib.profile_enable(False)
        res[v].append((ms / cnt * 1e3, dt * 1e6))
for v in (4, 5):
    print("variant", v, _lib.last_kernel_name() if v == 5 else "", " kernel us:", " ".join("%.1f" % k for k, _ in res[v]),
          " per call us (events on):", " ".join("%.1f" % c for _, c in res[v]))
