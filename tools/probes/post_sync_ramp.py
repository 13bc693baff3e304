"""Per-launch duration of the named batch's forward right after a device synchronisation (what a 20-step timed region
sees): is the first stretch after the idle gap slower?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda")
seeds = torch.from_numpy(synthetic.throughput_seeds(2000)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
for _ in range(2000):
    mcdbm.bound_forward(*args, **kw)
for gap_ms in (0, 1, 10, 100):
    torch.cuda.synchronize()
    if gap_ms:
        time.sleep(gap_ms / 1e3)
    n = 40
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for k in range(n):
        mcdbm.bound_forward(*args, **kw)
        ev[k + 1].record()
    torch.cuda.synchronize()
    d = [ev[k].elapsed_time(ev[k + 1]) * 1e3 for k in range(n)]
    print("gap %3d ms: per-call us" % gap_ms, " ".join("%.0f" % x for x in d[:12]), "... mean(first 20) %.1f mean(last 20) %.1f" % (sum(d[:20]) / 20, sum(d[20:]) / 20), flush=True)
    for _ in range(1500):
        mcdbm.bound_forward(*args, **kw)
