"""Host time of one bound_forward call (the Python boundary + ctypes + three launches), on the smallest configuration:
wall time per call of a long unsynchronised loop against the GPU's own time per call, and where the host time goes."""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
name = sys.argv[1] if len(sys.argv) > 1 else "gmm_n300_k8"
b = synthetic.build(name, device="cuda")
seeds = torch.from_numpy(synthetic.throughput_seeds(b["cfg"]["N"])).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
f = lambda: mcdbm.bound_forward(*args, **kw)
for _ in range(500): f()
torch.cuda.synchronize()
for label, ctx in (("default", None), ("fixed_parameters", mcdbm.fixed_parameters())):
    if ctx: ctx.__enter__()
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3000): f()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%-18s host issue %.2f us per call, wall incl. drain %.2f us per call" % (label, t_issue / 3000 * 1e6, t_all / 3000 * 1e6))
    if ctx: ctx.__exit__(None, None, None)
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): f()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
