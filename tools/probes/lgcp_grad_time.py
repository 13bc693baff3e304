"""Times forward vs value+gradient on the lgcp configuration (d = 1600, N = 20, K = 128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
counts = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts)
seeds = torch.from_numpy(synthetic.throughput_seeds(20)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
def timeit(f, reps=3):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
tf = timeit(lambda: mcdbm.compute_bound(*args, **kw))
tg = timeit(lambda: mcdbm.compute_bound_grad(*args, **kw))
g, (l, z) = mcdbm.compute_bound_grad(*args, **kw)
print("lgcp N=20 K=128: forward %.2f ms, value+gradient %.2f ms; mean loss %.3f, |grad| %.3e, finite %s" % (
    tf, tg, float(l.mean()), float(g.norm()), bool(torch.isfinite(g).all())))
