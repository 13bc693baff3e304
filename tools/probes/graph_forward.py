"""Does replaying the forward launch sequence (prep + trajectory kernel + finalize) from a HIP graph shorten a step?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda")
seeds = torch.from_numpy(synthetic.throughput_seeds(2000)).cuda()
f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
for _ in range(3): f()
torch.cuda.synchronize()
def timeit(fn, reps=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
print("eager  %.4f ms/step" % timeit(f))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = f()
print("graph  %.4f ms/step" % timeit(g.replay))
print("eager  %.4f ms/step" % timeit(f))
