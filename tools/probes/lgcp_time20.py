"""lgcp forward at the named batch only (N = 20, K = 128), repeated: for per-launch durations under rocprofv3."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
counts = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts)
seeds = torch.from_numpy(synthetic.throughput_seeds(20)).cuda()
f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
f(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10): out = f()
torch.cuda.synchronize()
print("lgcp n=20 K=128: %.3f ms per call" % ((time.perf_counter() - t) / 10 * 1e3))
