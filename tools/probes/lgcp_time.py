import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
counts = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts)
for n in (20, 32, 64, 128, 600):
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                    eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): out = f()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("lgcp n=%d K=128: %.2f ms per call, %.3e particle-steps/s, mean loss %.3f" % (n, dt * 1e3, n * 128 / dt, float(out[0].mean())))
