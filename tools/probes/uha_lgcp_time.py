"""MCD_CAIS_UHA_sn on lgcp (d = 1600, net width 3220): forward (and value + gradient with `grad`) at N = 20, K = 128."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
counts = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 128
b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts, boundmode="MCD_CAIS_UHA_sn", nbridges=K, init_eps=0.02,
                    init_gamma=5.0)
seeds = torch.from_numpy(synthetic.throughput_seeds(20)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
f = lambda: mcdbm.bound_forward(*args)
f(); torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5): out = f()
torch.cuda.synchronize()
print("UHA lgcp n=20 K=%d forward: %.2f ms per call, mean loss %.3f" % (K, (time.perf_counter() - t) / 5 * 1e3, float(out[0].mean())))
if "grad" in sys.argv:
    g = lambda: mcdbm.compute_bound_grad(*args)
    g(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): g()
    torch.cuda.synchronize()
    print("UHA lgcp n=20 K=%d value + gradient: %.2f ms per call" % (K, (time.perf_counter() - t) / 3 * 1e3))
