"""Forward time of the 2nd-order mode on the named shape (or argv[1]) — kernel variant and probes from the environment."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic, _lib
from cmcd_amd import mcdboundingmachine as mcdbm
name = sys.argv[1] if len(sys.argv) > 1 else "many_gmm_n2000_k256_dds"
over = dict(init_eps=0.2, init_gamma=2.0, init_sigma=15.0) if "many" in name else dict(init_eps=0.05, init_gamma=4.0)
b = synthetic.build(name, device="cuda", boundmode="MCD_CAIS_UHA_sn", **over)
n = int(sys.argv[2]) if len(sys.argv) > 2 else b["cfg"]["N"]
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
best = []
for rep in range(5):
    for _ in range(20):
        mcdbm.compute_bound(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        mcdbm.compute_bound(*args)
    torch.cuda.synchronize()
    best.append((time.perf_counter() - t0) / 200 * 1e3)
print("UHA_FWD", name, n, _lib.last_kernel_name(), "ms per call:", " ".join("%.4f" % x for x in best))
