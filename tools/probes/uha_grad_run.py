"""2nd-order mode value + gradient at a named shape, repeated (for rocprofv3 --kernel-trace --stats): argv = config n K [item 0|1]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
name = sys.argv[1] if len(sys.argv) > 1 else "many_gmm_n2000_k256_dds"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 256
if len(sys.argv) > 4:
    os.environ["CMCD_GRAD_ITEM"] = sys.argv[4]
over = dict(init_eps=0.2, init_gamma=2.0, init_sigma=15.0) if "many" in name else dict(init_gamma=3.0)
b = synthetic.build(name, device="cuda", boundmode="MCD_CAIS_UHA_sn", nbridges=K, **over)
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
for _ in range(3):
    mcdbm.compute_bound_grad(*args)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    mcdbm.compute_bound_grad(*args)
torch.cuda.synchronize()
print("UHA_GRAD", name, n, K, os.environ.get("CMCD_GRAD_ITEM"), "%.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
