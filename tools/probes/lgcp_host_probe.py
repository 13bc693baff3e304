"""Is the lgcp value+gradient host-bound?  Time until the call returns (launches enqueued) vs until the GPU is done."""
import os, sys, time
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
for name, fn in (("forward", mcdbm.compute_bound), ("value+gradient", mcdbm.compute_bound_grad)):
    fn(*args, **kw); torch.cuda.synchronize()
    enq, tot = [], []
    for _ in range(5):
        t0 = time.perf_counter(); fn(*args, **kw); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
    print("%s: enqueue %.2f ms, complete %.2f ms" % (name, min(enq), min(tot)))
