"""Crossover of the gradient's work-item path vs whole-chain path (CMCD_GRAD_ITEM=1/0) over the batch size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm

def timeit(f, reps=5):
    f(); f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

for mode, fn in (("MCD_CAIS_sn", mcdbm.compute_bound_grad), ("MCD_CAIS_var_sn", mcdbm.compute_log_var_grad)):
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", boundmode=mode, init_sigma=15.0)
    kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    for n in (500, 1000, 2000, 4000, 6000, 8000, 12000, 16000):
        seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
        args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
        t = {}
        for item in ("0", "1"):
            os.environ["CMCD_GRAD_ITEM"] = item
            t[item] = timeit(lambda: fn(*args, **kw))
        print("%-16s n=%6d  chain %.3f ms   item %.3f ms" % (mode, n, t["0"], t["1"]))

b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
for n in (2000, 4000, 16000):
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    t = {}
    for item in ("0", "1"):
        os.environ["CMCD_GRAD_ITEM"] = item
        t[item] = timeit(lambda: mcdbm.compute_log_var_grad(*args, **kw), reps=2)
    print("config 4 (132-wide, VarGrad) n=%6d  chain %.3f ms   item %.3f ms" % (n, t["0"], t["1"]))
