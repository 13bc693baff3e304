"""How much of the north-star training step is host time?  Wall per step against the GPU time of the same steps (events)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
for which in ("var", "bptt", "fwd"):
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", boundmode="MCD_CAIS_var_sn" if which == "var" else "MCD_CAIS_sn")
    kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    seeds = torch.from_numpy(synthetic.throughput_seeds(2000)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    fn = {"var": mcdbm.compute_log_var_grad, "bptt": mcdbm.compute_bound_grad, "fwd": mcdbm.compute_bound}[which]
    for _ in range(200):
        fn(*args, **kw)
    torch.cuda.synchronize()
    reps = 300
    t0 = time.perf_counter()
    for _ in range(reps):
        fn(*args, **kw)
    t_enq = time.perf_counter() - t0          # host time to enqueue (no sync)
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: host enqueue %.3f ms per step, wall %.3f ms per step" % (which, t_enq / reps * 1e3, t_all / reps * 1e3))
