"""Config 4's net (many_gmm, geffner width 132, K = 256) at shard sizes: forward kernel time per trajectory-kernel variant
(1 wave per tile, 3 cooperative on 16-particle tiles, 4 cooperative on 8-particle tiles) and the VarGrad step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm

b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])


def ktime(n, variant, reps=20):
    mcdbm.KERNEL_VARIANT = variant
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"], **kw)
    try:
        for _ in range(30):
            f()
    except NotImplementedError as e:
        return None
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_collect()
    _lib.profile_enable(False)
    return ms / cnt


def wall(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


for n in (1000, 2000, 4000, 8000, 16000):
    row = {v: ktime(n, v) for v in (1, 3, 4)}
    mcdbm.KERNEL_VARIANT = 0
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    tf = wall(lambda: mcdbm.compute_bound_var(*args, **kw))
    tg = wall(lambda: mcdbm.compute_log_var_grad(*args, **kw))
    print("N=%6d  kernel ms: wave/tile %s  coop16 %s  coop8 %s | auto forward call %.3f ms, VarGrad value+gradient %.3f ms" % (
        n, *["%.3f" % row[v] if row[v] is not None else "  -  " for v in (1, 3, 4)], tf, tg), flush=True)
