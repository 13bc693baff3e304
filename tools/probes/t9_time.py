"""Times forward and VarGrad value+gradient of config 4 (many_gmm, 132-wide geffner net, K = 256) at N = 2000 and 16000."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
def timeit(f, reps=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for n in (2000, 16000):
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    tf = timeit(lambda: mcdbm.compute_bound_var(*args, **kw))
    tg = timeit(lambda: mcdbm.compute_log_var_grad(*args, **kw))
    print("config 4, N=%d: forward %.3f ms, value+gradient %.3f ms" % (n, tf, tg))
