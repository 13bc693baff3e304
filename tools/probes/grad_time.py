"""Times forward vs value-and-gradient of the VarGrad loss on the dds many_gmm configuration."""
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

for n in (2000, 16000, 65536):
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", boundmode="MCD_CAIS_var_sn", init_sigma=15.0)
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    tf = timeit(lambda: mcdbm.compute_bound_var(*args, **kw))
    tg = timeit(lambda: mcdbm.compute_log_var_grad(*args, **kw))
    K = 256
    print("n=%6d  forward %.3f ms (%.3e steps/s)   value+grad %.3f ms (%.3e steps/s)  ratio %.2f" % (
        n, tf, n * K / tf * 1e3, tg, n * K / tg * 1e3, tg / tf))

for n in (2000, 16000, 65536):
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", init_sigma=15.0)
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    tf = timeit(lambda: mcdbm.compute_bound(*args, **kw))
    tg = timeit(lambda: mcdbm.compute_bound_grad(*args, **kw))
    print("MCD_CAIS_sn reparameterised n=%6d  forward %.3f ms   value+grad %.3f ms (%.3e steps/s)  ratio %.2f" % (
        n, tf, tg, n * 256 / tg * 1e3, tg / tf))

for n in (2000, 16000):
    b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    tf = timeit(lambda: mcdbm.compute_bound_var(*args, **kw), reps=3)
    tg = timeit(lambda: mcdbm.compute_log_var_grad(*args, **kw), reps=3)
    print("config 4 (132-wide) n=%6d  forward %.3f ms (%.3e steps/s)   value+grad %.3f ms (%.3e steps/s)  ratio %.2f" % (
        n, tf, n * 256 / tf * 1e3, tg, n * 256 / tg * 1e3, tg / tf))
