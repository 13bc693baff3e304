"""The overdamped modes' value + gradient on the named batch's shape (many_gmm, N = 2000, K = 256, dds), repeated for
rocprofv3 --kernel-trace --stats and timed: MCD_CAIS_sn (reparameterised) and MCD_CAIS_var_sn (VarGrad)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out = {}
for mode, fn in (("MCD_CAIS_sn", mcdbm.compute_bound_grad), ("MCD_CAIS_var_sn", mcdbm.compute_log_var_grad)):
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", boundmode=mode, init_sigma=15.0)
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    for _ in range(3):
        fn(*args, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        fn(*args, **kw)
    torch.cuda.synchronize()
    out[mode] = (time.perf_counter() - t0) / 20 * 1e3
print("GRAD_TIMES", json.dumps(out))
