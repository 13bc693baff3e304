"""rocprofv3 target: a few value+gradient calls at the north-star batch (both training modes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
for mode, fn in (("MCD_CAIS_sn", mcdbm.compute_bound_grad), ("MCD_CAIS_var_sn", mcdbm.compute_log_var_grad)):
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", boundmode=mode, init_sigma=15.0)
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    for _ in range(6):
        fn(*args, **kw)
    torch.cuda.synchronize()
