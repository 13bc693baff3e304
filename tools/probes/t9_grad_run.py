"""config 4's VarGrad step on a 2000-particle shard, repeated (for rocprofv3 --kernel-trace --stats)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
for _ in range(30):
    mcdbm.compute_log_var_grad(*args, **kw)
torch.cuda.synchronize()
