"""rocprofv3 target: two VarGrad value+gradient calls of config 4 (132-wide geffner net) at N = 2000."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
for _ in range(2):
    mcdbm.compute_log_var_grad(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                               eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
torch.cuda.synchronize()
