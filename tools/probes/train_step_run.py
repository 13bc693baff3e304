"""The north-star configuration's two training steps (value + gradient), repeated (for rocprofv3 --kernel-trace --stats).
usage: train_step_run.py [bptt|var] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
which = sys.argv[1] if len(sys.argv) > 1 else "bptt"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", boundmode="MCD_CAIS_var_sn" if which == "var" else "MCD_CAIS_sn")
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
fn = mcdbm.compute_log_var_grad if which == "var" else mcdbm.compute_bound_grad
for _ in range(50):
    fn(*args, **kw)
torch.cuda.synchronize()
