"""A launch sequence of the library, a few hundred calls back to back, for `rocprofv3 --kernel-trace` (tools/probes/seq_trace.sh):
    python tools/probes/seq_trace.py <config> <what: forward | prepared | grad | vargrad> [particles]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
name, what = sys.argv[1], sys.argv[2]
over = {}
if "lgcp" in name:
    over["lgcp_counts"] = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
if what == "vargrad" and "var" not in name:
    over["boundmode"] = "MCD_CAIS_var_sn"
b = synthetic.build(name, device="cuda", **over)
n = int(sys.argv[3]) if len(sys.argv) > 3 else b["cfg"]["N"]
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
fn = {"forward": mcdbm.bound_forward, "prepared": mcdbm.bound_forward, "grad": mcdbm.compute_bound_grad,
      "vargrad": mcdbm.compute_log_var_grad}[what]
reps = 60 if "lgcp" in name else 300
for _ in range(reps // 3): fn(*args, **kw)
torch.cuda.synchronize()
if what == "prepared":
    with mcdbm.fixed_parameters():
        for _ in range(reps): fn(*args, **kw)
else:
    for _ in range(reps): fn(*args, **kw)
torch.cuda.synchronize()
print("calls", reps)
