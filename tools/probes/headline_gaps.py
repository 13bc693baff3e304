"""The named batch, default path (prep launch in every call), a few hundred calls back to back: run under
`rocprofv3 --kernel-trace` (tools/probes/headline_gaps.sh) to read the per-kernel durations AND the gaps between them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build(synthetic.NORTH_STAR, device="cuda")
seeds = torch.from_numpy(synthetic.throughput_seeds(2000)).cuda()
f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
prepared = len(sys.argv) > 1 and sys.argv[1] == "prepared"
for _ in range(300): f()
torch.cuda.synchronize()
if prepared:
    with mcdbm.fixed_parameters():
        for _ in range(400): f()
else:
    for _ in range(400): f()
torch.cuda.synchronize()
