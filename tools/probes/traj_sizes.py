"""The wave-per-tile trajectory kernel (kernel variant 1) on saturating batches of several networks: A / B of kernel builds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
CASES = [("many_gmm_n2000_k256_dds", 262144, {}), ("funnel_n300_k64", 65536, {}), ("gmm_n300_k8", 262144, dict(nbridges=64)),
         ("many_gmm_var_n16000_k256", 131072, dict(emb_dim=40, nbridges=64)), ("many_gmm_var_n16000_k256", 16000, {})]
for name, n, over in CASES:
    b = synthetic.build(name, device="cuda", **over)
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    mcdbm.KERNEL_VARIANT = 1
    f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                    eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    f(); f(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
    K = b["params_fixed"][1]
    print("%-26s %-28s n = %7d  %.3f ms per launch  %.3e particle-steps/s" % (name, over, n, ms / cnt, n * K / (ms / cnt) * 1e3), flush=True)
