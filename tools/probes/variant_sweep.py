"""Times both trajectory-kernel variants on the BASELINE configs at several batch sizes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm

def time_cfg(name, n, variant, reps=5):
    mcdbm.KERNEL_VARIANT = variant
    b = synthetic.build(name, device="cuda")
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                    eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    f(); f(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps): f()
    ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
    return ms / cnt

for name, sizes in [("many_gmm_n2000_k256_dds", [512, 2000, 4096, 8192, 16384, 32768, 65536, 262144]),
                    ("many_gmm_var_n16000_k256", [2000, 4096, 16000, 65536]),
                    ("funnel_n300_k64", [300, 4096, 65536]),
                    ("gmm_n300_k8", [300, 4096, 65536])]:
    K = synthetic.CONFIGS[name]["nbridges"]
    for n in sizes:
        t1, t2 = time_cfg(name, n, 1), time_cfg(name, n, 2)
        print("%-26s n=%7d tiles=%6d  wave-per-tile %.4f ms  coop %.4f ms  best=%s  %.3e steps/s" % (
            name, n, (n + 15) // 16, t1, t2, "coop" if t2 < t1 else "wave", n * K / min(t1, t2) / 1e-3))
