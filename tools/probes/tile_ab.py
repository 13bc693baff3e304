"""Cooperative kernel on 16-particle tiles (variant 3) against 8-particle tiles (variant 4): kernel time from the
library's HIP events, and the largest per-particle difference between the two (reduction order of layer 3 only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm


def run(name, n, variant, reps=10):
    mcdbm.KERNEL_VARIANT = variant
    b = synthetic.build(name, device="cuda")
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                    eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    f(); f(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        out = f()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
    return ms / cnt, out[0].double().cpu().numpy()


for name, sizes in [("many_gmm_n2000_k256_dds", [256, 1024, 2000, 2048, 3000, 4096]),
                    ("funnel_n300_k64", [300, 2000]), ("gmm_n300_k8", [300, 2000])]:
    K = synthetic.CONFIGS[name]["nbridges"]
    for n in sizes:
        t3, l3 = run(name, n, 3)
        t4, l4 = run(name, n, 4)
        f = np.isfinite(l3)
        same_inf = np.array_equal(f, np.isfinite(l4))
        rel = np.abs(l3[f] - l4[f]) / np.maximum(1, np.abs(l3[f]))
        print("%-26s n=%5d  16-tile %.4f ms  8-tile %.4f ms  (%+.1f %%)  %.3e steps/s  inf-set equal %s  rel p50 %.1e p99 %.1e max %.1e"
              % (name, n, t3, t4, 100 * (t4 / t3 - 1), n * K / min(t3, t4) / 1e-3, same_inf, np.median(rel),
                 np.quantile(rel, 0.99), rel.max()), flush=True)
