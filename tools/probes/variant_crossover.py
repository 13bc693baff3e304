"""Cooperative against wave-per-tile kernel around the selection threshold (512 tiles of 16 particles):
python tools/probes/variant_crossover.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm


def time_cfg(name, n, variant, reps=20):
    mcdbm.KERNEL_VARIANT = variant
    b = synthetic.build(name, device="cuda")
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                    eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    for _ in range(10): f()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps): f()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
    return ms / cnt


for name, sizes in [("many_gmm_n2000_k256_dds", [4096, 6144, 8192, 10240, 12288, 16384, 24576, 32768, 40960, 49152, 65536, 98304]),
                    ("many_gmm_var_n16000_k256", [4096, 8192, 12288, 13000, 14000, 16000]),
                    ("funnel_n300_k64", [8192, 12288, 16384, 24576]), ("gmm_n300_k8", [8192, 12288, 16384, 24576])]:
    K = synthetic.CONFIGS[name]["nbridges"]
    for n in sizes:
        t1, t3 = time_cfg(name, n, 1), time_cfg(name, n, 3)
        print("%-26s n=%7d tiles=%6d  wave-per-tile %.4f ms  cooperative (16-particle tiles) %.4f ms  best=%s" % (
            name, n, (n + 15) // 16, t1, t3, "coop" if t3 < t1 else "wave"), flush=True)
