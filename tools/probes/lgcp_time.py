"""lgcp forward (K = 128) per batch size on both forms of the d = 1600 path: the 32-row launch sequence (variant 1) and the
wide-batch GEMM launches (variant 2); `auto` is the library's rule (wide from 128 particles).
    python tools/probes/lgcp_time.py [n ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cmcd_amd import synthetic, _lib
from cmcd_amd import mcdboundingmachine as mcdbm
counts = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts)
K = b["params_fixed"][1]
D, IN = 1600, 1620
flop_eval = 2.0 * (D * IN + IN * IN + IN * D + D * D)       # per particle and evaluation
ns = [int(a) for a in sys.argv[1:]] or [20, 32, 64, 128, 256, 600, 2048, 15000]
for n in ns:
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    for variant, tag in ((1, "32-row passes"), (2, "wide batch")):
        if (variant == 1 and n > 2048) or (variant == 2 and n < 32):
            continue
        mcdbm.KERNEL_VARIANT = variant
        f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                        eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
        f(); torch.cuda.synchronize()
        reps = 3 if n <= 2048 else 1
        t = time.perf_counter()
        for _ in range(reps): out = f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / reps
        tf = n * (K + 1) * flop_eval / dt / 1e12
        print("lgcp n=%d K=%d %-14s %8.2f ms per call, %.3e particle-steps/s, %.1f TFLOP/s = %.1f %% of fp32 peak, mean loss %.3f  [%s]"
              % (n, K, tag, dt * 1e3, n * K / dt, tf, 100 * tf / 157.3, float(out[0].mean()), _lib.last_kernel_name()), flush=True)
