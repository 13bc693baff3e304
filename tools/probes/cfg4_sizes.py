"""config 4's network (many_gmm, MCD_CAIS_var_sn, 132-wide geffner net, K = 256) on the wave-per-tile kernel at several batch sizes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
for n in [int(a) for a in sys.argv[1:]] or [4096, 16000, 32768, 65536, 131072]:
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    mcdbm.KERNEL_VARIANT = 1
    f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                    eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    f(); f(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5): f()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
    print("n = %7d  %s  %.3f ms per launch  %.3e particle-steps/s" % (n, _lib.last_kernel_name(), ms / cnt, n * 256 / (ms / cnt) * 1e3), flush=True)
