"""A small configuration's training iterations through cmcd_amd.opt.run (the replicate flags' shape), for rocprofv3
--kernel-trace --stats: which launches make up an iteration.  argv: config [mode] [iters] [nbridges]"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic, opt
from cmcd_amd import mcdboundingmachine as mcdbm
name = sys.argv[1] if len(sys.argv) > 1 else "funnel_n300_k64"
mode = sys.argv[2] if len(sys.argv) > 2 else "MCD_CAIS_sn"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 8
b = synthetic.build(name, device="cuda", boundmode=mode, nbridges=K)
gl, _ = mcdbm.make_grad_and_loss(mode, eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
torch.cuda.synchronize()
t0 = time.perf_counter()
opt.run(types.SimpleNamespace(N=b["cfg"]["N"]), 1e-3, iters, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"], gl,
        ("eps", "vd", "mgridref_y"), 1)
torch.cuda.synchronize()
print("SMALL_TRAIN", name, mode, "K", K, iters, "%.1f us per iteration" % ((time.perf_counter() - t0) / iters * 1e6))
