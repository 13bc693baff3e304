"""2nd-order mode, small-batch gradient path (Jacobian launch -> per-particle scan -> chunked sweep) against the whole-chain
sweep: the adjoint state (lz, lr, arpp) entering every point, as each path carries / loads it, and the gradients."""
import ctypes as C
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm

name = sys.argv[1] if len(sys.argv) > 1 else "many_gmm_n2000_k256_dds"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 96
K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
over = dict(init_eps=0.2, init_gamma=2.0, init_sigma=15.0) if "many" in name else dict(init_gamma=3.0)
b = synthetic.build(name, device="cuda", boundmode="MCD_CAIS_UHA_sn", nbridges=K, dense=True, **over)
D = b["params_fixed"][0]
seeds = torch.from_numpy(synthetic.parity_seeds(n)).cuda()
args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
L = _lib.lib()
out = {}
for mode in ("0", "1"):
    os.environ["CMCD_GRAD_ITEM"] = mode
    x = torch.zeros(K + 1, 3 * D, n, device="cuda")
    L.cmcd_debug_uha_xdump(C.c_void_p(x.data_ptr()))
    g, (losses, z) = mcdbm.compute_bound_grad(*args)
    torch.cuda.synchronize()
    L.cmcd_debug_uha_xdump(C.c_void_p(0))
    out[mode] = (x.cpu().numpy(), g.double().cpu().numpy(), losses.cpu().numpy())
xa, ga, la = out["0"]
xb, gb, lb = out["1"]
print("losses equal", np.array_equal(la, lb), "finite", int(np.isfinite(la).sum()), "of", n)
fin = np.isfinite(la)
sc = np.abs(xa[:, :, fin]).max()
for e in (K, K - 1, K - 2, K // 2, 1, 0):
    d = np.abs(xa[e][:, fin] - xb[e][:, fin]).max(axis=1)
    print("X_%d max abs diff per component" % e, np.array2string(d, precision=3), "scale", np.abs(xa[e][:, fin]).max())
print("state: max abs diff over all e", np.abs(xa[:, :, fin] - xb[:, :, fin]).max(), "scale", sc)
print("grad: max abs diff", np.abs(ga - gb).max(), "scale", np.abs(ga).max(), "cos", float(ga @ gb / np.linalg.norm(ga) / np.linalg.norm(gb)))
for mode in ("0", "1"):
    os.environ["CMCD_GRAD_ITEM"] = mode
    for _ in range(3):
        mcdbm.compute_bound_grad(*args)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        mcdbm.compute_bound_grad(*args)
    torch.cuda.synchronize()
    print("mode", mode, "value + gradient %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
