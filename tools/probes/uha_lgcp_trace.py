"""Trace of eps / gamma / their gradients while MCD_CAIS_UHA_sn trains on lgcp (README lgcp flags, lr from LR_DICT)."""
import os, sys, types
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmcd_amd import boundingmachine as bm, mcdboundingmachine as mcdbm, opt
from cmcd_amd.lgcp import load_model_lgcp

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
lr = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-3
eps0 = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-5
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 3000
cfg = types.SimpleNamespace(N=20)
counts = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
target, dim = load_model_lgcp("lgcp", cfg, flat_bin_counts=counts)
dev = torch.device("cuda")
gen = torch.Generator().manual_seed(1)
pf, un, fixed = bm.initialize(dim=dim, nbridges=0, trainable=("vd",), init_sigma=1.0, device=dev)
_, pf, _ = opt.run(cfg, 0.01, 20000, pf, un, fixed, target, bm.grad_and_loss, ("vd",), gen)
vd0 = {k: v.detach().cpu().clone() for k, v in un(pf)[0]["vd"].items()}
trainable = ("eta", "gamma", "eps", "vd", "mgridref_y")
pf, un, fixed = mcdbm.initialize(dim=dim, nbridges=K, vdparams=vd0, eta=0.0, eps=eps0, trainable=trainable,
                                 mode="MCD_CAIS_UHA_sn", emb_dim=20, nlayers=3, nn_arch="geffner", device=dev)
gl, lf = mcdbm.make_grad_and_loss("MCD_CAIS_UHA_sn")
o_eps, o_gam, o_fac = un.offset("eps"), un.offset("gamma"), un.offset("sn", "factor_sn")
optim = opt.create_optimizer(lr, trainable=trainable)
state = optim.init(pf)
for it in range(iters):
    seeds = torch.randint(1, 1000000, (20,), generator=gen, dtype=torch.int32).to(dev)
    g, (l, z) = gl(seeds, pf, un, fixed, target)
    if it % 100 == 0 or it < 5:
        print(it, "loss %.3f" % float(l.mean()), "eps %.3e gamma %.4f factor %.3e" % (float(pf[o_eps]), float(pf[o_gam]), float(pf[o_fac])),
              "g_eps %.3e g_gamma %.3e g_fac %.3e" % (float(g[o_eps]), float(g[o_gam]), float(g[o_fac])), "|g| %.3e" % float(g.norm()),
              "nan" if bool(torch.isnan(g).any()) else "", flush=True)
    optim.step(pf, g, state, un, trainable)
