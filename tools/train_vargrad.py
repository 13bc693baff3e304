"""End-to-end check of the VarGrad gradient: train a CMCD sampler with the log-variance loss
(`MCD_CAIS_var_sn`) through cmcd_amd.opt.run and report ELBO / ln Z before and after (true ln Z = 0)."""
import argparse, os, sys, time, types
from functools import partial
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import opt, synthetic, utils

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="gmm_n300_k8")
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--N", type=int, default=300)
ap.add_argument("--nbridges", type=int, default=None)
ap.add_argument("--init-sigma", type=float, default=2.0)
ap.add_argument("--emb-dim", type=int, default=None)
ap.add_argument("--init-eps", type=float, default=None)
args = ap.parse_args()
over = dict(boundmode="MCD_CAIS_var_sn", init_sigma=args.init_sigma)
if args.emb_dim:
    over["emb_dim"] = args.emb_dim
if args.init_eps:
    over["init_eps"] = args.init_eps
if args.nbridges:
    over["nbridges"] = args.nbridges
b = synthetic.build(args.config, device="cuda", **over)
# start from the reference's initial network (factor_sn = 0 / zero last layer), not the synthetic weights
flat, unflatten, fixed = mcdbm.initialize(dim=b["params_fixed"][0], nbridges=b["params_fixed"][1],
    vdparams={"mean": torch.zeros(b["params_fixed"][0]), "logdiag": torch.full((b["params_fixed"][0],), float(torch.log(torch.tensor(args.init_sigma))))},
    eps=b["cfg"]["init_eps"], trainable=("eta", "gamma", "mgridref_y"), mode="MCD_CAIS_var_sn",
    emb_dim=b["cfg"]["emb_dim"], nn_arch=b["cfg"]["nn_arch"], device="cuda")
kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
loss_fn = partial(mcdbm.compute_bound, **kw)

def evaluate(p):
    seeds = torch.from_numpy(synthetic.throughput_seeds(30 * 500, stream=99)).cuda()
    elbos, _ = utils.sample(None, 500, 30, p, unflatten, fixed, b["target"], loss_fn, seeds)
    return utils.log_final_losses(elbos.cpu()), dict(utils.log_final_losses.last)

def variance(p):
    seeds = torch.from_numpy(synthetic.throughput_seeds(4000, stream=5)).cuda()
    v, (l, _) = mcdbm.compute_bound_var(seeds, p, unflatten, fixed, b["target"], **kw)
    return float(v), int(torch.isfinite(l).sum())

(e0, z0), d0 = evaluate(flat)
v0 = variance(flat)
t = time.time()
losses, flat2, _ = opt.run(types.SimpleNamespace(N=args.N), args.lr, args.iters, flat, unflatten, fixed, b["target"],
                           partial(mcdbm.compute_log_var_grad, **kw), ("eta", "gamma", "mgridref_y"), 0)
torch.cuda.synchronize()
dt = time.time() - t
(e1, z1), d1 = evaluate(flat2)
print("config %s K=%d N=%d iters=%d lr=%g: %.1f s (%.2f ms/iter)" % (args.config, fixed[1], args.N, args.iters, args.lr, dt, dt / args.iters * 1e3))
v1 = variance(flat2)
print("log-variance loss (4000 fresh particles): before %.4f (finite %d)  after %.4f (finite %d)" % (v0[0], v0[1], v1[0], v1[1]))
print("recorded mean losses:", ["%.3f" % x for x in losses[:: max(1, len(losses) // 8)]])
print("before: ELBO %.4f  lnZ %.4f (+- %.4f)" % (e0, z0, d0["final_ln_Z_std"]))
print("after : ELBO %.4f  lnZ %.4f (+- %.4f)" % (e1, z1, d1["final_ln_Z_std"]))
