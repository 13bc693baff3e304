"""The reference driver's flow (/root/reference/src/main.py:76-230, without W&B / plotting) on the HIP path:
  1. mean-field pre-training of q          bm.initialize(nbridges=0) + opt.run(bm.grad_and_loss)     main.py:82-109
  2. MCD machine from the pre-trained q    mcdbm.initialize(vdparams=vdparams_init, ...)               main.py:138-159
  3. training                              opt.run(grad_and_loss) — MCD_CAIS_sn: reparameterised gradient,
                                           MCD_CAIS_var_sn: VarGrad                                   main.py:161-214
  4. evaluation                            utils.sample + log_final_losses (n_input_dist_seeds x n_samples)
Flags follow configs/base.py names.  Example (the README many_gmm run, shortened):
  python tools/train.py --model many_gmm --boundmode MCD_CAIS_sn --N 2000 --nbridges 256 --nn_arch dds \
      --init_sigma 60 --init_eps 1.0 --eps_schedule cos_sq --grad_clipping --iters 2000 --lr 1e-3
"""
import argparse, os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import boundingmachine as bm
from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import opt, utils
from cmcd_amd.model_handler import load_model

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="many_gmm")
ap.add_argument("--boundmode", default="MCD_CAIS_sn")
ap.add_argument("--N", type=int, default=300)
ap.add_argument("--nbridges", type=int, default=8)
ap.add_argument("--nn_arch", default="geffner")
ap.add_argument("--emb_dim", type=int, default=20)
ap.add_argument("--init_sigma", type=float, default=1.0)
ap.add_argument("--init_eps", type=float, default=0.01)
ap.add_argument("--eps_schedule", default="")
ap.add_argument("--grad_clipping", action="store_true")
ap.add_argument("--pretrain_mfvi", action="store_true")
ap.add_argument("--mfvi_iters", type=int, default=2000)
ap.add_argument("--mfvi_lr", type=float, default=1e-2)
ap.add_argument("--train_eps", action="store_true")
ap.add_argument("--train_vi", action="store_true")
ap.add_argument("--train_betas", action="store_true")
ap.add_argument("--iters", type=int, default=2000)
ap.add_argument("--lr", type=float, default=1e-3)
ap.add_argument("--n_samples", type=int, default=500)
ap.add_argument("--n_input_dist_seeds", type=int, default=30)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--file_path", default="", help="lgcp: pines.csv (default: the bin-count fixture under tests/golden)")
cfg = ap.parse_args()

# under torchrun (one process per GPU): particles of every iteration are sharded over the ranks, gradients all-reduced
import torch.distributed as dist
WORLD, RANK = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
if "RANK" in os.environ and "MASTER_PORT" in os.environ:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
if RANK != 0:
    sys.stdout = open(os.devnull, "w")

if "lgcp" in cfg.model and not cfg.file_path:
    # the Finnish-pines point set is reference content; its 40 x 40 bin counts ship as a test fixture
    import numpy as np
    from cmcd_amd.lgcp import load_model_lgcp
    res = load_model_lgcp(cfg.model, cfg, flat_bin_counts=np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy")))
else:
    res = load_model(cfg.model, cfg)
log_prob_model, dim = res[0], res[1]
gen = torch.Generator().manual_seed(cfg.seed)
eval_seeds = torch.randint(1, 1000000, (cfg.n_samples * cfg.n_input_dist_seeds,), generator=torch.Generator().manual_seed(cfg.seed + 1),
                           dtype=torch.int32).cuda()

# 1. mean-field q
flat, unflatten, fixed = bm.initialize(dim=dim, nbridges=0, trainable=("vd",), init_sigma=cfg.init_sigma, device="cuda")
if cfg.pretrain_mfvi:
    t = time.time()
    losses, flat, _ = opt.run(cfg, cfg.mfvi_lr, cfg.mfvi_iters, flat, unflatten, fixed, log_prob_model, bm.grad_and_loss,
                              ("vd",), gen, log_prefix="pretrain")
    torch.cuda.synchronize()
    elbo_init = -sum(losses[-500:]) / len(losses[-500:])
    print("Done training initial parameters, got ELBO %.2f.  (%.2f ms/iter)" % (elbo_init, (time.time() - t) / cfg.mfvi_iters * 1e3))
vdparams_init = {k: v.detach().cpu().clone() for k, v in unflatten(flat)[0]["vd"].items()}

# 2. MCD machine
trainable = ("eta", "gamma")
if cfg.train_eps:
    trainable += ("eps",)
if cfg.train_vi:
    trainable += ("vd",)
if cfg.train_betas:
    trainable += ("mgridref_y",)
print(f"Params being trained : {trainable}")
flat, unflatten, fixed = mcdbm.initialize(dim=dim, nbridges=cfg.nbridges, vdparams=vdparams_init, eps=cfg.init_eps,
                                          trainable=trainable, mode=cfg.boundmode, emb_dim=cfg.emb_dim,
                                          nn_arch=cfg.nn_arch, device="cuda")
grad_and_loss, loss_fn = mcdbm.make_grad_and_loss(cfg.boundmode, eps_schedule=cfg.eps_schedule, grad_clipping=cfg.grad_clipping)
if WORLD > 1:
    from cmcd_amd import parallel
    grad_and_loss = parallel.make_sharded_grad_and_loss(cfg.boundmode, eps_schedule=cfg.eps_schedule,
                                                        grad_clipping=cfg.grad_clipping)


def evaluate(p, tag):
    elbos, _ = utils.sample(cfg, cfg.n_samples, cfg.n_input_dist_seeds, p, unflatten, fixed, log_prob_model, loss_fn, eval_seeds)
    e, z = utils.log_final_losses(elbos.cpu())
    d = utils.log_final_losses.last
    print("%s: ELBO %.4f (+- %.4f)   ln Z %.4f (+- %.4f)" % (tag, e, d["elbo_final_std"], z, d["final_ln_Z_std"]))


evaluate(flat, "before")
# 3. training
t = time.time()
losses, flat, _ = opt.run(cfg, cfg.lr, cfg.iters, flat, unflatten, fixed, log_prob_model, grad_and_loss, trainable, gen)
torch.cuda.synchronize()
dt = time.time() - t
print("%s %s K=%d N=%d: %d iterations in %.1f s (%.2f ms/iter)" % (cfg.model, cfg.boundmode, cfg.nbridges, cfg.N, cfg.iters, dt, dt / cfg.iters * 1e3))
print("recorded mean losses:", ["%.3f" % x for x in losses[:: max(1, len(losses) // 8)]])
# 4. evaluation
evaluate(flat, "after ")
if dist.is_initialized():
    dist.destroy_process_group()
