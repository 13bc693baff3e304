"""Mean-field VI (cmcd_amd.boundingmachine, C ABI cmcd_mfvi_bound_grad) on the GPU vs the float64 restatement."""
import types

import numpy as np
import pytest
import torch

from cmcd_amd import boundingmachine as bm
from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import opt, synthetic
from cmcd_amd.lgcp import load_model_lgcp
from cmcd_amd.model_handler import load_model
from oracle import cmcd_oracle as orc

from helpers import compare_losses, lgcp_counts_fixture, oracle_target

pytestmark = pytest.mark.gpu


def _setup(model, n, seed=3):
    rng = np.random.default_rng(seed)
    if model == "lgcp":
        target, dim = load_model_lgcp("lgcp", None, flat_bin_counts=lgcp_counts_fixture())
        mean = np.full(dim, np.log(126.0) - 0.955) + 0.05 * rng.standard_normal(dim)
        logdiag = np.full(dim, np.log(0.5)) + 0.05 * rng.standard_normal(dim)
        otarget = oracle_target({"model": "lgcp"}, lgcp_counts_fixture())
    else:
        target, dim, _ = load_model(model, types.SimpleNamespace())
        sig = 15.0 if model == "many_gmm" else 1.0
        mean = 0.3 * rng.standard_normal(dim)
        logdiag = np.log(sig) + 0.1 * rng.standard_normal(dim)
        otarget = oracle_target({"model": model})
    vdp = {"mean": torch.tensor(mean, dtype=torch.float32), "logdiag": torch.tensor(logdiag, dtype=torch.float32)}
    flat, unflatten, fixed = bm.initialize(dim=dim, nbridges=0, vdparams=vdp, trainable=("vd",), device="cuda")
    vd64 = {k: v.double().numpy() for k, v in vdp.items()}
    return target, otarget, dim, flat, unflatten, fixed, vd64


@pytest.mark.parametrize("model,n", [("gmm", 300), ("funnel", 301), ("many_gmm", 2000), ("many_gmm", 7), ("lgcp", 29)])
def test_mfvi_bound_and_gradient_match_the_oracle(hip_lib, model, n):
    target, otarget, dim, flat, unflatten, fixed, vd64 = _setup(model, n)
    seeds = synthetic.parity_seeds(n)
    grad, (losses, z) = bm.grad_and_loss(torch.from_numpy(seeds).cuda(), flat, unflatten, fixed, target)
    mean, (losses2, z2) = bm.compute_bound(torch.from_numpy(seeds).cuda(), flat, unflatten, fixed, target)
    torch.cuda.synchronize()
    assert torch.equal(losses, losses2) and torch.equal(z, z2)
    l_ref, z_ref = orc.mfvi_losses(seeds, vd64, dim, otarget)
    compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"mfvi {model}", K=0)
    f = np.isfinite(l_ref)
    if f.all():
        assert abs(float(mean) - l_ref.mean()) <= 1e-3 * max(1.0, abs(l_ref.mean()))
        g_ref = orc.mfvi_grad(seeds, vd64, dim, otarget)
        g = grad.double().cpu().numpy()
        for leaf in ("mean", "logdiag"):
            off = unflatten.offset("vd", leaf)
            a, r = g[off:off + dim], g_ref[leaf]
            assert np.abs(a - r).max() <= 2e-3 * max(np.abs(r).max(), 1e-3), (leaf, np.abs(a - r).max(), np.abs(r).max())
        other = np.ones(g.shape[0], bool)
        for leaf in ("mean", "logdiag"):
            off = unflatten.offset("vd", leaf)
            other[off:off + dim] = False
        assert not g[other].any()


def test_mfvi_draws_the_z0_of_the_mcd_machine(hip_lib):
    """Same key usage (split(PRNGKey(seed))[0] -> sample_rep) in both machines: bit-identical z."""
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", nbridges=1, init_eps=1e-12)
    train, _ = b["unflatten"](b["params_flat"])
    target = b["target"]
    flat, unflatten, fixed = bm.initialize(dim=2, nbridges=0, vdparams={k: v.cpu() for k, v in train["vd"].items()} if "vd" in train
                                           else None, trainable=("vd",), init_sigma=60.0, device="cuda")
    seeds = torch.from_numpy(synthetic.parity_seeds(100)).cuda()
    _, (_, z_mf) = bm.compute_bound(seeds, flat, unflatten, fixed, target)
    l, z, _ = mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], target)
    # one step of size ~1e-12 leaves z_1 = z_0 + sqrt(2e-12) * noise: equal to ~1e-5
    assert float((z - z_mf).abs().max()) < 2e-5


def test_mfvi_pretraining_raises_the_elbo(hip_lib):
    """main.py:82-109: opt.run on bm.compute_bound with trainable = ("vd",)."""
    target, dim, _ = load_model("gmm", types.SimpleNamespace())
    flat, unflatten, fixed = bm.initialize(dim=dim, nbridges=0, trainable=("vd",), init_sigma=1.0, device="cuda")
    fresh = torch.from_numpy(synthetic.throughput_seeds(4000, stream=9)).cuda()
    v0 = float(bm.compute_bound(fresh, flat, unflatten, fixed, target)[0])
    losses, flat2, _ = opt.run(types.SimpleNamespace(N=500), 1e-2, 400, flat, unflatten, fixed, target,
                               bm.grad_and_loss, ("vd",), 0)
    v1 = float(bm.compute_bound(fresh, flat2, unflatten, fixed, target)[0])
    print("mean-field -ELBO", v0, "->", v1)
    assert np.isfinite(v1) and v1 < v0 - 0.3


def test_mfvi_unsupported(hip_lib):
    target, dim, _ = load_model("gmm", types.SimpleNamespace())
    flat, unflatten, fixed = bm.initialize(dim=dim, nbridges=4, trainable=("vd",), device="cuda")
    with pytest.raises(NotImplementedError):
        bm.compute_bound(torch.arange(1, 9, dtype=torch.int32).cuda(), flat, unflatten, fixed, target)
    flat, unflatten, fixed = bm.initialize(dim=dim, nbridges=0, trainable=("vd",), device="cpu")
    with pytest.raises(RuntimeError):
        bm.compute_bound(torch.arange(1, 9, dtype=torch.int32), flat, unflatten, fixed, target)
