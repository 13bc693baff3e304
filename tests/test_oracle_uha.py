"""Pins the restatements of `MCD_CAIS_UHA_sn` (2nd-order CMCD, /root/reference/src/mcd_under_lp_a_cais.py:6-115) without the
reference: NumPy float32 / float64 / torch agree; importance weights are unbiased for a normalised target whatever the
network, friction and step size (any error in either kernel's mean, scale or log-ratio breaks this); a longhand
restatement with the network switched off; autograd through the torch twin == finite differences of the NumPy forward."""
import copy

import numpy as np
import pytest

from cmcd_amd import synthetic
from oracle import cmcd_oracle as orc
from oracle import cmcd_oracle_torch as ot
from oracle import prng

from helpers import oracle_target, run_oracle

MODE = "MCD_CAIS_UHA_sn"
CASES = [("gmm_n300_k8", dict(nbridges=4)),
         ("funnel_n300_k64", dict(nbridges=3, init_eps=0.05, init_gamma=4.0)),
         ("many_gmm_n2000_k256_dds", dict(nbridges=4, init_eps=0.2, init_gamma=2.0, init_sigma=10.0))]


def test_network_is_built_with_rho_dim():
    """initialize_network(..., rho_dim=dim) (/root/reference/src/mcdboundingmachine.py:82-98): geffner width
    2 dim + emb_dim (src/nn.py:43), dds first layer [2 dim + 64, 64] (src/nn_dds.py:56,121-123,159)."""
    b = synthetic.build("funnel_n300_k64", device="cpu", boundmode=MODE, nbridges=2)
    spec = b["params_fixed"][3]
    assert (spec.rho_dim, spec.width) == (10, 68)
    assert b["unflatten"].shape("sn", "nn", 0, 0) == (68, 68) and b["unflatten"].shape("sn", "nn", 2, 0) == (68, 10)
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cpu", boundmode=MODE, nbridges=2)
    assert b["unflatten"].shape("sn", "drift_net/~/linear_2", "w") == (2 + 2 + 64, 64)
    assert b["unflatten"].shape("sn", "drift_net/~/linear_zero", "w") == (64, 2)


def test_key_chain_has_the_extra_split_and_normal():
    """/root/reference/src/mcd_under_lp_a_cais.py:92-93,100: evolve's key C yields (R, G') = split(C), rho_0 = normal(R),
    gen_0 = second(split(G')) — one split and one normal more than the overdamped chain (mcd_cais.py:94), whose gen_0
    is second(split(C)); z_0's noise is the same draw in both."""
    seeds = np.arange(1, 6)
    e0, rho0, noise = prng.particle_noise_uha(seeds, 3, 2)
    e0_c, noise_c = prng.particle_noise(seeds, 3, 2)
    np.testing.assert_array_equal(e0, e0_c)
    a, b = prng.split(prng.prng_key(seeds))
    c, _ = prng.split(b)
    r, gp = prng.split(c)
    np.testing.assert_array_equal(rho0, prng.normal(r, 3))
    _, gen = prng.split(gp)
    g, h = prng.split(gen)
    np.testing.assert_array_equal(noise[:, 0], prng.normal(g, 3))
    _, gen = prng.split(h)
    g, _ = prng.split(gen)
    np.testing.assert_array_equal(noise[:, 1], prng.normal(g, 3))
    assert not np.array_equal(noise, noise_c)


@pytest.mark.parametrize("name,over", CASES)
def test_three_restatements_agree(param_set, name, over):
    b = synthetic.build(name, device="cpu", boundmode=MODE, **over)
    seeds = synthetic.parity_seeds(24)
    l64, z64 = run_oracle(b, seeds, dtype=np.float64)
    l32, z32 = run_oracle(b, seeds, dtype=np.float32)
    dim, K, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    _, lt, zt, _ = ot.bound_and_grad(seeds, p, dim, K, mode, spec.arch, b["cfg"]["model"])
    np.testing.assert_allclose(lt, l64, rtol=1e-5, atol=1e-5)    # torch erf / softplus vs scipy / logaddexp
    np.testing.assert_allclose(zt, z64, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(l32, l64, rtol=2e-4, atol=2e-4)
    assert l32.dtype == np.float32 and z32.dtype == np.float32


def test_schedule_and_clip_do_not_follow_the_config():
    """The function body takes neither eps_schedule nor grad_clipping (:6-17): cos^2 (:33-40,48) and clip 1e2 (:23-30,46)
    whatever the descriptor says."""
    seeds = synthetic.parity_seeds(8)
    ref = None
    for sched, clip in (("", False), ("linear", True), ("cos_sq", True)):
        b = synthetic.build("gmm_n300_k8", device="cpu", boundmode=MODE, eps_schedule=sched, grad_clipping=clip)
        l, _ = run_oracle(b, seeds, dtype=np.float64)
        ref = l if ref is None else ref
        np.testing.assert_array_equal(l, ref)


def test_zero_network_longhand():
    """factor_sn = 0: underdamped Langevin AIS with partial momentum refresh, written here without the oracle's loop."""
    b = synthetic.build("gmm_n300_k8", device="cpu", boundmode=MODE, init_eps=0.05, init_gamma=3.0)
    train, _ = b["unflatten"](b["params_flat"])
    train["sn"]["factor_sn"].zero_()
    seeds = synthetic.parity_seeds(30)
    loss, z = run_oracle(b, seeds, dtype=np.float64)
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    tgt = oracle_target(b["cfg"])
    K, d = 8, 2
    e0, rho, n = prng.particle_noise_uha(seeds, d, K)
    e0, rho, n = e0.astype(np.float64), rho.astype(np.float64), n.astype(np.float64)
    mean, std = p["vd"]["mean"], np.exp(p["vd"]["logdiag"])
    x = mean + std * e0
    w = -np.sum(-0.5 * e0 ** 2 - np.log(std) - 0.5 * np.log(2 * np.pi), -1)
    w += 0.5 * np.sum(rho ** 2, -1)                          # - log N(rho_0; 0, 1) up to the constant that returns at the end
    for i in range(K):
        beta = (i + 1) / (K + 1)
        eps = float(p["eps"]) * np.cos((i / K + 0.008) / 1.008 * np.pi / 2) ** 2
        eta = float(p["gamma"]) * eps
        gu = lambda y: -(beta * np.clip(tgt(y)[1], -100, 100) + (1 - beta) * (-(y - mean) / std ** 2))
        rp = rho * (1 - eta) + np.sqrt(2 * eta) * n[:, i]
        w += (-np.sum((rho - rp * (1 - eta)) ** 2, -1) + 2 * eta * np.sum(n[:, i] ** 2, -1)) / (4 * eta)
        rpp = rp - eps * gu(x) / 2
        x = x + eps * rpp
        rho = rpp - eps * gu(x) / 2
    w += -0.5 * np.sum(rho ** 2, -1) + tgt(x)[0]
    np.testing.assert_allclose(loss, -w, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(z, x, rtol=1e-10)


@pytest.mark.parametrize("over", [dict(init_eps=0.05, init_gamma=6.0), dict(init_eps=0.1, init_gamma=1.5, nbridges=5)])
def test_unbiasedness_normalised_target(over):
    """E[exp(-loss)] = Z = 1 on the extended space (z, rho) for ANY network, gamma, eps: the forward kernel is a normalised
    Gaussian in rho' followed by a volume-preserving leap-frog, the backward kernel a normalised Gaussian in rho."""
    b = synthetic.build("gmm_n300_k8", device="cpu", boundmode=MODE, init_sigma=2.0, dense=True, **over)
    seeds = synthetic.throughput_seeds(20000, stream=5)
    loss, _ = run_oracle(b, seeds, dtype=np.float64)
    wts = np.exp(-loss)
    est, se = wts.mean(), wts.std() / np.sqrt(len(wts))
    assert abs(est - 1.0) < 5 * se + 1e-3, (est, se)
    assert se < 0.1     # ... and the check has power: the weights are not degenerate


def _value(b, p, seeds):
    dim, K, mode, spec = b["params_fixed"]
    loss, _ = orc.compute_log_elbo_batch(seeds, p, dim, K, mode, spec.arch, oracle_target(b["cfg"]), dtype=np.float64)
    return loss.mean()


@pytest.mark.parametrize("name,over", CASES)
def test_gradient_matches_finite_differences(param_set, name, over):
    """No stop_gradient in this mode: autograd through the torch twin must be the derivative of the NumPy forward value
    (every trainable leaf: network, eps, gamma, q, the beta grid)."""
    b = synthetic.build(name, device="cpu", boundmode=MODE, **over)
    seeds = synthetic.parity_seeds(12)
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    dim, K, mode, spec = b["params_fixed"]
    _, _, _, g = ot.bound_and_grad(seeds, p, dim, K, mode, spec.arch, b["cfg"]["model"])
    rng = np.random.default_rng(0)
    last = "s_w3" if spec.arch == "dds" else "W3"
    first = "s_w1" if spec.arch == "dds" else "W1"
    for paths in ([("sn", last)], [("sn", first)], [("eps",)], [("gamma",)], [("vd", "mean"), ("vd", "logdiag")],
                  [("mgridref_y",)]):
        direction, analytic = [], 0.0
        for path in paths:
            node_g, node_p = g, p
            for k in path:
                node_g, node_p = node_g[k], node_p[k]
            d = rng.standard_normal(np.shape(node_p))
            direction.append((path, d))
            analytic += float(np.sum(node_g * d))

        def shifted(sign, h=1e-5):
            q = copy.deepcopy(p)
            for path, d in direction:
                node = q
                for k in path[:-1]:
                    node = node[k]
                node[path[-1]] = node[path[-1]] + sign * h * d
            return _value(b, q, seeds)
        fd = (shifted(+1) - shifted(-1)) / 2e-5
        assert abs(fd - analytic) <= 3e-4 * max(1.0, abs(fd)), (paths, fd, analytic)
