"""Analytic identities that pin the oracle without the reference."""
import numpy as np
import pytest

from cmcd_amd import synthetic
from oracle import cmcd_oracle as orc
from oracle import prng

from helpers import oracle_target, run_oracle


@pytest.mark.parametrize("name,over", [
    ("gmm_n300_k8", {}), ("funnel_n300_k64", dict(nbridges=8)),
    ("many_gmm_n2000_k256_dds", dict(nbridges=8)), ("many_gmm_var_n16000_k256", dict(nbridges=4)),
])
def test_reuse_identity(name, over):
    """One evaluation per step carried forward == the reference's two evaluations (bit-identical)."""
    b = synthetic.build(name, device="cpu", **over)
    seeds = synthetic.parity_seeds(24)
    for dt in (np.float32, np.float64):
        l1, z1 = run_oracle(b, seeds, dtype=dt, reuse=True)
        l2, z2 = run_oracle(b, seeds, dtype=dt, reuse=False)
        np.testing.assert_array_equal(l1, l2)
        np.testing.assert_array_equal(z1, z2)


def test_zero_network_reduces_to_ula_ais():
    """factor_sn = 0: the step is plain ULA-AIS (mcd_over_orig.py with an eps schedule), written
    here independently of the oracle's evolve loop."""
    b = synthetic.build("gmm_n300_k8", device="cpu")
    train, _ = b["unflatten"](b["params_flat"])
    train["sn"]["factor_sn"].zero_()
    seeds = synthetic.parity_seeds(40)
    loss, z = run_oracle(b, seeds, dtype=np.float64)

    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    tgt = oracle_target(b["cfg"])
    dim, K = 2, 8
    e0, e = prng.particle_noise(seeds, dim, K)
    mean, std = p["vd"]["mean"], np.exp(p["vd"]["logdiag"])
    x = mean + std * e0
    logq = lambda y: np.sum(-0.5 * ((y - mean) / std) ** 2 - np.log(std) - 0.5 * np.log(2 * np.pi), -1)
    w = -logq(x)
    eps = float(p["eps"])
    for i in range(K):
        beta = (i + 1) / (K + 1)
        gu = lambda y: -(beta * tgt(y)[1] + (1 - beta) * (-(y - mean) / std ** 2))
        fk = x - eps * gu(x)
        xn = fk + np.sqrt(2 * eps) * e[:, i]
        bk = xn - eps * gu(xn)
        w += (-np.sum((x - bk) ** 2, -1) + np.sum((xn - fk) ** 2, -1)) / (4 * eps)
        x = xn
    w += tgt(x)[0]
    np.testing.assert_allclose(loss, -w, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(z, x, rtol=1e-10)


def test_unbiasedness_normalised_target():
    """E[exp(-loss)] = Z = 1 for any network parameters (both kernels are normalised Gaussians)."""
    b = synthetic.build("gmm_n300_k8", device="cpu", init_sigma=2.0, init_eps=0.05)
    seeds = synthetic.throughput_seeds(20000, stream=3)
    loss, _ = run_oracle(b, seeds, dtype=np.float64)
    wts = np.exp(-loss)
    est, se = wts.mean(), wts.std() / np.sqrt(len(wts))
    assert abs(est - 1.0) < 5 * se + 1e-3, (est, se)


def test_p_equals_q_zero_eps_limit():
    """Gaussian target == q, zero net, eps -> 0: every increment vanishes and loss -> 0."""
    class SameAsQ:
        def __call__(self, z):
            return (np.sum(-0.5 * z * z - 0.5 * np.log(2 * np.pi), -1)), -z

    b = synthetic.build("gmm_n300_k8", device="cpu", init_eps=1e-7)
    train, _ = b["unflatten"](b["params_flat"])
    train["sn"]["factor_sn"].zero_()
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    loss, _ = orc.compute_log_elbo_batch(synthetic.parity_seeds(50), p, 2, 8, "MCD_CAIS_sn", "geffner",
                                          SameAsQ(), dtype=np.float64)
    assert np.abs(loss).max() < 1e-5


def test_schedules():
    b = orc.betas_from_grid(np.ones(9), np.linspace(0, 1, 10), np.linspace(0, 1, 10)[1:-1], np.float64)
    np.testing.assert_allclose(b, np.arange(1, 9) / 9.0, rtol=1e-12)          # ones => (i+1)/(K+1)
    e = orc.eps_table(1.0, 4, "cos_sq", np.float64)
    np.testing.assert_allclose(e, np.cos((np.arange(4) / 4 + 0.008) / 1.008 * np.pi / 2) ** 2)
    e = orc.eps_table(0.1, 5, "linear", np.float64)
    np.testing.assert_allclose(e[[0, -1]], [0.1, 1e-4])
    assert np.all(orc.eps_table(0.3, 5, "", np.float64) == 0.3)


def test_unknown_mode_raises():
    b = synthetic.build("gmm_n300_k8", device="cpu")
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    with pytest.raises(NotImplementedError, match="Mode not implemented."):
        orc.compute_log_elbo_batch(np.array([1]), p, 2, 8, "MCD_U_a-lp", "geffner", oracle_target(b["cfg"]))


def test_reductions_and_inf_semantics():
    loss = np.array([1.0, 2.0, np.inf, 0.5])
    s = orc.stats5(loss)
    assert s[0] == 3 and np.isinf(s[1]) and s[3] == -0.5
    assert np.isclose(orc.ln_z(loss), np.log(np.exp(-1) + np.exp(-2) + np.exp(-.5)) - np.log(4))
    e = orc.log_final_losses(np.array([[1.0, 2.0], [3.0, 5.0]]))
    assert np.isclose(e[0], -2.75)


def test_ula_modes_against_cais_and_longhand():
    """MCD_ULA (mcd_over_orig.py, use_sn=False) == MCD_CAIS_sn with a zero network, constant eps, no clip
    (same formulas); MCD_ULA_sn differs from CAIS only by dropping the forward drift and shifting the
    backward index — checked against a longhand loop."""
    b = synthetic.build("gmm_n300_k8", device="cpu")
    seeds = synthetic.parity_seeds(30)
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    tgt = oracle_target(b["cfg"])
    dim, K = 2, 8
    p0 = dict(p)
    p0["sn"] = dict(p["sn"]); p0["sn"]["factor_sn"] = np.zeros(())
    l_cais, z_cais = orc.compute_log_elbo_batch(seeds, p0, dim, K, "MCD_CAIS_sn", "geffner", tgt, dtype=np.float64)
    p_nosn = {k: v for k, v in p.items() if k != "sn"}
    l_ula, z_ula = orc.compute_log_elbo_batch(seeds, p_nosn, dim, K, "MCD_ULA", "geffner", tgt, dtype=np.float64,
                                              eps_schedule="cos_sq", grad_clipping=True)   # both ignored
    np.testing.assert_allclose(l_ula, l_cais, rtol=1e-12)
    np.testing.assert_allclose(z_ula, z_cais, rtol=1e-12)

    l_sn, z_sn = orc.compute_log_elbo_batch(seeds, p, dim, K, "MCD_ULA_sn", "geffner", tgt, dtype=np.float64)
    e0, e = prng.particle_noise(seeds, dim, K)
    mean, std = p["vd"]["mean"], np.exp(p["vd"]["logdiag"])
    x = mean + std * e0
    w = -np.sum(-0.5 * ((x - mean) / std) ** 2 - np.log(std) - 0.5 * np.log(2 * np.pi), -1)
    eps = float(p["eps"])
    for i in range(K):
        beta = (i + 1) / (K + 1)
        gu = lambda y: -(beta * tgt(y)[1] + (1 - beta) * (-(y - mean) / std ** 2))
        fk = x - eps * gu(x)
        xn = fk + np.sqrt(2 * eps) * e[:, i]
        bk = xn - eps * gu(xn) + eps * orc.apply_geffner(p["sn"], xn, i, np.float64)     # index i  (:44)
        w += (-np.sum((x - bk) ** 2, -1) + np.sum((xn - fk) ** 2, -1)) / (4 * eps)
        x = xn
    w += tgt(x)[0]
    np.testing.assert_allclose(l_sn, -w, rtol=1e-9, atol=1e-9)
    assert np.abs(l_sn - l_ula).max() > 1e-4     # the network matters


def test_dds_timestep_coefficients_follow_jnp_linspace():
    """nn_dds.py:108 calls jax.numpy's linspace: float32 `start (1 - s) + stop s`, not NumPy's float64 `start + i step`
    rounded afterwards.  The two differ in 31 of the 64 entries by up to 7.6e-6 (2e-3 rad of phase at t = 256)."""
    from oracle import cmcd_oracle as orc
    c = orc.timestep_coeff()
    assert c.dtype == np.float32 and c.shape == (64,) and c[0] == np.float32(0.1) and c[-1] == np.float32(100.0)
    i = np.arange(64, dtype=np.float64)
    exact = 0.1 + i * (99.9 / 63.0)
    assert np.abs(c - exact).max() < 1.6e-5            # a few float32 ulps at 100
    assert (c != np.linspace(0.1, 100.0, 64).astype(np.float32)).sum() == 31
    # hand-evaluated entries of the float32 formula
    s1 = np.float32(1) / np.float32(63)
    assert c[1] == np.float32(np.float32(np.float32(0.1) * np.float32(np.float32(1) - s1)) + np.float32(np.float32(100) * s1))
