"""Closed-form target gradients of the oracle vs torch.autograd on independently written densities."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import targets

GOLD = os.path.join(os.path.dirname(__file__), "golden")
LOG2PI = math.log(2 * math.pi)


def _autograd(fn, z):
    zt = torch.tensor(z, dtype=torch.float64, requires_grad=True)
    lp = fn(zt)
    (g,) = torch.autograd.grad(lp.sum(), zt)
    return lp.detach().numpy(), g.numpy()


def torch_gmm(z):
    means = torch.tensor([[3.0, 0.0], [-2.5, 0.0], [2.0, 3.0]], dtype=torch.float64)
    covs = torch.tensor([[[0.7, 0.0], [0.0, 0.05]], [[0.7, 0.0], [0.0, 0.05]], [[1.0, 0.95], [0.95, 1.0]]],
                        dtype=torch.float64)

    def raw(x):
        comp = torch.distributions.MultivariateNormal(means, covariance_matrix=covs)
        return torch.logsumexp(comp.log_prob(x[:, None, :]) + math.log(1 / 3), dim=1)

    return torch.logaddexp(raw(z), raw(z.flip(-1))) - math.log(2.0)


def torch_funnel(z):
    v = z[:, 0]
    lp_v = torch.distributions.Normal(torch.zeros((), dtype=torch.float64), torch.tensor(3.0, dtype=torch.float64)).log_prob(v)
    lp_o = torch.distributions.Normal(torch.zeros((), dtype=torch.float64), torch.exp(0.5 * v)[:, None]).log_prob(z[:, 1:]).sum(-1)
    return lp_v + lp_o


def torch_many_gmm(z):
    mu = torch.tensor(np.asarray(targets.many_gmm_means(), np.float64))
    s = math.log1p(math.exp(0.1))
    comp = torch.distributions.Normal(mu, s).log_prob(z[:, None, :]).sum(-1)
    return torch.logsumexp(comp - math.log(40.0), dim=1)


@pytest.mark.parametrize("tgt,fn,scale,d", [
    (targets.Gmm(), torch_gmm, 2.5, 2),
    (targets.Funnel(10), torch_funnel, 1.5, 10),
    (targets.ManyGmm(), torch_many_gmm, 30.0, 2),
])
def test_grad_matches_autograd(tgt, fn, scale, d):
    z = np.random.default_rng(0).normal(0, scale, (200, d))
    lp, g = tgt(z)
    lp_t, g_t = _autograd(fn, z)
    np.testing.assert_allclose(lp, lp_t, rtol=1e-10, atol=1e-10)
    np.testing.assert_allclose(g, g_t, rtol=1e-8, atol=1e-9)


def test_many_gmm_floor():
    tgt = targets.ManyGmm()
    z = np.array([[1000.0, 1000.0], [0.0, 0.0]])
    lp, g = tgt(z)
    assert lp[0] == -np.inf and np.all(g[0] == 0)        # model_handler.py:279-280
    assert np.isfinite(lp[1])


def test_normalised_targets_integrate_to_one():
    # ln Z = 0 for gmm: 2-D quadrature
    xs = np.linspace(-9, 9, 721)
    xx, yy = np.meshgrid(xs, xs, indexing="ij")
    z = np.stack([xx.ravel(), yy.ravel()], 1)
    lp, _ = targets.Gmm()(z)
    assert abs(np.exp(lp).sum() * (xs[1] - xs[0]) ** 2 - 1.0) < 2e-3


def test_lgcp_against_cholesky_formulation():
    counts = np.load(os.path.join(GOLD, "lgcp_bin_counts.npy"))
    assert counts.shape == (1600,) and counts.sum() == 127
    tgt = targets.Lgcp(counts)
    assert abs(-tgt.lognorm - 800 * LOG2PI - 225.7055) < 1e-3     # sum log L_ii, SURVEY A.5
    rng = np.random.default_rng(1)
    z = tgt.mu0 + 0.3 * rng.normal(size=(3, 1600))

    def fn(zt):  # the reference's formulation: whiten with the Cholesky factor (cp_utils.py:153)
        chol = torch.linalg.cholesky(torch.tensor(tgt.gram))
        white = torch.linalg.solve_triangular(chol, (zt - tgt.mu0).T, upper=False).T
        prior = -0.5 * (white * white).sum(-1) - 800 * LOG2PI - torch.log(torch.diagonal(chol)).sum()
        c = torch.tensor(tgt.counts)
        return prior + (zt * c - tgt.a * torch.exp(zt)).sum(-1)

    lp, g = tgt(z)
    lp_t, g_t = _autograd(fn, z)
    np.testing.assert_allclose(lp, lp_t, rtol=1e-9)
    np.testing.assert_allclose(g, g_t, rtol=1e-6, atol=1e-7)
