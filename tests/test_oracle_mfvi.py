"""CPU checks of the mean-field VI restatement (oracle.cmcd_oracle.mfvi_*): consistency with the
nbridges = 0 limit of the MCD restatement and the closed-form gradient against finite differences."""
import numpy as np
import pytest

from oracle import cmcd_oracle as orc
from oracle import targets as otg

from helpers import lgcp_counts_fixture


def _target(name):
    if name == "gmm":
        return otg.Gmm(), 2
    if name == "funnel":
        return otg.Funnel(10), 10
    if name == "many_gmm":
        return otg.ManyGmm(), 2
    return otg.Lgcp(lgcp_counts_fixture()), 1600


def _vd(name, dim):
    rng = np.random.default_rng(3)
    if name == "lgcp":
        return {"mean": np.full(dim, np.log(126.0) - 0.955) + 0.05 * rng.standard_normal(dim),
                "logdiag": np.full(dim, np.log(0.5)) + 0.05 * rng.standard_normal(dim)}
    sig = 15.0 if name == "many_gmm" else 1.0
    return {"mean": 0.3 * rng.standard_normal(dim), "logdiag": np.log(sig) + 0.1 * rng.standard_normal(dim)}


@pytest.mark.parametrize("name", ["gmm", "funnel", "many_gmm"])
def test_mfvi_is_the_zero_bridge_limit(name):
    """boundingmachine.compute_log_elbo with nbridges = 0 and the MCD machine with no steps draw the same z
    from the same key and add the same two terms."""
    target, dim = _target(name)
    vd = _vd(name, dim)
    seeds = np.arange(1, 40, dtype=np.int32)
    l, z = orc.mfvi_losses(seeds, vd, dim, target)
    p = {"vd": vd, "eps": np.float64(0.1), "mgridref_y": np.ones(1), "gridref_x": np.linspace(0, 1, 2),
         "target_x": np.zeros(0)}
    l2, z2 = orc.compute_log_elbo_batch(seeds, p, dim, 0, "MCD_ULA", "dds", target, dtype=np.float64)
    np.testing.assert_array_equal(z, z2)
    np.testing.assert_allclose(l, l2, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("name", ["gmm", "funnel", "many_gmm", "lgcp"])
def test_mfvi_gradient_matches_finite_differences(name):
    target, dim = _target(name)
    vd = _vd(name, dim)
    seeds = np.arange(1, 13 if name == "lgcp" else 65, dtype=np.int32)
    g = orc.mfvi_grad(seeds, vd, dim, target)
    rng = np.random.default_rng(0)
    coords = range(dim) if dim <= 10 else rng.choice(dim, 5, replace=False)
    h = 1e-5
    for leaf in ("mean", "logdiag"):
        for j in coords:
            vp = {k: v.copy() for k, v in vd.items()}
            vm = {k: v.copy() for k, v in vd.items()}
            vp[leaf][j] += h
            vm[leaf][j] -= h
            fd = (orc.mfvi_losses(seeds, vp, dim, target)[0].mean() - orc.mfvi_losses(seeds, vm, dim, target)[0].mean()) / (2 * h)
            assert abs(fd - g[leaf][j]) <= 1e-6 * max(1.0, abs(fd)), (leaf, j, fd, g[leaf][j])
