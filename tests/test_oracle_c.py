"""Two independent restatements of the reference — NumPy (oracle/cmcd_oracle.py) and plain C
(oracle/cmcd_oracle.c) — must agree; the C one is also the multi-core CPU baseline of bench.py."""
import numpy as np
import pytest

from cmcd_amd import synthetic

from helpers import compare_losses, run_c_oracle, run_oracle


@pytest.mark.parametrize("name,n,over", [
    ("gmm_n300_k8", 64, {}),
    ("funnel_n300_k64", 48, dict(nbridges=16)),
    ("many_gmm_n2000_k256_dds", 96, dict(nbridges=32)),
    ("many_gmm_n2000_k256_dds", 40, dict(nbridges=8, eps_schedule="linear", init_eps=0.05)),
    ("many_gmm_var_n16000_k256", 32, dict(nbridges=8)),
    # 2nd-order CMCD (mcd_under_lp_a_cais.py): network on concat(z, rho), three restatements
    ("gmm_n300_k8", 64, dict(boundmode="MCD_CAIS_UHA_sn")),
    ("funnel_n300_k64", 48, dict(boundmode="MCD_CAIS_UHA_sn", nbridges=8, init_eps=0.05, init_gamma=4.0)),
    ("many_gmm_n2000_k256_dds", 64, dict(boundmode="MCD_CAIS_UHA_sn", nbridges=16, init_eps=0.2, init_gamma=2.0, init_sigma=15.0)),
])
def test_c_oracle_matches_numpy_oracle(param_set, name, n, over):
    b = synthetic.build(name, device="cpu", **over)
    seeds = synthetic.parity_seeds(n)
    lc, zc = run_c_oracle(b, seeds)
    l64, z64 = run_oracle(b, seeds, dtype=np.float64)
    rep = compare_losses(lc, l64, zc, z64, tag=f"C oracle {name}", K=b["params_fixed"][1])
    assert rep["rel_p99"] < 1e-3
    l32, _ = run_oracle(b, seeds, dtype=np.float32, reuse=False)       # same arithmetic type
    f = np.isfinite(l32)
    assert np.array_equal(f, np.isfinite(lc))
    assert np.max(np.abs(lc[f] - l32[f]) / np.maximum(1, np.abs(l32[f]))) < 2e-3


def test_c_oracle_refuses_what_it_does_not_cover():
    from oracle import c_oracle
    from cmcd_amd import _lib as abi
    d = abi.Desc(dim=1600, nbridges=2, mode=0, arch=0, emb_dim=20, target=3, eps_schedule=0, grad_clipping=0,
                 ngrid=2, reserved=0)
    lay = abi.Layout(*([-1] * len(abi.LAYOUT_FIELDS)))
    with pytest.raises(NotImplementedError):
        c_oracle.bound(d, lay, np.array([1], np.int32), np.zeros(4, np.float32), None)
    assert c_oracle.threads() >= 1
