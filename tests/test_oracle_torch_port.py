"""The torch-CPU float32 port timed by bench.py's cpu_baseline leg agrees with the NumPy restatement."""
import numpy as np
import pytest

from cmcd_amd import synthetic
from oracle import torch_port

from helpers import oracle_target, run_oracle


@pytest.mark.parametrize("name,over", [("many_gmm_n2000_k256_dds", dict(nbridges=24)), ("gmm_n300_k8", {}),
                                       ("many_gmm_var_n16000_k256", dict(nbridges=8))])
def test_port_matches_numpy_restatement(param_set, name, over):
    b = synthetic.build(name, device="cpu", **over)
    seeds = synthetic.parity_seeds(200)
    dim, K, mode, spec = b["params_fixed"]
    p = torch_port.Prepared(seeds, synthetic.oracle_params(b["unflatten"], b["params_flat"]), dim, K, mode, spec.arch,
                            b["cfg"]["model"], oracle_target(b["cfg"]), b["cfg"]["eps_schedule"], b["cfg"]["grad_clipping"])
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64)
    for reuse in (False, True):
        l, z = torch_port.run(p, reuse=reuse)
        l, z = l.numpy().astype(np.float64), z.numpy().astype(np.float64)
        assert np.array_equal(np.isinf(l), np.isinf(l_ref))
        f = np.isfinite(l_ref)
        rel = np.abs(l[f] - l_ref[f]) / np.maximum(1.0, np.abs(l_ref[f]))
        assert np.quantile(rel, 0.99) < 5e-3 and abs(l[f].mean() - l_ref[f].mean()) < 2e-3 * max(1.0, abs(l_ref[f].mean()))
        assert np.quantile(np.abs(z - z_ref)[f], 0.99) < 5e-2
