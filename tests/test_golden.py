"""The oracle against the committed golden vectors (tests/golden, made by tools/make_golden.py)."""
import ast
import os

import numpy as np
import pytest

from cmcd_amd import synthetic

from helpers import run_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["gmm_k8", "funnel_k64", "many_gmm_dds_k256", "many_gmm_var_k32"]
CASES += ["dense_" + c for c in CASES]


def load_case(tag):
    g = np.load(os.path.join(GOLD, f"oracle_{tag}.npz"))
    return g, str(g["config"]), ast.literal_eval(str(g["overrides"]))


@pytest.mark.parametrize("tag", CASES)
def test_oracle_float64_reproduces_golden(tag):
    g, name, over = load_case(tag)
    b = synthetic.build(name, device="cpu", **over)
    loss, z = run_oracle(b, g["seeds"], dtype=np.float64)
    np.testing.assert_allclose(loss, g["loss"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(z, g["z"], rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_float32_within_tolerance_of_golden(tag):
    """The float32 restatement (the reference's arithmetic type) vs the float64 golden vectors."""
    g, name, over = load_case(tag)
    b = synthetic.build(name, device="cpu", **over)
    loss, _ = run_oracle(b, g["seeds"], dtype=np.float32)
    ref = g["loss"]
    assert np.array_equal(np.isinf(loss), np.isinf(ref))
    f = np.isfinite(ref)
    rel = np.abs(loss[f] - ref[f]) / np.maximum(1, np.abs(ref[f]))
    assert np.quantile(rel, 0.99) < 5e-3 and abs(loss[f].mean() - ref[f].mean()) < 1e-3 * max(1, abs(ref[f].mean()))
