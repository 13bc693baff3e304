import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def hip_lib():
    from cmcd_amd import build, _lib
    build.build()
    return _lib.lib()


@pytest.fixture(params=["sparse", "dense"])
def param_set(request, monkeypatch):
    """Runs the test once with the measurement inputs of SURVEY.md section 8d (biases / timestep_phase / q mean zero,
    one sigma, uniform mgridref_y) and once with every parameter leaf non-trivial (cmcd_amd.synthetic.build(dense=True)):
    a kernel that dropped a bias, folded the time coder wrongly or ignored per-dimension q scales passes the first."""
    from cmcd_amd import synthetic
    monkeypatch.setattr(synthetic, "DENSE_DEFAULT", request.param == "dense")
    return request.param
