"""Target registry: config.model -> target descriptor for the HIP kernels.

Mirrors the routing of /root/reference/src/model_handler.py:30-43.  The one deliberate difference
from the reference: `log_prob_model` is not an arbitrary traceable callable (that cannot cross
into a HIP kernel) but a `Target` descriptor naming one of the registered densities and carrying
its constant tensors.
"""
import math
import types

import numpy as np
import torch

from . import prng
from ._lib import TARGET

TRACTABLE_DISTS = ["nice", "funnel", "gmm", "many_gmm"]  # /root/reference/src/configs/base.py:74


def _cfg(config, key, default):
    if config is None:
        return default
    if isinstance(config, dict):
        return config.get(key, default)
    return getattr(config, key, default)


class Target:
    """What `log_prob_model` is in this framework: (name, dim, constants)."""

    def __init__(self, name, dim, consts=None, n_mixes=0):
        self.name = name
        self.dim = dim
        self.target_id = TARGET[name]
        self.n_mixes = n_mixes
        self._consts = None if consts is None else torch.as_tensor(np.asarray(consts, np.float32))
        self._on = {}

    def consts_on(self, device):
        if self._consts is None:
            return None
        key = str(device)
        if key not in self._on:
            self._on[key] = self._consts.to(device).contiguous()
        return self._on[key]

    def __call__(self, z):
        raise NotImplementedError(
            "Target descriptors are evaluated inside the HIP trajectory kernel "
            "(cmcd_amd.mcdboundingmachine.compute_bound); there is no host-side log_prob.")

    def __hash__(self):
        return hash((self.name, self.dim, self.n_mixes))

    def __eq__(self, other):
        return isinstance(other, Target) and (self.name, self.dim, self.n_mixes) == (other.name, other.dim, other.n_mixes)

    def __repr__(self):
        return f"Target({self.name!r}, dim={self.dim})"


def _no_sampler(*a, **k):
    raise NotImplementedError("sampling from the target is outside the CMCD hot path")


# Exact samplers of the tractable targets (the third return value of the reference's load_model; main.py:185-190
# draws target samples for the W2 metrics).  `rng` is an int seed or a numpy Generator: the reference's jax key
# streams are not reproduced (these samples only feed a distance between point clouds).
def _gen(rng):
    return rng if isinstance(rng, np.random.Generator) else np.random.default_rng(int(rng))


def _funnel_sampler(d, sig=3.0, clip_y=11.0):
    def sample_data(rng, n_samples):
        """/root/reference/src/model_handler.py:145-152 (including its exp(-y/2) scale)."""
        g = _gen(rng)
        y = np.clip(sig * g.standard_normal((n_samples, 1)), -clip_y, clip_y)
        x = g.standard_normal((n_samples, d - 1)) * np.exp(-y / 2)
        return np.concatenate((y, x), axis=1).astype(np.float32)
    return sample_data


def _gmm_sampler(rng, num_samples):
    """/root/reference/src/model_handler.py:204-229: the 3-component mixture (not its flip-symmetrised density)."""
    g = _gen(rng)
    means = np.array([[3.0, 0.0], [-2.5, 0.0], [2.0, 3.0]])
    covs = np.array([[[0.7, 0.0], [0.0, 0.05]], [[0.7, 0.0], [0.0, 0.05]], [[1.0, 0.95], [0.95, 1.0]]])
    idx = g.integers(0, 3, num_samples)
    chol = np.linalg.cholesky(covs)
    e = g.standard_normal((num_samples, 2))
    return (means[idx] + np.einsum("nij,nj->ni", chol[idx], e)).astype(np.float32)


def _many_gmm_sampler(consts, n_mixes):
    def sample(seed, sample_shape):
        """distrax.MixtureSameFamily(Categorical(uniform), Normal(mean, scale)).sample (model_handler.py:283-284)."""
        g = _gen(seed)
        n = int(np.prod(sample_shape))
        mean = np.asarray(consts[1:], np.float64).reshape(n_mixes, 2)
        idx = g.integers(0, n_mixes, n)
        return (mean[idx] + float(consts[0]) * g.standard_normal((n, 2))).astype(np.float32).reshape(tuple(sample_shape) + (2,))
    return sample


def load_model_funnel(model="funnel", config=None):
    """/root/reference/src/model_handler.py:124-154"""
    d = int(_cfg(config, "funnel_d", 10))
    return Target("funnel", d), d, _funnel_sampler(d, float(_cfg(config, "funnel_sig", 3)), float(_cfg(config, "funnel_clipy", 11)))


def load_model_gmm(model="gmm", config=None):
    """/root/reference/src/model_handler.py:157-242"""
    return Target("gmm", 2), 2, _gmm_sampler


def many_gmm_constants(n_mixes=40, loc_scaling=40.0, log_var_scaling=0.1, seed=0):
    """{scale, means[n_mixes,2]}: /root/reference/src/model_handler.py:255-267.  `var` there is
    softplus(log_var) but is handed to distrax.Normal as *scale*."""
    mean = prng.uniform(seed, (n_mixes, 2), -1.0, 1.0) * np.float32(loc_scaling)
    scale = np.float32(math.log1p(math.exp(log_var_scaling)))
    return np.concatenate([[scale], mean.reshape(-1)]).astype(np.float32)


def load_model_manygmm(model="many_gmm", config=None):
    """/root/reference/src/model_handler.py:245-281"""
    n_mixes = int(_cfg(config, "n_mixes", 40))
    loc_scaling = float(_cfg(config, "loc_scaling", 40))
    consts = many_gmm_constants(n_mixes, loc_scaling)
    return Target("many_gmm", 2, consts, n_mixes=n_mixes), 2, _many_gmm_sampler(consts, n_mixes)


def load_model(model="many_gmm", config=None):
    """Same substring routing order as /root/reference/src/model_handler.py:30-43."""
    if "funnel" in model:
        return load_model_funnel(model, config)
    if "lgcp" in model:
        from .lgcp import load_model_lgcp
        return load_model_lgcp(model, config)
    if "many_gmm" in model:
        return load_model_manygmm(model, config)
    if "gmm" in model:
        return load_model_gmm(model, config)
    raise NotImplementedError(f"model {model!r} is outside the CMCD hot path (SURVEY.md section 8)")
