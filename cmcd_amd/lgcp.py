"""Log-Gaussian Cox process target (d = 1600) — host-side constants and the call into the library.

Mirrors /root/reference/src/model_handler.py:287-409 and /root/reference/src/cp_utils.py:16-155
(un-whitened posterior, `config.use_whitened = False`, /root/reference/src/configs/base.py:134).
The reference evaluates the Gaussian prior through two triangular solves with the Cholesky factor
and differentiates with jax.grad; here the prior precision K^-1 is formed once on the host in
float64 (cond(K) ~ 28) and shipped as float32, so that grad log p = -K^-1 (x - mu) + counts - a e^x
is one dense matrix product per bridge on the GPU.
"""
import math
import os

import numpy as np
from .model_handler import Target, _cfg

M_GRID = 40


def get_bin_counts(points, num_bins_per_dim=M_GRID):
    """/root/reference/src/cp_utils.py:16-42: points on the upper edge go into the last bin."""
    counts = np.zeros((num_bins_per_dim, num_bins_per_dim))
    for elem in np.asarray(points, np.float64) * num_bins_per_dim:
        r, c = int(math.floor(elem[0])), int(math.floor(elem[1]))
        r -= r == num_bins_per_dim
        c -= c == num_bins_per_dim
        counts[r, c] += 1
    return counts.reshape(-1)


def lgcp_constants(flat_bin_counts, m=M_GRID, signal_variance=1.91, beta=1.0 / 33):
    """-> float32 vector {Kinv[d,d], counts[d], mu0, a, lognorm}, d = m*m
    (model_handler.py:304-346)."""
    d = m * m
    idx = np.array([(i, j) for i in range(m) for j in range(m)], np.float64)       # cp_utils.py:45-50
    dist = np.sqrt(((idx[:, None, :] - idx[None]) ** 2).sum(-1))
    gram = signal_variance * np.exp(-dist / (m * beta))                               # cp_utils.py:81-84
    chol = np.linalg.cholesky(gram)
    kinv = np.linalg.inv(gram)
    kinv = 0.5 * (kinv + kinv.T)
    lognorm = -0.5 * d * math.log(2 * math.pi) - np.sum(np.log(np.abs(np.diag(chol))))  # :341-345
    mu0 = math.log(126.0) - 0.5 * signal_variance                                     # :346
    a = 1.0 / d                                                                       # :321
    return np.concatenate([kinv.reshape(-1), np.asarray(flat_bin_counts, np.float64).reshape(-1),
                           [mu0, a, lognorm]]).astype(np.float32)


def load_model_lgcp(model="lgcp", config=None, flat_bin_counts=None):
    """/root/reference/src/model_handler.py:405-409 -> (log_prob_model, dim).  The point set comes
    from `config.file_path` (pines.csv, as in the reference) unless bin counts are passed in."""
    if flat_bin_counts is None:
        path = _cfg(config, "file_path", None)
        if not path or not os.path.exists(path):
            raise ValueError("Please specify a path in config for the Finnish pines data csv.")
        flat_bin_counts = get_bin_counts(np.genfromtxt(path, delimiter=","))
    consts = lgcp_constants(flat_bin_counts)
    return Target("lgcp", M_GRID * M_GRID, consts), M_GRID * M_GRID
