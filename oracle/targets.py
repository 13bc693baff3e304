"""Oracle targets: log p(z) and closed-form grad log p(z).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED (see oracle/__init__.py).  Each target follows the cited
reference lines; the reference differentiates with ``jax.grad``
(/root/reference/src/mcd_cais.py:24-30), here the gradients are closed forms
(checked against torch.autograd in tests/test_oracle_targets.py).

All functions are vectorised over particles: z is [N, d], dtype float32 or
float64; returns (logp[N], grad[N, d]) in the same dtype.
"""
import numpy as np

from . import prng

LOG_2PI = 1.8378770664093453


def _logsumexp(a, axis):
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0)
    with np.errstate(divide="ignore"):
        return (np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m).squeeze(axis)


class Gmm:
    """2-D symmetrised 3-component mixture.  /root/reference/src/model_handler.py:157-200."""
    name = "gmm"
    dim = 2

    def __init__(self):
        means = np.array([[3.0, 0.0], [-2.5, 0.0], [2.0, 3.0]])
        covs = np.array([[[0.7, 0.0], [0.0, 0.05]],
                         [[0.7, 0.0], [0.0, 0.05]],
                         [[1.0, 0.95], [0.95, 1.0]]])
        self.means = means
        self.prec = np.linalg.inv(covs)
        l = np.linalg.cholesky(covs)
        # normalizing_term (:185-187) + log weight (:178)
        self.logc = -LOG_2PI - np.log(np.diagonal(l, axis1=-2, axis2=-1)).sum(1) + np.log(1.0 / 3.0)

    def _raw(self, x, dt):
        means = self.means.astype(dt)
        prec = self.prec.astype(dt)
        diff = x[:, None, :] - means[None]                    # [N,3,2]
        pd = np.einsum("kij,nkj->nki", prec, diff)           # Sigma^-1 (x - mu)
        logit = dt(-0.5) * np.sum(diff * pd, -1) + self.logc.astype(dt)[None]
        f = _logsumexp(logit, 1)
        r = np.exp(logit - f[:, None])
        g = -np.sum(r[:, :, None] * pd, 1)
        return f, g

    def __call__(self, z):
        dt = z.dtype.type
        fa, ga = self._raw(z, dt)
        zf = z[:, ::-1]
        fb, gb = self._raw(zf, dt)
        m = np.maximum(fa, fb)
        lse = m + np.log(np.exp(fa - m) + np.exp(fb - m))
        logp = lse - dt(np.log(2.0))                          # :192-195
        wa = np.exp(fa - lse)[:, None]
        wb = np.exp(fb - lse)[:, None]
        grad = wa * ga + wb * gb[:, ::-1]
        return logp.astype(z.dtype), grad.astype(z.dtype)


class Funnel:
    """Neal's funnel, scale of v hard-coded 3.0.  /root/reference/src/model_handler.py:124-143."""
    name = "funnel"

    def __init__(self, dim=10):
        self.dim = dim

    def __call__(self, z):
        dt = z.dtype.type
        d1 = self.dim - 1
        v = z[:, 0]
        x = z[:, 1:]
        ss = np.sum(x * x, 1)
        with np.errstate(over="ignore", invalid="ignore"):
            emv = np.exp(-v)
            logp = (dt(-0.5 * LOG_2PI - np.log(3.0)) - v * v / dt(18.0)
                    + dt(-0.5 * d1 * LOG_2PI) - dt(0.5 * d1) * v - dt(0.5) * emv * ss)
            g = np.empty_like(z)
            g[:, 0] = -v / dt(9.0) - dt(0.5 * d1) + dt(0.5) * emv * ss
            g[:, 1:] = -x * emv[:, None]
        return logp.astype(z.dtype), g


def many_gmm_means(n_mixes=40, dim=2, loc_scaling=40.0, seed=0):
    """/root/reference/src/model_handler.py:255-261 (uniform from PRNGKey(seed))."""
    key = prng.prng_key(np.array(seed))
    return prng.uniform(key, (n_mixes, dim), -1.0, 1.0) * np.float32(loc_scaling)


class ManyGmm:
    """40-mode mixture with the -1e4 floor.  /root/reference/src/model_handler.py:245-281."""
    name = "many_gmm"

    def __init__(self, n_mixes=40, dim=2, loc_scaling=40.0, log_var_scaling=0.1):
        self.dim = dim
        self.n_mixes = n_mixes
        self.means = many_gmm_means(n_mixes, dim, loc_scaling).astype(np.float64)
        # named `var` in the reference but passed as *scale* (:262-266)
        self.scale = float(np.log1p(np.exp(log_var_scaling)))

    def __call__(self, z):
        dt = z.dtype.type
        mu = self.means.astype(z.dtype)
        s = dt(self.scale)
        diff = (z[:, None, :] - mu[None]) / s                 # [N,K,d]
        logit = (dt(-0.5) * np.sum(diff * diff, -1)
                 - dt(self.dim * (np.log(self.scale) + 0.5 * LOG_2PI)) - dt(np.log(self.n_mixes)))
        logp = _logsumexp(logit, 1)
        r = np.exp(logit - logp[:, None])
        g = -np.sum(r[:, :, None] * diff, 1) / s
        valid = logp > dt(-1e4)                               # :279-280
        logp = np.where(valid, logp, dt(-np.inf))
        g = np.where(valid[:, None], g, dt(0.0))              # grad of the constant branch
        return logp.astype(z.dtype), g.astype(z.dtype)


def lgcp_bin_counts(points, m=40):
    """/root/reference/src/cp_utils.py:16-42."""
    counts = np.zeros((m, m))
    for elem in np.asarray(points) * m:
        r, c = int(np.floor(elem[0])), int(np.floor(elem[1]))
        r -= r == m
        c -= c == m
        counts[r, c] += 1
    return counts.reshape(-1)


class Lgcp:
    """Log-Gaussian Cox process on a 40x40 grid (un-whitened).

    /root/reference/src/model_handler.py:304-396, /root/reference/src/cp_utils.py:45-155.
    The reference does two triangular solves through autodiff; here the prior
    precision K^-1 is formed once in float64.
    """
    name = "lgcp"

    def __init__(self, flat_bin_counts, m=40):
        self.dim = m * m
        self.counts = np.asarray(flat_bin_counts, np.float64)
        idx = np.array([(i, j) for i in range(m) for j in range(m)], np.float64)  # cp_utils.py:45-50
        dist = np.sqrt(((idx[:, None, :] - idx[None]) ** 2).sum(-1))
        self.gram = 1.91 * np.exp(-dist / (m * (1.0 / 33)))                        # model_handler.py:325-335
        chol = np.linalg.cholesky(self.gram)
        self.kinv = np.linalg.inv(self.gram)
        self.kinv = 0.5 * (self.kinv + self.kinv.T)
        self.lognorm = -0.5 * self.dim * LOG_2PI - np.sum(np.log(np.abs(np.diag(chol))))  # :341-345
        self.mu0 = np.log(126.0) - 0.5 * 1.91                                       # :346
        self.a = 1.0 / self.dim                                                     # :321

    def __call__(self, z):
        dt = z.dtype.type
        kinv = self.kinv.astype(z.dtype)
        c = self.counts.astype(z.dtype)
        r = z - dt(self.mu0)
        kr = r @ kinv
        ez = np.exp(z)
        logp = dt(-0.5) * np.sum(r * kr, 1) + dt(self.lognorm) + np.sum(z * c - dt(self.a) * ez, 1)
        g = -kr + c[None] - dt(self.a) * ez
        return logp.astype(z.dtype), g.astype(z.dtype)
