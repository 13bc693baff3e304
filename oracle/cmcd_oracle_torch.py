"""Differentiable restatement of the bound (torch float64 + autograd) — the oracle for GRADIENTS.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the reference obtains its
gradients from `jax.grad(compute_bound_fn, 1, has_aux=True)` (/root/reference/src/main.py:174-176),
which cannot be run here.  This file restates the forward path in torch with the reference's
`stop_gradient` placement — `MCD_CAIS_var_sn` detaches `z` at the top of every step and `z_new` right
after sampling (/root/reference/src/mcd_cais_var.py:59,79), `MCD_CAIS_sn` does not (full
reparameterised gradient, /root/reference/src/mcd_cais.py:46-89) — and lets autograd differentiate it.
Its forward values are checked against oracle/cmcd_oracle.py in tests/test_oracle_grad.py.

Parameters: the same dict layout as oracle/cmcd_oracle.py (see its docstring), as float64 torch
tensors with requires_grad where a gradient is wanted.  Targets: log-densities written in torch so
that the full `MCD_CAIS_sn` gradient (through grad log p) is available too.
"""
import math

import numpy as np
import torch

from . import cmcd_oracle as _np_oracle
from . import prng
from .targets import many_gmm_means

LOG_2PI = math.log(2 * math.pi)


# --------------------------------------------------------------------------- targets (log p only)
def logp_gmm(z):
    means = torch.tensor([[3.0, 0.0], [-2.5, 0.0], [2.0, 3.0]], dtype=z.dtype)
    covs = torch.tensor([[[0.7, 0.0], [0.0, 0.05]], [[0.7, 0.0], [0.0, 0.05]], [[1.0, 0.95], [0.95, 1.0]]],
                        dtype=z.dtype)

    def raw(x):
        comp = torch.distributions.MultivariateNormal(means, covariance_matrix=covs)
        return torch.logsumexp(comp.log_prob(x[:, None, :]) + math.log(1 / 3), dim=1)

    return torch.logaddexp(raw(z), raw(z.flip(-1))) - math.log(2.0)


def logp_funnel(z):
    v = z[:, 0]
    d1 = z.shape[1] - 1
    return (-0.5 * LOG_2PI - math.log(3.0) - v * v / 18.0 - 0.5 * d1 * LOG_2PI - 0.5 * d1 * v
            - 0.5 * torch.exp(-v) * (z[:, 1:] ** 2).sum(-1))


def logp_many_gmm(z, n_mixes=40, loc_scaling=40.0):
    mu = torch.tensor(np.asarray(many_gmm_means(n_mixes, 2, loc_scaling), np.float64), dtype=z.dtype)
    s = math.log1p(math.exp(0.1))
    comp = (-0.5 * ((z[:, None, :] - mu) / s) ** 2 - math.log(s) - 0.5 * LOG_2PI).sum(-1)
    lp = torch.logsumexp(comp - math.log(mu.shape[0]), dim=1)
    return torch.where(lp > -1e4, lp, torch.full_like(lp, -math.inf))   # model_handler.py:279-280


def make_logp_lgcp(flat_bin_counts, m=40):
    """Log-Gaussian Cox process on an m x m grid, un-whitened (/root/reference/src/model_handler.py:304-396,
    /root/reference/src/cp_utils.py:45-155), from the constants of oracle.targets.Lgcp (float64)."""
    from .targets import Lgcp
    t = Lgcp(flat_bin_counts, m)
    kinv = torch.tensor(t.kinv)
    counts = torch.tensor(t.counts)

    def logp(z):
        r = z - t.mu0
        return -0.5 * ((r @ kinv) * r).sum(-1) + (z * counts - t.a * torch.exp(z)).sum(-1) + t.lognorm
    return logp


TARGETS = {"gmm": logp_gmm, "funnel": logp_funnel, "many_gmm": logp_many_gmm}


def grad_logp(logp_fn, z, create_graph):
    """jax.grad(log_prob_model)(z) per particle; zero where log p is floored to -inf."""
    zz = z if z.requires_grad else z.detach().requires_grad_(True)
    lp = logp_fn(zz)
    fin = torch.isfinite(lp)
    (g,) = torch.autograd.grad(torch.where(fin, lp, torch.zeros_like(lp)).sum(), zz, create_graph=create_graph)
    return torch.where(fin[:, None], g, torch.zeros_like(g))


# --------------------------------------------------------------------------- nets
def gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def apply_dds(sn, z, t):
    coeff = torch.tensor(_np_oracle.timestep_coeff().astype(np.float64), dtype=z.dtype)
    arg = coeff * float(t) + sn["timestep_phase"].reshape(-1)
    emb = torch.cat([torch.sin(arg), torch.cos(arg)])
    tau = gelu(emb @ sn["t_w1"] + sn["t_b1"]) @ sn["t_w2"] + sn["t_b2"]
    x = torch.cat([z, tau.expand(z.shape[0], -1)], 1)
    h = gelu(x @ sn["s_w1"] + sn["s_b1"])
    h = gelu(h @ sn["s_w2"] + sn["s_b2"])
    return torch.clamp(h @ sn["s_w3"] + sn["s_b3"], -1e4, 1e4)


def apply_geffner(sn, z, i):
    nb = sn["emb"].shape[0]
    emb = sn["emb"][min(max(int(i), 0), nb - 1)]
    u = torch.cat([z, emb.expand(z.shape[0], -1)], 1)
    u = u + torch.nn.functional.softplus(u @ sn["W1"] + sn["b1"])
    u = u + torch.nn.functional.softplus(u @ sn["W2"] + sn["b2"])
    return (u @ sn["W3"] + sn["b3"]) * sn["factor_sn"]


# --------------------------------------------------------------------------- schedules
def betas_from_grid(m, K):
    """interp(target_x, gridref_x, [0, cumsum(m)/sum(m)]) with both grids uniform
    (/root/reference/src/mcdboundingmachine.py:107-118,146-149), differentiable in m."""
    G = m.shape[0] - 1
    gy = torch.cat([torch.zeros(1, dtype=m.dtype), torch.cumsum(m, 0) / m.sum()])
    x = torch.arange(1, K + 1, dtype=m.dtype) / (K + 1)
    pos = x * (G + 1)
    j = torch.clamp(torch.floor(pos).long() + 1, 1, G + 1)
    frac = pos - (j - 1).to(m.dtype)
    return gy[j - 1] + frac * (gy[j] - gy[j - 1])


def eps_table(eps0, K, schedule):
    i = torch.arange(K, dtype=eps0.dtype)
    if schedule == "cos_sq":
        return eps0 * torch.cos((i / K + 0.008) / 1.008 * 0.5 * math.pi) ** 2
    if schedule == "linear":
        return (0.0001 - eps0) / (K - 1) * i + eps0
    return eps0 * torch.ones(K, dtype=eps0.dtype)


def to_torch(params, requires_grad=True):
    if isinstance(params, dict):
        return {k: to_torch(v, requires_grad) for k, v in params.items()}
    t = torch.tensor(np.asarray(params, np.float64), dtype=torch.float64)
    return t.requires_grad_(requires_grad)


def losses(seeds, p, dim, nbridges, mode, arch, target_name, eps_schedule=None, grad_clipping=False):
    """Per-particle losses [N] (float64), differentiable wrt the leaves of `p`."""
    if mode == "MCD_CAIS_UHA_sn":
        return losses_uha(seeds, p, dim, nbridges, arch, target_name)
    var_mode = mode == "MCD_CAIS_var_sn"
    # MCD_ULA / MCD_ULA_sn (/root/reference/src/mcd_over_orig.py:6-65 via mcd_utils.py:35-58): constant eps, no
    # clipping, no network in the forward kernel; the backward kernel's network (ULA_sn only) takes index i
    ula = mode in ("MCD_ULA", "MCD_ULA_sn")
    if ula:
        grad_clipping, eps_schedule = False, None
    logp_fn = TARGETS[target_name] if isinstance(target_name, str) else target_name   # or a log-density callable
    e0, noise = prng.particle_noise(np.asarray(seeds), dim, nbridges)
    e0 = torch.tensor(e0.astype(np.float64))
    noise = torch.tensor(noise.astype(np.float64))
    vd, sn = p["vd"], p.get("sn")
    std = torch.exp(vd["logdiag"])
    betas = betas_from_grid(p["mgridref_y"], nbridges)
    eps_tab = eps_table(p["eps"], nbridges, eps_schedule)
    apply = apply_dds if arch == "dds" else apply_geffner

    def log_q(z):
        return (-((z - vd["mean"]) ** 2) / (2 * std * std) - torch.log(std) - 0.5 * LOG_2PI).sum(-1)

    def log_kernel(x, mean, scale):
        return (-((x - mean) ** 2) / (2 * scale * scale) - torch.log(scale) - 0.5 * LOG_2PI).sum(-1)

    clip = 1e2 if var_mode else 1e3

    def grad_u(z, beta):
        gq = -(z - vd["mean"]) / (std * std)           # jax.grad of vd.log_prob wrt z
        gp = grad_logp(logp_fn, z, create_graph=not var_mode)
        if grad_clipping:
            gp = torch.clamp(gp, -clip, clip)
            if var_mode:
                gq = torch.clamp(gq, -clip, clip)
        return -1.0 * (beta * gp + (1.0 - beta) * gq)

    z = std * e0 + vd["mean"]                          # reparameterised sample
    w = -log_q(z)
    for i in range(nbridges):
        beta, eps = betas[i], eps_tab[i]
        if var_mode:
            z = z.detach()                             # mcd_cais_var.py:59
        uf = grad_u(z, beta)
        fk = z - eps * uf if ula else z - eps * uf - eps * apply(sn, z, i)
        scale = torch.sqrt(2 * eps)
        z_new = fk + scale * noise[:, i, :]
        if var_mode:
            z_new = z_new.detach()                     # mcd_cais_var.py:79
        ub = grad_u(z_new, beta)
        if mode == "MCD_ULA":
            bk = z_new - eps * ub
        else:
            bk = z_new - eps * ub + eps * apply(sn, z_new, i if ula else i + 1)
        w = w + log_kernel(z, bk, scale) - log_kernel(z_new, fk, scale)
        z = z_new
    w = w + logp_fn(z)
    return -w, z


def losses_uha(seeds, p, dim, nbridges, arch, target_name):
    """``MCD_CAIS_UHA_sn`` (/root/reference/src/mcd_under_lp_a_cais.py:6-115 under mcdboundingmachine.py:126-179), no
    stop_gradient anywhere: the gradient is the full reparameterised one through (z, rho).  Same statement as
    oracle/cmcd_oracle.py:compute_log_elbo_batch_uha (cos^2 schedule always on, clip 1e2 on grad log p only, both
    network calls at time index i on concat(z, rho) / concat(z, rho'))."""
    logp_fn = TARGETS[target_name] if isinstance(target_name, str) else target_name
    e0, rho0, noise = prng.particle_noise_uha(np.asarray(seeds), dim, nbridges)
    e0 = torch.tensor(e0.astype(np.float64))
    rho = torch.tensor(rho0.astype(np.float64))
    noise = torch.tensor(noise.astype(np.float64))
    vd, sn = p["vd"], p["sn"]
    std = torch.exp(vd["logdiag"])
    betas = betas_from_grid(p["mgridref_y"], nbridges)
    eps_tab = eps_table(p["eps"], nbridges, "cos_sq")
    gamma = p["gamma"]
    apply = apply_dds if arch == "dds" else apply_geffner

    def log_q(z):
        return (-((z - vd["mean"]) ** 2) / (2 * std * std) - torch.log(std) - 0.5 * LOG_2PI).sum(-1)

    def log_kernel(x, mean, scale):
        return (-((x - mean) ** 2) / (2 * scale * scale) - torch.log(scale) - 0.5 * LOG_2PI).sum(-1)

    def grad_u(z, beta):
        gq = -(z - vd["mean"]) / (std * std)
        gp = torch.clamp(grad_logp(logp_fn, z, create_graph=True), -1e2, 1e2)
        return -1.0 * (beta * gp + (1.0 - beta) * gq)

    one = torch.ones((), dtype=torch.float64)
    z = std * e0 + vd["mean"]
    w = -log_q(z)
    w = w - log_kernel(rho, torch.zeros_like(rho), one)
    for i in range(nbridges):
        beta, eps = betas[i], eps_tab[i]
        uf = grad_u(z, beta)
        eta_aux = gamma * eps
        fk = rho * (1.0 - eta_aux) - 2.0 * eta_aux * apply(sn, torch.cat([z, rho], 1), i)
        scale = torch.sqrt(2.0 * eta_aux)
        rho_prime = fk + scale * noise[:, i, :]
        rho_pp = rho_prime - eps * uf / 2.0
        z_new = z + eps * rho_pp
        ub = grad_u(z_new, beta)
        rho_new = rho_pp - eps * ub / 2.0
        bk = rho_prime * (1.0 - eta_aux) + 2.0 * eta_aux * apply(sn, torch.cat([z, rho_prime], 1), i)
        w = w + log_kernel(rho, bk, scale) - log_kernel(rho_prime, fk, scale)
        z, rho = z_new, rho_new
    w = w + log_kernel(rho, torch.zeros_like(rho), one)
    w = w + logp_fn(z)
    return -w, z


def bound_and_grad(seeds, params_np, dim, nbridges, mode, arch, target_name, eps_schedule=None, grad_clipping=False):
    """value = var(losses, ddof=0) for MCD_CAIS_var_sn, mean(losses) otherwise; grads = d value / d leaf
    as a dict with the layout of `params_np` (what jax.grad(compute_bound_fn, 1) returns, leaf by leaf)."""
    p = to_torch(params_np)
    l, z = losses(seeds, p, dim, nbridges, mode, arch, target_name, eps_schedule, grad_clipping)
    value = l.var(unbiased=False) if mode == "MCD_CAIS_var_sn" else l.mean()
    leaves = []

    def collect(d, prefix=()):
        for k, v in d.items():
            if isinstance(v, dict):
                collect(v, prefix + (k,))
            else:
                leaves.append((prefix + (k,), v))
    collect(p)
    gs = torch.autograd.grad(value, [v for _, v in leaves], allow_unused=True)
    grads = {}
    for (path, v), g in zip(leaves, gs):
        d = grads
        for k in path[:-1]:
            d = d.setdefault(k, {})
        d[path[-1]] = np.zeros(tuple(v.shape)) if g is None else g.detach().numpy()
    return float(value.detach()), l.detach().numpy(), z.detach().numpy(), grads
