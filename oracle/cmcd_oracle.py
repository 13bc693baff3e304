"""CPU restatement of the CMCD ``MCD_CAIS_sn`` / ``MCD_CAIS_var_sn`` bound (and the sibling modes
``MCD_ULA``, ``MCD_ULA_sn``, ``MCD_CAIS_UHA_sn``).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED: the
reference cannot be imported here and holds no tests; this file follows the
cited lines and is pinned by PRNG known answers + analytic identities only.

Vectorised over particles with NumPy; ``dtype`` float32 mirrors the reference's
arithmetic type, float64 is the high-precision check.  The PRNG streams are
float32/uint32 in both (oracle/prng.py).

Parameter dict (same keys as cmcd_amd's ``unflatten`` output):
  vd: {mean[d], logdiag[d]}; eps: scalar; mgridref_y[G+1]; gridref_x[G+2]; target_x[K]
  gamma: scalar (``MCD_CAIS_UHA_sn`` only; that mode's nets take [z; rho]: s_w1[2d+64,64], in = 2d + e)
  sn (dds):     timestep_phase[1,64], t_w1[128,64], t_b1[64], t_w2[64,64], t_b2[64],
                s_w1[d+64,64], s_b1[64], s_w2[64,64], s_b2[64], s_w3[64,d], s_b3[d]
  sn (geffner): emb[K,e], factor_sn, W1[in,in], b1[in], W2[in,in], b2[in], W3[in,d], b3[d]
"""
import math

import numpy as np

try:  # exact erf for GELU (/root/reference/src/nn_dds.py:176 uses jax.scipy.special.erf)
    from scipy.special import erf as _erf
except Exception:  # pragma: no cover
    _erf = np.vectorize(math.erf)

from . import prng

LOG_2PI = 1.8378770664093453
MODES = ("MCD_CAIS_sn", "MCD_CAIS_var_sn", "MCD_ULA", "MCD_ULA_sn", "MCD_CAIS_UHA_sn")


# --------------------------------------------------------------------------- schedules
def betas_from_grid(mgridref_y, gridref_x, target_x, dtype):
    """/root/reference/src/mcdboundingmachine.py:146-149."""
    m = np.asarray(mgridref_y, dtype)
    gy = np.cumsum(m) / np.sum(m)
    gy = np.concatenate([np.zeros(1, dtype), gy])
    return np.interp(np.asarray(target_x, dtype), np.asarray(gridref_x, dtype), gy).astype(dtype)


def eps_table(eps0, nbridges, schedule, dtype):
    """/root/reference/src/mcd_cais.py:34-44,54-59."""
    dt = np.dtype(dtype).type
    i = np.arange(nbridges).astype(dtype)
    e0 = dt(eps0)
    if schedule == "cos_sq":
        phase = i / dt(nbridges)
        decay = np.cos((phase + dt(0.008)) / dt(1.008) * dt(0.5) * dt(np.pi)) ** 2
        return (e0 * decay).astype(dtype)
    if schedule == "linear":
        return ((dt(0.0001) - e0) / dt(nbridges - 1) * i + e0).astype(dtype)
    return np.full(nbridges, e0, dtype)


# --------------------------------------------------------------------------- q
def q_sample(vd, noise):
    """/root/reference/src/vardist/diag_gauss.py:49-62."""
    return np.exp(vd["logdiag"]) * noise + vd["mean"]


def q_log_prob(vd, z):
    """/root/reference/src/vardist/diag_gauss.py:26-33 (numpyro Normal.log_prob)."""
    dt = z.dtype.type
    s = np.exp(vd["logdiag"])
    return np.sum(-((z - vd["mean"]) ** 2) / (dt(2.0) * s * s) - np.log(s) - dt(0.5 * LOG_2PI), -1)


def q_grad(vd, z):
    s = np.exp(vd["logdiag"])
    return -(z - vd["mean"]) / (s * s)


# --------------------------------------------------------------------------- nets
def gelu(x):
    """/root/reference/src/nn_dds.py:167-176."""
    dt = x.dtype.type
    return x * dt(0.5) * (dt(1.0) + _erf(x / np.sqrt(dt(2.0))).astype(x.dtype))


def softplus(x):
    """stax.Softplus = logaddexp(x, 0)  (/root/reference/src/nn.py:46)."""
    return np.logaddexp(x, x.dtype.type(0.0))


def timestep_coeff():
    """`np.linspace(start=0.1, stop=100, num=64)` of /root/reference/src/nn_dds.py:108 is jax.numpy's: float32
    arithmetic, `start * (1 - step) + stop * step` with `step = iota(63) / 63`, the end point appended exactly."""
    step = np.arange(63, dtype=np.float32) / np.float32(63)
    out = np.float32(0.1) * (np.float32(1) - step) + np.float32(100) * step
    return np.concatenate([out, np.array([100.0], np.float32)]).astype(np.float32)


def dds_time_embedding(sn, t, dtype):
    """/root/reference/src/nn_dds.py:108,131-143,155-158.  t: integer bridge index."""
    dt = np.dtype(dtype).type
    coeff = timestep_coeff()
    arg = coeff * np.float32(t) + np.asarray(sn["timestep_phase"], np.float32).reshape(-1)
    arg = arg.astype(np.float32).astype(np.float64)                 # fp32-rounded argument
    emb = np.concatenate([np.sin(arg), np.cos(arg)]).astype(dtype)
    h = gelu(emb @ sn["t_w1"] + sn["t_b1"])
    return (h @ sn["t_w2"] + sn["t_b2"]).astype(dtype)


def apply_dds(sn, z, t, dtype):
    """PISNet.__call__  /root/reference/src/nn_dds.py:145-164."""
    dt = np.dtype(dtype).type
    tau = dds_time_embedding(sn, t, dtype)
    x = np.concatenate([z, np.broadcast_to(tau, (z.shape[0], tau.shape[0]))], 1)
    h = gelu(x @ sn["s_w1"] + sn["s_b1"])
    h = gelu(h @ sn["s_w2"] + sn["s_b2"])
    out = h @ sn["s_w3"] + sn["s_b3"]
    return np.clip(out, dt(-1e4), dt(1e4))


def apply_geffner(sn, z, i, dtype):
    """/root/reference/src/nn.py:42-72; out-of-range i clamps like a JAX gather."""
    nb = sn["emb"].shape[0]
    emb = sn["emb"][min(max(int(i), 0), nb - 1)]
    u = np.concatenate([z, np.broadcast_to(emb, (z.shape[0], emb.shape[0]))], 1)
    u = u + softplus(u @ sn["W1"] + sn["b1"])
    u = u + softplus(u @ sn["W2"] + sn["b2"])
    return (u @ sn["W3"] + sn["b3"]) * sn["factor_sn"]


def apply_sn(arch, sn, z, i, dtype):
    return apply_dds(sn, z, i, dtype) if arch == "dds" else apply_geffner(sn, z, i, dtype)


# --------------------------------------------------------------------------- kernels
def log_prob_kernel(x, mean, scale):
    """/root/reference/src/mcd_utils.py:19-21."""
    dt = x.dtype.type
    return np.sum(-((x - mean) ** 2) / (dt(2.0) * scale * scale) - np.log(scale) - dt(0.5 * LOG_2PI), -1)


def cast_params(params, dtype):
    if isinstance(params, dict):
        return {k: cast_params(v, dtype) for k, v in params.items()}
    return np.asarray(params, dtype)


# --------------------------------------------------------------------------- the bound
def compute_log_elbo_batch(seeds, params, dim, nbridges, mode, arch, target,
                           eps_schedule=None, grad_clipping=False, dtype=np.float32, reuse=False):
    """Per-particle loss and final sample.  /root/reference/src/mcdboundingmachine.py:126-179
    with the evolve loop of /root/reference/src/mcd_cais.py:46-96 (``MCD_CAIS_sn``) or
    /root/reference/src/mcd_cais_var.py:57-112 (``MCD_CAIS_var_sn``; forward values differ
    only through the clip rule :33-40).

    reuse=False evaluates grad/net twice per step like the reference; reuse=True carries the
    backward evaluation into the next step's forward kernel (bit-identical).
    Returns (loss[N], z[N, dim]) in ``dtype``.
    """
    if mode not in MODES:
        raise NotImplementedError("Mode not implemented.")
    if mode == "MCD_CAIS_UHA_sn":
        return compute_log_elbo_batch_uha(seeds, params, dim, nbridges, arch, target, dtype=dtype)
    dt = np.dtype(dtype).type
    p = cast_params(params, dtype)
    vd, sn = p["vd"], p.get("sn")
    seeds = np.asarray(seeds)
    eps0_noise, noise = prng.particle_noise(seeds, dim, nbridges)
    betas = betas_from_grid(p["mgridref_y"], p["gridref_x"], p["target_x"], dtype) if nbridges >= 1 else None
    eps_tab = eps_table(p["eps"], nbridges, eps_schedule, dtype) if nbridges >= 1 else None

    z = q_sample(vd, eps0_noise.astype(dtype))
    w = -q_log_prob(vd, z)

    var_mode = mode == "MCD_CAIS_var_sn"
    clip = dt(1e2) if var_mode else dt(1e3)
    # MCD_ULA / MCD_ULA_sn: /root/reference/src/mcd_over_orig.py:6-65 via mcd_utils.py:35-58 — constant eps,
    # no clipping (the dispatcher does not even pass eps_schedule / grad_clipping), no network in the
    # forward kernel, and the backward kernel's network (ULA_sn only) takes index i, not i + 1.
    ula = mode in ("MCD_ULA", "MCD_ULA_sn")
    if ula:
        grad_clipping = False
        eps_tab = np.full(nbridges, dt(p["eps"]), dtype) if nbridges >= 1 else None
        reuse = False

    def grads(zz):
        _, gp = target(zz)
        gq = q_grad(vd, zz)
        if grad_clipping:
            gp = np.clip(gp, -clip, clip)
            if var_mode:
                gq = np.clip(gq, -clip, clip)
        return gp, gq

    def grad_u(g, beta):
        gp, gq = g
        return dt(-1.0) * (beta * gp + (dt(1.0) - beta) * gq)

    carried = None
    for i in range(nbridges):
        beta, eps = betas[i], eps_tab[i]
        if reuse and carried is not None:
            g_z, s_z = carried
        else:
            g_z, s_z = grads(z), (None if ula else apply_sn(arch, sn, z, i, dtype))
        uf = grad_u(g_z, beta)
        if ula:
            fk_mean = z - eps * uf
        else:
            fk_mean = z - eps * uf - eps * s_z
        scale = np.sqrt(dt(2.0) * eps)
        z_new = fk_mean + scale * noise[:, i, :].astype(dtype)
        if mode == "MCD_ULA":
            g_n, s_n = grads(z_new), None
        else:
            g_n, s_n = grads(z_new), apply_sn(arch, sn, z_new, i if ula else i + 1, dtype)
        ub = grad_u(g_n, beta)
        bk_mean = z_new - eps * ub if s_n is None else z_new - eps * ub + eps * s_n
        w = w + (log_prob_kernel(z, bk_mean, scale) - log_prob_kernel(z_new, fk_mean, scale))
        z = z_new
        carried = (g_n, s_n)
    logp, _ = target(z)
    w = w + logp
    return (dt(-1.0) * w).astype(dtype), z.astype(dtype)


def compute_log_elbo_batch_uha(seeds, params, dim, nbridges, arch, target, dtype=np.float32):
    """``MCD_CAIS_UHA_sn`` — second-order (underdamped) CMCD.  /root/reference/src/mcdboundingmachine.py:126-179 with
    the evolve loop of /root/reference/src/mcd_under_lp_a_cais.py:6-115: state (z, rho), score network on
    ``concat(z, rho)`` with the SAME time index i in the forward and the backward kernel (:51-54,77-80), momentum
    refresh with ``eta_aux = gamma * eps`` (:50), one leap-frog step (:59-66), the cos^2 step-size schedule always on
    (:33-40,48) and ``grad log p`` always clipped at 1e2 (``stable=True``, :23-30,46,64) — the function takes neither
    ``eps_schedule`` nor ``grad_clipping``.  (The reference's dispatcher passes both keywords,
    /root/reference/src/mcd_utils.py:174-188, which this signature would reject; the body is what is restated.)
    Network built with ``rho_dim = dim`` (/root/reference/src/mcdboundingmachine.py:82-98, src/nn.py:42-43,
    src/nn_dds.py:55-56): geffner input width 2 d + emb_dim, dds first layer [2 d + 64, 64]; output width d.
    Returns (loss[N], z[N, dim]) in ``dtype``."""
    dt = np.dtype(dtype).type
    p = cast_params(params, dtype)
    vd, sn = p["vd"], p["sn"]
    seeds = np.asarray(seeds)
    e0, rho0, noise = prng.particle_noise_uha(seeds, dim, nbridges)
    betas = betas_from_grid(p["mgridref_y"], p["gridref_x"], p["target_x"], dtype)
    eps_tab = eps_table(p["eps"], nbridges, "cos_sq", dtype)          # :33-40,48
    gamma = dt(p["gamma"])
    clip = dt(1e2)

    z = q_sample(vd, e0.astype(dtype))
    w = -q_log_prob(vd, z)                                            # mcdboundingmachine.py:157
    rho = rho0.astype(dtype)                                          # :93
    zero = np.zeros_like(rho)
    w = w - log_prob_kernel(rho, zero, dt(1.0))                       # :96-97

    def grad_u(zz, beta):                                             # :23-30
        _, gp = target(zz)
        gq = q_grad(vd, zz)
        return dt(-1.0) * (beta * np.clip(gp, -clip, clip) + (dt(1.0) - beta) * gq)

    for i in range(nbridges):
        beta, eps = betas[i], eps_tab[i]
        uf = grad_u(z, beta)                                          # :46
        eta_aux = gamma * eps                                         # :50
        s_old = apply_sn(arch, sn, np.concatenate([z, rho], 1), i, dtype)
        fk_rho_mean = rho * (dt(1.0) - eta_aux) - dt(2.0) * eta_aux * s_old          # :52-54
        scale = np.sqrt(dt(2.0) * eta_aux)                            # :56
        rho_prime = fk_rho_mean + scale * noise[:, i, :].astype(dtype)               # :58-59
        rho_pp = rho_prime - eps * uf / dt(2.0)                       # :62
        z_new = z + eps * rho_pp                                      # :63
        ub = grad_u(z_new, beta)                                      # :65
        rho_new = rho_pp - eps * ub / dt(2.0)                         # :67
        s_new = apply_sn(arch, sn, np.concatenate([z, rho_prime], 1), i, dtype)      # :77-80: old z, new momentum
        bk_rho_mean = rho_prime * (dt(1.0) - eta_aux) + dt(2.0) * eta_aux * s_new
        w = w + (log_prob_kernel(rho, bk_rho_mean, scale) - log_prob_kernel(rho_prime, fk_rho_mean, scale))  # :83-88
        z, rho = z_new, rho_new
    w = w + log_prob_kernel(rho, zero, dt(1.0))                       # :112
    logp, _ = target(z)
    w = w + logp                                                      # mcdboundingmachine.py:178
    return (dt(-1.0) * w).astype(dtype), z.astype(dtype)


def compute_bound(seeds, params, dim, nbridges, mode, arch, target, **kw):
    """/root/reference/src/mcdboundingmachine.py:183-205 -> (mean, (losses, z))."""
    loss, z = compute_log_elbo_batch(seeds, params, dim, nbridges, mode, arch, target, **kw)
    with np.errstate(invalid="ignore"):
        return loss.mean(), (loss, z)


def compute_bound_var(seeds, params, dim, nbridges, mode, arch, target, **kw):
    """/root/reference/src/mcdboundingmachine.py:208-231 -> (clip(var, +-1e7), (losses, z))."""
    loss, z = compute_log_elbo_batch(seeds, params, dim, nbridges, mode, arch, target, **kw)
    with np.errstate(invalid="ignore"):
        return np.clip(loss.var(ddof=0), -1e7, 1e7), (loss, z)


def mfvi_losses(seeds, vd, dim, target, dtype=np.float64):
    """Mean-field VI bound, per seed: /root/reference/src/boundingmachine.py:73-111 with nbridges = 0
    (rng_key, _ = split(PRNGKey(seed)); z = vd.sample_rep(rng_key); w = -vd.log_prob(z) + log_prob(z); -w).
    `vd` = {"mean", "logdiag"} float arrays.  -> (losses[N], z[N, dim])."""
    vdp = {k: np.asarray(v, dtype) for k, v in vd.items()}
    e0, _ = prng.particle_noise(np.asarray(seeds), dim, 0)
    z = q_sample(vdp, e0.astype(dtype))
    logp, _ = target(z)
    return (q_log_prob(vdp, z) - logp).astype(dtype), z


def mfvi_grad(seeds, vd, dim, target, dtype=np.float64):
    """d mean(losses) / d {mean, logdiag} of `mfvi_losses`, i.e. what jax.grad(bm.compute_bound, 1)
    (/root/reference/src/main.py:87-89) returns for the "vd" leaves: with z = mean + std e,
    log q(z) = -|e|^2/2 - sum logdiag - c does not depend on mean, so
    d/d mean = -E[grad log p(z)],  d/d logdiag = -1 - E[grad log p(z) * std e].
    (tests/test_oracle_mfvi.py checks this against finite differences of `mfvi_losses`.)"""
    vdp = {k: np.asarray(v, dtype) for k, v in vd.items()}
    e0, _ = prng.particle_noise(np.asarray(seeds), dim, 0)
    z = q_sample(vdp, e0.astype(dtype))
    _, gp = target(z)
    return {"mean": -gp.mean(0), "logdiag": (-1.0 - gp * (z - vdp["mean"])).mean(0)}


def ln_z(loss):
    """logsumexp(-loss) - log n   (/root/reference/src/utils.py:233-235)."""
    a = -np.asarray(loss, np.float64)
    m = np.max(a)
    if not np.isfinite(m):
        return float(m)
    return float(m + np.log(np.sum(np.exp(a - m))) - np.log(a.shape[0]))


def stats5(loss):
    """[n_finite, sum, sum of squares, max(-loss), sum exp(-loss - max)] in float64 — the
    partial-statistics vector of the C ABI (include/cmcd_hip.h), over FINITE and +inf entries
    exactly as the reductions of mcdboundingmachine.py:205,231 / utils.py:233 see them."""
    l = np.asarray(loss, np.float64)
    a = -l
    m = np.max(a) if l.size else -np.inf
    with np.errstate(invalid="ignore", over="ignore"):
        s = np.sum(np.exp(a - m)) if np.isfinite(m) else 0.0
        return np.array([np.sum(np.isfinite(l)), np.sum(l), np.sum(l * l), m, s])


def log_final_losses(eval_losses):
    """/root/reference/src/utils.py:219-248 -> (elbo, elbo_std, lnZ, lnZ_std)."""
    e = np.asarray(eval_losses, np.float64)
    elbos = -e.mean(1)
    lnzs = np.array([ln_z(r) for r in e])
    return elbos.mean(), elbos.std(), lnzs.mean(), lnzs.std()
