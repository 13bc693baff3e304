"""Vectorised torch-CPU float32 port of the forward bound (`MCD_CAIS_sn` / `MCD_CAIS_var_sn`), for the `cpu_baseline` leg
of bench.py only ("torch-CPU fp32 restatement, vectorised over particles, reference-faithful and
reuse variants, all cores and one thread").  TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (oracle/__init__.py).

Follows the same cited lines as oracle/cmcd_oracle.py (/root/reference/src/mcd_cais.py:46-89, mcdboundingmachine.py:
126-179, nn.py:42-72, nn_dds.py:91-164, model_handler.py:245-281); tests/test_oracle_torch_port.py holds it to the NumPy
restatement.  Targets: many_gmm (the named batch and configs[3]) and gmm; other targets raise NotImplementedError.
The PRNG streams (oracle/prng.py) are drawn before the timed region: what is timed is the arithmetic of the K bridges."""
import math

import numpy as np
import torch

from . import cmcd_oracle as _np_oracle
from . import prng
from .targets import Gmm, ManyGmm

LOG_2PI = math.log(2 * math.pi)


def _target(name, tgt):
    if name == "many_gmm":
        mu = torch.tensor(np.asarray(tgt.means, np.float32))
        s, k, d = float(tgt.scale), tgt.n_mixes, tgt.dim
        c = -d * (math.log(s) + 0.5 * LOG_2PI) - math.log(k)

        def f(z):
            diff = (z[:, None, :] - mu[None]) / s
            logit = -0.5 * (diff * diff).sum(-1) + c
            lp = torch.logsumexp(logit, 1)
            g = -(torch.softmax(logit, 1)[:, :, None] * diff).sum(1) / s
            ok = lp > -1e4
            return torch.where(ok, lp, torch.full_like(lp, -math.inf)), torch.where(ok[:, None], g, torch.zeros_like(g))
        return f
    if name == "gmm":
        means, prec = torch.tensor(tgt.means, dtype=torch.float32), torch.tensor(tgt.prec, dtype=torch.float32)
        logc = torch.tensor(tgt.logc, dtype=torch.float32)

        def raw(x):
            diff = x[:, None, :] - means[None]
            pd = torch.einsum("kij,nkj->nki", prec, diff)
            logit = -0.5 * (diff * pd).sum(-1) + logc[None]
            fv = torch.logsumexp(logit, 1)
            return fv, -(torch.softmax(logit, 1)[:, :, None] * pd).sum(1)

        def f(z):
            fa, ga = raw(z)
            fb, gb = raw(z.flip(-1))
            lse = torch.logaddexp(fa, fb)
            return lse - math.log(2.0), torch.exp(fa - lse)[:, None] * ga + torch.exp(fb - lse)[:, None] * gb.flip(-1)
        return f
    raise NotImplementedError(f"torch port: target {name!r} not covered")


class Prepared:
    """Everything that is not the K-step arithmetic: parameters as float32 tensors, schedules, the PRNG streams."""

    def __init__(self, seeds, params, dim, nbridges, mode, arch, model, target, eps_schedule=None, grad_clipping=False):
        if mode not in ("MCD_CAIS_sn", "MCD_CAIS_var_sn"):
            raise NotImplementedError("torch port: CAIS modes only")
        f32 = lambda a: torch.tensor(np.asarray(a, np.float32))
        self.dim, self.K, self.arch = dim, nbridges, arch
        self.var_mode, self.clip_on = mode == "MCD_CAIS_var_sn", bool(grad_clipping)
        self.sn = {k: f32(v) for k, v in params["sn"].items()}
        self.mean, self.std = f32(params["vd"]["mean"]), torch.exp(f32(params["vd"]["logdiag"]))
        self.betas = f32(_np_oracle.betas_from_grid(params["mgridref_y"], params["gridref_x"], params["target_x"], np.float32))
        self.eps = f32(_np_oracle.eps_table(params["eps"], nbridges, eps_schedule, np.float32))
        e0, noise = prng.particle_noise(np.asarray(seeds), dim, nbridges)
        self.e0, self.noise = torch.from_numpy(e0), torch.from_numpy(noise)
        self.target = _target(model, target)
        if arch == "dds":   # the time path does not depend on the particle: one table per call
            sn = self.sn
            coeff = torch.from_numpy(_np_oracle.timestep_coeff())
            t = torch.arange(nbridges + 1, dtype=torch.float32)[:, None]
            arg = coeff[None] * t + sn["timestep_phase"].reshape(1, -1)
            emb = torch.cat([torch.sin(arg), torch.cos(arg)], 1)
            h = emb @ sn["t_w1"] + sn["t_b1"]
            self.tau = (h * 0.5 * (1 + torch.erf(h / math.sqrt(2)))) @ sn["t_w2"] + sn["t_b2"]


def _net(p, z, i):
    sn = p.sn
    if p.arch == "dds":
        gelu = lambda x: x * 0.5 * (1 + torch.erf(x / math.sqrt(2)))
        x = torch.cat([z, p.tau[i].expand(z.shape[0], -1)], 1)
        h = gelu(gelu(x @ sn["s_w1"] + sn["s_b1"]) @ sn["s_w2"] + sn["s_b2"])
        return torch.clamp(h @ sn["s_w3"] + sn["s_b3"], -1e4, 1e4)
    emb = sn["emb"][min(i, sn["emb"].shape[0] - 1)]
    u = torch.cat([z, emb.expand(z.shape[0], -1)], 1)
    u = u + torch.nn.functional.softplus(u @ sn["W1"] + sn["b1"])
    u = u + torch.nn.functional.softplus(u @ sn["W2"] + sn["b2"])
    return (u @ sn["W3"] + sn["b3"]) * sn["factor_sn"]


@torch.no_grad()
def run(p, reuse=False, max_bridges=None, max_particles=None):
    """-> (loss[N], z[N, dim]) float32.  reuse=False: two network and two gradient evaluations per bridge as the
    reference executes them; reuse=True: one of each (the backward evaluation opens the next bridge).
    max_bridges / max_particles: walk only a prefix of the chain / of the batch (bench.py's bounded timing samples; the
    returned loss is then not the bound)."""
    clip = 1e2 if p.var_mode else 1e3
    iv = 1.0 / (p.std * p.std)

    def ev(z, i):
        _, gp = p.target(z)
        gq = -(z - p.mean) * iv
        if p.clip_on:
            gp = gp.clamp(-clip, clip)
            if p.var_mode:
                gq = gq.clamp(-clip, clip)
        return gp, gq, _net(p, z, i)

    m = p.e0.shape[0] if max_particles is None else min(int(max_particles), p.e0.shape[0])
    e0 = p.e0[:m]
    z = p.std * e0 + p.mean
    w = -(-0.5 * e0 * e0 - torch.log(p.std) - 0.5 * LOG_2PI).sum(-1)
    carried = None
    for i in range(p.K if max_bridges is None else min(int(max_bridges), p.K)):
        beta, eps = p.betas[i], p.eps[i]
        gp, gq, s = carried if (reuse and carried is not None) else ev(z, i)
        fk = z + eps * (beta * gp + (1 - beta) * gq) - eps * s
        sig = torch.sqrt(2 * eps)
        n = p.noise[:m, i, :]
        zn = fk + sig * n
        gp2, gq2, s2 = ev(zn, i + 1)
        bk = zn + eps * (beta * gp2 + (1 - beta) * gq2) + eps * s2
        w = w + (-((z - bk) ** 2).sum(-1) / (4 * eps) + 0.5 * (((zn - fk) / sig) ** 2).sum(-1))
        z, carried = zn, (gp2, gq2, s2)
    w = w + p.target(z)[0]
    return -w, z
