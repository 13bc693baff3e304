"""Particles shard across GPUs; the only exchange is the 5-number statistics vector.

The reference is single-device (no pmap/pjit anywhere under /root/reference/src).  Every particle
trajectory is independent (`jax.vmap` over seeds, /root/reference/src/mcdboundingmachine.py:193-203);
the only cross-particle operations are the batch mean / var (:205,231) and logsumexp
(/root/reference/src/utils.py:233).  So each rank runs the trajectory kernel on a contiguous block
of seeds and one all-gather of `stats[5]` (float64) per call merges the reductions, in rank order
(deterministic regardless of the collective's algorithm).  One process per GPU, RCCL via
torch.distributed (backend "nccl"); the same code runs on gloo for the CPU tests.
"""
import math

import torch
import torch.distributed as dist

NSTATS = 5


def shard_range(n, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`, balanced: n // world particles per rank and one more on the first
    n % world ranks, so a rank is empty only when there are fewer particles than ranks."""
    per, extra = divmod(n, world_size)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def _check_every_rank_has_particles(n, world):
    """Evaluated identically on every rank BEFORE any collective is entered: a ValueError raised on the empty ranks
    only would leave the others waiting in the all-gather / all-reduce."""
    if n < world:
        raise ValueError(f"fewer particles ({n}) than ranks ({world}): every rank needs at least one particle for the "
                         "gradient")


def empty_stats(device=None):
    return torch.tensor([0.0, 0.0, 0.0, -math.inf, 0.0], dtype=torch.float64, device=device)


def merge_stats(stats_rows):
    """Fixed-order merge of [R, 5] statistics rows, no host sync.  On a ROCm device this is ONE launch
    of the library's merge kernel (cmcd_stats_merge_device); on CPU tensors (gloo tests) the same
    arithmetic in torch ops."""
    s = stats_rows
    if s.is_cuda:
        from . import _lib
        s = s.contiguous()
        out = torch.empty(NSTATS, dtype=torch.float64, device=s.device)
        with torch.cuda.device(s.device):
            _lib.check(_lib.lib().cmcd_stats_merge_device(s.data_ptr(), s.shape[0], out.data_ptr(),
                                                           torch.cuda.current_stream().cuda_stream))
        return out
    m = torch.max(s[:, 3])
    scale = torch.where(torch.isfinite(s[:, 3]) & torch.isfinite(m), torch.exp(s[:, 3] - m),
                        (s[:, 3] == m).to(s.dtype))
    out = torch.stack([s[:, 0].sum(), s[:, 1].sum(), s[:, 2].sum(), m, (s[:, 4] * scale).sum()])
    return out


def finalize(stats, n_total):
    """-> dict(mean, var (ddof=0, clipped +-1e7 like compute_bound_var), ln_z, n_finite)."""
    n = float(n_total)
    mean = stats[1] / n
    var = torch.clamp(stats[2] / n - mean * mean, -1e7, 1e7)
    ln_z = stats[3] + torch.log(stats[4]) - math.log(n)
    return dict(mean=mean, var=var, ln_z=ln_z, n_finite=stats[0])


def all_gather_stats(local_stats, group=None):
    """[5] -> [world, 5] in rank order."""
    world = dist.get_world_size(group)
    out = torch.empty(world * NSTATS, dtype=local_stats.dtype, device=local_stats.device)
    dist.all_gather_into_tensor(out, local_stats.contiguous().view(-1), group=group)
    return out.view(world, NSTATS)


def sharded_bound(seeds_global, forward_fn, group=None):
    """Runs `forward_fn(local_seeds) -> (losses, z, stats[5])` on this rank's block of the global
    seed vector and merges the statistics across ranks.

    Returns dict(losses, z (local shards), lo, hi, stats (global), mean, var, ln_z)."""
    if not (dist.is_available() and dist.is_initialized()):
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = int(seeds_global.shape[0])
    lo, hi = shard_range(n, world, rank)
    if hi > lo:
        losses, z, stats = forward_fn(seeds_global[lo:hi])
    else:
        losses, z, stats = None, None, None
    if world > 1:
        if stats is None:   # forward only: an empty rank contributes the neutral element, on the device the collective uses
            dev = seeds_global.device
            if dist.get_backend(group) == "nccl" and not dev.type == "cuda":
                dev = torch.device("cuda", torch.cuda.current_device())
            stats = empty_stats(dev)
        rows = all_gather_stats(stats, group)
        stats = merge_stats(rows)
    out = dict(losses=losses, z=z, lo=lo, hi=hi, stats=stats)
    out.update(finalize(stats, n))
    return out


def sharded_var_grad(seeds_global, forward_fn, grad_fn, group=None):
    """VarGrad value-and-gradient with particles sharded over ranks.

    `forward_fn(local_seeds) -> (losses, z, stats[5])`, then the global mean comes from the merged
    statistics, then `grad_fn(local_seeds, losses, stats_global, n_total) -> grad_flat` (the local
    sum over this rank's particles of omega_n d w_n / d params), then ONE all-reduce(sum) of the
    gradient vector (RCCL over xGMI on GPUs; 84 KB for the dds net).  Returns dict(grad, value, ...)."""
    if not (dist.is_available() and dist.is_initialized()):
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = int(seeds_global.shape[0])
    lo, hi = shard_range(n, world, rank)
    _check_every_rank_has_particles(n, world)
    local = seeds_global[lo:hi]
    losses, z, stats = forward_fn(local)
    if world > 1:
        stats = merge_stats(all_gather_stats(stats, group))
    grad = grad_fn(local, losses, stats, n)
    if world > 1:
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    out = dict(grad=grad, losses=losses, z=z, lo=lo, hi=hi, stats=stats)
    out.update(finalize(stats, n))
    return out


def sharded_bound_grad(seeds_global, value_and_grad_fn, group=None):
    """Reparameterised `MCD_CAIS_sn` value-and-gradient with particles sharded over ranks.

    `value_and_grad_fn(local_seeds, n_total) -> (grad_flat, (losses, z), stats[5])` — e.g.
    `partial(mcdbm.compute_bound_grad, ..., return_stats=True)` with `n_total` forwarded — returns this rank's
    sum over its particles of (1 / n_total) d loss_n / d params; the weights do not depend on the other
    ranks (mean loss), so the forward statistics and the gradient travel together: one all-gather of the
    5-double statistics and ONE all-reduce(sum) of the gradient vector.  Returns dict(grad, mean, ...)."""
    if not (dist.is_available() and dist.is_initialized()):
        world, rank = 1, 0
    else:
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    n = int(seeds_global.shape[0])
    lo, hi = shard_range(n, world, rank)
    _check_every_rank_has_particles(n, world)
    grad, (losses, z), stats = value_and_grad_fn(seeds_global[lo:hi], n)
    if world > 1:
        stats = merge_stats(all_gather_stats(stats, group))
        dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
    out = dict(grad=grad, losses=losses, z=z, lo=lo, hi=hi, stats=stats)
    out.update(finalize(stats, n))
    return out


def make_sharded_grad_and_loss(boundmode, eps_schedule=None, grad_clipping=False, group=None):
    """`grad_and_loss` for `opt.run` under torchrun (one process per GPU): every rank draws the SAME global seed
    vector (same generator seed), works on its contiguous block, and the gradient is all-reduced — the data-parallel
    form of /root/reference/src/main.py:161-176.

    * MCD_CAIS_var_sn: local forward -> all-gather + merge of the 5-double statistics (the global mean loss the
      VarGrad weights need: BASELINE's "RCCL log-w all-reduce") -> local gradient -> one all-reduce of grad_flat;
    * MCD_CAIS_sn / MCD_ULA_sn / MCD_ULA: weights 1 / N_total are known up front -> local value-and-gradient -> one
      all-reduce of grad_flat.
    Returns (grad_flat, (local losses, local z[, global statistics])); with no process group it is the plain single-GPU
    call.  With more than one rank the third entry of the aux tuple is the MERGED 5-double statistics vector: `opt.run`
    takes its divergence decision (isnan(mean loss), opt.py:122-124) and its logged loss from it, so that every rank
    skips or applies the same update — the local shard's losses alone could be NaN on one rank only."""
    from . import mcdboundingmachine as mcdbm

    def grad_and_loss(seeds_global, params_flat, unflatten, params_fixed, log_prob):
        on = dist.is_available() and dist.is_initialized()
        world, rank = (dist.get_world_size(group), dist.get_rank(group)) if on else (1, 0)
        n = int(seeds_global.shape[0])
        lo, hi = shard_range(n, world, rank)
        _check_every_rank_has_particles(n, world)
        local = seeds_global[lo:hi]
        merged = {}
        if "var" in boundmode:
            def merge(st):
                merged["stats"] = merge_stats(all_gather_stats(st, group))
                return merged["stats"]
            grad, aux = mcdbm.compute_log_var_grad(local, params_flat, unflatten, params_fixed, log_prob,
                                                   eps_schedule=eps_schedule, grad_clipping=grad_clipping,
                                                   n_total=n, stats_total=merge if world > 1 else None)
        elif world > 1:
            grad, aux, stats = mcdbm.compute_bound_grad(local, params_flat, unflatten, params_fixed, log_prob,
                                                        eps_schedule=eps_schedule, grad_clipping=grad_clipping,
                                                        n_total=n, return_stats=True)
            merged["stats"] = merge_stats(all_gather_stats(stats, group))
        else:
            grad, aux = mcdbm.compute_bound_grad(local, params_flat, unflatten, params_fixed, log_prob,
                                                 eps_schedule=eps_schedule, grad_clipping=grad_clipping, n_total=n)
        if world > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM, group=group)
            return grad, (aux[0], aux[1], merged["stats"])
        return grad, aux
    return grad_and_loss
