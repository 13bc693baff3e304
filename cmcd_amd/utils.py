"""Final-metric arithmetic and the evaluation harness around the hot path.

`log_final_losses` mirrors /root/reference/src/utils.py:219-248 (without the W&B logging);
`sample` mirrors /root/reference/src/opt.py:167-197 — the call sequence whose throughput
BASELINE.json's metric counts — but evaluates all `n_input_dist_seeds * n_samples` particles in ONE
launch sequence instead of a Python loop with a device->host sync per element
(`[x.item() for x in loss_list]`, opt.py:193).
"""
import math

import torch


def log_final_losses(eval_losses, log_prefix=""):
    """eval_losses: [n_input_dist_seeds, n_samples] -> (final_elbo, final_ln_Z) like the reference;
    the standard deviations over seed groups are returned as attributes of the result tuple's
    companion dict via `log_final_losses.last`."""
    e = torch.as_tensor(eval_losses, dtype=torch.float64)
    n_samples = e.shape[1]
    final_elbos = -e.mean(dim=1)
    final_ln_zs = torch.logsumexp(-e, dim=1) - math.log(n_samples)
    out = {
        f"elbo_final{log_prefix}": float(final_elbos.mean()),
        f"final_ln_Z{log_prefix}": float(final_ln_zs.mean()),
        f"elbo_final_std{log_prefix}": float(final_elbos.std(unbiased=False)),   # jnp.std: ddof = 0
        f"final_ln_Z_std{log_prefix}": float(final_ln_zs.std(unbiased=False)),
    }
    log_final_losses.last = out
    return out[f"elbo_final{log_prefix}"], out[f"final_ln_Z{log_prefix}"]


log_final_losses.last = {}


def sample(info, n_samples, n_input_dist_seeds, params_flat, unflatten, params_fixed, log_prob_model, loss_fn,
           eval_seeds, log_prefix=""):
    """/root/reference/src/opt.py:167-197 -> (elbos [n_input_dist_seeds][n_samples], zs [n*m, dim]).

    `eval_seeds` replaces the reference's `jax.random.randint(rng_key_gen, (n,), 1, 1e6)` (opt.py:182):
    an int32 tensor of n_samples * n_input_dist_seeds seeds.  `loss_fn` is called exactly like the
    reference's jitted callable, once, on all seeds; the per-group split happens afterwards."""
    eval_seeds = torch.as_tensor(eval_seeds)
    assert eval_seeds.numel() == n_samples * n_input_dist_seeds
    _, (loss_list, z) = loss_fn(eval_seeds, params_flat, unflatten, params_fixed, log_prob_model)
    elbos = loss_list.view(n_input_dist_seeds, n_samples)
    return elbos, z
