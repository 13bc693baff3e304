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


def params_to_numpy(params_flat, unflatten):
    """`params = {**params_train, **params_notrain}` (/root/reference/src/main.py:283-284) as nested dicts / lists
    of NumPy arrays — the structure the reference pickles as `params.pkl` (:286-296), minus the jax array type."""
    import numpy as np

    def conv(t):
        if isinstance(t, dict):
            return {k: conv(v) for k, v in t.items()}
        if isinstance(t, (list, tuple)):
            return type(t)(conv(v) for v in t)
        return np.asarray(t.detach().cpu().numpy())
    train, notrain = unflatten(params_flat.detach().cpu())
    return conv({**train, **notrain})


def save_params(path, params_flat, unflatten):
    """Writes `params.pkl` like the reference's W&B artifact (main.py:286-296): one pickle of the merged dict."""
    import pickle
    with open(path, "wb") as f:
        pickle.dump(params_to_numpy(params_flat, unflatten), f)


def load_params(path_or_dict, params_flat, unflatten):
    """Inverse of `save_params`: copies every leaf of a merged `params` dict (a pickle path, or the dict itself,
    e.g. a reference run's `jax.tree_util.tree_map(np.asarray, params)`) into a copy of `params_flat` at the
    offsets of `unflatten`.  Leaves are matched by name and must have the reference's shapes."""
    import pickle
    import numpy as np
    params = path_or_dict
    if not isinstance(params, dict):
        with open(path_or_dict, "rb") as f:
            params = pickle.load(f)
    out = params_flat.detach().clone()

    def walk(node, prefix):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(v, prefix + (k,))
        elif isinstance(node, (list, tuple)):
            for i, v in enumerate(node):
                walk(v, prefix + (i,))
        else:
            arr = np.asarray(node, np.float32)
            hit = [(path, off, shape) for path, (off, shape) in unflatten.layout.items() if path[1:] == prefix]
            if not hit:
                raise KeyError(f"leaf {prefix} is not part of this parameter tree")
            _, off, shape = hit[0]
            if tuple(arr.shape) != tuple(shape):
                raise ValueError(f"leaf {prefix}: shape {arr.shape} != {tuple(shape)}")
            out[off:off + max(arr.size, 1)] = torch.from_numpy(arr.reshape(-1)).to(out.device)
    walk(params, ())
    return out


def W2_distance(x, y, reg=0.01, num_iter_max=10000, stop_thr=1e-16):
    """/root/reference/src/utils.py:207-216: entropic OT cost `ot.sinkhorn2(a, b, M / M.max(), reg)` between two equal-size
    point clouds with uniform weights — the Sinkhorn-Knopp iteration of POT's default solver (POT itself is not in this
    image), in float64 torch on the device of `x`: u <- a / (K v), v <- b / (K^T u), K = exp(-M / reg); stops when the
    marginal violation drops below `stop_thr` (checked every 10 iterations) or after `num_iter_max` iterations."""
    x = torch.as_tensor(x, dtype=torch.float64)
    y = torch.as_tensor(y, dtype=torch.float64, device=x.device)
    n = x.shape[0]
    a = torch.full((n,), 1.0 / n, dtype=torch.float64, device=x.device)
    b = a.clone()
    M = torch.cdist(x, y) ** 2                      # ot.dist default: squared Euclidean
    M = M / M.max()
    Kmat = torch.exp(-M / reg)
    u, v = torch.ones_like(a) / n, torch.ones_like(b) / n
    for it in range(num_iter_max):
        KtU = Kmat.t() @ u
        v = b / KtU
        u = a / (Kmat @ v)
        if it % 10 == 0:
            err = torch.linalg.norm(v * (Kmat.t() @ u) - b) ** 2
            if float(err) < stop_thr:
                break
    P = u[:, None] * Kmat * v[None, :]
    return float((P * M).sum())


def calculate_W2_distances(samples, target_samples, other_target_samples, n_samples, n_input_dist_seeds, n_sinkhorn,
                           log_prefix=""):
    """/root/reference/src/utils.py:251-282, returning the four numbers it logs to W&B."""
    import numpy as np
    w2, self_w2 = [], []
    assert n_sinkhorn <= n_samples
    for i in range(n_input_dist_seeds):
        sl = slice(i * n_samples, i * n_samples + n_sinkhorn)
        w2.append(W2_distance(samples[sl], target_samples[sl]))
        self_w2.append(W2_distance(target_samples[sl], other_target_samples[sl]))
    return {f"w2_dist{log_prefix}": float(np.mean(w2)), f"w2_dist_std{log_prefix}": float(np.std(w2)),
            f"self_w2_dist{log_prefix}": float(np.mean(self_w2)), f"self_w2_dist_std{log_prefix}": float(np.std(self_w2))}
