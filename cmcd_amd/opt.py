"""Optimiser loop around the hot path.

Mirrors /root/reference/src/opt.py:14-35 (`project`, `create_optimizer`: optax.chain(clip(5.0), adam)) and
:67-164 (`run`), with torch tensors on the device; W&B logging and sample plots are left out.  The
gradient comes from `grad_and_loss(seeds, params_flat, unflatten, params_fixed, log_prob_model)` exactly as
in the reference (e.g. functools.partial(mcdboundingmachine.compute_log_var_grad, eps_schedule=...)).
"""
import copy

import torch


def project(x, unflatten, trainable):
    """/root/reference/src/opt.py:14-24 (in place on the views of `x`)."""
    x_train, _ = unflatten(x)
    if "eps" in trainable:
        x_train["eps"].clamp_(0.0000001, 0.5)
    if "eta" in trainable:
        x_train["eta"].clamp_(0, 0.99)
    if "gamma" in trainable:
        x_train["gamma"].clamp_(min=0.001)
    if "mgridref_y" in trainable:
        x_train["mgridref_y"].copy_(torch.relu(x_train["mgridref_y"] - 0.001) + 0.001)
    return x


class _ClipAdam:
    """optax.chain(optax.clip(5.0), optax.adam(lr, b1, b2, eps)): elementwise clip, then Adam with bias
    correction and eps outside the square root."""

    def __init__(self, step_size, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = step_size, b1, b2, eps

    def init(self, params):
        return {"count": 0, "mu": torch.zeros_like(params), "nu": torch.zeros_like(params)}

    def update(self, grad, state, params=None):
        g = grad.clamp(-5.0, 5.0)
        state["count"] += 1
        t = state["count"]
        state["mu"].mul_(self.b1).add_(g, alpha=1 - self.b1)
        state["nu"].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
        mu_hat = state["mu"] / (1 - self.b1 ** t)
        nu_hat = state["nu"] / (1 - self.b2 ** t)
        return -self.lr * mu_hat / (nu_hat.sqrt() + self.eps), state


def _project_ranges(unflatten, trainable):
    """The reference's `project` (opt.py:14-24) as (offset, length, kind, lo, hi) records of params_flat."""
    from . import _lib
    out = []

    def add(name, kind, lo, hi):
        if name in trainable and (0, name) in unflatten.layout:
            off, shape = unflatten.layout[(0, name)]
            n = 1
            for d in shape:
                n *= d
            out.append(_lib.ProjectRange(off, max(n, 1), kind, 0, lo, hi))
    add("eps", 0, 0.0000001, 0.5)
    add("eta", 0, 0.0, 0.99)
    add("gamma", 0, 0.001, float("inf"))
    add("mgridref_y", 1, 0.001, 0.0)
    return out


class _FusedClipAdam(_ClipAdam):
    """The same optimiser, with clip -> Adam -> apply -> project (-> EMA) as ONE launch of the library's
    `cmcd_adam_step` on device tensors (the reference gets this fusion from jit; eager torch would issue ~15
    small kernels per iteration, which at 1 ms per training step is a third of the loop)."""

    def step(self, params, grad, state, unflatten, trainable, ema=None, ema_step=0.001, device_counter=None,
             losses=None, diverged=None):
        """`device_counter`: int64 device scalar holding the number of completed steps — the graph-replay form
        (`cmcd_adam_step_dev`), in which no launch argument changes from one iteration to the next.
        `losses` / `diverged`: the reference's `if isnan(mean(loss)): return` (opt.py:122-124) evaluated inside the
        launch — a step whose mean loss is NaN is skipped and sets the sticky int32 device flag `diverged`."""
        from . import _lib
        L = _lib.lib()
        ranges = state.get("ranges")
        if ranges is None:
            rl = _project_ranges(unflatten, trainable)
            ranges = state["ranges"] = (_lib.ProjectRange * max(len(rl), 1))(*rl), len(rl)
        arr, cnt = ranges
        lptr, ln = (losses.data_ptr(), losses.numel()) if losses is not None else (None, 0)
        dptr = diverged.data_ptr() if diverged is not None else None
        if device_counter is not None:
            with torch.cuda.device(params.device):
                _lib.check(L.cmcd_adam_step_dev(
                    params.data_ptr(), grad.data_ptr(), state["mu"].data_ptr(), state["nu"].data_ptr(),
                    ema.data_ptr() if ema is not None else None, params.numel(), self.lr, self.b1, self.b2, self.eps,
                    5.0, device_counter.data_ptr(), ema_step, arr, cnt, lptr, ln, dptr,
                    torch.cuda.current_stream().cuda_stream))
            _bump(params, ema)
            return
        state["count"] += 1
        with torch.cuda.device(params.device):
            _lib.check(L.cmcd_adam_step(
                params.data_ptr(), grad.data_ptr(), state["mu"].data_ptr(), state["nu"].data_ptr(),
                ema.data_ptr() if ema is not None else None, params.numel(), self.lr, self.b1, self.b2, self.eps,
                5.0, state["count"], ema_step, arr, cnt, lptr, ln, dptr, torch.cuda.current_stream().cuda_stream))
        _bump(params, ema)


def _bump(*tensors):
    """The fused step writes its parameters through raw pointers: tell torch (the version counter is what the forward's
    prepared-table cache keys on, cmcd_amd/mcdboundingmachine.py:_prepared)."""
    for t in tensors:
        if t is not None:
            torch.autograd.graph.increment_version(t)


def create_optimizer(step_size, b1=0.9, b2=0.999, eps=1e-8, trainable=None):
    """/root/reference/src/opt.py:27-35"""
    return _FusedClipAdam(step_size, b1, b2, eps)


def run(info, lr, iters, params_flat, unflatten, params_fixed, log_prob_model, grad_and_loss, trainable, rng_key_gen,
        extra=True, log_prefix="", target_samples=None, use_ema=False, use_graph=None):
    """/root/reference/src/opt.py:67-164 -> (losses, params_flat, ema_params).

    `rng_key_gen`: an int seed (a torch generator draws the per-iteration particle seeds with
    randint(1, 1e6), opt.py:93-94) or a ready torch.Generator.  `info.N` particles per iteration.  A NaN
    mean loss stops the run ("Diverged", opt.py:122-124) and returns what the reference *meant* to
    return, a 3-tuple.

    `use_graph` (default off; CMCD_TRAIN_GRAPH=1 or use_graph=True): after three eager iterations the whole
    iteration — gradient launch sequence + fused optimiser step — is captured once in a HIP graph and replayed
    with fresh seeds copied into a static buffer.  Measured on MI355X it removes the host from the loop but not
    the ~5 us dependent-launch floor of each of the ~20 kernels, which is what bounds the small configurations
    (gmm K=8, N=300: 0.10 ms per iteration either way), so it is an option, not the default."""
    optimizer = create_optimizer(lr, trainable=trainable)
    params_flat = params_flat.clone()
    opt_state = optimizer.init(params_flat)
    ema_params = copy.deepcopy(params_flat) if use_ema else None
    gen = rng_key_gen if isinstance(rng_key_gen, torch.Generator) else torch.Generator().manual_seed(int(rng_key_gen))
    losses = []
    n = info.N if hasattr(info, "N") else info["N"]
    every = max(iters // 1000, 1)
    fused = params_flat.is_cuda and params_flat.dtype == torch.float32 and params_flat.is_contiguous()
    # the reference tests isnan(mean(loss)) on the host every iteration, before the update (opt.py:122-124).  Here the
    # fused step makes that decision on the device (a NaN step is skipped and raises `diverged`), and the host polls the
    # flag together with the logged loss every 0.1 % of the run and once at the end: same returned parameters — the
    # last ones before the NaN loss — without a host sync per iteration.
    diverged = torch.zeros(1, dtype=torch.int32, device=params_flat.device) if fused else None
    guard = lambda l: l if (l.is_cuda and l.dtype == torch.float32 and l.is_contiguous()) else l.float().contiguous()
    if use_graph is None:
        import os
        use_graph = fused and os.environ.get("CMCD_TRAIN_GRAPH", "0") == "1"
    use_graph = bool(use_graph) and fused and iters > 8
    graph = None
    n_eager = 3
    # seeds for a block of iterations are drawn at once (same generator stream as one draw per iteration) and
    # shipped to the device in one copy
    block = max(1, min(iters, (1 << 22) // max(n, 1)))
    for i in range(iters):
        if i % block == 0:
            nb = min(block, iters - i)
            seed_block = torch.randint(1, 1000000, (nb * n,), generator=gen, dtype=torch.int32).to(params_flat.device)
        seeds = seed_block[(i % block) * n:(i % block + 1) * n]
        if use_graph and i >= n_eager:
            if graph is None:
                # capture: static seeds in, (loss, params, moments, EMA) updated in place
                static_seeds = seeds.clone()
                counter = torch.full((1,), opt_state["count"], dtype=torch.int64, device=params_flat.device)
                torch.cuda.synchronize()
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        g_grad, g_aux = grad_and_loss(static_seeds, params_flat, unflatten, params_fixed,
                                                      log_prob_model)
                        g_loss = g_aux[0] if len(g_aux) <= 2 else (g_aux[2][1:2] / float(n)).to(torch.float32)
                        optimizer.step(params_flat, g_grad, opt_state, unflatten, trainable,
                                       ema=ema_params if use_ema else None, device_counter=counter,
                                       losses=guard(g_loss), diverged=diverged)
                except Exception as exc:   # capture refused (e.g. a gradient path that synchronises): stay eager
                    print(f"opt.run: graph capture unavailable ({exc}); running eagerly")
                    use_graph, graph = False, None
                # capture does not execute: this iteration's replay follows like every other
            if graph is None:
                grad, aux = grad_and_loss(seeds, params_flat, unflatten, params_fixed, log_prob_model)
                loss = aux[0] if len(aux) <= 2 else (aux[2][1:2] / float(n)).to(torch.float32)
                if i % every == 0:
                    mean_loss = float(loss.mean())
                    if mean_loss != mean_loss or int(diverged) != 0:
                        print("Diverged")
                        return losses, params_flat, ema_params
                    losses.append(mean_loss)
                optimizer.step(params_flat, grad, opt_state, unflatten, trainable, ema=ema_params if use_ema else None,
                               losses=guard(loss), diverged=diverged)
                continue
            static_seeds.copy_(seeds)
            graph.replay()
            _bump(params_flat, ema_params if use_ema else None)     # the replayed step wrote them through raw pointers
            opt_state["count"] += 1
            if i % every == 0:
                mean_loss = float(g_loss.mean())
                if mean_loss != mean_loss or int(diverged) != 0:
                    print("Diverged")
                    return losses, params_flat, ema_params
                losses.append(mean_loss)
            continue
        grad, aux = grad_and_loss(seeds, params_flat, unflatten, params_fixed, log_prob_model)
        loss = aux[0]
        if len(aux) > 2:
            # particles sharded over ranks (parallel.make_sharded_grad_and_loss): `loss` is this rank's shard, aux[2]
            # the merged statistics.  The guard and the logged value come from the GLOBAL sum of losses so that all
            # ranks take the same decision.
            loss = (aux[2][1:2] / float(n)).to(torch.float32)
        if i % every == 0:
            mean_loss = float(loss.mean())                   # the only host sync, every 0.1 % of the steps
            if mean_loss != mean_loss or (diverged is not None and int(diverged) != 0):
                print("Diverged")
                return losses, params_flat, ema_params
            losses.append(mean_loss)
            if iters >= 20000 and i % (every * 50) == 0:     # long runs: a sign of life every 5 % (stderr)
                import sys
                print(f"[opt.run {log_prefix}] iteration {i} / {iters}: mean loss {mean_loss:.4f}", file=sys.stderr, flush=True)
        elif not fused:
            if bool(torch.isnan(loss.mean())):               # eager path: the reference's per-iteration check
                print("Diverged")
                return losses, params_flat, ema_params
        if fused:
            optimizer.step(params_flat, grad, opt_state, unflatten, trainable, ema=ema_params if use_ema else None,
                           losses=guard(loss), diverged=diverged)
        else:
            updates, opt_state = optimizer.update(grad, opt_state, params_flat)
            params_flat.add_(updates)
            project(params_flat, unflatten, trainable)
            if use_ema:
                ema_params.mul_(1 - 0.001).add_(params_flat, alpha=0.001)   # optax.incremental_update(step_size=0.001)
    if diverged is not None and int(diverged) != 0:
        print("Diverged")
    return losses, params_flat, ema_params
