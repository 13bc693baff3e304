"""world_size-2 gloo run of the particle sharding + statistics merge (the N>1 path of bench.py)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cmcd_amd import parallel, synthetic

from helpers import run_oracle


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_forward(built):
    from oracle.cmcd_oracle import stats5

    def fwd(seeds):
        loss, z = run_oracle(built, seeds.numpy(), dtype=np.float32)
        return torch.from_numpy(loss), torch.from_numpy(z), torch.from_numpy(stats5(loss))
    return fwd


def _worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cpu", nbridges=4)
    seeds = torch.from_numpy(synthetic.parity_seeds(n))
    r = parallel.sharded_bound(seeds, _oracle_forward(b))
    out[rank] = (r["lo"], r["hi"], float(r["mean"]), float(r["var"]), float(r["ln_z"]), float(r["n_finite"]),
                 None if r["losses"] is None else r["losses"].numpy())
    dist.destroy_process_group()


def _run(world, n):
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, out), nprocs=world, join=True)
    return dict(out)


def test_two_ranks_match_single_process():
    n = 70
    out = _run(2, n)
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cpu", nbridges=4)
    loss, _ = run_oracle(b, synthetic.parity_seeds(n), dtype=np.float32)
    assert (out[0][0], out[0][1], out[1][0], out[1][1]) == (0, 35, 35, 70)
    np.testing.assert_array_equal(np.concatenate([out[0][6], out[1][6]]), loss)   # contiguous shards
    from oracle.cmcd_oracle import ln_z
    fin = np.isfinite(loss)
    for r in (0, 1):                                                   # every rank holds the global result
        assert out[r][5] == fin.sum()
        assert abs(out[r][4] - ln_z(loss)) < 1e-9
        if fin.all():
            assert abs(out[r][2] - loss.astype(np.float64).mean()) < 1e-9
            assert abs(out[r][3] - loss.astype(np.float64).var()) < 1e-7
    assert np.array_equal(out[0][2:6], out[1][2:6], equal_nan=True)    # identical on every rank
    assert np.isinf(out[0][2]) and np.isnan(out[0][3])                 # +inf particle => mean inf, var nan


def test_empty_shard():
    out = _run(2, 1)                                                   # rank 1 gets no particles
    assert (out[1][0], out[1][1]) == (1, 1) and out[1][6] is None
    assert np.array_equal(out[0][2:6], out[1][2:6], equal_nan=True)


def test_shard_range_covers_everything():
    for n in (1, 7, 2000, 16001):
        for w in (1, 2, 3, 8):
            r = [parallel.shard_range(n, w, k) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n and all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in r]
            assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)     # balanced
    assert all(hi > lo for lo, hi in (parallel.shard_range(9, 8, k) for k in range(8)))      # nobody empty at n >= world


def _too_few_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seeds = torch.arange(1, 2, dtype=torch.int32)              # one particle, two ranks
    got = []
    for fn in (lambda: parallel.sharded_bound_grad(seeds, lambda s, n: 1 / 0),
               lambda: parallel.sharded_var_grad(seeds, lambda s: 1 / 0, lambda *a: 1 / 0)):
        try:
            fn()
            got.append("no error")
        except ValueError as e:
            got.append(str(e))
    out[rank] = got
    dist.destroy_process_group()


def test_fewer_particles_than_ranks_is_refused_on_every_rank():
    """The check runs before any collective and gives the same answer on all ranks (a ValueError on the empty rank
    alone would leave the other one waiting in the all-gather)."""
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_too_few_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert out[0] == out[1] and all("fewer particles" in m for m in out[0])


def test_merge_stats_matches_c_abi():
    from cmcd_amd import _lib, build
    from oracle.cmcd_oracle import stats5
    build.build()
    rng = np.random.default_rng(1)
    parts = [rng.normal(2, 3, k) for k in (5, 1, 100)]
    rows = [stats5(p) for p in parts]
    merged_c, mean, var, lnz = _lib.stats_merge(rows, [len(p) for p in parts])
    m = parallel.merge_stats(torch.tensor(np.array(rows)))
    np.testing.assert_allclose(m.numpy(), merged_c, rtol=1e-14)
    f = parallel.finalize(m, 106)
    assert abs(float(f["mean"]) - mean) < 1e-13 and abs(float(f["ln_z"]) - lnz) < 1e-13


def _grad_worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cmcd_oracle_torch as ot
    from oracle.cmcd_oracle import stats5
    b = synthetic.build("gmm_n300_k8", device="cpu", boundmode="MCD_CAIS_var_sn", nbridges=3)
    dim, K, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])

    def fwd(seeds):
        loss, z = run_oracle(b, seeds.numpy(), dtype=np.float64)
        return torch.from_numpy(loss), torch.from_numpy(z), torch.from_numpy(stats5(loss))

    def grad(seeds, losses, stats, n_total):
        # local sum_n omega_n dw_n/dtheta with omega from the GLOBAL mean: autograd of sum(omega * w)
        pt = ot.to_torch(p)
        l, _ = ot.losses(seeds.numpy(), pt, dim, K, mode, spec.arch, "gmm", b["cfg"]["eps_schedule"], False)
        mean = float(stats[1]) / n_total
        omega = (-2.0 / n_total) * (l.detach() - mean)
        (g,) = torch.autograd.grad((omega * (-l)).sum(), [pt["sn"]["W3"]])
        return g.reshape(-1).clone()

    seeds = torch.from_numpy(synthetic.parity_seeds(n))
    r = parallel.sharded_var_grad(seeds, fwd, grad)
    out[rank] = (r["grad"].numpy(), float(r["var"]))
    dist.destroy_process_group()


def test_sharded_vargrad_equals_single_process():
    """Two ranks, global mean through the statistics merge, one gradient all-reduce == one process."""
    from oracle import cmcd_oracle_torch as ot
    n = 22
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_grad_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    b = synthetic.build("gmm_n300_k8", device="cpu", boundmode="MCD_CAIS_var_sn", nbridges=3)
    dim, K, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    val, _, _, g = ot.bound_and_grad(synthetic.parity_seeds(n), p, dim, K, mode, spec.arch, "gmm",
                                     b["cfg"]["eps_schedule"], False)
    for r in (0, 1):
        np.testing.assert_allclose(out[r][0], g["sn"]["W3"].reshape(-1), rtol=1e-9, atol=1e-12)
        assert abs(out[r][1] - val) < 1e-9


def _bptt_worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cmcd_oracle_torch as ot
    from oracle.cmcd_oracle import stats5
    b = synthetic.build("gmm_n300_k8", device="cpu", nbridges=3)
    dim, K, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])

    def value_and_grad(seeds, n_total):
        pt = ot.to_torch(p)
        l, z = ot.losses(seeds.numpy(), pt, dim, K, mode, spec.arch, "gmm", b["cfg"]["eps_schedule"], False)
        (g,) = torch.autograd.grad(l.sum() / n_total, [pt["sn"]["W3"]])
        return g.reshape(-1).clone(), (l.detach(), z.detach()), torch.from_numpy(stats5(l.detach().numpy()))

    r = parallel.sharded_bound_grad(torch.from_numpy(synthetic.parity_seeds(n)), value_and_grad)
    out[rank] = (r["grad"].numpy(), float(r["mean"]))
    dist.destroy_process_group()


def test_sharded_reparameterised_gradient_equals_single_process():
    """MCD_CAIS_sn: local gradients weighted 1 / N_total + one all-reduce == jax.grad of the global mean."""
    from oracle import cmcd_oracle_torch as ot
    n = 21
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_bptt_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    b = synthetic.build("gmm_n300_k8", device="cpu", nbridges=3)
    dim, K, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    val, _, _, g = ot.bound_and_grad(synthetic.parity_seeds(n), p, dim, K, mode, spec.arch, "gmm",
                                     b["cfg"]["eps_schedule"], False)
    for r in (0, 1):
        np.testing.assert_allclose(out[r][0], g["sn"]["W3"].reshape(-1), rtol=1e-9, atol=1e-12)
        assert abs(out[r][1] - val) < 1e-9


def _uha_worker(rank, world, port, n, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import cmcd_oracle_torch as ot
    from oracle.cmcd_oracle import stats5
    b = synthetic.build("gmm_n300_k8", device="cpu", boundmode="MCD_CAIS_UHA_sn", nbridges=3, init_gamma=3.0)
    dim, K, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])

    def value_and_grad(seeds, n_total):
        pt = ot.to_torch(p)
        l, z = ot.losses(seeds.numpy(), pt, dim, K, mode, spec.arch, "gmm", b["cfg"]["eps_schedule"], False)
        g_w3, g_gam = torch.autograd.grad(l.sum() / n_total, [pt["sn"]["W3"], pt["gamma"]])
        flat = torch.cat([g_w3.reshape(-1), g_gam.reshape(-1)]).clone()
        return flat, (l.detach(), z.detach()), torch.from_numpy(stats5(l.detach().numpy()))

    r = parallel.sharded_bound_grad(torch.from_numpy(synthetic.parity_seeds(n)), value_and_grad)
    out[rank] = (r["grad"].numpy(), float(r["mean"]), r["lo"], r["hi"])
    dist.destroy_process_group()


def test_sharded_second_order_gradient_equals_single_process():
    """MCD_CAIS_UHA_sn (2nd-order CMCD, state (z, rho), no stop_gradient): the sharded value-and-gradient — local gradients
    weighted 1 / N_total, statistics all-gather, ONE all-reduce — equals the single-process gradient of the global mean, for a
    network leaf and for the friction gamma, which only this mode trains."""
    from oracle import cmcd_oracle_torch as ot
    n = 23                                                    # ragged: 12 + 11
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_uha_worker, args=(2, _free_port(), n, out), nprocs=2, join=True)
    b = synthetic.build("gmm_n300_k8", device="cpu", boundmode="MCD_CAIS_UHA_sn", nbridges=3, init_gamma=3.0)
    dim, K, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    val, _, _, g = ot.bound_and_grad(synthetic.parity_seeds(n), p, dim, K, mode, spec.arch, "gmm",
                                     b["cfg"]["eps_schedule"], False)
    want = np.concatenate([np.asarray(g["sn"]["W3"]).reshape(-1), np.asarray(g["gamma"]).reshape(-1)])
    assert (out[0][2], out[0][3], out[1][2], out[1][3]) == (0, 12, 12, 23)
    assert abs(want[-1]) > 0                                  # the friction's gradient is not trivially zero
    for r in (0, 1):
        np.testing.assert_allclose(out[r][0], want, rtol=1e-9, atol=1e-12)
        assert abs(out[r][1] - val) < 1e-9
