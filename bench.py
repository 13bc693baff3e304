#!/usr/bin/env python3
"""bench.py — throughput of the CMCD annealed-Langevin bound on MI355X.

A "step" is one `compute_bound` forward over one batch of synthetic particles (the hot path named
by BASELINE.json: many_gmm, MCD_CAIS_sn, N=2000, nbridges=256, dds net).  Multi-GPU: one process per
GPU (torch.distributed, backend nccl = RCCL); every rank runs the named batch on its own seeds
(weak scaling) and one all-gather of the 5-number statistics vector per step merges the ELBO mean /
ln Z across ranks.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

PEAK_FP32_TFLOPS = 157.3   # MI355X fp32 vector == fp32 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def flops_per_particle_step(cfg, dim, width):
    """Algorithmic work of one particle-bridge-step AFTER the two value-preserving restructurings of
    DESIGN.md (one net + one target-gradient evaluation per step; time path folded into a per-step
    bias): 2 * MAC_net + F_target + 24 d.  `survey` is SURVEY.md section 8d's figure (un-folded first layer)."""
    if cfg["nn_arch"] == "dds":
        mac = dim * 64 + 64 * 64 + 64 * dim
        mac_survey = (dim + 64) * 64 + 64 * 64 + 64 * dim
    else:
        mac = dim * width + width * width + width * dim
        mac_survey = 2 * width * width + width * dim
    f_target = {"gmm": 200, "funnel": 60, "many_gmm": 800, "lgcp": 2 * dim * dim}[cfg["model"]]
    return 2 * mac + f_target + 24 * dim, 2 * mac_survey + f_target + 24 * dim


def cpu_baseline(built, seeds_np, losses_hip, max_particles, min_seconds=8.0):
    """CPU baseline = the oracle timed on this box's host cores on a bounded sample of the same
    workload: the plain-C restatement (oracle/cmcd_oracle.c: scalar float32, reference-faithful 2 net
    + 2 gradient evaluations per bridge step, OpenMP over particles) run repeatedly on the named batch
    for >= min_seconds; the NumPy restatement is timed once beside it.  The first call's output also
    gives the ELBO / ln Z absolute errors of the HIP path on identical seeds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import run_c_oracle, run_oracle
    from oracle import c_oracle
    from oracle import cmcd_oracle as orc
    n = min(len(seeds_np), max_particles)
    K = built["params_fixed"][1]
    calls, t0 = 0, time.perf_counter()
    l_ref = None
    while True:
        l_c, _ = run_c_oracle(built, seeds_np[:n])
        l_ref = l_c if l_ref is None else l_ref
        calls += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds and calls >= 3:
            break
    threads = c_oracle.threads()
    out = {
        "value": calls * n * K / dt, "unit": "bridge-steps*particles/s", "cores": threads, "kind": "port",
        "sample": f"{calls} compute_bound calls of {n} particles x {K} bridges in {dt:.1f}s; plain-C float32 oracle, "
                  f"2 net + 2 grad evaluations per step as the reference, OpenMP threads={threads} "
                  f"(os.cpu_count()={os.cpu_count()})",
        "seconds": dt,
    }
    t1 = time.perf_counter()
    run_oracle(built, seeds_np[:n], dtype=np.float32, reuse=False)
    dt_np = time.perf_counter() - t1
    out["numpy_port_value"] = n * K / dt_np
    lh = losses_hip[:n].astype(np.float64)
    lr = l_ref.astype(np.float64)
    fin = np.isfinite(lr)
    parity = {
        "elbo_abs_err": float(abs(lh[fin].mean() - lr[fin].mean())),
        "lnz_abs_err": float(abs(orc.ln_z(lh) - orc.ln_z(lr))),
        "inf_set_equal": bool(np.array_equal(np.isinf(lh), np.isinf(lr))),
        "n": int(n), "against": "plain-C float32 oracle (reference-faithful)",
    }
    return out, parity


def _strict(o):
    """Strict JSON: non-finite floats become null (json.dumps would print -Infinity / NaN, which is not JSON)."""
    if isinstance(o, dict):
        return {k: _strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v) for v in o]
    if isinstance(o, float) and not math.isfinite(o):
        return None
    return o


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of timed work — 20 steps (7 ms) end before the GPU's clocks have settled
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--spinup", type=int, default=300, help="untimed launches before the warm-up steps (clock settling)")
    ap.add_argument("--config", default=None, help="name in cmcd_amd.synthetic.CONFIGS")
    ap.add_argument("--particles", type=int, default=None, help="override N per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--forward-only", action="store_true",
                    help="skip the vargrad / training_step legs (their trajectory-keeping forward launches would be averaged "
                         "into the same kernel name by rocprofv3 --stats)")
    ap.add_argument("--cpu-particles", type=int, default=2000)
    ap.add_argument("--saturated", type=int, default=1 << 18,
                    help="also time a saturating batch of this many particles (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a ROCm GPU (no CPU fallback for the hot path)")
    # test hook (tests/test_gpu_bench.py): all ranks on device 0 over gloo, to exercise the N > 1 code path on a 1-GPU box
    shared_gpu = os.environ.get("CMCD_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if shared_gpu else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    use_dist = "RANK" in os.environ and "MASTER_PORT" in os.environ   # launched by torch.distributed.run
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    from cmcd_amd import _lib, build, synthetic
    from cmcd_amd import mcdboundingmachine as mcdbm
    from cmcd_amd import parallel
    if local_rank == 0:
        build.build()          # only one process per node may (re)build the in-tree library
    if use_dist:
        dist.barrier()

    name = args.config or synthetic.NORTH_STAR
    over = {"N": args.particles} if args.particles else {}
    if "lgcp" in name:   # the 40 x 40 bin counts of the point set ship as a fixture
        over["lgcp_counts"] = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
    b = synthetic.build(name, device=device, **over)
    cfg = b["cfg"]
    dim, K, mode, spec = b["params_fixed"]
    n = cfg["N"]
    seeds_np = synthetic.throughput_seeds(n, stream=rank)
    seeds = torch.from_numpy(seeds_np).to(device)  # resident in HBM before the timed region

    def forward(s):
        return mcdbm.bound_forward(s, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                   eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])

    # Multi-GPU: one RCCL all-gather of the 40-byte statistics vector + one merge kernel per step.  The collective is
    # latency-only (~20 us against a 340 us step), so it is taken off the launch stream: torch's process group runs it on
    # its own stream (async_op=True) behind an event on the forward of step k, and the launch stream waits for it only
    # after the forward of step k+1 has been enqueued — the statistics of step k are merged one step late, every step's
    # all-gather and merge still run inside the timed region (the last one is drained before the closing barrier).
    gathered = [torch.zeros(world * parallel.NSTATS, dtype=torch.float64, device=device) for _ in range(2)]
    pending = []          # [(work, buffer, stats kept alive)] of the step whose all-gather is in flight

    def drain():
        work, buf, _ = pending.pop()
        work.wait()
        return parallel.merge_stats(buf.view(world, parallel.NSTATS))

    def step(k):
        losses, z, stats = forward(seeds)
        if use_dist:
            work = dist.all_gather_into_tensor(gathered[k & 1], stats, async_op=True)
            merged = drain() if pending else None
            pending.append((work, gathered[k & 1], stats))
            stats = merged
        return losses, z, stats

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # device spin-up (untimed, before the W warm-up steps): ~0.1 s of launches so that the GPU's clocks have settled even
    # when the caller asks for a handful of steps (20 steps = 6 ms read 0.305 ms per kernel against 0.267 ms settled)
    for k in range(args.spinup):
        step(k)
    if pending:
        drain()
    for k in range(args.warmup):
        step(k)
    if pending:
        drain()
    barrier()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        losses, z, stats = step(k)
    if pending:
        stats = drain()   # global statistics of the last step
    barrier()
    elapsed = time.perf_counter() - t0
    kern_ms, launches = _lib.profile_collect()
    _lib.profile_enable(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    units_per_step = n * K * world
    value = units_per_step * args.steps / elapsed
    f_alg, f_survey = flops_per_particle_step(cfg, dim, spec.width)
    if launches == 0:   # lgcp: a launch sequence, no single trajectory kernel — the whole call is the unit
        kern_s, launches = elapsed / args.steps, args.steps
    else:
        kern_s = kern_ms / 1e3 / max(launches, 1)
    achieved = n * K * f_alg / kern_s / 1e12
    fin = parallel.finalize(stats, n * world)

    # HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc
    # passes of this same command; profiles/r01_pmc/summary.json) — a measured constant of the kernel,
    # bench.py cannot collect counters on itself.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc", "summary.json")))
        if name == synthetic.NORTH_STAR and n == 2000:
            traffic = pmc["coop_kernel"]["hbm_bytes_per_launch"]
    except Exception:
        pass

    # which trajectory kernel the library's auto-selection ran (cmcd_kernels.hip: cooperative up to 512 tiles,
    # 256 for nets wider than 128; CMCD_KERNEL_VARIANT pins it)
    tiles = (n + 15) // 16
    coop = mcdbm.KERNEL_VARIANT in (2, 3, 4) or (mcdbm.KERNEL_VARIANT == 0 and tiles <= (256 if spec.width >= 128 else 512))
    kernel_name = "coop_kernel" if coop else "traj_kernel"
    if coop:   # 8-particle tiles (twice the workgroups) while each still gets a CU; instances exist for widths <= 64
        half = mcdbm.KERNEL_VARIANT == 4 or (mcdbm.KERNEL_VARIANT != 3 and n <= 2048 and spec.width <= 64)
        kernel_name += "<8-particle tiles>" if half else "<16-particle tiles>"

    if cfg["model"] == "lgcp":
        kernel_name = "lgcp launch sequence (skinny GEMMs + state kernels)"

    result = {
        "metric": "bridge-steps*particles/sec", "value": value, "unit": "bridge-steps*particles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup": args.spinup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": name, "model": cfg["model"], "boundmode": cfg["boundmode"],
                   "particles_per_gpu": n, "nbridges": K, "nn_arch": cfg["nn_arch"], "dim": dim,
                   "global_particles": n * world, "parallelism": f"particles sharded x{world}, stats all-gather"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                     "kernel": kernel_name, "kernel_ms": kern_s * 1e3, "launches": launches,
                     "flop_per_particle_step": f_alg, "flop_per_particle_step_survey": f_survey,
                     "achieved_survey_flops": n * K * f_survey / kern_s / 1e12,
                     "state_bytes_gbs": n * K * (8 * dim + 8) / kern_s / 1e9},
        "elbo": float(-fin["mean"]), "ln_z": float(fin["ln_z"]), "n_finite": float(fin["n_finite"]),
    }
    # untrained net at init_sigma = 60: some particles leave float32 range exactly as in the reference (parity.inf_set_equal),
    # so the plain mean is -inf; the mean over this rank's finite particles is reported beside it
    lfin = losses[torch.isfinite(losses)]
    result["elbo_finite_particles"] = float(-lfin.double().mean()) if lfin.numel() else None

    if rank == 0 and args.saturated and world == 1:
        ns = args.saturated
        sseeds = torch.from_numpy(synthetic.throughput_seeds(ns, stream=7)).to(device)
        forward(sseeds)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        reps = 3
        for _ in range(reps):
            forward(sseeds)
        torch.cuda.synchronize()
        ms, cnt = _lib.profile_collect()
        _lib.profile_enable(False)
        ks = ms / 1e3 / cnt
        result["saturated"] = {"particles": ns, "kernel_ms": ks * 1e3, "value": ns * K / ks,
                               "achieved": ns * K * f_alg / ks / 1e12,
                               "frac": ns * K * f_alg / ks / 1e12 / PEAK_FP32_TFLOPS}

    if cfg["model"] == "lgcp":
        # weight-bandwidth bound: every evaluation streams K^-1 and the three weight matrices
        IN = dim + cfg["emb_dim"]
        wbytes = 4.0 * (dim * dim + 2 * dim * IN + IN * IN) * (K + 1) * -(-n // 32)
        # measured L2 <-> fabric bytes of the three GEMM launches of one evaluation (tools/probes/pmc_hbm_lgcp.sh: separate
        # rocprofv3 --pmc passes, FETCH_SIZE doubled per the gfx950 note), x (K + 1) evaluations x passes — a constant of the
        # named shape (N <= 32 per pass, K = 128, width 1620): weights + the operand slices every workgroup re-reads + slabs
        traffic = None
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc", "lgcp_summary.json")))
            if dim == 1600 and IN == 1620:
                traffic = sum(v["hbm_bytes_per_launch"] for v in pm.values()) * (K + 1) * -(-n // 32)
        except Exception:
            pass
        result["roofline"].update({"bound": "hbm", "achieved": wbytes / kern_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": wbytes / kern_s / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
                                   "traffic_unit": "bytes/call (PMC)", "weight_bytes_per_call": wbytes})

    if rank == 0 and world == 1 and name == synthetic.NORTH_STAR and not args.forward_only:
        # value-and-gradient of the VarGrad loss on the same batch (boundmode MCD_CAIS_var_sn, same net/target)
        try:
            bv = synthetic.build(name, device=device, boundmode="MCD_CAIS_var_sn", **over)
            gargs = (seeds, bv["params_flat"], bv["unflatten"], bv["params_fixed"], bv["target"])
            gkw = dict(eps_schedule=bv["eps_schedule"], grad_clipping=bv["grad_clipping"])
            for _ in range(2):
                mcdbm.compute_log_var_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(5):
                mcdbm.compute_log_var_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg0) / 5
            result["vargrad"] = {"ms_per_value_and_grad": tg * 1e3, "value": n * K / tg,
                                 "unit": "bridge-steps*particles/s (forward + backward)"}
        except NotImplementedError as e:
            result["vargrad"] = {"error": str(e)}
        # value-and-gradient of the north-star's own training loss (MCD_CAIS_sn, reparameterised gradient:
        # forward with stored trajectory + reverse sweep) on the same batch
        try:
            gargs = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
            gkw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
            for _ in range(2):
                mcdbm.compute_bound_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(5):
                mcdbm.compute_bound_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg0) / 5
            result["training_step"] = {"ms_per_value_and_grad": tg * 1e3, "value": n * K / tg,
                                       "unit": "bridge-steps*particles/s (forward + reverse sweep)"}
        except NotImplementedError as e:
            result["training_step"] = {"error": str(e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, parity = cpu_baseline(b, seeds_np, losses.cpu().numpy(), args.cpu_particles)
        result["cpu_baseline"] = base
        result["parity"] = parity
        result["speedup_vs_cpu"] = value / base["value"]

    if rank == 0:
        print(json.dumps(_strict(result)))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
