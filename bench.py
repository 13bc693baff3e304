#!/usr/bin/env python3
"""bench.py — throughput of the CMCD annealed-Langevin bound on MI355X.

A "step" is one `compute_bound` forward over one batch of synthetic particles.  The HEADLINE (`value`, `config`,
`roofline`) is the same workload at every N: the hot path named by BASELINE.json (many_gmm, MCD_CAIS_sn, N=2000,
nbridges=256, dds net) with 2000 particles PER GPU ("scaling": "weak"), so a driver-built 1 -> 8 curve divides like by
like.  N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), particles sharded, one all-gather of the
5-number statistics vector per step merges the ELBO mean / ln Z across ranks.  At every N the headline is the PER-CALL form
(`headline_leg` = weak): every step's all-gather and merge COMPLETE, in stream order, before the next step's forward — what
one `compute_bound` call that returns the merged scalar costs (SURVEY.md section 8d times "the launch sequence incl. the
collective").  The throughput form (all-gather of step k beside the forward of step k + 1) is reported beside it in
legs.weak_pipelined and the collective's own cost in `collective`; the north_star's STRONG-scaling figures of the same job are
at top level in `strong_scaling`.  Legs timed in the same job and reported under "legs":

  weak                 the headline measurement at every N.  N > 1: all-gather + merge completed inside every step
  weak_prepared        the same loop inside fixed_parameters(): cmcd_bound_forward_prepared, no prep launch (NOT the headline)
  weak_pipelined       N > 1: the all-gather issued async and merged one step late — the rate of a loop of independent
                       calls that does not consume each scalar at once (NOT the headline)
  strong_named         the named batch split over the ranks (2000 / N each) — north_star's "strong scaling", latency-bound
  strong_sharded_cfg4  BASELINE.json configs[3]: many_gmm, MCD_CAIS_var_sn, 16000 particles x 132-wide net split over
                       the ranks (the configuration BASELINE names for 8 GPUs), with its sharded VarGrad training step

Each strong leg also times the un-split batch on rank 0's GPU alone inside the same job (`single_gpu_ms`,
`speedup_vs_single_gpu`), so the line carries its own strong-scaling ratios; `collective` holds the measured latency of
the statistics all-gather + merge.  Rank 0 prints ONE JSON line.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus 8 --steps 20 --warmup 3          # starts the launcher below itself, as a child process
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 20 --warmup 3

`value` is the DEFAULT product path at every N (one plain compute_bound per step, prep launch included); the loop inside
`fixed_parameters()` (prep launch skipped on unchanged parameters) is the side leg legs.weak_prepared.  `roofline`,
`cpu_baseline` and `parity` are in the line at every N (rank 0 times the CPU baseline, the other ranks wait).
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
CFG4 = "many_gmm_var_n16000_k256"


SHA_FAMILIES = {
    # which HIP sources a stored PMC figure depends on: the trajectory kernels of configs 1 - 4, and the d = 1600 launch
    # sequences (r04: separate, so that work on one family does not void the other's counters)
    "traj": ("cmcd_coop.hip", "cmcd_kernels.hip", "cmcd_device.h", "cmcd_common.h"),
    "lgcp": ("cmcd_lgcp.hip", "cmcd_lgcp_wide.hip", "cmcd_device.h", "cmcd_common.h"),
}


def kernel_sources_sha(family="traj"):
    """Identifies the kernel build a stored PMC figure belongs to: sha1 over the HIP sources of that kernel family."""
    import hashlib
    h = hashlib.sha1()
    for f in SHA_FAMILIES[family]:
        with open(os.path.join(ROOT, "cmcd_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def stored_traffic(key):
    """HBM bytes per launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of
    this same command, tools/probes/pmc_hbm.sh) — bench.py cannot collect counters on itself.  The summary records the
    sha of the kernel sources it was measured on; a figure from another build is reported as null, not as current."""
    for rnd in ("r05_pmc", "r04_pmc", "r03_pmc", "r02_pmc", "r01_pmc"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", rnd, "summary.json")))
        except Exception:
            continue
        if pmc.get("kernel_sources_sha") == kernel_sources_sha() and key in pmc:
            return pmc[key].get("hbm_bytes_per_launch"), rnd
    return None, None


def stored_issue_occupancy(key):
    """Issue-slot occupancy of the kernel's SIMDs from the same stored PMC passes (sha-matched like stored_traffic):
    on gfx950 an fp32 matrix instruction and fp32 VALU work of a SIMD do not overlap (profiles/r04_mfma_valu_overlap_probe.txt),
    so (matrix busy cycles + VALU active cycles) / (SIMDs x launch duration) is the fraction of the launch in which a SIMD issued
    arithmetic at all — the bound a latency chain like this one is held against, beside the matrix-peak fraction."""
    for rnd in ("r05_pmc", "r04_pmc"):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", rnd, "summary.json")))
        except Exception:
            continue
        d = pmc.get(key)
        if pmc.get("kernel_sources_sha") != kernel_sources_sha() or not d:
            continue
        try:
            simd_cycles = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0          # the counter sums the 8 XCDs; 256 CUs x 4 SIMDs
            mfma = d["SQ_VALU_MFMA_BUSY_CYCLES"]
            valu = 4.0 * (d["SQ_ACTIVE_INST_VALU"] - d["SQ_INSTS_MFMA"])   # quad-cycles -> cycles; matrix issues counted once
            return {"mfma_busy_frac": mfma / simd_cycles, "valu_active_frac": valu / simd_cycles,
                    "frac": (mfma + valu) / simd_cycles, "wait_any_frac": d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"],
                    "source": f"profiles/{rnd}/summary.json (kernel sources {kernel_sources_sha()})"}
        except Exception:
            return None
    return None


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
    # BASELINE.md section 3's other lines, each on a bounded sample: the same C port on ONE thread, and the vectorised
    # torch-CPU float32 port (oracle/torch_port.py) on one thread, reference-faithful (2 + 2 evaluations per
    # bridge) and with the backward evaluation reused (1 + 1)
    lines = {}
    try:
        n1 = min(n, 128)
        c_oracle.set_threads(1)
        t1 = time.perf_counter()
        run_c_oracle(built, seeds_np[:n1])
        lines["c_port_1_thread"] = {"value": n1 * K / (time.perf_counter() - t1), "cores": 1,
                                    "sample": f"1 call of {n1} particles x {K} bridges"}
    finally:
        c_oracle.set_threads(threads)
    try:
        from cmcd_amd import synthetic
        from helpers import oracle_target
        from oracle import torch_port
        dim, _, mode, spec = built["params_fixed"]
        cfg = built["cfg"]
        params_np = synthetic.oracle_params(built["unflatten"], built["params_flat"])
        prep = torch_port.Prepared(seeds_np[:n], params_np, dim, K, mode, spec.arch, cfg["model"], oracle_target(cfg),
                                   cfg["eps_schedule"], cfg["grad_clipping"])
        ncpu, was = os.cpu_count() or 1, torch.get_num_threads()
        # (r03 on the MI355X host: one torch thread per logical CPU — 256 — on these 64-wide matrices ran at 6 particle-steps/s:
        # thread hand-over, not arithmetic, against 3.8e5 on ONE thread.  That line was an oversubscription artefact, not a
        # baseline, and is no longer printed; r04: 16 threads 2.6e5 against 3.7e5 on ONE — the vectorised port is timed at the
        # thread count where it is fastest, one)
        for tag, nt, reuse in (("torch_cpu_1_thread", 1, False), ("torch_cpu_1_thread_reuse", 1, True)):
            torch.set_num_threads(nt)
            # bounded samples: a 4-bridge probe of 64 particles gives the rate, the timed sample is then the largest
            # (particles x bridges) prefix of the same batch that fits ~3 s at that rate (128 threads on small matrices can
            # be SLOWER than one: the sample must not be sized for the fast case)
            pm, pk = min(n, 16), min(K, 2)
            torch_port.run(prep, reuse=reuse, max_bridges=1, max_particles=pm)
            t1 = time.perf_counter()
            torch_port.run(prep, reuse=reuse, max_bridges=pk, max_particles=pm)
            per_step = max((time.perf_counter() - t1) / (pm * pk), 1e-9)     # seconds per particle-bridge-step
            budget = 3.0 / per_step                                        # particle-steps that fit the budget
            sm = int(min(n, max(pm, budget / K))) if budget >= pm * K else pm
            sk = int(min(K, max(pk, budget / sm)))
            t1 = time.perf_counter()
            torch_port.run(prep, reuse=reuse, max_bridges=sk, max_particles=sm)
            lines[tag] = {"value": sm * sk / (time.perf_counter() - t1), "cores": nt,
                          "sample": f"{sm} particles x the first {sk} of {K} bridges in one call, float32, PRNG streams drawn "
                                    f"outside the timed call"}
        torch.set_num_threads(was)
    except NotImplementedError as e:
        lines["torch_cpu"] = {"error": str(e)}
    out["other_lines"] = lines
    out["fastest_cpu_value"] = max([out["value"], out["numpy_port_value"]] + [v["value"] for v in lines.values() if "value" in v])
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


def self_launch(n_gpus):
    """`python3 bench.py --gpus N` with N > 1 and no launcher in the environment: run
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <argv>`
    as a child process and return its exit code.  Called before this process has initialised the GPU
    (torch.cuda.device_count() does not); the child's stdout (rank 0's one JSON line) and stderr are inherited."""
    import socket
    import subprocess
    shared = os.environ.get("CMCD_BENCH_SHARED_GPU") == "1"      # test hook: every rank on device 0 (see main)
    have = torch.cuda.device_count()
    if have < (1 if shared else n_gpus):
        print(f"bench.py: --gpus {n_gpus} but {have} GPU(s) visible on this node", file=sys.stderr)
        return 2
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def _strict(o):
    """Strict JSON: non-finite floats become null (json.dumps would print -Infinity / NaN, which is not JSON)."""
    if isinstance(o, dict):
        return {k: _strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v) for v in o]
    if isinstance(o, float) and not math.isfinite(o):
        return None
    return o


class Leg:
    """One workload of the job: `n_global` particles of a synthetic configuration, sharded contiguously over the ranks
    (or `n_global` per rank when weak)."""

    def __init__(self, name, cfg_name, n_global, weak, device, rank, world, over=None):
        from cmcd_amd import mcdboundingmachine as mcdbm
        from cmcd_amd import parallel, synthetic
        self.name, self.cfg_name, self.weak, self.world, self.rank = name, cfg_name, weak, world, rank
        self.b = synthetic.build(cfg_name, device=device, **(over or {}))
        self.dim, self.K, self.mode, self.spec = self.b["params_fixed"]
        if weak:
            self.n_local, self.n_global = n_global, n_global * world
            seeds_np = synthetic.throughput_seeds(n_global, stream=rank)
        else:
            lo, hi = parallel.shard_range(n_global, world, rank)
            self.n_local, self.n_global = hi - lo, n_global
            seeds_np = synthetic.throughput_seeds(n_global, stream=0)[lo:hi]     # every rank draws the same global vector
        self.seeds_np = seeds_np
        self.seeds = torch.from_numpy(seeds_np).to(device)       # resident in HBM before any timed region
        self.full_seeds_np = synthetic.throughput_seeds(n_global, stream=0) if not weak else seeds_np
        self._mcdbm = mcdbm

    def forward(self, seeds=None):
        b = self.b
        return self._mcdbm.bound_forward(self.seeds if seeds is None else seeds, b["params_flat"], b["unflatten"],
                                         b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                         grad_clipping=b["grad_clipping"])

    def kernel_name(self):
        """What the library launched for this leg's last forward (cmcd_last_kernel_name): not re-derived here."""
        from cmcd_amd import _lib
        return _lib.last_kernel_name()


def time_leg(leg, steps, warmup, use_dist, device, world, sharded=True, spinup=0, pipelined=False):
    """W untimed + K timed forward steps of `leg` on this rank's shard: barrier + synchronize on both sides, MAX over ranks.
    (The untimed part ends with one rehearsal of the timed region's own shape — barrier, K steps, barrier — so that the timed
    region is not the process's first region of that shape: r04, 0.191 - 0.202 -> 0.183 - 0.186 ms per step over five interleaved
    pairs of runs with the driver's flags, the kernel's own time unchanged.)
    Multi-GPU: one RCCL all-gather of the 40-byte statistics vector + one merge kernel per step.
    pipelined=False (the headline at every N): the all-gather and the merge of step k are enqueued behind its forward and
    complete, in stream order, before the forward of step k + 1 — the cost of a call that returns the merged scalar.
    pipelined=True (legs.weak_pipelined only): the collective is latency-only, so it is taken off the launch stream: torch's
    process group runs it on its own stream (async_op=True) behind an event on the forward of step k, and the launch stream
    waits for it only after the forward of step k+1 has been enqueued — the statistics of step k are merged one step late,
    every step's all-gather and merge still run inside the timed region (the last one is drained before the closing barrier).
    The timed region contains NO measurement code: the per-launch HIP-event hook (cmcd_profile_enable) is off; the kernel's
    own duration (`kern_ms` / `launches`) comes from a SECOND, separate loop of the same forward calls behind the closing
    barrier, with the hook on and no collective."""
    from cmcd_amd import _lib, parallel
    gathered = [torch.zeros(world * parallel.NSTATS, dtype=torch.float64, device=device) for _ in range(2)]
    pending = []
    gather = use_dist and sharded

    def drain():
        work, buf, _ = pending.pop()
        work.wait()
        return parallel.merge_stats(buf.view(world, parallel.NSTATS))

    def step(k):
        losses, z, stats = leg.forward()
        if gather and not pipelined:
            dist.all_gather_into_tensor(gathered[0], stats)      # same stream order as the forward: done before step k + 1
            stats = parallel.merge_stats(gathered[0].view(world, parallel.NSTATS))
        elif gather:
            work = dist.all_gather_into_tensor(gathered[k & 1], stats, async_op=True)
            merged = drain() if pending else None
            pending.append((work, gathered[k & 1], stats))
            stats = merged
        return losses, z, stats

    def barrier():
        if use_dist and sharded:
            dist.barrier()
        torch.cuda.synchronize()

    # device spin-up (untimed, in front of the W warm-up steps without a synchronisation in between): ~0.1 s of launches
    # so that the GPU's clocks have settled even when the caller asks for a handful of steps (20 steps = 5 ms)
    # the collector runs BEFORE the spin-up: between the opening barrier and the first timed launch nothing may idle the
    # GPU (tools/probes/post_sync_ramp.py: a 10 ms gap drops the clocks for the next > 40 calls, 203 -> 218 us per call;
    # r02's 20-step driver run read 0.2245 ms per step with a gc.collect() sitting in exactly that gap)
    import gc
    gc.collect()
    gc_was = gc.isenabled()
    gc.disable()          # a 20-step timed region is 4 ms: one collector pause on the launching thread would be 10 % of it
    _lib.profile_enable(False)
    for k in range(spinup + warmup):
        step(k)
    if pending:
        drain()
    if os.environ.get("CMCD_BENCH_REHEARSE", "1") == "1":
        # one untimed rehearsal of the timed region's own shape (barrier, K steps, barrier): the first synchronise of a process
        # and the first launches behind it run on cold host paths, and the idle gap they leave is long enough for the clocks to
        # step down (tools/probes/post_sync_ramp.py) — the every-call-prep loop, timed second, used to beat the headline loop,
        # timed first, in two runs of three.  Nothing of the timed region changes.
        barrier()
        for k in range(steps):
            step(k)
        if pending:
            drain()
    barrier()
    t0 = time.perf_counter()
    for k in range(steps):
        losses, z, stats = step(k)
    if pending:
        stats = drain()   # global statistics of the last step
    barrier()
    elapsed = time.perf_counter() - t0
    # kernel duration: a separate loop of the same forward calls (no collective), HIP events around every trajectory-kernel
    # launch on the launch stream — outside the region timed above.  The loop runs inside fixed_parameters(): with the prep
    # launch in front of it, the first event of a pair fires when the PREP kernel retires, and the pair then holds the
    # dispatch gap between two dependent kernels on top of the trajectory kernel's own time (r05, under rocprofv3: 194.8 us
    # by events with the prep launch in every call, 185.4 us without, 186.4 us = the profiler's own average of that kernel
    # over both loops, profiles/r05_c_kernel_stats_bench_forward_only.csv).  The kernel's duration does not depend on who
    # formed its tables; `value` / `ms_per_step` above stay the default path with the prep launch in every call.
    from cmcd_amd import mcdboundingmachine as _mcdbm
    with _mcdbm.fixed_parameters():
        leg.forward()
        _lib.profile_enable(True)
        for k in range(min(steps, 500)):      # the hook holds 512 event pairs
            leg.forward()
        torch.cuda.synchronize()
        kern_ms, launches = _lib.profile_collect()
        _lib.profile_enable(False)
    if gc_was:
        gc.enable()
    if use_dist and sharded:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return dict(elapsed=elapsed, kern_ms=kern_ms, launches=launches, losses=losses, stats=stats, kernel=_lib.last_kernel_name())


def leg_report(leg, t, steps):
    """The per-leg object of the JSON line."""
    from cmcd_amd import parallel
    cfg = leg.b["cfg"]
    units = leg.n_global * leg.K
    f_alg, f_survey = flops_per_particle_step(cfg, leg.dim, leg.spec.width)
    if t["launches"] == 0:
        kern_s = t["elapsed"] / steps
    else:
        kern_s = t["kern_ms"] / 1e3 / t["launches"]
    fin = parallel.finalize(t["stats"], leg.n_global)
    var = float(fin["var"])
    return {
        "workload": leg.cfg_name, "scaling": "weak" if leg.weak else "strong", "global_particles": leg.n_global,
        "particles_per_gpu": leg.n_local, "nbridges": leg.K, "value": units * steps / t["elapsed"],
        "ms_per_step": t["elapsed"] / steps * 1e3, "kernel": t.get("kernel") or leg.kernel_name(), "kernel_ms": kern_s * 1e3,
        "kernel_frac_of_fp32_peak": leg.n_local * leg.K * f_alg / kern_s / 1e12 / PEAK_FP32_TFLOPS,
        "elbo": float(-fin["mean"]), "ln_z": float(fin["ln_z"]), "loss_var": var, "n_finite": float(fin["n_finite"]),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 0.1 s of timed work — 20 steps (7 ms) end before the GPU's clocks have settled
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--spinup", type=int, default=-1,
                    help="untimed launches before the warm-up steps (clock settling); -1 = as many as fill --spinup-seconds")
    ap.add_argument("--spinup-seconds", type=float, default=0.35,
                    help="device time of the untimed spin-up: a count-based spin-up (300 launches = 0.07 s) left the clocks "
                         "of a fresh box unsettled for a 20-step run (r02: 0.202 ms against 0.191 ms for 300 steps)")
    ap.add_argument("--config", default=None, help="name in cmcd_amd.synthetic.CONFIGS")
    ap.add_argument("--particles", type=int, default=None, help="override N per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train-step", action="store_true",
                    help="with --config: also time value-and-gradient of that configuration's own training loss (training_step)")
    ap.add_argument("--forward-only", action="store_true",
                    help="skip the vargrad / training_step legs (their trajectory-keeping forward launches would be averaged "
                         "into the same kernel name by rocprofv3 --stats)")
    ap.add_argument("--no-legs", action="store_true", help="N = 1: skip the strong_sharded_cfg4 leg (profiling runs)")
    ap.add_argument("--cpu-particles", type=int, default=2000)
    ap.add_argument("--saturated", type=int, default=1 << 18,
                    help="also time a saturating batch of this many particles (0 = skip)")
    args = ap.parse_args()
    if os.environ.get("CMCD_BENCH_TRACE_AFTER"):     # diagnostics: dump every thread's stack if the run is still going then
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CMCD_BENCH_TRACE_AFTER"]), exit=True)

    if args.gpus > 1 and "RANK" not in os.environ:
        # typed as `python3 bench.py --gpus N` (no launcher): start the one-process-per-GPU job as a CHILD, before anything in
        # this process has touched the GPU, relay its output and leave with its return code
        raise SystemExit(self_launch(args.gpus))

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
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if use_dist:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rank 0 times the CPU baseline (tens of seconds of host work) while the other ranks wait at a barrier
        patience = datetime.timedelta(minutes=30)
        if shared_gpu:
            dist.init_process_group("gloo", timeout=patience)
        else:
            dist.init_process_group("nccl", device_id=device, timeout=patience)

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
    n_named = args.particles or synthetic.CONFIGS[name]["N"]
    weak = Leg("weak", name, n_named, True, device, rank, world, over)
    b, cfg = weak.b, weak.b["cfg"]
    dim, K, mode, spec = b["params_fixed"]
    n = weak.n_local
    seeds_np, seeds = weak.seeds_np, weak.seeds
    forward = weak.forward

    legs = {}
    if args.spinup < 0:
        # time-based spin-up: a first synchronised batch gives the step time, the spin-up is then that many launches
        for _ in range(10):
            forward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(40):
            forward()
        torch.cuda.synchronize()
        est = max((time.perf_counter() - t0) / 40, 1e-5)
        args.spinup = int(min(max(args.spinup_seconds / est, 50), 5000))
        if use_dist:    # every rank must issue the SAME number of steps (each carries a collective): take the largest count
            cnt = torch.tensor([args.spinup], dtype=torch.int64, device=device)
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX)
            args.spinup = int(cnt.item())
    # HEADLINE = the default product path: every step is one plain compute_bound call (cmcd_bound_forward: prep launch +
    # trajectory kernel + statistics), what a training step or any caller without further statements pays.
    mcdbm.PREP_CACHE = False
    tw = time_leg(weak, args.steps, args.warmup, use_dist, device, world, spinup=args.spinup)
    legs["weak"] = leg_report(weak, tw, args.steps)
    # Side leg, at every N: the same loop inside `with mcdbm.fixed_parameters():` — an evaluation loop that STATES its
    # parameters are unchanged (the shape of the reference's 30 loss_fn calls on one params_flat,
    # /root/reference/src/opt.py:185-190), so all but the first call skip the prep launch (cmcd_bound_forward_prepared).
    # Never the headline (r04 reported it as `value`; no default caller of this package takes that path).
    calls0 = dict(mcdbm.PREP_CALLS)
    mcdbm.PREP_CACHE = True
    try:
        tprep = time_leg(weak, args.steps, args.warmup, use_dist, device, world, spinup=min(args.spinup, 300))
    finally:
        mcdbm.PREP_CACHE = False
    legs["weak_prepared"] = leg_report(weak, tprep, args.steps)
    legs["weak_prepared"]["what"] = ("the headline loop inside fixed_parameters(): per-parameter tables of the first call reused, "
                                     "prep launch skipped; results bit-identical (tests/test_gpu_fullsize.py)")
    legs["weak_prepared"]["prepared_calls"] = mcdbm.PREP_CALLS["prepared"] - calls0["prepared"]
    elapsed, kern_ms, launches, losses, stats = tw["elapsed"], tw["kern_ms"], tw["launches"], tw["losses"], tw["stats"]
    default_workload = name == synthetic.NORTH_STAR and not args.particles

    head_leg, head_t = weak, tw
    collective = None
    if world > 1:
        # the same batch with the statistics all-gather taken off the critical path (merged one step late)
        tp = time_leg(weak, args.steps, args.warmup, use_dist, device, world, spinup=0, pipelined=True)
        legs["weak_pipelined"] = leg_report(weak, tp, args.steps)
        # the collective alone: all-gather of the 5 doubles + merge kernel, back to back on the launch stream, HIP events
        buf = torch.zeros(world * parallel.NSTATS, dtype=torch.float64, device=device)
        st5 = tw["stats"].clone()
        for _ in range(20):
            dist.all_gather_into_tensor(buf, st5)
            parallel.merge_stats(buf.view(world, parallel.NSTATS))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200
        dist.barrier()
        e0.record()
        for _ in range(reps):
            dist.all_gather_into_tensor(buf, st5)
            parallel.merge_stats(buf.view(world, parallel.NSTATS))
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        tt = torch.tensor([us], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        collective = {"what": "all_gather_into_tensor of 5 float64 per rank + fixed-order merge kernel, stream-ordered",
                      "backend": dist.get_backend(), "us_per_call": float(tt.item()), "bytes_per_rank": 40}
    if default_workload and not args.no_legs:
        # strong scaling on the named batch (2000 / N per rank) and on configs[3] (16000 x 132-wide net split over the ranks)
        strong = Leg("strong_named", name, n_named, False, device, rank, world) if world > 1 else None
        cfg4 = Leg("strong_sharded_cfg4", CFG4, synthetic.CONFIGS[CFG4]["N"], False, device, rank, world)
        for leg in (strong, cfg4):
            if leg is None:
                continue
            if leg.n_global < world:
                raise SystemExit(f"{leg.name}: fewer particles than ranks")
            t = time_leg(leg, args.steps, args.warmup, use_dist, device, world,
                         spinup=args.spinup if leg.n_local <= 4096 else 0)    # short launches only: clock settling
            legs[leg.name] = leg_report(leg, t, args.steps)
            if world > 1:
                # the same job's single-GPU reference: rank 0 runs the UN-split batch alone, the others wait at the barrier
                if rank == 0:
                    full = Leg(leg.name + "_single", leg.cfg_name, leg.n_global, True, device, 0, 1)
                    ts = time_leg(full, max(args.steps // 4, 3), max(args.warmup // 4, 1), False, device, 1, sharded=False)
                    single_ms = ts["elapsed"] / max(args.steps // 4, 3) * 1e3
                    legs[leg.name]["single_gpu_ms"] = single_ms
                    legs[leg.name]["speedup_vs_single_gpu"] = single_ms / legs[leg.name]["ms_per_step"]
                    del full
                dist.barrier()
            if leg is cfg4:
                cfg4_t = t
        if world == 1:
            legs["strong_named"] = dict(legs["weak"], scaling="strong")      # N = 1: the same measurement
        # configs[3]'s training step on its shard: VarGrad value + gradient, statistics all-gather ("RCCL log-w
        # all-reduce") between forward and gradient, one all-reduce of grad_flat
        try:
            gl = parallel.make_sharded_grad_and_loss("MCD_CAIS_var_sn", eps_schedule=cfg4.b["eps_schedule"],
                                                     grad_clipping=cfg4.b["grad_clipping"])
            gs = torch.from_numpy(cfg4.full_seeds_np).to(device)
            gargs = (gs, cfg4.b["params_flat"], cfg4.b["unflatten"], cfg4.b["params_fixed"], cfg4.b["target"])
            reps = max(min(args.steps, 20) // 2, 2)
            for _ in range(2):
                gl(*gargs)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(reps):
                gl(*gargs)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg0) / reps
            if use_dist:
                tt = torch.tensor([tg], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tg = float(tt.item())
            legs["strong_sharded_cfg4"]["train_step_ms"] = tg * 1e3
            legs["strong_sharded_cfg4"]["train_value"] = cfg4.n_global * cfg4.K / tg
            if world > 1:
                # the same job's single-GPU training step: rank 0 runs the un-split VarGrad value + gradient alone
                if rank == 0:
                    sargs = gargs + ()
                    skw = dict(eps_schedule=cfg4.b["eps_schedule"], grad_clipping=cfg4.b["grad_clipping"])
                    mcdbm.compute_log_var_grad(*sargs, **skw)
                    torch.cuda.synchronize()
                    ts0 = time.perf_counter()
                    for _ in range(2):
                        mcdbm.compute_log_var_grad(*sargs, **skw)
                    torch.cuda.synchronize()
                    single_train_ms = (time.perf_counter() - ts0) / 2 * 1e3
                    legs["strong_sharded_cfg4"]["train_step_single_gpu_ms"] = single_train_ms
                    legs["strong_sharded_cfg4"]["train_step_speedup_vs_single_gpu"] = single_train_ms / (tg * 1e3)
                dist.barrier()
        except NotImplementedError as e:
            legs["strong_sharded_cfg4"]["train_step_error"] = str(e)

    # ---- headline: the named batch per GPU at every N (weak scaling), in the PER-CALL form: every step's statistics
    # all-gather and merge complete in stream order inside the step (legs.weak) — the cost of one compute_bound call that
    # returns the merged scalar, which is what SURVEY.md section 8d's metric times.  The throughput form (all-gather of step k
    # beside the forward of step k + 1) stays beside it in legs.weak_pipelined, `collective.us_per_call` is the collective alone,
    # and `strong_scaling` lifts the strong legs' same-job speed-ups to the top level.
    hl = legs["weak"]
    hcfg = head_leg.b["cfg"]
    f_alg, f_survey = flops_per_particle_step(hcfg, head_leg.dim, head_leg.spec.width)
    value, n, K, dim = hl["value"], head_leg.n_local, head_leg.K, head_leg.dim
    kern_s = hl["kernel_ms"] / 1e3
    achieved = n * K * f_alg / kern_s / 1e12
    launches = head_t["launches"] or args.steps
    traffic, traffic_src = (None, None)
    if head_leg is weak and default_workload:
        traffic, traffic_src = stored_traffic("coop_kernel")

    result = {
        "metric": "bridge-steps*particles/sec", "value": value, "unit": "bridge-steps*particles/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup": args.spinup,
        "rehearsal_steps": args.steps if os.environ.get("CMCD_BENCH_REHEARSE", "1") == "1" else 0,   # untimed, see time_leg
        "ms_per_step": hl["ms_per_step"], "higher_is_better": True, "scaling": hl["scaling"],
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": head_leg.cfg_name, "model": hcfg["model"], "boundmode": hcfg["boundmode"],
                   "particles_per_gpu": n, "nbridges": K, "nn_arch": hcfg["nn_arch"], "dim": dim,
                   "global_particles": head_leg.n_global,
                   "parallelism": f"particles sharded x{world}, stats all-gather"},
        "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_unit": "bytes/launch (PMC)",
                     "traffic_source": (f"profiles/{traffic_src}/summary.json (kernel sources {kernel_sources_sha()})"
                                        if traffic is not None else "none for this kernel build"),
                     "kernel": hl["kernel"], "kernel_ms": kern_s * 1e3, "launches": launches,
                     "flop_per_particle_step": f_alg, "flop_per_particle_step_survey": f_survey,
                     "achieved_survey_flops": n * K * f_survey / kern_s / 1e12,
                     "state_bytes_gbs": n * K * (8 * dim + 8) / kern_s / 1e9,
                     "issue_occupancy": stored_issue_occupancy("coop_kernel") if traffic is not None else None},
        "elbo": hl["elbo"], "ln_z": hl["ln_z"], "n_finite": hl["n_finite"],
        "legs": legs,
    }
    result["headline_leg"] = "weak"
    if world > 1:
        result["collective"] = collective
        result["value_per_call"] = legs["weak"]["value"]
        result["value_pipelined"] = legs["weak_pipelined"]["value"]
        # the north_star's strong-scaling figures, measured in this job (one batch split over the ranks against the same batch
        # on rank 0's GPU alone): named = 2000 particles x 256 bridges (latency-bound by construction), cfg4 = BASELINE
        # configs[3] forward, cfg4_train_step = its sharded VarGrad value + gradient + all-reduce
        sl = legs.get("strong_sharded_cfg4", {})
        result["strong_scaling"] = {"named": legs.get("strong_named", {}).get("speedup_vs_single_gpu"),
                                    "cfg4": sl.get("speedup_vs_single_gpu"),
                                    "cfg4_train_step": sl.get("train_step_speedup_vs_single_gpu"),
                                    "n_gpus": world, "what": "single-GPU time / sharded time, same job"}
        result["scaling_note"] = ("headline = the named batch (2000 particles) per GPU with the statistics all-gather + merge "
                                  "completed inside every step (legs.weak: the per-call latency of a compute_bound that returns "
                                  "the merged scalar); legs.weak_pipelined = the throughput form (all-gather of step k beside the "
                                  "forward of step k + 1), collective.us_per_call = the collective alone; strong_scaling = "
                                  "legs.strong_named / legs.strong_sharded_cfg4 (ONE batch split over the ranks) against the same "
                                  "job's single-GPU time")
    # untrained net at init_sigma = 60: some particles leave float32 range exactly as in the reference (parity.inf_set_equal),
    # so the plain mean is -inf; the mean over this rank's finite particles is reported beside it
    lfin = losses[torch.isfinite(losses)]
    result["elbo_finite_particles"] = float(-lfin.double().mean()) if lfin.numel() else None
    n, K, dim = weak.n_local, weak.K, weak.dim
    f_alg, f_survey = flops_per_particle_step(cfg, dim, spec.width)
    kern_s = legs["weak"]["kernel_ms"] / 1e3

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

    if rank == 0 and world == 1 and default_workload and not args.no_legs:
        # The reference's 2nd-order mode (config.boundmode = "MCD_CAIS_UHA_sn") on the SAME batch shape:
        # an extra measured line, not the headline (BASELINE.json's metric is quoted on MCD_CAIS_sn).  Two network
        # evaluations per bridge on concat(z, rho); unit = the headline's (particles x bridge steps per second).
        from cmcd_amd import mcdboundingmachine as _m
        ub = synthetic.build(name, device=device, boundmode="MCD_CAIS_UHA_sn", init_eps=0.2,
                             init_gamma=2.0, init_sigma=15.0)
        useeds = torch.from_numpy(synthetic.throughput_seeds(n, stream=0)).to(device)
        uargs = (useeds, ub["params_flat"], ub["unflatten"], ub["params_fixed"], ub["target"])
        for _ in range(50):
            _m.bound_forward(*uargs)
        torch.cuda.synchronize()
        ureps = 200
        tu0 = time.perf_counter()
        for _ in range(ureps):
            ur = _m.bound_forward(*uargs)
        torch.cuda.synchronize()
        tu = (time.perf_counter() - tu0) / ureps
        result["second_order"] = {"workload": f"{weak.cfg_name}:MCD_CAIS_UHA_sn", "particles": n, "nbridges": ub["params_fixed"][1],
                                  "ms_per_step": tu * 1e3, "value": n * ub["params_fixed"][1] / tu, "steps": ureps,
                                  "kernel": _lib.last_kernel_name(), "n_finite": int(torch.isfinite(ur[0]).sum())}
        # algorithmic work of a 2nd-order particle-bridge-step: TWO network evaluations on concat(z, rho) (first layer 2 d
        # wide, time path folded as for the headline), one target gradient, ~24 flops per state dimension of (z, rho)
        umac = (2 * dim * 64 + 64 * 64 + 64 * dim) if cfg["nn_arch"] == "dds" else \
               (2 * dim * (2 * dim + cfg["emb_dim"]) + (2 * dim + cfg["emb_dim"]) ** 2 + (2 * dim + cfg["emb_dim"]) * dim)
        uf = 2 * 2 * umac + {"gmm": 200, "funnel": 60, "many_gmm": 800}.get(cfg["model"], 0) + 24 * 2 * dim
        result["second_order"]["flop_per_particle_step"] = uf
        result["second_order"]["achieved_tflops_per_call"] = result["second_order"]["value"] * uf / 1e12
        result["second_order"]["frac_of_fp32_peak_per_call"] = result["second_order"]["value"] * uf / 1e12 / PEAK_FP32_TFLOPS
        if not args.forward_only:
            for _ in range(3):
                _m.compute_bound_grad(*uargs)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(20):
                _m.compute_bound_grad(*uargs)
            torch.cuda.synchronize()
            result["second_order"]["value_and_grad_ms"] = (time.perf_counter() - tg0) / 20 * 1e3

    if cfg["model"] == "lgcp":
        IN = dim + cfg["emb_dim"]
        if n >= 225:     # cmcd_common.h: kLgcpWideMin
            # wide batches (cmcd_lgcp_wide.hip; the reference's evaluation batches): every particle shares ONE weight pass per
            # evaluation, intensity 2 n FLOP per 4 weight bytes >> the machine balance => matrix-pipe bound
            fl = 2.0 * (dim * dim + 2 * dim * IN + IN * IN) * (K + 1) * n
            result["roofline"].update({"bound": "mfma", "achieved": fl / kern_s / 1e12, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                                       "frac": fl / kern_s / 1e12 / PEAK_FP32_TFLOPS, "traffic": None,
                                       "flop_per_call": fl, "kernel": legs["weak"]["kernel"]})
        else:
            # weight-bandwidth bound: every evaluation streams K^-1 and the three weight matrices once
            # per pass of <= 32 particles (ALGORITHMIC bytes: the un-packed matrices, whatever the kernels re-read)
            wbytes = 4.0 * (dim * dim + 2 * dim * IN + IN * IN) * (K + 1) * -(-n // 32)
            # measured L2 <-> fabric bytes of the three GEMM launches of one evaluation (tools/probes/round_end_r04.sh: separate
            # rocprofv3 --pmc passes, FETCH_SIZE doubled per the gfx950 note), x (K + 1) evaluations x passes
            traffic = None
            for rnd in ("r05_pmc", "r04_pmc", "r03_pmc", "r02_pmc"):
                try:   # only a summary measured on this build of the kernels counts (kernel_sources_sha)
                    pm = json.load(open(os.path.join(ROOT, "profiles", rnd, "lgcp_summary.json")))
                    if pm.get("kernel_sources_sha") == kernel_sources_sha("lgcp") and dim == 1600 and IN == 1620:
                        traffic = sum(v["hbm_bytes_per_launch"] for v in pm.values() if isinstance(v, dict)) * (K + 1) * -(-n // 32)
                        break
                except Exception:
                    pass
            result["roofline"].update({"bound": "hbm", "achieved": wbytes / kern_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": wbytes / kern_s / 1e9 / PEAK_HBM_GBS, "traffic": traffic,
                                       "traffic_unit": "bytes/call (PMC)", "weight_bytes_per_call": wbytes,
                                       "kernel": legs["weak"]["kernel"]})

    if rank == 0 and world == 1 and name == synthetic.NORTH_STAR and not args.forward_only:
        # value-and-gradient of the VarGrad loss on the same batch (boundmode MCD_CAIS_var_sn, same net/target)
        try:
            bv = synthetic.build(name, device=device, boundmode="MCD_CAIS_var_sn", **over)
            gargs = (seeds, bv["params_flat"], bv["unflatten"], bv["params_fixed"], bv["target"])
            gkw = dict(eps_schedule=bv["eps_schedule"], grad_clipping=bv["grad_clipping"])
            for _ in range(10):
                mcdbm.compute_log_var_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(40):
                mcdbm.compute_log_var_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg0) / 40
            result["vargrad"] = {"ms_per_value_and_grad": tg * 1e3, "value": n * K / tg,
                                 "unit": "bridge-steps*particles/s (forward + backward)"}
        except NotImplementedError as e:
            result["vargrad"] = {"error": str(e)}
        # value-and-gradient of the north-star's own training loss (MCD_CAIS_sn, reparameterised gradient:
        # forward with stored trajectory + reverse sweep) on the same batch
        try:
            gargs = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
            gkw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
            for _ in range(10):
                mcdbm.compute_bound_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(40):
                mcdbm.compute_bound_grad(*gargs, **gkw)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg0) / 40
            result["training_step"] = {"ms_per_value_and_grad": tg * 1e3, "value": n * K / tg,
                                       "unit": "bridge-steps*particles/s (forward + reverse sweep)"}
        except NotImplementedError as e:
            result["training_step"] = {"error": str(e)}

    if rank == 0 and world == 1 and args.train_step and name != synthetic.NORTH_STAR:
        # --train-step: value-and-gradient of THIS configuration's own training loss on the same batch (the VarGrad loss for
        # MCD_CAIS_var_sn, the reparameterised gradient otherwise) — the per-configuration lines of profiles/*_all_configs.jsonl
        try:
            fn = mcdbm.compute_log_var_grad if b["params_fixed"][2] == "MCD_CAIS_var_sn" else mcdbm.compute_bound_grad
            gargs = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
            gkw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
            for _ in range(5):
                fn(*gargs, **gkw)
            torch.cuda.synchronize()
            tg0 = time.perf_counter()
            for _ in range(20):
                fn(*gargs, **gkw)
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg0) / 20
            result["training_step"] = {"ms_per_value_and_grad": tg * 1e3, "value": n * K / tg, "loss": fn.__name__,
                                       "unit": "bridge-steps*particles/s (forward + reverse sweep)"}
        except NotImplementedError as e:
            result["training_step"] = {"error": str(e)}

    if not args.no_cpu_baseline:
        # at EVERY N: rank 0 times the CPU baseline on its own shard's seeds (bounded sample, host cores of this box) and checks
        # its GPU losses against it; the other ranks wait at the barrier (process-group timeout raised in main's init)
        if rank == 0:
            base, parity = cpu_baseline(b, seeds_np, losses.cpu().numpy(), args.cpu_particles)
            result["cpu_baseline"] = base
            result["parity"] = parity
            result["speedup_vs_cpu"] = value / base["fastest_cpu_value"]     # against the FASTEST of the CPU lines
        if use_dist:
            dist.barrier()

    if rank == 0:
        print(json.dumps(_strict(result)))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
