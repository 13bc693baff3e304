"""bench.py's contract on a GPU box: one JSON line with the fields the driver reads, at N = 1 and — through the
test hook that puts every rank on device 0 over gloo — on the N = 2 code path (all-gather of the statistics, barrier,
max-over-ranks timing, whole-job value)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

FIELDS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
          "dtype", "data", "config", "roofline")


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def test_bench_single_gpu_line(hip_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                          "--saturated", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _line(out.stdout)
    for f in FIELDS:
        assert f in r, f
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["scaling"] == "weak" and r["value"] > 1e8 and r["headline_leg"] == "weak"
    assert "torch_cpu_all_cores" not in json.dumps(r)
    assert r["roofline"]["frac"] > 0 and r["config"]["workload"] == "many_gmm_n2000_k256_dds"
    # PMC-derived figures are reported for the build they were measured on only (sha of the kernel sources), never as current
    occ = r["roofline"]["issue_occupancy"]
    assert (occ is None) == (r["roofline"]["traffic"] is None)
    if occ is not None:
        assert 0.0 < occ["mfma_busy_frac"] < occ["frac"] < 1.0
    # the N = 1 points of the scaling legs ride in the same line
    legs = r["legs"]
    assert set(legs) == {"weak", "weak_prepared", "strong_named", "strong_sharded_cfg4"}
    assert legs["weak"]["value"] == pytest.approx(r["value"]) and legs["strong_named"]["global_particles"] == 2000
    # the headline is the default path (prep launch in every call); the fixed_parameters() loop is a side leg and skipped it
    assert legs["weak_prepared"]["prepared_calls"] >= 3 and legs["weak_prepared"]["value"] > 1e8
    c4 = legs["strong_sharded_cfg4"]
    assert c4["workload"] == "many_gmm_var_n16000_k256" and c4["global_particles"] == 16000 and c4["value"] > 1e8
    assert c4["train_step_ms"] > c4["ms_per_step"]
    # the reference's 2nd-order mode on the same batch shape rides along as an extra measured line
    so = r["second_order"]
    assert so["workload"] == "many_gmm_n2000_k256_dds:MCD_CAIS_UHA_sn" and so["value"] > 1e8 and so["particles"] == 2000
    assert so["kernel"] == "uha_coop_kernel<8-particle tiles>" and so["n_finite"] > 1000
    assert so["value_and_grad_ms"] > so["ms_per_step"]


def test_bench_two_ranks_share_the_gpu(hip_lib):
    env = dict(os.environ, CMCD_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--saturated", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = _line(out.stdout)
    # the headline is ONE workload at every N: the named batch per GPU (weak scaling) in the PER-CALL form (all-gather +
    # merge completed inside every step, legs.weak); the throughput form rides beside it in legs.weak_pipelined
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["config"]["workload"] == "many_gmm_n2000_k256_dds"
    assert r["config"]["global_particles"] == 4000 and r["config"]["particles_per_gpu"] == 2000 and r["value"] > 1e6
    assert r["collective"]["us_per_call"] > 0 and r["collective"]["bytes_per_rank"] == 40
    legs = r["legs"]
    assert set(legs) == {"weak", "weak_prepared", "weak_pipelined", "strong_named", "strong_sharded_cfg4"}
    assert legs["weak"]["value"] == pytest.approx(r["value"]) and r["headline_leg"] == "weak"
    assert r["ms_per_step"] == pytest.approx(legs["weak"]["ms_per_step"])
    assert r["value_per_call"] == pytest.approx(r["value"]) and r["value_pipelined"] == pytest.approx(legs["weak_pipelined"]["value"])
    assert legs["weak_pipelined"]["global_particles"] == 4000
    # the north_star's strong-scaling figures at the top level of the line
    ss = r["strong_scaling"]
    assert ss["n_gpus"] == 2 and ss["named"] == pytest.approx(legs["strong_named"]["speedup_vs_single_gpu"])
    assert ss["cfg4"] == pytest.approx(legs["strong_sharded_cfg4"]["speedup_vs_single_gpu"]) and ss["cfg4_train_step"] > 0
    # whole-job units: both ranks' particles counted
    # (two processes time-slice one GPU and gather over gloo through the host here: the rates themselves mean nothing)
    assert legs["weak"]["global_particles"] == 4000 and legs["weak"]["particles_per_gpu"] == 2000
    assert legs["strong_named"]["global_particles"] == 2000 and legs["strong_named"]["particles_per_gpu"] == 1000
    for k in ("strong_named", "strong_sharded_cfg4"):
        assert legs[k]["single_gpu_ms"] > 0 and legs[k]["speedup_vs_single_gpu"] > 0
    assert legs["strong_sharded_cfg4"]["global_particles"] == 16000 and legs["strong_sharded_cfg4"]["particles_per_gpu"] == 8000
    assert legs["strong_sharded_cfg4"]["train_step_ms"] > 0
    assert r["roofline"]["kernel"].startswith("coop_kernel<8-particle tiles")       # asked of the library, not re-derived


def test_bench_plain_command_starts_its_own_ranks(hip_lib):
    """`python3 bench.py --gpus 2 ...` as typed — no launcher on the command line, the form of the driver's N = 1 command: bench.py
    starts torch.distributed.run itself as a child process (before touching the GPU) and the one JSON line is complete at N > 1:
    roofline AND cpu_baseline / parity (rank 0, the other rank waits), the collective and the strong-scaling figures."""
    env = dict(os.environ, CMCD_BENCH_SHARED_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = _line(out.stdout)
    for f in FIELDS:
        assert f in r, f
    assert r["n_gpus"] == 2 and r["steps"] == 5 and r["warmup"] == 2 and r["config"]["global_particles"] == 4000
    assert r["roofline"]["frac"] > 0 and r["roofline"]["kernel"].startswith("coop_kernel<8-particle tiles")
    cb = r["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "sample" in cb
    assert r["parity"]["inf_set_equal"] and r["parity"]["elbo_abs_err"] <= 1e-3 and r["parity"]["lnz_abs_err"] <= 1e-3
    assert r["collective"]["us_per_call"] > 0
    ss = r["strong_scaling"]
    assert ss["n_gpus"] == 2 and ss["named"] > 0 and ss["cfg4"] > 0 and ss["cfg4_train_step"] > 0
    assert "weak_prepared" in r["legs"] and r["headline_leg"] == "weak"


def test_bench_refuses_more_ranks_than_gpus(hip_lib):
    """Without the shared-GPU test hook a one-GPU box cannot run --gpus 2: a clear refusal, not a crash of a rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("CMCD_BENCH_SHARED_GPU", "RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode == 2 and "GPU(s) visible" in out.stderr


def test_bench_single_rank_over_rccl(hip_lib):
    """The `nccl` (= RCCL) branch of bench.py on a one-GPU box: a single-rank process group — RCCL initialisation, the
    per-step all-gather of the statistics, the all-reduce of the timings and the barriers all go through the library the
    8-GPU run uses (two ranks cannot share one device under RCCL, so the two-rank test above runs over gloo)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29657", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--saturated", "0", "--no-legs"]
    env = {k: v for k, v in os.environ.items() if k != "CMCD_BENCH_SHARED_GPU"}
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = _line(out.stdout)
    assert r["n_gpus"] == 1 and r["config"]["workload"] == "many_gmm_n2000_k256_dds" and r["value"] > 1e8


@pytest.mark.parametrize("cfg,loss", [("gmm_n300_k8", "compute_bound_grad"), ("lgcp_n20_k128", "compute_bound_grad")])
def test_bench_config_line_carries_its_training_step(hip_lib, cfg, loss):
    """`--config C --train-step` (what profiles/*_all_configs.jsonl is made of): the configuration's forward line plus value +
    gradient of its own training loss on the same batch."""
    # (20 steps: a 3-step timed region of a 25 us call is mostly the closing synchronisation)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "20", "--warmup", "3",
                          "--no-cpu-baseline", "--saturated", "0", "--no-legs", "--train-step"], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    r = _line(out.stdout)
    assert r["config"]["workload"] == cfg and r["value"] > 0
    ts = r["training_step"]
    assert ts["loss"] == loss and ts["ms_per_value_and_grad"] > r["roofline"]["kernel_ms"]
