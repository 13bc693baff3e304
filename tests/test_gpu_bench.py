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
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["scaling"] == "weak" and r["value"] > 1e8
    assert r["roofline"]["frac"] > 0 and r["config"]["workload"]


def test_bench_two_ranks_share_the_gpu(hip_lib):
    env = dict(os.environ, CMCD_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29653", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--saturated", "0"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    r = _line(out.stdout)
    assert r["n_gpus"] == 2 and r["scaling"] == "weak"
    # whole-job units: both ranks' particles counted
    # (two processes time-slice one GPU and gather over gloo through the host here: the rate itself means nothing)
    assert r["config"]["global_particles"] == 4000 and r["value"] > 1e6
