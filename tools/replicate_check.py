"""Runs the reference's replicate command lines for every row of tests/golden/reference_notebook_tables.json that is asked
for and prints this build's numbers next to the stored ones (the GPU test tests/test_gpu_reference_tables.py asserts a
25-second subset; this tool is for the long rows — lgcp trains for minutes).

usage (GPU box): python tools/replicate_check.py lgcp:MCD_CAIS_sn:8 lgcp:MCD_ULA_sn:8 funnel:MCD_CAIS_sn:256 [--seeds 1,2,3]
                 (--seeds=1,2,3 works too; --iters N overrides the row's iteration count for a quick look)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmcd_amd import main as cli  # noqa: E402

TABLES = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_notebook_tables.json")))


def run(model, mode, k, seed, ref=None, iters=None, init_eps=None):
    hp = dict(TABLES[model]["hparams"])
    if ref and "lr" in ref:      # a row with its own learning rate (LR_DICT of /root/reference/src/configs/base.py:5-63)
        hp["lr"] = ref["lr"]
    if iters:
        hp["iters"] = iters
    if init_eps is not None:
        hp["init_eps"] = init_eps
    argv = ["--config.boundmode", mode, "--config.model", model, "--config.N", str(hp["N"]), "--config.emb_dim",
            str(hp["emb_dim"]), "--config.init_sigma", str(hp["init_sigma"]), "--config.iters", str(hp["iters"]),
            "--config.n_samples", str(hp["n_samples"]), "--config.nbridges", str(k), "--config.seed", str(seed),
            "--config.train_vi"]
    argv += ["--config.pretrain_mfvi", "--config.mfvi_iters", str(hp["mfvi_iters"])] if hp["pretrain_mfvi"] else ["--noconfig.pretrain_mfvi"]
    argv += ["--config.train_eps"] if hp["train_eps"] else ["--noconfig.train_eps"]
    if model == "funnel":
        argv += ["--config.init_eps", "0.1", "--config.lr", "0.01", "--config.eps_schedule", "cos_sq"]
    else:
        argv += ["--config.init_eps", str(hp["init_eps"]), "--config.lr", str(hp["lr"])]
    return cli.main(cli.parse_flags(argv, cli.get_config()))


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("rows", nargs="+", help="model:boundmode:nbridges")
    ap.add_argument("--seeds", default="1", help="comma-separated training seeds")
    ap.add_argument("--iters", type=int, default=None)
    ap.add_argument("--init_eps", type=float, default=None, help="override the row's init_eps (rows whose flags the reference does not hold)")
    ns = ap.parse_args()
    args, seeds = ns.rows, tuple(int(s) for s in ns.seeds.split(","))
    for spec in args:
        model, mode, k = spec.split(":")
        k = int(k)
        ref = next(r for r in TABLES[model]["rows"] if r["nbridges"] == k and r.get("boundmode", "MCD_CAIS_sn") == mode)
        t0 = time.time()
        runs = np.array([run(model, mode, k, s, ref, ns.iters, ns.init_eps) for s in seeds])
        line = dict(model=model, boundmode=mode, nbridges=k, seeds=list(seeds), elbo=runs[:, 0].tolist(), ln_Z=runs[:, 1].tolist(),
                    elbo_mean=float(runs[:, 0].mean()), reference_elbo=ref["elbo"], reference_elbo_std=ref["elbo_std"],
                    reference_ln_Z=ref.get("ln_Z"), cite=ref["cite"], wall_s=round(time.time() - t0, 1), init_eps=ns.init_eps,
                    iters=ns.iters)
        print("REPLICATE", json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
