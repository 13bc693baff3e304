"""End-to-end pin to the only numbers the reference holds: the final ELBO / ln Z of its own trained runs, stored as
outputs of /root/reference/src/notebooks/plotting_rebuttal.ipynb (tests/golden/reference_notebook_tables.json, made by
tools/make_notebook_tables.py with the .ipynb line of every value).

The reference's replicate command lines (/root/reference/README.md:53,63,73) are run flag for flag through
cmcd_amd.main — HIP forward, reparameterised HIP gradient, fused Adam, 30 x n_samples evaluation — with three training
seeds each for the funnel rows (all six bridge counts), eight for the bimodal gmm row (mode-aware check), one for the
three lgcp modes.  The stored value is ONE trained model of the reference (sigma_notebook = the spread of its 30 evaluation
groups), so the difference between it and the mean of n training seeds of this build has variance
sigma_notebook^2 + sigma_train^2 (1 + 1 / n); the test holds it to 3 of those sigmas, with sigma_train = the sample sigma
of the n runs, floored for gmm by this build's measured 10-seed spread (the gmm runs are bimodal over training seeds:
three seeds that share a mode have a sample sigma three times too small; tests/golden/reference_notebook_tables.json
`train_seed_spread`, profiles/r02_gmm_training_seed_spread.txt).  Not a bitwise pin (the initial weights and
the per-iteration particle seeds come from torch generators, not from jax's), but a wrong score network, schedule,
target or gradient moves these numbers by many sigmas (the untrained bound is ELBO ~ -2.3 on funnel K = 8)."""
import json
import os

import numpy as np
import pytest

from cmcd_amd import main as cli

pytestmark = pytest.mark.gpu

TABLES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_notebook_tables.json")))
SEEDS = (1, 2, 3)


def _row(model, k):
    return next(r for r in TABLES[model]["rows"] if r["nbridges"] == k and r.get("boundmode", "MCD_CAIS_sn") == "MCD_CAIS_sn")


def _run(model, k, seed):
    hp = TABLES[model]["hparams"]
    argv = ["--config.boundmode", "MCD_CAIS_sn", "--config.model", model, "--config.N", str(hp["N"]),
            "--config.alpha", "0.05", "--config.emb_dim", str(hp["emb_dim"]), "-config.init_sigma", str(hp["init_sigma"]),
            "--config.iters", str(hp["iters"]), "--noconfig.pretrain_mfvi", "--config.train_vi", "--noconfig.train_eps",
            "--config.n_samples", str(hp["n_samples"]), "--config.nbridges", str(k), "--config.seed", str(seed),
            "--noconfig.compute_w2"]      # the Sinkhorn W2 block (30 x 2000-point problems) is not what is being pinned
    if model == "funnel":   # README.md:53; init_eps / lr are overwritten from FUNNEL_EPS_DICT by setup_config
        argv += ["--config.init_eps", "0.1", "--config.lr", "0.01", "--config.eps_schedule", "cos_sq"]
    else:                   # README.md:73
        argv += ["--config.init_eps", str(hp["init_eps"]), "--config.lr", str(hp["lr"])]
    return cli.main(cli.parse_flags(argv, cli.get_config()))


@pytest.mark.parametrize("model,k", [("funnel", 8), ("funnel", 16), ("funnel", 32), ("funnel", 64), ("funnel", 128),
                                     ("funnel", 256)])
def test_trained_bound_reproduces_the_reference_notebook_table(hip_lib, model, k):
    ref = _row(model, k)
    seeds = SEEDS if k < 128 else SEEDS[:2]                      # the long chains: two training seeds (the suite's time budget)
    runs = np.array([_run(model, k, s) for s in seeds])          # [seed, (elbo, ln Z)]
    mean, std = runs.mean(0), runs.std(0, ddof=1)
    print(f"{model} K={k}: ELBO {mean[0]:.4f} +- {std[0]:.4f} (reference {ref['elbo']:.4f} +- {ref['elbo_std']:.4f}, "
          f"ipynb:{ref['cite']}), ln Z {mean[1]:.4f} +- {std[1]:.4f} (reference {ref['ln_Z']:.4f} +- {ref['ln_Z_std']:.4f}); "
          f"per seed {runs.tolist()}")
    prior = TABLES[model].get("train_seed_spread", {})
    n = len(seeds)
    for q, key in ((0, "elbo"), (1, "ln_Z")):
        s_train = max(std[q], prior.get(key + "_std", 0.0))
        tol = 3.0 * np.sqrt(ref[key + "_std"] ** 2 + s_train ** 2 * (1.0 + 1.0 / n))
        assert abs(mean[q] - ref[key]) <= tol, (key, mean[q], ref[key], tol)
    # the targets are normalised (true ln Z = 0, Appendix A.6 of SURVEY.md) and the ELBO is a lower bound
    assert mean[0] < mean[1] + 0.02 and abs(mean[1]) < 0.5


def test_gmm_runs_land_in_the_two_training_modes_and_the_lower_one_is_the_notebooks(hip_lib):
    """gmm K = 8 (README.md:73 flags) is BIMODAL over training seeds in this build (profiles/r02_gmm_training_seed_spread.txt:
    five of ten seeds end at ELBO -0.651 +- 0.026, five at -0.478 +- 0.034; which one flips with last-bit changes of the
    gradient kernels).  A +-3 sigma interval around the seed mean pins nothing there, so the check is mode-aware: every one of
    eight seeds must land in one of the two modes, and the mode the notebook's single stored run sits in (-0.6937 +- 0.0525,
    ipynb:554; ln Z -0.1358 +- 0.0835) must be reproduced by the seeds that reach it."""
    ref = _row("gmm", 8)
    runs = np.array([_run("gmm", 8, s) for s in range(1, 9)])
    elbo, lnz = runs[:, 0], runs[:, 1]
    lower = elbo < -0.57
    print("gmm K=8 per seed", runs.tolist(), "lower mode", int(lower.sum()), "of", len(elbo))
    assert np.all(((elbo > -0.76) & (elbo < -0.58)) | ((elbo > -0.565) & (elbo < -0.37))), elbo
    assert lower.sum() >= 1, "no seed reached the mode of the notebook's run (probability 2^-8 under the measured split)"
    n_lo = int(lower.sum())
    tol_e = 3.0 * np.sqrt(ref["elbo_std"] ** 2 + 0.026 ** 2 / n_lo)
    tol_z = 3.0 * np.sqrt(ref["ln_Z_std"] ** 2 + 0.04 ** 2 / n_lo)
    assert abs(elbo[lower].mean() - ref["elbo"]) <= tol_e, (elbo[lower].mean(), ref["elbo"], tol_e)
    assert abs(lnz[lower].mean() - ref["ln_Z"]) <= tol_z, (lnz[lower].mean(), ref["ln_Z"], tol_z)
    assert np.all(elbo < lnz + 0.05) and np.all(np.abs(lnz) < 0.5)      # normalised target: ELBO <= ln Z = 0


@pytest.mark.parametrize("mode", ["MCD_ULA", "MCD_ULA_sn", "MCD_CAIS_sn"])
def test_lgcp_modes_reproduce_the_reference_notebook_table(hip_lib, mode):
    """lgcp (d = 1600), K = 8 with the README's lgcp flags (/root/reference/README.md:63: 20000 mean-field iterations, 37500
    training iterations, lr 1e-4; 20 s for `MCD_ULA`, ~50 s for the network modes on MI355X).  Stored ELBOs: MCD_ULA 447.81
    +- 0.39 (ipynb:3482), MCD_ULA_sn 458.21 +- 0.32 (:3491), MCD_CAIS_sn 469.48 +- 0.26 (:3500); the mean-field start is
    391.3, so the 56 / 67 / 78-nat gains over 8 annealed steps — and the reference's own ORDERING of the three modes — are
    what is being checked.  One training seed against one stored run: 3.5 of the notebook's evaluation sigmas (r02 measured
    447.66 / 458.95 / 468.99, profiles/r02_o_*)."""
    ref = next(r for r in TABLES["lgcp"]["rows"] if r["boundmode"] == mode and r["nbridges"] == 8)
    hp = TABLES["lgcp"]["hparams"]
    argv = ["--config.boundmode", mode, "--config.model", "lgcp", "--config.N", str(hp["N"]), "--config.emb_dim",
            str(hp["emb_dim"]), "--config.init_eps", str(hp["init_eps"]), "--config.init_sigma", str(hp["init_sigma"]),
            "--config.iters", str(hp["iters"]), "--config.pretrain_mfvi", "--config.mfvi_iters", str(hp["mfvi_iters"]),
            "--config.train_vi", "--config.train_eps", "--config.lr", str(hp["lr"]), "--config.n_samples",
            str(hp["n_samples"]), "--config.nbridges", "8", "--config.seed", "1"]
    elbo, ln_z = cli.main(cli.parse_flags(argv, cli.get_config()))
    print(f"lgcp {mode} K=8: ELBO {elbo:.3f} (reference {ref['elbo']:.3f} +- {ref['elbo_std']:.3f}, ipynb:{ref['cite']}), ln Z {ln_z:.3f}")
    assert abs(elbo - ref["elbo"]) <= 3.5 * ref["elbo_std"], (elbo, ref["elbo"])
    assert ln_z > elbo
