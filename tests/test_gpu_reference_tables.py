"""End-to-end pin to the only numbers the reference holds: the final ELBO / ln Z of its own trained runs, stored as
outputs of /root/reference/src/notebooks/plotting_rebuttal.ipynb (tests/golden/reference_notebook_tables.json, made by
tools/make_notebook_tables.py with the .ipynb line of every value).

The reference's replicate command lines (/root/reference/README.md:53,63,73) are run flag for flag through
cmcd_amd.main — HIP forward, reparameterised HIP gradient, fused Adam, 30 x n_samples evaluation — with three training
seeds each for the funnel rows (all six bridge counts), eight for the widely spread gmm row (unselected seed mean), one for the
three lgcp modes.  The stored value is ONE trained model of the reference (sigma_notebook = the spread of its 30 evaluation
groups), so the difference between it and the mean of n training seeds of this build has variance
sigma_notebook^2 + sigma_train^2 (1 + 1 / n); the test holds it to 3 of those sigmas, with sigma_train = the sample sigma
of the n runs (gmm has its own test below: its runs spread between -0.69 and -0.42 over training seeds,
profiles/r02_gmm_training_seed_spread.txt).  Not a bitwise pin (the initial weights and
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


def test_gmm_seed_mean_against_the_notebook_row(hip_lib):
    """gmm K = 8 (README.md:73 flags) spreads WIDELY over training seeds in this build: ten seeds of r02 ended between -0.69 and
    -0.43 (profiles/r02_gmm_training_seed_spread.txt), eight seeds of r03 between -0.660 and -0.425 (mean -0.556, sample sigma
    0.083: a continuum; where a seed lands flips with last-bit changes — the gradient tails sum with float atomics).  The
    notebook stores ONE trained model (-0.6937 +- 0.0525 over its 30 evaluation groups, ipynb:554; ln Z -0.1358 +- 0.0835).

    The comparison is UNSELECTED (r04; the r03 form compared the three lowest of eight seeds with the notebook, which passes
    by construction): the mean of ALL eight seeds against the stored run, whose difference has variance
    sigma_notebook^2 + sigma_train^2 (1 + 1 / n) when the notebook's model is one more draw from the same training-seed
    distribution — the same rule as the funnel rows above.  Measured: the seed mean sits ~0.14 ABOVE the notebook's run
    (1.3 of those sigmas; 2.6 sigma_notebook alone).  The restatement-trained model (tests/golden/oracle_trained_rows.json:
    -0.533) lands at the same place as the HIP-trained ones, so the offset is not the kernels'; it is recorded as a known
    deviation in CHANGELOG.md (DESIGN r04 section 5b), and this test fails if it grows beyond 3 sigma of the combined spread."""
    ref = _row("gmm", 8)
    n = 8
    runs = np.array([_run("gmm", 8, s) for s in range(1, n + 1)])
    elbo, lnz = runs[:, 0], runs[:, 1]
    mean, std = runs.mean(0), runs.std(0, ddof=1)
    z = []
    for q, key in ((0, "elbo"), (1, "ln_Z")):
        sig = np.sqrt(ref[key + "_std"] ** 2 + std[q] ** 2 * (1.0 + 1.0 / n))
        z.append((mean[q] - ref[key]) / sig)
    print(f"gmm K=8 over {n} training seeds: ELBO {mean[0]:.4f} +- {std[0]:.4f} (reference {ref['elbo']:.4f} +- {ref['elbo_std']:.4f}: "
          f"{z[0]:+.2f} combined sigma), ln Z {mean[1]:.4f} +- {std[1]:.4f} (reference {ref['ln_Z']:.4f} +- {ref['ln_Z_std']:.4f}: "
          f"{z[1]:+.2f}); per seed {runs.tolist()}")
    assert abs(z[0]) <= 3.0 and abs(z[1]) <= 3.0, z
    # the seed spread itself must stay what was measured (a collapsed or exploded spread would make the rule above vacuous)
    assert 0.02 < std[0] < 0.15, std
    assert np.all((elbo > -0.80) & (elbo < -0.35)), elbo
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
