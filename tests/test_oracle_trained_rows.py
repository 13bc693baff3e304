"""The pin of the ORACLE itself to numbers the reference holds: tests/golden/oracle_trained_rows.json records models
trained entirely by the restatements (tools/oracle_train.py: torch float64 autograd through oracle/cmcd_oracle_torch.py,
the eager clip + Adam loop of cmcd_amd/opt.py, evaluation through oracle/cmcd_oracle.py float64; no HIP code runs) with the
reference README's flags, next to the values the reference stores in its notebook
(tests/golden/reference_notebook_tables.json).  A wrong kernel mean, scale, schedule, network or target in the restatement
moves the trained ELBO / ln Z by many sigmas (the untrained funnel bound is ELBO -2.3).

What the pin says, plainly (r04: six training seeds of funnel K = 8 instead of one run against one run):
  * the restatement + optax-style Adam trains funnel K = 8 to ELBO -1.0105 +- 0.0080 over training seeds where the notebook
    holds ONE run at -1.063 +- 0.025 (its spread over 30 evaluation groups): the seed MEAN sits +0.052 above the stored run =
    2.1 notebook sigmas, 2.0 sigmas of the combined spread sqrt(sigma_nb^2 + sigma_seed^2 (1 + 1 / n));
  * ln Z agrees (-0.30 +- 0.05 against -0.304 +- 0.151), and so does the ELBO from K = 32 up (-0.687 against -0.681 +- 0.020);
  * K = 8 and K = 16 ELBOs are 0.03 - 0.06 HIGH, cause unknown — the HIP-trained models land on the same value as the
    restatement-trained ones (-1.010 +- 0.006, CHANGELOG.md (DESIGN r04 section 5b)), so it is not the kernels', but which optimiser /
    initialisation produced the notebook's rows is not recorded anywhere in the reference, and no claim is made about it."""
import json
import os

import numpy as np
import pytest

ROWS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_trained_rows.json")))
FULL = [r for r in ROWS if r["iters"] >= 11000]


def _group(model, k):
    return [r for r in FULL if r["model"] == model and r["nbridges"] == k]


def test_funnel_k8_seed_mean_against_the_notebook_row():
    rows = _group("funnel", 8)
    assert len(rows) >= 5, "the seed-mean pin needs >= 5 training seeds (tools/oracle_train.py funnel 8 --seed s)"
    elbo = np.array([r["elbo"] for r in rows])
    lnz = np.array([r["ln_Z"] for r in rows])
    ref, ref_std = rows[0]["reference_elbo"], rows[0]["reference_elbo_std"]
    n = len(rows)
    mean, sd = elbo.mean(), elbo.std(ddof=1)
    z = (mean - ref) / np.sqrt(ref_std ** 2 + sd ** 2 * (1.0 + 1.0 / n))
    print(f"funnel K=8, {n} restatement-trained seeds: ELBO {mean:.4f} +- {sd:.4f} (notebook {ref:.4f} +- {ref_std:.4f}: {z:+.2f} "
          f"combined sigma), ln Z {lnz.mean():.4f} +- {lnz.std(ddof=1):.4f} (notebook {rows[0]['reference_ln_Z']:.4f})")
    # the seed mean, not a single run: within 3 sigma of the combined spread — and the measured offset is recorded, not hidden
    assert abs(z) <= 3.0, (mean, ref, z)
    assert 0.02 < mean - ref < 0.09, "the K = 8 ELBO offset against the notebook moved: re-read CHANGELOG.md (DESIGN r04 section 5b)"
    assert sd < 0.02, sd                       # training seeds agree with each other to ~0.008
    assert abs(lnz.mean() - rows[0]["reference_ln_Z"]) <= 0.15, lnz.mean()


def test_gmm_k8_restatement_trained_seeds_sit_where_the_hip_trained_ones_do():
    """The advisor's question (r03): why does this build train gmm K = 8 to a higher ELBO (-0.54 +- 0.09 over eight HIP
    training seeds) than the notebook's single run (-0.694 +- 0.052)?  Five seeds trained by the RESTATEMENTS alone (no HIP
    code): -0.555 +- 0.088, ln Z -0.04 — the same distribution.  So the offset belongs to the training recipe this repo runs
    (optax-style Adam, torch initial weights) against whatever produced the notebook's row, not to the kernels; against the
    notebook the seed mean is +1.3 combined sigma.  Asserted: the restatement-trained mean against the notebook (3 combined
    sigma) and against the HIP path's recorded seed statistics."""
    rows = _group("gmm", 8)
    assert len(rows) >= 5
    elbo = np.array([r["elbo"] for r in rows])
    n, mean, sd = len(rows), elbo.mean(), elbo.std(ddof=1)
    ref, ref_std = rows[0]["reference_elbo"], rows[0]["reference_elbo_std"]
    z = (mean - ref) / np.sqrt(ref_std ** 2 + sd ** 2 * (1.0 + 1.0 / n))
    print(f"gmm K=8, {n} restatement-trained seeds: ELBO {mean:.4f} +- {sd:.4f} (notebook {ref:.4f} +- {ref_std:.4f}: {z:+.2f} combined sigma)")
    assert abs(z) <= 3.0, (mean, ref, z)
    hip_mean, hip_sd, hip_n = -0.5383, 0.0870, 8      # tests/test_gpu_reference_tables.py, r04 run (profiles/r04_gpu_test_suite.log)
    assert abs(mean - hip_mean) <= 3.0 * np.sqrt(sd ** 2 / n + hip_sd ** 2 / hip_n), (mean, hip_mean)


@pytest.mark.parametrize("row", [r for r in FULL if not (r["model"] == "funnel" and r["nbridges"] == 8)],
                         ids=lambda r: f"{r['model']}_k{r['nbridges']}_seed{r['seed']}")
def test_oracle_trained_model_reaches_the_reference_notebook_row(row):
    # ln Z: the reference's own spread over its 30 evaluation groups is the only sigma it holds (0.15 for funnel K = 8)
    assert abs(row["ln_Z"] - row["reference_ln_Z"]) <= 0.15, (row["ln_Z"], row["reference_ln_Z"])
    if row["model"] == "gmm":
        # gmm K = 8 spreads between -0.69 and -0.42 over training seeds of the HIP path (CHANGELOG.md (DESIGN r04 section 5b); the notebook's
        # single run, -0.694, sits at the lower end): the restatement-trained models land INSIDE that spread — the width
        # belongs to the training dynamics, not to the kernels.  An envelope, not a pin (tests/test_gpu_reference_tables.py
        # holds the unselected seed mean of the HIP path against the notebook).
        assert -0.70 < row["elbo"] < -0.41, row["elbo"]
    else:
        # one training run of the restatement against one of the reference: 3 notebook sigmas
        assert abs(row["elbo"] - row["reference_elbo"]) <= 3.0 * row["reference_elbo_std"], (row["elbo"], row["reference_elbo"])
    # the bound is a bound, and training got there (first logged loss of an untrained funnel model is ~2.3)
    assert row["elbo"] < row["ln_Z"] and max(row["last_training_losses"]) < 1.3
