"""Extracts the final ELBO / ln Z tables the reference keeps as stored notebook OUTPUTS
(/root/reference/src/notebooks/plotting_rebuttal.ipynb) into tests/golden/reference_notebook_tables.json.

Run in the BUILD container (the only place /root/reference exists).  Only numbers are taken (results of the reference's
own trained MCD_CAIS_sn / MCD_CAIS_UHA_sn / MCD_ULA_sn / MCD_ULA runs, with the .ipynb line each one sits on); no notebook source."""
import json
import os
import re

NB = "/root/reference/src/notebooks/plotting_rebuttal.ipynb"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    lines = open(NB).read().split("\n")
    out = {
        "_about": "Final ELBO / ln Z of the reference's own trained runs, as stored in the outputs of "
                  "/root/reference/src/notebooks/plotting_rebuttal.ipynb (30 evaluation seed groups x n_samples particles, "
                  "/root/reference/src/utils.py:219-248; *_std = spread over the 30 groups of ONE trained model). Values only "
                  "(data); 'cite' = 1-based line of the .ipynb file. Extracted by tools/make_notebook_tables.py.",
        "funnel": {"hparams": {"boundmode": "MCD_CAIS_sn", "N": 300, "emb_dim": 48, "init_sigma": 1, "iters": 11000,
                               "pretrain_mfvi": False, "train_vi": True, "train_eps": False, "n_samples": 2000,
                               "eps_schedule": "cos_sq", "readme_cite": "/root/reference/README.md:53",
                               "note": "init_eps and lr come from FUNNEL_EPS_DICT[nbridges] "
                                       "(/root/reference/src/configs/base.py:65-72)"}, "rows": []},
        "gmm": {"hparams": {"boundmode": "MCD_CAIS_sn", "N": 300, "emb_dim": 20, "init_eps": 0.01, "init_sigma": 1,
                            "iters": 11000, "pretrain_mfvi": False, "train_vi": True, "train_eps": False, "lr": 0.001,
                            "n_samples": 500, "readme_cite": "/root/reference/README.md:73",
                            "note": "the README's replicate flags; the notebook's own table lists init_sigma 2 for its runs "
                                    "(ipynb:563-564) without their iteration count — run with init_sigma 2 for the README's "
                                    "11000 iterations this build ends at ELBO -1.02 (K = 8), i.e. the README flags, not "
                                    "the table's column, are what reproduces the stored numbers"}, "rows": []},
        "lgcp": {"hparams": {"N": 20, "emb_dim": 20, "init_eps": 1e-5, "init_sigma": 1, "iters": 37500,
                             "pretrain_mfvi": True, "mfvi_iters": 20000, "train_vi": True, "train_eps": True, "lr": 1e-4,
                             "n_samples": 500, "readme_cite": "/root/reference/README.md:63"}, "rows": []}}
    for i, l in enumerate(lines, 1):
        m = re.search(r"\{'boundmode': 'MCD_CAIS_sn', 'model': 'funnel', 'nbridges': (\d+), 'elbo': ([-\d.e]+), "
                      r"'elbo_std': ([-\d.e]+), 'ln_Z': ([-\d.e]+), 'ln_Z_std': ([-\d.e]+)", l)
        if m and i < 500:
            out["funnel"]["rows"].append(dict(nbridges=int(m[1]), elbo=float(m[2]), elbo_std=float(m[3]), ln_Z=float(m[4]),
                                              ln_Z_std=float(m[5]), cite=i))
        m = re.search(r'"\d  MCD_CAIS_sn   gmm\s+(\d+)\s+([-\d.]+)\s+([-\d.]+)\s+([-\d.]+)\s+([-\d.]+)', l)
        if m and i < 600:
            out["gmm"]["rows"].append(dict(nbridges=int(m[1]), elbo=float(m[2]), elbo_std=float(m[3]), ln_Z=float(m[4]),
                                           ln_Z_std=float(m[5]), cite=i))
    for mode in ("MCD_ULA", "MCD_ULA_sn", "MCD_CAIS_sn"):
        rows = []
        for i, l in enumerate(lines, 1):
            if 3470 < i < 3520:
                m = re.search(r'"%s lgcp (\d+) \[([\d.]+)\]' % mode, l)
                if m:
                    rows.append([int(m[1]), float(m[2]), i])
        last = rows[-1][2]                      # the two lines after "(6,) (6,)" hold [elbos] [stds]
        blob = (lines[last + 1] + lines[last + 2]).replace("\\n", "").replace('"', "").replace(",", "")
        stds = [float(x) for x in re.findall(r"\[([^\]]+)\]", blob)[1].split()]
        for (k, e, c), s in zip(rows, stds):
            out["lgcp"]["rows"].append(dict(boundmode=mode, nbridges=k, elbo=e, elbo_std=s, cite=c, std_cite=last + 3))
    # 2nd-order CMCD on lgcp (ipynb:3536-3550): "MCD_CAIS_UHA_sn lgcp K [elbo]" (each value printed twice: the line with the
    # number is kept), then one line "[elbos] [stds]"; lr from LR_DICT["lgcp"]["MCD_CAIS_UHA_sn"] (configs/base.py:56)
    rows = []
    for i, l in enumerate(lines, 1):
        if 3520 < i < 3560:
            m = re.search(r'"MCD_CAIS_UHA_sn lgcp (\d+) \[([\d.]+)\]', l)
            if m:
                rows.append([int(m[1]), float(m[2]), i])
    last = rows[-1][2]
    blob = (lines[last] + lines[last + 1]).replace("\\n", "").replace('"', "").replace(",", "")
    stds = [float(x) for x in re.findall(r"\[([^\]]+)\]", blob)[1].split()]
    for (k, e, c), sd in zip(rows, stds):
        out["lgcp"]["rows"].append(dict(boundmode="MCD_CAIS_UHA_sn", nbridges=k, elbo=e, elbo_std=sd, cite=c, std_cite=last + 2,
                                        lr=1e-3))
    out["funnel"]["rows"].sort(key=lambda r: r["nbridges"])
    path = os.path.join(ROOT, "tests", "golden", "reference_notebook_tables.json")
    if os.path.exists(path):     # this build's own measured training-seed spread rides along (not reference data; see its "source")
        old = json.load(open(path))
        for k in ("gmm", "funnel", "lgcp"):
            if "train_seed_spread" in old.get(k, {}):
                out[k]["train_seed_spread"] = old[k]["train_seed_spread"]
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print({k: len(v["rows"]) for k, v in out.items() if k != "_about"})


if __name__ == "__main__":
    main()
