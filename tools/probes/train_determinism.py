"""Is a training run reproducible bit for bit?  Runs the README's gmm K = 8 / funnel K = 8 flags twice with one seed and
compares the final ELBO / ln Z and the trained parameters' checksum."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import main as cli
TABLES = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_notebook_tables.json")))
for model in sys.argv[1:] or ["gmm", "funnel"]:
    hp = TABLES[model]["hparams"]
    argv = ["--config.boundmode", "MCD_CAIS_sn", "--config.model", model, "--config.N", str(hp["N"]), "--config.alpha", "0.05",
            "--config.emb_dim", str(hp["emb_dim"]), "-config.init_sigma", str(hp["init_sigma"]), "--config.iters", str(hp["iters"]),
            "--noconfig.pretrain_mfvi", "--config.train_vi", "--noconfig.train_eps", "--config.n_samples", str(hp["n_samples"]),
            "--config.nbridges", "8", "--config.seed", "3", "--noconfig.compute_w2"]
    if model == "funnel":
        argv += ["--config.init_eps", "0.1", "--config.lr", "0.01", "--config.eps_schedule", "cos_sq"]
    else:
        argv += ["--config.init_eps", str(hp["init_eps"]), "--config.lr", str(hp["lr"])]
    res = [cli.main(cli.parse_flags(argv, cli.get_config())) for _ in range(3)]
    print("DETERMINISM", model, ["%.9f %.9f" % (e, z) for e, z in res], "identical:", all(r == res[0] for r in res), flush=True)
