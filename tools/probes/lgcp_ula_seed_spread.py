"""lgcp MCD_ULA K = 8 with the README's flags: final ELBO over training seeds, on the split-K forward (CMCD_KERNEL_VARIANT=3)
and on the no-split-K forward (0).  Is the difference between the two forms inside the seed-to-seed spread?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmcd_amd import main as cli
from cmcd_amd import mcdboundingmachine as mcdbm
TABLES = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_notebook_tables.json")))
hp = TABLES["lgcp"]["hparams"]
mode = sys.argv[1] if len(sys.argv) > 1 else "MCD_ULA"
for variant in (3, 0):
    mcdbm.KERNEL_VARIANT = variant
    for seed in (1, 2, 3):
        argv = ["--config.boundmode", mode, "--config.model", "lgcp", "--config.N", str(hp["N"]), "--config.emb_dim",
                str(hp["emb_dim"]), "--config.init_eps", str(hp["init_eps"]), "--config.init_sigma", str(hp["init_sigma"]),
                "--config.iters", str(hp["iters"]), "--config.pretrain_mfvi", "--config.mfvi_iters", str(hp["mfvi_iters"]),
                "--config.train_vi", "--config.train_eps", "--config.lr", str(hp["lr"]), "--config.n_samples",
                str(hp["n_samples"]), "--config.nbridges", "8", "--config.seed", str(seed)]
        elbo, ln_z = cli.main(cli.parse_flags(argv, cli.get_config()))
        print("RESULT", mode, "variant", variant, "seed", seed, "ELBO %.3f ln Z %.3f" % (elbo, ln_z), flush=True)
