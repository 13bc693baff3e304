"""lgcp MCD_ULA K = 8 trained with the README's flags; the final 30 x 500 evaluation on the wide-batch form (2) and on the
32-row passes (1): does the ELBO depend on the form of the evaluation?"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmcd_amd import main as cli, utils
from cmcd_amd import mcdboundingmachine as mcdbm
TABLES = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_notebook_tables.json")))
hp = TABLES["lgcp"]["hparams"]
mode = sys.argv[1] if len(sys.argv) > 1 else "MCD_ULA"
orig = utils.sample
def sample_both(*a, **k):
    out = None
    for v in (1, 2, 0):
        mcdbm.KERNEL_VARIANT = v
        out = orig(*a, **k)
        el = out[0]
        print("EVAL variant", v, "ELBO %.4f" % float(-el.double().mean()), flush=True)
    return out
utils.sample = sample_both
argv = ["--config.boundmode", mode, "--config.model", "lgcp", "--config.N", str(hp["N"]), "--config.emb_dim",
        str(hp["emb_dim"]), "--config.init_eps", str(hp["init_eps"]), "--config.init_sigma", str(hp["init_sigma"]),
        "--config.iters", str(hp["iters"]), "--config.pretrain_mfvi", "--config.mfvi_iters", str(hp["mfvi_iters"]),
        "--config.train_vi", "--config.train_eps", "--config.lr", str(hp["lr"]), "--config.n_samples",
        str(hp["n_samples"]), "--config.nbridges", "8", "--config.seed", "1"]
elbo, ln_z = cli.main(cli.parse_flags(argv, cli.get_config()))
print("RESULT", mode, "ELBO %.3f ln Z %.3f" % (elbo, ln_z), flush=True)
