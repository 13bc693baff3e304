"""s_setprio levels of the cooperative kernel's roles (2 bits each: MLP, TGT, RNG, ACC from bit 0) on the north-star batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm

name = sys.argv[1] if len(sys.argv) > 1 else synthetic.NORTH_STAR
b = synthetic.build(name, device="cuda")
n = int(sys.argv[2]) if len(sys.argv) > 2 else b["cfg"]["N"]
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
L = _lib.lib()


def t(prio, reps=10):
    L.cmcd_debug_set_coop_prio(prio)
    f(); f(); torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
    return ms / cnt


def enc(mlp, tgt, rng, acc):
    return mlp | tgt << 2 | rng << 4 | acc << 6


for variant in ((int(sys.argv[3]),) if len(sys.argv) > 3 else (4,)):
    mcdbm.KERNEL_VARIANT = variant
    t(0, 50)   # clocks settle
    cands = [(0, 0, 0, 0), (0, 1, 0, 1), (1, 1, 0, 1), (1, 0, 0, 0), (1, 0, 0, 1), (2, 1, 0, 1), (0, 1, 0, 0), (0, 0, 0, 1), (0, 2, 0, 1),
             (0, 1, 0, 2), (0, 2, 0, 2), (0, 0, 1, 0), (0, 1, 1, 0), (0, 1, 2, 0), (0, 0, 2, 0), (0, 1, 3, 0), (0, 2, 3, 0), (1, 2, 3, 0),
             (2, 0, 0, 0), (3, 0, 0, 0), (2, 1, 0, 0), (2, 0, 0, 1), (2, 0, 1, 0), (1, 0, 1, 0), (3, 1, 1, 1), (3, 2, 1, 2), (2, 1, 1, 1)]
    res = {c: [] for c in cands}
    for rnd in range(4):   # interleaved rounds: drift hits every candidate alike
        for c in cands:
            res[c].append(t(enc(*c), 30))
    base = min(res[cands[0]])
    print("variant %d" % variant)
    for c in cands:
        print("  MLP %d TGT %d RNG %d ACC %d: " % c + " ".join("%.4f" % v for v in res[c]) + "  min %.4f (%+.1f %%)" % (
            min(res[c]), 100 * (min(res[c]) / base - 1)), flush=True)
