"""MCD_CAIS_UHA_sn on the named batch's shape (many_gmm, N = 2000, K = 256, dds net on concat(z, rho)): forward and
value + gradient, repeated (for rocprofv3 --kernel-trace --stats) and timed."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cmcd_amd import synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
out = {}
CASES = (("many_gmm_n2000_k256_dds", dict(init_eps=0.2, init_gamma=2.0, init_sigma=15.0)),
         ("funnel_n300_k64", dict(init_eps=0.05, init_gamma=4.0)),
         ("lgcp_n20_k128", dict(init_eps=0.02, init_gamma=5.0)))
if "nolgcp" in sys.argv[2:]:
    CASES = CASES[:2]
if "manyonly" in sys.argv[2:]:
    CASES = CASES[:1]
for name, over in CASES:
    kw = {}
    if "lgcp" in name:
        import numpy as np
        kw["lgcp_counts"] = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
    b = synthetic.build(name, device="cuda", boundmode="MCD_CAIS_UHA_sn", **over, **kw)
    m = n if "many" in name else b["cfg"]["N"]
    seeds = torch.from_numpy(synthetic.throughput_seeds(m)).cuda()
    args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    for fn, tag, reps in ((mcdbm.compute_bound, "forward", 30), (mcdbm.compute_bound_grad, "value_and_grad", 10)):
        for _ in range(3):
            fn(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[f"{name}:{tag}"] = {"ms": dt * 1e3, "particles": m, "nbridges": b["params_fixed"][1],
                                "particle_steps_per_s": m * b["params_fixed"][1] / dt}
print("UHA_TIMES", json.dumps(out))
