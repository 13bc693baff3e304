"""Wide-batch vs 32-row forms of the lgcp forward on the same seeds, per mode and batch size: max |loss difference|."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from cmcd_amd import synthetic, _lib
from cmcd_amd import mcdboundingmachine as mcdbm
counts = np.load(os.path.join(ROOT, "tests", "golden", "lgcp_bin_counts.npy"))
for mode in ("MCD_ULA", "MCD_ULA_sn", "MCD_CAIS_sn", "MCD_CAIS_var_sn"):
    b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts, boundmode=mode, nbridges=8, init_eps=2e-3)
    dim, K, _, spec = b["params_fixed"]
    if mode == "MCD_ULA":
        flat, unflatten, fixed = mcdbm.initialize(dim=dim, nbridges=K, vdparams=None, eps=b["cfg"]["init_eps"], trainable=("eps",),
                                                  mode="MCD_ULA", device="cuda")
        b = dict(b, params_flat=flat, unflatten=unflatten, params_fixed=fixed)
    for n in (500, 2048, 15000):
        seeds = torch.from_numpy(synthetic.throughput_seeds(n, stream=2)).cuda()
        out = {}
        for v in (1, 2):
            mcdbm.KERNEL_VARIANT = v
            l, z, st = mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                           eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
            out[v] = (l.double().cpu(), z.double().cpu(), _lib.last_kernel_name()[:16])
        d = (out[1][0] - out[2][0]).abs()
        print(mode, n, "mean loss", float(out[1][0].mean()), float(out[2][0].mean()), "max |dl|", float(d.max()), "argmax", int(d.argmax()),
              "max |dz|", float((out[1][1] - out[2][1]).abs().max()), out[1][2], out[2][2], flush=True)
