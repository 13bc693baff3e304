"""A/B of two builds of libcmcd_hip.so on the GPU box: runs the forward of several configurations with each library in
its own process and compares loss[N], z[N, d] and the statistics BIT FOR BIT.  A pure re-layout of a kernel (same
arithmetic in the same order) must print 'identical' everywhere.
  python tools/probes/bitwise_ab.py tools/probes/_prev/libcmcd_hip_prev.so cmcd_amd/libcmcd_hip.so"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [  # (config, N, variant, dense)
    ("many_gmm_n2000_k256_dds", 2000, 0, False), ("many_gmm_n2000_k256_dds", 1000, 0, True),
    ("many_gmm_n2000_k256_dds", 4000, 0, False),
    ("gmm_n300_k8", 300, 0, True), ("funnel_n300_k64", 300, 0, True),
    ("many_gmm_var_n16000_k256", 2000, 0, True), ("many_gmm_var_n16000_k256", 4000, 0, False),
]


def child(out):
    sys.path.insert(0, ROOT)
    import torch
    from cmcd_amd import synthetic
    from cmcd_amd import mcdboundingmachine as mcdbm
    res = {}
    for k, (name, n, variant, dense) in enumerate(CASES):
        b = synthetic.build(name, device="cuda", dense=dense)
        seeds = torch.from_numpy(synthetic.parity_seeds(n)).cuda()
        r = mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
        torch.cuda.synchronize()
        for j, t in enumerate(r if isinstance(r, (tuple, list)) else [r]):
            if hasattr(t, "cpu"):
                res["c%d_%d" % (k, j)] = t.detach().cpu().numpy()
    np.savez(out, **res)


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2])
        sys.exit(0)
    outs = []
    for i, lib in enumerate(sys.argv[1:3]):
        out = "/tmp/bitwise_ab_%d.npz" % i
        env = dict(os.environ, CMCD_LIB_PATH=os.path.abspath(lib))
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out], check=True, env=env)
        outs.append(np.load(out))
    a, b = outs
    ok = True
    for key in a.files:
        x, y = a[key], b[key]
        same = x.shape == y.shape and np.array_equal(x.view(np.uint8), y.view(np.uint8))
        if not same:
            ok = False
            d = np.abs(x.astype(np.float64) - y.astype(np.float64))
            fin = np.isfinite(d)
            print(key, CASES[int(key[1:].split("_")[0])], "DIFFERENT: max abs diff", d[fin].max() if fin.any() else None,
                  "mismatching elements", int((x != y).sum()), "of", x.size)
        else:
            print(key, CASES[int(key[1:].split("_")[0])], "identical", x.shape)
    print("ALL IDENTICAL" if ok else "DIFFERENCES FOUND")
