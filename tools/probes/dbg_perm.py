import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from cmcd_amd import mcdboundingmachine as mcdbm, synthetic
for name, n, over in [("many_gmm_var_n16000_k256", 16000, {}), ("many_gmm_var_n16000_k256", 4096, {}), ("many_gmm_var_n16000_k256", 16000, dict(nbridges=4))]:
  for variant in (1, 2):
    mcdbm.KERNEL_VARIANT = variant
    b = synthetic.build(name, device="cuda", **over)
    seeds = synthetic.throughput_seeds(n, stream=5)
    def fwd(s):
        o = mcdbm.bound_forward(torch.as_tensor(s).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"]); torch.cuda.synchronize(); return o[0].cpu().numpy()
    l1 = fwd(seeds); l2 = fwd(seeds)
    perm = np.random.default_rng(0).permutation(n)
    lp = fwd(seeds[perm])
    d = lp - l1[perm]
    bad = np.flatnonzero(d != 0)
    print(name, n, over, "variant", variant, "repeat equal:", np.array_equal(l1, l2), "perm mismatches:", len(bad), "max abs diff", np.abs(d).max() if len(bad) else 0, "first bad idx", bad[:8], "their positions mod 16:", (bad[:8] % 16), "orig pos mod 16", perm[bad[:8]] % 16)
