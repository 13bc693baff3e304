"""Forward kernel time of config 4's shard sizes for two library builds: python tools/probes/t9_fwd_ab.py libA.so libB.so"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from cmcd_amd import _lib, synthetic
    from cmcd_amd import mcdboundingmachine as mcdbm
    b = synthetic.build("many_gmm_var_n16000_k256", device="cuda")
    out = []
    for n in (2000, 4000, 8000):
        seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
        f = lambda: mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                        eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
        for _ in range(60): f()
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(60): r = f()
        torch.cuda.synchronize()
        ms, cnt = _lib.profile_collect(); _lib.profile_enable(False)
        out.append("n=%d %.4f ms (var %.6f)" % (n, ms / cnt, float(r[0][torch.isfinite(r[0])].var())))
    print(" | ".join(out)); sys.exit(0)
for r in range(2):
    for l in sys.argv[1:3]:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, CMCD_LIB_PATH=os.path.abspath(l)), capture_output=True, text=True)
        print(os.path.basename(l), o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:], flush=True)
