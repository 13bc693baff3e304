"""Interleaved A/B of two library builds on the training steps of the named batch and of config 4's 2000-particle shard:
python tools/probes/train_ab.py libA.so libB.so [rounds]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch
    from cmcd_amd import synthetic
    from cmcd_amd import mcdboundingmachine as mcdbm
    def timeit(f, reps=30):
        for _ in range(5): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): f()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / reps
    out = []
    for name, n in (("many_gmm_n2000_k256_dds", 2000), ("many_gmm_var_n16000_k256", 2000)):
        b = synthetic.build(name, device="cuda")
        kw = dict(eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
        seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
        args = (seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
        if b["params_fixed"][2] == "MCD_CAIS_sn":
            out.append("bptt %.4f" % timeit(lambda: mcdbm.compute_bound_grad(*args, **kw)))
            bv = synthetic.build(name, device="cuda", boundmode="MCD_CAIS_var_sn")
            argv = (seeds, bv["params_flat"], bv["unflatten"], bv["params_fixed"], bv["target"])
            out.append("var %.4f" % timeit(lambda: mcdbm.compute_log_var_grad(*argv, **kw)))
        else:
            out.append("cfg4-shard var %.4f" % timeit(lambda: mcdbm.compute_log_var_grad(*args, **kw)))
    print(" | ".join(out))
    sys.exit(0)
libs = [os.path.abspath(p) for p in sys.argv[1:3]]
for r in range(int(sys.argv[3]) if len(sys.argv) > 3 else 2):
    for l in libs:
        o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=dict(os.environ, CMCD_LIB_PATH=l), capture_output=True, text=True)
        print(os.path.basename(l), o.stdout.strip().splitlines()[-1] if o.stdout.strip() else o.stderr[-300:], flush=True)
