"""Where does a saturating batch spend its time?  Builds diagnostic copies of the library with one ingredient of
the wave-per-tile kernel removed (-DCMCD_TRAJ_ABL=mask: 1 layer-2 MFMAs, 2 activations, 4 target gradient,
8 noise generation) and times forward calls of 2^18 particles with each (child processes, CMCD_LIB_PATH)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
csrc = os.path.join(ROOT, "cmcd_amd", "csrc")
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from cmcd_amd import synthetic
    from cmcd_amd import mcdboundingmachine as mcdbm
    n = 262144
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda")
    seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
    f = lambda: mcdbm.compute_bound(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                    eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); f(); e.record(); torch.cuda.synchronize()
    print("mask %2s: %.3f ms per call" % (os.environ.get("ABL"), s.elapsed_time(e) / 2))
    sys.exit(0)
outdir = os.path.join(ROOT, "gpurun_out", "abl")
os.makedirs(outdir, exist_ok=True)
for mask in ([int(x) for x in sys.argv[1:]] or (0, 1, 2, 4, 8, 3, 6, 15)):
    lib = os.path.join(outdir, "libcmcd_abl%d.so" % mask)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                    "-DCMCD_TRAJ_ABL=%d" % mask, "-I", os.path.join(ROOT, "include"), "-I", csrc, "-Wno-format-security",
                    "-o", lib] + [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith(".hip")], check=True)
    subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, CMCD_LIB_PATH=lib, ABL=str(mask), CMCD_KERNEL_VARIANT="1"))
