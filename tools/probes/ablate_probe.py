"""Diagnostic: whole-kernel time of coop_kernel with roles ablated (wrong results, timing only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = "/tmp/libcmcd_hip_stamps.so"
csrc = os.path.join(ROOT, "cmcd_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCMCD_STAMPS",
                "-I", os.path.join(ROOT, "include"), "-I", csrc, "-Wno-format-security", "-o", lib,
                os.path.join(csrc, "cmcd_kernels.hip"), os.path.join(csrc, "cmcd_coop.hip"),
                os.path.join(csrc, "cmcd_lgcp.hip"), os.path.join(csrc, "cmcd_grad.hip")], check=True)
os.environ["CMCD_LIB_PATH"] = lib
os.environ["CMCD_KERNEL_VARIANT"] = "2"
import torch
from cmcd_amd import _lib, synthetic
from cmcd_amd import mcdboundingmachine as mcdbm
b = synthetic.build(synthetic.NORTH_STAR, device="cuda")
seeds = torch.from_numpy(synthetic.throughput_seeds(b["cfg"]["N"])).cuda()
names = {0: "full", 256: "prio: aux high", 512: "prio: MLP high", 768: "prio: TGT high", 1: "no TGT", 7: "no aux at all", 31: "skeleton only", 31 + 256: "skeleton, aux high"}
for mask, nm in names.items():
    os.environ["CMCD_ABLATE"] = str(mask)
    for _ in range(2):
        mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                            eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5):
        mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                            eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    ms, cnt = _lib.profile_collect()
    _lib.profile_enable(False)
    print("%-16s %.4f ms  (%.0f ns / bridge)" % (nm, ms / cnt, ms / cnt / 257 * 1e6))
