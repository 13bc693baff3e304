"""Diagnostic: where does one bridge step of coop_kernel spend its cycles?  Builds a separate
library with -DCMCD_STAMPS (s_memtime around every phase), runs the north-star batch once and prints
per-wave cycle shares of workgroup 0.  Read SHARES, not totals (stamps serialise the schedule)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
lib = os.path.join(ROOT, "gpurun_out", "libcmcd_hip_stamps.so")
os.makedirs(os.path.dirname(lib), exist_ok=True)
csrc = os.path.join(ROOT, "cmcd_amd", "csrc")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                "-DCMCD_STAMPS", "-I", os.path.join(ROOT, "include"), "-I", csrc, "-Wno-format-security",
                "-o", lib, os.path.join(csrc, "cmcd_kernels.hip"), os.path.join(csrc, "cmcd_coop.hip"), os.path.join(csrc, "cmcd_coop_wide.hip"), os.path.join(csrc, "cmcd_uha.hip"),
                os.path.join(csrc, "cmcd_lgcp.hip"), os.path.join(csrc, "cmcd_lgcp_wide.hip"), os.path.join(csrc, "cmcd_grad.hip"), os.path.join(csrc, "cmcd_bptt.hip"),
                os.path.join(csrc, "cmcd_mfvi.hip"), os.path.join(csrc, "cmcd_opt.hip")], check=True)
os.environ["CMCD_LIB_PATH"] = lib
os.environ.setdefault("CMCD_KERNEL_VARIANT", "2")   # 3 / 4 pin the 16- / 8-particle tiling
import torch  # noqa: E402
from cmcd_amd import _lib, synthetic  # noqa: E402
from cmcd_amd import mcdboundingmachine as mcdbm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else synthetic.NORTH_STAR
b = synthetic.build(name, device="cuda")
n = int(sys.argv[2]) if len(sys.argv) > 2 else b["cfg"]["N"]
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
for _ in range(3):
    mcdbm.bound_forward(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                        eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
torch.cuda.synchronize()
L = _lib.lib()
buf = (C.c_ulonglong * 256)()
K = b["params_fixed"][1]
print("kernel:", _lib.last_kernel_name())
if _lib.last_kernel_name().startswith("coop_wide8"):
    # cmcd_coop_wide.hip: slot 6 = the step (MLP: z_i formed in the wave; TGT: log-weight terms + z_i), 0 = interval 1 work (MLP: layer 1;
    # TGT: grad log p; RNG: chain segment; ACC: first conversion), 1 = wait at barrier 1, 7 = MLP layer 2, 2 = rest of interval 2
    # (MLP: layer 3; TGT: base; RNG: chain segment; ACC: second conversion), 3 = wait at barrier 2
    L.cmcd_debug_read_stamps_wide(buf)
    Tw = (b["params_fixed"][3].width + 15) // 16 if b["params_fixed"][3].arch == "geffner" else 4
    Tw = 2 if Tw <= 2 else 4
    cols = [("step", 6), ("int1 work", 0), ("wait bar1", 1), ("layer 2", 7), ("int2 work", 2), ("wait bar2", 3)]
    print("cycles per bridge step, workgroup 0 (coop_wide8_kernel):")
    for wv in range(Tw + 4):
        role = "MLP%d" % wv if wv < Tw else ["TGT0", "TGT1", "RNG", "ACC"][wv - Tw]
        vals = [buf[wv * 16 + k] / (K + 1) for _, k in cols]
        print("%5s " % role + "  ".join("%s=%6.0f" % (nm, v) for (nm, _), v in zip(cols, vals)) + "  total=%7.0f" % sum(vals))
    sys.exit(0)
L.cmcd_debug_read_stamps(buf)
names = ["int1 work", "wait bar1", "int2 tail", "wait bar2", "phaseC tail"]
T = (b["params_fixed"][3].width + 15) // 16 if b["params_fixed"][3].arch == "geffner" else 4
print("cycles per bridge step, workgroup 0:")
nw = T + 4
if buf[(T + 3) * 16] == 0 and buf[(T + 3) * 16 + 2] == 0:
    nw = T + 3           # merged RNG / ACC wave (9-tile instance on 8-particle tiles)
for wv in range(nw):
    row = [buf[wv * 16 + k] / (K + 1) for k in range(5)]
    fine = [buf[wv * 16 + k] / (K + 1) for k in range(5, 10)]
    role = "MLP%d" % wv if wv < T else (["TGT0", "TGT1", "RNG", "ACC"] if nw == T + 4 else ["TGT0", "TGT1", "RNG+ACC"])[wv - T]
    print("%5s " % role + "  ".join("%s=%7.0f" % (nm, v) for nm, v in zip(names, row)) + "  total=%7.0f" % (sum(row) + sum(fine)) + "  | [wait bar3 (d>4), ACC z publish (d>4), int2 reads+MFMA, int2 fold+act, phaseC rows]=" + " ".join("%5.0f" % v for v in fine))
