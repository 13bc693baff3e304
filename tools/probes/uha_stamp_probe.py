"""Diagnostic: where does one bridge of uha_coop_kernel (2nd-order mode, cooperative form) spend its cycles?  Builds a
separate library with -DCMCD_STAMPS (s_memtime around every phase of both passes), runs the named batch's shape once and
prints per-wave cycle shares of workgroup 0.  Read SHARES, not totals (stamps serialise the schedule)."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmcd_amd import build as B  # noqa: E402

lib = os.path.join(ROOT, "cmcd_amd", "libcmcd_hip_stamps.so")   # git-ignored (*.so); travels with the gpurun snapshot
os.makedirs(os.path.dirname(lib), exist_ok=True)
if not os.path.exists(lib) or os.environ.get("CMCD_STAMPS_REBUILD"):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCMCD_STAMPS",
           "-fno-slp-vectorize", "-I", os.path.join(ROOT, "include"), "-I", B.CSRC, "-Wno-format-security", "-o", lib]
    subprocess.run(cmd + [os.path.join(B.CSRC, s) for s in B.SOURCES], check=True)
if "--build-only" in sys.argv:
    sys.exit(0)
os.environ["CMCD_LIB_PATH"] = lib
os.environ.setdefault("CMCD_KERNEL_VARIANT", "2")
import torch  # noqa: E402
from cmcd_amd import _lib, synthetic  # noqa: E402
from cmcd_amd import mcdboundingmachine as mcdbm  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "many_gmm_n2000_k256_dds"
over = dict(init_eps=0.2, init_gamma=2.0, init_sigma=15.0) if "many" in name else dict(init_eps=0.05, init_gamma=4.0)
b = synthetic.build(name, device="cuda", boundmode="MCD_CAIS_UHA_sn", **over)
n = int(sys.argv[2]) if len(sys.argv) > 2 else b["cfg"]["N"]
seeds = torch.from_numpy(synthetic.throughput_seeds(n)).cuda()
for _ in range(3):
    mcdbm.compute_bound(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
torch.cuda.synchronize()
print("kernel", _lib.last_kernel_name())
L = _lib.lib()
buf = (C.c_ulonglong * 256)()
L.cmcd_debug_read_uha_stamps(buf)
K = b["params_fixed"][1]
names = ["int1", "bar1", "int2", "bar2", "comb", "bar3"]
print("cycles per bridge, workgroup 0 (pass 0 | pass 1):")
nw = max(w for w in range(16) if buf[w * 16] or buf[w * 16 + 1]) + 1
dealt = name.startswith("funnel") and os.environ["CMCD_KERNEL_VARIANT"] != "3" and n <= 2048   # two state waves
tailw = dealt and b["cfg"].get("emb_dim", 48) == 48 and os.environ["CMCD_KERNEL_VARIANT"] != "5" and nw == 8   # r05: tail wave
T = nw - (4 if tailw else 3 if dealt else 2)
for wv in range(nw):
    row = [buf[wv * 16 + k] / K for k in range(12)]
    role = "MLP%d" % wv if wv < T else ["STATE", "RNG", "STAT2", "TAILW"][wv - T]
    print("%5s " % role + "  ".join("%s=%5.0f" % (nm, v) for nm, v in zip(names, row[:6])) + "  |  " +
          "  ".join("%s=%5.0f" % (nm, v) for nm, v in zip(names, row[6:])) + "   total=%6.0f" % sum(row))
