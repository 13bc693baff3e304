"""A / B builds: libcmcd_hip with ONE source recompiled under extra flags, the other objects taken from cmcd_amd/build/.
    python tools/probes/build_variant.py gpurun_out/libcmcd_hip_x.so cmcd_lgcp_wide.hip -DCMCD_WIDE_DEPTH=4
Run a probe on it with CMCD_LIB_PATH=<that file> (exported in the shell, not through `env`, under rocprofv3)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmcd_amd import build
out, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
build.build()
obj = os.path.join(ROOT, "gpurun_out", "variant_" + src.replace(".hip", ".o"))
os.makedirs(os.path.dirname(obj), exist_ok=True)
hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"), "-I", build.CSRC,
                "-Wno-format-security"] + build.EXTRA_FLAGS.get(src, []) + flags + ["-c", os.path.join(build.CSRC, src), "-o", obj], check=True)
objs = [obj if s == src else build._obj(s) for s in build.SOURCES]
subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
