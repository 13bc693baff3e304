"""Builds libcmcd_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcmcd_hip.so")
SOURCES = ["cmcd_kernels.hip", "cmcd_coop.hip", "cmcd_lgcp.hip", "cmcd_grad.hip", "cmcd_mfvi.hip", "cmcd_opt.hip"]
HEADERS = ["cmcd_device.h", os.path.join(ROOT, "include", "cmcd_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "-Wno-format-security", "-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
