"""Builds an A / B variant of the library next to the product one: python tools/probes/build_variant.py NAME -DFLAG [...]
-> cmcd_amd/libcmcd_hip_NAME.so (git-ignored, travels with the gpurun snapshot; select it with CMCD_LIB_PATH)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cmcd_amd import build as B  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
# --only=a.hip,b.hip: compile these sources with the flags, take the product build's objects for the rest
only = [f.split("=", 1)[1].split(",") for f in flags if f.startswith("--only=")]
only = only[0] if only else None
flags = [f for f in flags if not f.startswith("--only=")]
obj = os.path.join("/tmp", "cmcd_variant_" + name)
os.makedirs(obj, exist_ok=True)
common = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
          "-I", B.CSRC, "-Wno-format-security"] + flags


def one(src):
    if only is not None and src not in only:
        return os.path.join(B.BUILD if hasattr(B, "BUILD") else os.path.join(ROOT, "cmcd_amd", "build"), src.replace(".hip", ".o"))
    o = os.path.join(obj, src.replace(".hip", ".o"))
    subprocess.run(common + B.EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(B.CSRC, src), "-o", o], check=True)
    return o


with ThreadPoolExecutor(max_workers=6) as ex:
    objs = list(ex.map(one, B.SOURCES))
lib = os.path.join(ROOT, "cmcd_amd", "libcmcd_hip_%s.so" % name)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
print(lib)
