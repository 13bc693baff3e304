"""Builds libcmcd_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Each source is compiled to its own object (in parallel) and the objects are linked: per-file flags are possible and a
rebuild after touching one kernel takes that file's compile time, not the sum."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libcmcd_hip.so")
SOURCES = ["cmcd_kernels.hip", "cmcd_uha.hip", "cmcd_coop.hip", "cmcd_coop_wide.hip", "cmcd_lgcp.hip", "cmcd_lgcp_wide.hip", "cmcd_grad.hip", "cmcd_bptt.hip", "cmcd_mfvi.hip", "cmcd_opt.hip"]
HEADERS = ["cmcd_device.h", os.path.join(ROOT, "include", "cmcd_hip.h"), os.path.join(ROOT, "include", "cmcd_hip_diag.h")]
# Per-file flags.  cmcd_kernels.hip holds the wave-per-tile trajectory kernel, which is VALU-issue bound at 4 waves per
# SIMD: there a packed fp32 instruction holds the pipe ~1.8x as long as a plain one and the SLP vectoriser pays v_mov
# shuffles to form its operands (ISA reading r02: 126 v_pk_* + 4 v_mov per pair of mixture components), so it is off
# for that file.  The cooperative kernel is issue-bound per wave (one instruction per ~5 cycles whatever it is) and
# keeps the packed forms.
EXTRA_FLAGS = {"cmcd_kernels.hip": ["-fno-slp-vectorize"],
               "cmcd_uha.hip": ["-fno-slp-vectorize"],   # same wave-per-tile mapping as cmcd_kernels.hip
               # Jacobian / scan kernels of the work-item reparameterised gradient: -11 % / -5 % without SLP (cmcd_bptt.hip)
               "cmcd_bptt.hip": ["-fno-slp-vectorize"]}


def _deps():
    deps = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.abspath(__file__))
    return deps


def _obj(src):
    return os.path.join(OBJ, src.replace(".hip", ".o"))


def _stale_obj(src):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps() + [os.path.join(CSRC, src)])


def _lib_current():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return not any(os.path.getmtime(d) > t for d in _deps() + [os.path.join(CSRC, s) for s in SOURCES])


def build(force=False, verbose=False):
    if not force and _lib_current():
        return LIB          # (the objects need not exist: the GPU box receives the library, not build/)
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
              "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-format-security"]
    if os.environ.get("CMCD_DIAG_HOOKS", "1") == "0":     # the boundary only: no measurement / diagnostic exports (cmcd_hip_diag.h)
        common.append("-DCMCD_NO_DIAG_HOOKS")
    todo = [s for s in SOURCES if force or _stale_obj(s)]

    def compile_one(src):
        cmd = common + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", _obj(src)]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)

    if todo:
        with ThreadPoolExecutor(max_workers=min(len(todo), int(os.environ.get("CMCD_BUILD_JOBS", "6")))) as ex:
            list(ex.map(compile_one, todo))
    objs = [_obj(s) for s in SOURCES]
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs, check=True)
    return LIB


BOUNDARY_LIB = os.path.join(HERE, "libcmcd_hip_boundary.so")
HOOK_SOURCES = ("cmcd_kernels.hip", "cmcd_coop.hip", "cmcd_uha.hip")     # the translation units that define hooks of cmcd_hip_diag.h


def build_boundary_only(force=False):
    """libcmcd_hip_boundary.so: the same library with the measurement / diagnostic hooks compiled out (-DCMCD_NO_DIAG_HOOKS) —
    its export list is include/cmcd_hip.h only.  Only the three sources that define hooks are recompiled; the other objects are
    the product build's.  Load it with CMCD_LIB_PATH (tests/test_gpu_fullsize.py runs the boundary and bench.py on it)."""
    build()
    srcs = [os.path.join(CSRC, s) for s in HOOK_SOURCES]
    if not force and os.path.exists(BOUNDARY_LIB) and \
            not any(os.path.getmtime(d) > os.path.getmtime(BOUNDARY_LIB) for d in _deps() + srcs + [LIB]):
        return BOUNDARY_LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DCMCD_NO_DIAG_HOOKS",
              "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wno-format-security"]

    def compile_one(src):
        out = os.path.join(OBJ, src.replace(".hip", ".boundary.o"))
        subprocess.run(common + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", out], check=True)
        return out

    with ThreadPoolExecutor(max_workers=3) as ex:
        special = dict(zip(HOOK_SOURCES, ex.map(compile_one, HOOK_SOURCES)))
    objs = [special.get(s, _obj(s)) for s in SOURCES]
    missing = [o for o in objs if not os.path.exists(o)]
    if missing:      # the product library was current but its objects are gone (a fresh checkout of a built tree): rebuild them
        build(force=True)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", BOUNDARY_LIB] + objs, check=True)
    return BOUNDARY_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(LIB)
    if "--boundary" in sys.argv:
        print(build_boundary_only(force="--force" in sys.argv))
