"""ctypes binding of libcmcd_hip.so (include/cmcd_hip.h).  No CPU fallback: if the library is
missing or a call fails, this raises."""
import ctypes as C
import os

import torch  # noqa: F401  (first: torch ships its own HIP runtime; loading libcmcd_hip.so before it would bring in /opt/rocm's
#                            copy as a second runtime in the process, and launches through it fail with "no ROCm-capable device")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CMCD_LIB_PATH", os.path.join(_HERE, "libcmcd_hip.so"))  # override: diagnostic builds

MODE = {"MCD_CAIS_sn": 0, "MCD_CAIS_var_sn": 1, "MCD_ULA": 2, "MCD_ULA_sn": 3, "MCD_CAIS_UHA_sn": 4}
ARCH = {"geffner": 0, "dds": 1}
TARGET = {"gmm": 0, "funnel": 1, "many_gmm": 2, "lgcp": 3}
EPS_SCHEDULE = {None: 0, "": 0, "none": 0, "linear": 1, "cos_sq": 2}
NSTATS = 5


class Desc(C.Structure):
    _fields_ = [(k, C.c_int32) for k in (
        "dim", "nbridges", "mode", "arch", "emb_dim", "target", "eps_schedule", "grad_clipping",
        "ngrid", "reserved")]


LAYOUT_FIELDS = (
    "vd_mean", "vd_logdiag", "eps", "mgridref_y", "gamma",
    "g_emb", "g_factor", "g_w1", "g_b1", "g_w2", "g_b2", "g_w3", "g_b3",
    "d_phase", "d_tw1", "d_tb1", "d_tw2", "d_tb2", "d_sw1", "d_sb1", "d_sw2", "d_sb2", "d_sw3", "d_sb3")


class Layout(C.Structure):
    _fields_ = [(k, C.c_int64) for k in LAYOUT_FIELDS]


class ProjectRange(C.Structure):
    """cmcd_project_range of include/cmcd_hip.h."""
    _fields_ = [("offset", C.c_int64), ("length", C.c_int64), ("kind", C.c_int32), ("reserved", C.c_int32),
                ("lo", C.c_float), ("hi", C.c_float)]


_lib = None
HAS_DIAG = False     # set by lib(): whether the loaded library carries the hooks of include/cmcd_hip_diag.h


def lib():
    """Loads the HIP library; raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m cmcd_amd.build` "
                "(there is no CPU fallback for the CMCD hot path)")
        L = C.CDLL(LIB_PATH)
        L.cmcd_version.restype = C.c_int
        L.cmcd_last_error.restype = C.c_char_p
        L.cmcd_workspace_bytes.restype = C.c_int64
        L.cmcd_workspace_bytes.argtypes = [C.POINTER(Desc), C.c_int64]
        L.cmcd_target_floats.restype = C.c_int64
        L.cmcd_target_floats.argtypes = [C.POINTER(Desc), C.c_int32]
        L.cmcd_bound_forward.restype = C.c_int
        L.cmcd_bound_forward.argtypes = [
            C.POINTER(Desc), C.POINTER(Layout), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
            C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmcd_bound_forward_prepared.restype = C.c_int
        L.cmcd_bound_forward_prepared.argtypes = L.cmcd_bound_forward.argtypes
        L.cmcd_stats_merge.restype = C.c_int
        L.cmcd_stats_merge.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int32,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.cmcd_grad_workspace_bytes.restype = C.c_int64
        L.cmcd_grad_workspace_bytes.argtypes = [C.POINTER(Desc), C.c_int64]
        L.cmcd_vargrad_weights.restype = C.c_int
        L.cmcd_vargrad_weights.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        L.cmcd_bound_var_grad.restype = C.c_int
        L.cmcd_bound_var_grad.argtypes = [C.POINTER(Desc), C.POINTER(Layout), C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p]
        L.cmcd_bound_var_grad_kept.restype = C.c_int
        L.cmcd_bound_var_grad_kept.argtypes = L.cmcd_bound_var_grad.argtypes
        L.cmcd_bound_var_forward.restype = C.c_int
        L.cmcd_bound_var_forward.argtypes = L.cmcd_bound_forward.argtypes
        L.cmcd_bound_grad_workspace_bytes.restype = C.c_int64
        L.cmcd_bound_grad_workspace_bytes.argtypes = [C.POINTER(Desc), C.c_int64]
        L.cmcd_bound_grad.restype = C.c_int
        L.cmcd_bound_grad.argtypes = [C.POINTER(Desc), C.POINTER(Layout), C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmcd_mfvi_workspace_bytes.restype = C.c_int64
        L.cmcd_mfvi_workspace_bytes.argtypes = [C.c_int32, C.c_int32, C.c_int64]
        L.cmcd_mfvi_bound_grad.restype = C.c_int
        L.cmcd_mfvi_bound_grad.argtypes = [C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_void_p,
                                           C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmcd_adam_step.restype = C.c_int
        L.cmcd_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                     C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_float,
                                     C.POINTER(ProjectRange), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.cmcd_adam_step_dev.restype = C.c_int
        L.cmcd_adam_step_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                         C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_float,
                                         C.POINTER(ProjectRange), C.c_int32, C.c_void_p, C.c_int64, C.c_void_p,
                                         C.c_void_p]
        L.cmcd_stats_merge_device.restype = C.c_int
        L.cmcd_stats_merge_device.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
        # measurement / diagnostic hooks (include/cmcd_hip_diag.h): absent from a boundary-only build (-DCMCD_NO_DIAG_HOOKS)
        global HAS_DIAG
        HAS_DIAG = hasattr(L, "cmcd_profile_enable")
        if HAS_DIAG:
            L.cmcd_debug_capture_noise.restype = C.c_int
            L.cmcd_debug_capture_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
            L.cmcd_debug_grad_item.restype = C.c_int
            L.cmcd_debug_grad_item.argtypes = [C.c_int]
            L.cmcd_last_kernel_name.restype = C.c_char_p
            L.cmcd_profile_enable.restype = C.c_int
            L.cmcd_profile_enable.argtypes = [C.c_int]
            L.cmcd_profile_collect.restype = C.c_int
            L.cmcd_profile_collect.argtypes = [C.POINTER(C.c_double), C.POINTER(C.c_int64)]
        _lib = L
    return _lib


def last_error():
    return lib().cmcd_last_error().decode()


def check(rc):
    if rc == 0:
        return
    msg = last_error()
    if rc == -2:
        raise NotImplementedError(msg)
    if rc == -1:
        raise ValueError(msg)
    raise RuntimeError(f"libcmcd_hip error {rc}: {msg}")


def stats_merge(stats_rows, n_per):
    """Host-side fixed-order merge of per-rank statistics -> (merged5, mean, var, lnZ)."""
    cnt = len(n_per)
    flat = (C.c_double * (NSTATS * cnt))(*[float(x) for row in stats_rows for x in row])
    ns = (C.c_int64 * cnt)(*[int(x) for x in n_per])
    merged = (C.c_double * NSTATS)()
    out3 = (C.c_double * 3)()
    check(lib().cmcd_stats_merge(flat, ns, cnt, merged, out3))
    return list(merged), out3[0], out3[1], out3[2]


_grad_item_sent = None


def sync_grad_item_override():
    """Forwards CMCD_GRAD_ITEM (unset / "0" / "1"; tests and tools/probes) to the library when it changed since the
    last gradient call of this process: the library reads no environment on its per-call path."""
    global _grad_item_sent
    want = os.environ.get("CMCD_GRAD_ITEM")
    if want != _grad_item_sent:
        L = lib()
        if HAS_DIAG:
            check(L.cmcd_debug_grad_item(-1 if want is None else int(want != "0")))
        elif want is not None:
            raise RuntimeError("CMCD_GRAD_ITEM needs a library built with the diagnostic hooks (include/cmcd_hip_diag.h)")
        _grad_item_sent = want


def last_kernel_name():
    """The trajectory kernel (or launch sequence) the last forward call of this thread enqueued ("" without the hooks)."""
    L = lib()
    return L.cmcd_last_kernel_name().decode() if HAS_DIAG else ""


def profile_enable(on=True):
    L = lib()
    if HAS_DIAG:
        check(L.cmcd_profile_enable(int(bool(on))))


def profile_collect():
    """-> (total trajectory-kernel milliseconds, launches) since the last enable/collect."""
    ms, cnt = C.c_double(), C.c_int64()
    L = lib()
    if not HAS_DIAG:
        return 0.0, 0        # callers fall back to the wall time per step (bench.py: leg_report)
    check(L.cmcd_profile_collect(C.byref(ms), C.byref(cnt)))
    return ms.value, cnt.value
