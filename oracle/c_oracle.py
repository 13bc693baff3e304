"""ctypes loader of the plain-C restatement (oracle/cmcd_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libcmcd_oracle_c.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        deps = [os.path.join(HERE, "cmcd_oracle.c"), os.path.join(HERE, "..", "include", "cmcd_hip.h")]
        if not os.path.exists(LIB) or any(os.path.exists(d) and os.path.getmtime(LIB) < os.path.getmtime(d) for d in deps):
            subprocess.run(["make", "-s", "-C", HERE], check=True)   # (the shared structs live in the header)
        _lib = C.CDLL(LIB)
        _lib.cmcd_oracle_threads.restype = C.c_int
        _lib.cmcd_oracle_bound.restype = C.c_int
    return _lib


def threads():
    return lib().cmcd_oracle_threads()


def set_threads(n):
    lib().cmcd_oracle_set_threads(C.c_int(int(n)))


def bound(desc, layout, seeds, params_flat, target_consts):
    """desc/layout: the ctypes structs of cmcd_amd._lib (same ABI structs); arrays: NumPy, host.
    -> (loss[n] f32, z[n, dim] f32)"""
    seeds = np.ascontiguousarray(seeds, np.int32)
    P = np.ascontiguousarray(params_flat, np.float32)
    tc = np.zeros(1, np.float32) if target_consts is None else np.ascontiguousarray(target_consts, np.float32)
    n, d = len(seeds), desc.dim
    loss = np.empty(n, np.float32)
    z = np.empty((n, d), np.float32)
    rc = lib().cmcd_oracle_bound(C.byref(desc), C.byref(layout), seeds.ctypes.data_as(C.c_void_p), C.c_int64(n),
                                 P.ctypes.data_as(C.c_void_p), tc.ctypes.data_as(C.c_void_p),
                                 C.c_int64(0 if target_consts is None else len(tc)),
                                 loss.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise NotImplementedError(f"C oracle does not cover this configuration (rc={rc})")
    return loss, z
