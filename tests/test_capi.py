"""The C-ABI library: loads, exports every symbol include/cmcd_hip.h declares, validates its
arguments before touching a GPU, and merges statistics exactly (no compute calls without a GPU)."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from cmcd_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header="cmcd_hip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cmcd_[a-z_0-9]+)\s*\(", src)))


def test_exports_every_declared_symbol(hip_lib):
    names = declared_functions()
    assert {"cmcd_version", "cmcd_last_error", "cmcd_workspace_bytes", "cmcd_bound_forward", "cmcd_bound_forward_prepared",
            "cmcd_stats_merge", "cmcd_target_floats", "cmcd_bound_grad", "cmcd_adam_step"} <= set(names)
    # the boundary header declares the boundary only: measurement / diagnostic hooks live in cmcd_hip_diag.h
    assert not [n for n in names if n.startswith(("cmcd_debug_", "cmcd_profile_")) or n == "cmcd_last_kernel_name"]
    for n in names:
        assert hasattr(hip_lib, n), f"{n} declared in cmcd_hip.h but not exported"
    diag = declared_functions("cmcd_hip_diag.h")
    assert {"cmcd_profile_enable", "cmcd_profile_collect", "cmcd_last_kernel_name", "cmcd_debug_capture_noise",
            "cmcd_debug_grad_item", "cmcd_debug_uha_xdump", "cmcd_debug_set_coop_prio"} == set(diag)
    for n in diag:      # the in-tree build (tests, bench) carries them
        assert hasattr(hip_lib, n), f"{n} declared in cmcd_hip_diag.h but not exported"
    assert hip_lib.cmcd_version() == 3


def test_boundary_library_exports_exactly_the_boundary_header():
    """cmcd_amd/libcmcd_hip_boundary.so (built by __graft_entry__.build() / `python -m cmcd_amd.build --boundary`): its dynamic
    export list IS include/cmcd_hip.h — every declared function, nothing else of the library's own."""
    import subprocess
    from cmcd_amd import build
    if not os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
        pytest.skip("hipcc not installed")
    lib = build.build_boundary_only()
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l and l.split()[-1].startswith("cmcd_")}
    assert exported == set(declared_functions()), exported ^ set(declared_functions())


def test_boundary_only_build_exports_no_hooks(tmp_path):
    """-DCMCD_NO_DIAG_HOOKS (CMCD_DIAG_HOOKS=0 python -m cmcd_amd.build): every symbol of the hooks' header is gone from the three
    translation units that define them, every boundary symbol they define is still there (checked on the objects: no link, no GPU)."""
    import subprocess
    from cmcd_amd import build
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    diag = set(declared_functions("cmcd_hip_diag.h"))
    seen = set()
    for src in ("cmcd_kernels.hip", "cmcd_coop.hip"):       # (cmcd_uha.hip holds one more hook behind the same guard)
        obj = tmp_path / (src + ".o")
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-DCMCD_NO_DIAG_HOOKS", "-I",
                        os.path.join(ROOT, "include"), "-I", build.CSRC, "-Wno-format-security"] + build.EXTRA_FLAGS.get(src, []) +
                       ["-c", os.path.join(build.CSRC, src), "-o", str(obj)], check=True)
        out = subprocess.run(["nm", "--defined-only", str(obj)], capture_output=True, text=True, check=True).stdout
        seen |= {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert not (seen & diag), seen & diag
    assert {"cmcd_bound_forward", "cmcd_bound_forward_prepared", "cmcd_version", "cmcd_stats_merge"} <= seen


def _desc(**kw):
    base = dict(dim=2, nbridges=8, mode=0, arch=1, emb_dim=64, target=2, eps_schedule=2, grad_clipping=1,
                ngrid=8, reserved=0)
    base.update(kw)
    return _lib.Desc(**base)


def test_struct_sizes_match_header():
    assert C.sizeof(_lib.Desc) == 40
    assert C.sizeof(_lib.Layout) == 8 * len(_lib.LAYOUT_FIELDS) == 8 * 24


def test_workspace_bytes_and_plugin_switch(hip_lib):
    d = _desc()
    nb = hip_lib.cmcd_workspace_bytes(C.byref(d), 2000)
    assert nb > 0 and nb % 4 == 0
    assert hip_lib.cmcd_workspace_bytes(C.byref(d), 4000) > nb          # per-wave partials grow
    # config.boundmode plugin surface: anything but the two CAIS modes is refused with the
    # reference's message (mcd_utils.py:190)
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(mode=7)), 2000) == 0
    assert _lib.last_error() == "Mode not implemented."
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(mode=2)), 2000) > 0          # MCD_ULA (arch placeholder dds)
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(mode=3, arch=0, emb_dim=20, target=0)), 300) > 0   # MCD_ULA_sn
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(mode=2, arch=0, emb_dim=20)), 300) == 0
    # MCD_CAIS_UHA_sn (2nd-order CMCD): network on concat(z, rho) -> geffner width 2 dim + emb_dim
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(mode=4)), 2000) > 0
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(mode=4, arch=0, emb_dim=48, target=1, dim=10)), 300) > \
        hip_lib.cmcd_workspace_bytes(C.byref(_desc(mode=0, arch=0, emb_dim=48, target=1, dim=10)), 300)
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(arch=5)), 2000) == 0
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(nbridges=0)), 2000) == 0
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(arch=0, emb_dim=20, target=0)), 300) > 0
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(arch=0, emb_dim=130)), 300) > 0
    assert hip_lib.cmcd_workspace_bytes(C.byref(_desc(arch=0, emb_dim=48, target=1, dim=10)), 300) > 0


def test_target_floats(hip_lib):
    assert hip_lib.cmcd_target_floats(C.byref(_desc()), 40) == 81
    assert hip_lib.cmcd_target_floats(C.byref(_desc(target=0)), 0) == 0
    assert hip_lib.cmcd_target_floats(C.byref(_desc(target=3, dim=1600)), 0) == 1600 * 1600 + 1600 + 3


def test_forward_rejects_bad_arguments_before_any_gpu_work(hip_lib):
    d = _desc()
    lay = _lib.Layout(*([-1] * len(_lib.LAYOUT_FIELDS)))
    rc = hip_lib.cmcd_bound_forward(C.byref(d), C.byref(lay), None, 16, None, 0, None, 0, None, 0, None, None,
                                    None, None)
    assert rc == -1 and "null pointer" in _lib.last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    rc = hip_lib.cmcd_bound_forward(C.byref(_desc(mode=9)), C.byref(lay), None, 16, None, 0, None, 0, None, 0,
                                    None, None, None, None)
    assert rc == -2
    with pytest.raises(NotImplementedError, match="Mode not implemented."):
        _lib.check(rc)


def test_stats_merge_matches_numpy():
    rng = np.random.default_rng(0)
    loss = rng.normal(3.0, 2.0, 1000)
    loss[[5, 700]] = np.inf
    from oracle.cmcd_oracle import ln_z, stats5
    parts = np.array_split(loss, [100, 333, 900])
    rows = [stats5(p) for p in parts]
    merged, mean, var, lnz = _lib.stats_merge(rows, [len(p) for p in parts])
    whole = stats5(loss)
    assert merged[0] == whole[0] == 998
    assert math.isinf(mean) and math.isnan(var)                       # inf semantics of the reference
    assert abs(lnz - ln_z(loss)) < 1e-12
    fin = loss[np.isfinite(loss)]
    parts = np.array_split(fin, 7)
    merged, mean, var, lnz = _lib.stats_merge([stats5(p) for p in parts], [len(p) for p in parts])
    assert abs(mean - fin.mean()) < 1e-12 and abs(var - fin.var()) < 1e-10 and abs(lnz - ln_z(fin)) < 1e-12


def test_stats_merge_empty_rank_and_all_inf():
    from oracle.cmcd_oracle import stats5
    empty = [0.0, 0.0, 0.0, -math.inf, 0.0]
    a = stats5(np.array([1.0, 2.0]))
    merged, mean, var, lnz = _lib.stats_merge([empty, a, empty], [0, 2, 0])
    assert merged[0] == 2 and abs(mean - 1.5) < 1e-15
    merged, mean, var, lnz = _lib.stats_merge([stats5(np.array([np.inf, np.inf]))], [2])
    assert merged[0] == 0 and lnz == -math.inf


def test_gradient_and_optimiser_entry_points_validate_before_any_gpu_work(hip_lib):
    """Workspace sizing and argument checks of the training entry points run on the host."""
    lay = _lib.Layout(*([-1] * len(_lib.LAYOUT_FIELDS)))
    # workspace sizes: the reparameterised gradient exists for CAIS_sn / ULA / ULA_sn, VarGrad for CAIS_var_sn
    assert hip_lib.cmcd_bound_grad_workspace_bytes(C.byref(_desc()), 2000) > hip_lib.cmcd_workspace_bytes(C.byref(_desc()), 2000)
    assert hip_lib.cmcd_bound_grad_workspace_bytes(C.byref(_desc(mode=1)), 2000) == 0
    assert hip_lib.cmcd_bound_grad_workspace_bytes(C.byref(_desc(mode=2)), 2000) > 0
    assert hip_lib.cmcd_bound_grad_workspace_bytes(C.byref(_desc(mode=3, arch=0, emb_dim=20, target=0)), 300) > 0
    assert hip_lib.cmcd_bound_grad_workspace_bytes(C.byref(_desc(arch=0, emb_dim=20, target=3, dim=1600, nbridges=4)), 20) > 0
    assert hip_lib.cmcd_grad_workspace_bytes(C.byref(_desc(mode=1)), 2000) > 0
    assert hip_lib.cmcd_grad_workspace_bytes(C.byref(_desc(mode=1, arch=0, emb_dim=31)), 2000) > 0        # width 33 runs padded to 64
    assert hip_lib.cmcd_grad_workspace_bytes(C.byref(_desc(mode=1, arch=0, emb_dim=200)), 2000) == 0      # width 202 > 144: no instance
    assert hip_lib.cmcd_mfvi_workspace_bytes(2, 2, 1000) > 0 and hip_lib.cmcd_mfvi_workspace_bytes(3, 1600, 20) > 0
    assert hip_lib.cmcd_mfvi_workspace_bytes(1, 7, 100) == 0                                              # funnel d = 7
    # null pointers / wrong modes are refused with a message
    rc = hip_lib.cmcd_bound_grad(C.byref(_desc()), C.byref(lay), None, 16, None, 0, None, 0, 1.0, None, 0, None, None,
                                 None, None, None)
    assert rc == -1
    rc = hip_lib.cmcd_bound_grad(C.byref(_desc(mode=1)), C.byref(lay), None, 16, None, 0, None, 0, 1.0, None, 0, None,
                                 None, None, C.c_void_p(16), None)
    assert rc == -2 and "MCD_CAIS_var_sn" in _lib.last_error()
    rc = hip_lib.cmcd_bound_var_grad(C.byref(_desc()), C.byref(lay), C.c_void_p(16), 16, C.c_void_p(16), 0, None, 0,
                                     C.c_void_p(16), C.c_void_p(16), 0, C.c_void_p(16), None)
    assert rc == -2
    rc = hip_lib.cmcd_mfvi_bound_grad(2, 2, 0, 2, None, 16, None, 4, None, 0, 1.0, None, 0, None, None, None, None, None)
    assert rc == -1
    rc = hip_lib.cmcd_adam_step(None, None, None, None, None, 10, 1e-3, 0.9, 0.999, 1e-8, 5.0, 1, 1e-3, None, 0, None, 0,
                                None, None)
    assert rc == -1
    rng = (_lib.ProjectRange * 9)()
    rc = hip_lib.cmcd_adam_step(C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), C.c_void_p(16), None, 10, 1e-3, 0.9, 0.999,
                                1e-8, 5.0, 1, 1e-3, rng, 9, None, 0, None, None)
    assert rc == -1 and "8 projection ranges" in _lib.last_error()
    assert C.sizeof(_lib.ProjectRange) == 32
