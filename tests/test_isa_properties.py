"""Code-generation properties of the hot kernels that the measured speed depends on (CHANGELOG.md (DESIGN r04 section 4), "second pass
of round 2"), checked on the gfx950 ISA that hipcc emits — no GPU needed.  Each assertion names a regression that was
worth several percent when it was found by reading the ISA:

* the cooperative kernel runs ONE LOOP PER ROLE (the roles as branches inside a shared loop body cost the MLP waves ~40
  phi copies and ~35 scalar branches per bridge);
* every 4x4x1 matrix instruction of the 8-particle instance takes its B operand by row broadcast (`blgp:4..7`), i.e.
  the layer-2 activations are fetched once per 16-lane row;
* the per-bridge schedule row is the hand-issued `s_load` (not a compiler-scheduled load with its own wait);
* the headline instance does not spill;
* the wave-per-tile kernel keeps the erfinv tail inside a branch (the compiler flattens it back into straight-line
  code unless its input passes through a volatile statement) and is built without the SLP vectoriser.
"""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cmcd_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def _asm(tmp_path_factory, src, extra=()):
    out = tmp_path_factory.mktemp("isa") / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
           "--cuda-device-only", "-S", "-o", str(out), os.path.join(CSRC, src), *extra]
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text().split("\n")


def _kernel_whole(lines, mangled_prefix):
    """-> (every line of the function up to its .Lfunc_end, the metadata lines behind it): for kernels with more than one
    `s_endpgm` (a role that leaves early sits in front of the loops)."""
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix + ":"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    tail = next(i for i in range(end, len(lines)) if "ScratchSize" in lines[i])
    return lines[start:end + 1], lines[end:tail + 1]


def _kernel(lines, mangled_prefix):
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix) and l.rstrip().endswith(":") or
                 (l.startswith(mangled_prefix) and ": " in l and "; @" in l))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].strip() == "s_endpgm")
    tail = next(i for i in range(end, len(lines)) if lines[i].startswith("; ScratchSize:") or "ScratchSize" in lines[i])
    return lines[start:end + 1], lines[end:tail + 1]


@pytest.fixture(scope="module")
def coop_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "cmcd_coop.hip")


@pytest.fixture(scope="module")
def traj_asm(tmp_path_factory):
    from cmcd_amd import build
    return _asm(tmp_path_factory, "cmcd_kernels.hip", build.EXTRA_FLAGS.get("cmcd_kernels.hip", []))


# many_gmm (2), dds (1), D = 2, T = 4, 8-particle tiles, separate RNG / ACC waves: the kernel the named batch runs
HEADLINE = "_ZN4cmcd11coop_kernelILi2ELi1ELi2ELi4ELb1ELb0EEEvNS_8TrajArgsE"


def test_headline_kernel_has_one_bridge_loop_per_role(coop_asm):
    body, _ = _kernel(coop_asm, HEADLINE)
    # top-level loops that contain a workgroup barrier = bridge loops; roles: MLP, TGT (two flavours), RNG, ACC
    loops, cur = [], None
    for l in body:
        if "Loop Header: Depth=1" in l:
            cur = {"barriers": 0}
            loops.append(cur)
        elif cur is not None and l.strip() == "s_barrier":
            cur["barriers"] += 1
    bridge_loops = [lp for lp in loops if lp["barriers"] >= 2]
    assert len(bridge_loops) >= 5, f"expected one bridge loop per role, found {len(bridge_loops)}"


def test_headline_kernel_broadcasts_the_layer2_operand(coop_asm):
    body, _ = _kernel(coop_asm, HEADLINE)
    mfma = [l for l in body if "v_mfma_f32_4x4x1_16b_f32" in l]
    assert len(mfma) >= 64          # 32 per bridge, two loop-body instantiations (even / odd evaluation)
    assert all(re.search(r"blgp:[4-7]", l) for l in mfma), "a 4x4x1 matrix instruction without the row broadcast"
    # per bridge body: the chain is fed by two 16-byte LDS reads per lane (it was eight)
    idx = [i for i, l in enumerate(body) if "v_mfma_f32_4x4x1_16b_f32" in l]
    first = idx[0]
    window = body[max(0, first - 12):first]
    assert sum("ds_read_b128" in l for l in window) == 2, window


def test_headline_kernel_schedule_row_is_the_hand_issued_scalar_load(coop_asm):
    body, _ = _kernel(coop_asm, HEADLINE)
    hand = 0
    for i, l in enumerate(body):
        if "#ASMSTART" in l and "s_load_dwordx4" in body[i + 1] and "s_load_dwordx4" in body[i + 2]:
            hand += 1
    assert hand >= 4, "the per-bridge schedule row should be requested by inline asm in every role loop that reads it"


def test_headline_kernel_does_not_spill(coop_asm):
    _, meta = _kernel(coop_asm, HEADLINE)
    scratch = next(l for l in meta if "ScratchSize" in l)
    assert re.search(r"ScratchSize:\s*0\b", scratch), scratch


def test_wave_per_tile_kernel_keeps_the_erfinv_tail_in_a_branch(traj_asm):
    body, _ = _kernel(traj_asm, "_ZN4cmcd11traj_kernelILi2ELi1ELi2ELi4ELb1ELb0EEEvNS_8TrajArgsE")
    marks = [i for i, l in enumerate(body) if "; erfinv tail" in l]
    assert marks, "marker of the tail branch not found"
    for i in marks:
        # the IEEE square root of the tail must come AFTER the marker (inside the branch), not be hoisted above it
        after = body[i:i + 12]
        assert any("v_sqrt_f32" in l for l in after), "the tail's square root was hoisted out of its branch"
    loop_start = max(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
    in_loop = body[loop_start:]
    sqrt_in_loop = sum("v_sqrt_f32" in l for l in in_loop)
    tails_in_loop = sum("; erfinv tail" in l for l in in_loop)
    assert sqrt_in_loop == tails_in_loop, "a square root outside a tail branch in the bridge loop"


def test_wave_per_tile_kernel_is_built_without_slp(traj_asm):
    body, _ = _kernel(traj_asm, "_ZN4cmcd11traj_kernelILi2ELi1ELi2ELi4ELb1ELb0EEEvNS_8TrajArgsE")
    packed = sum(1 for l in body if re.match(r"\s*v_pk_(fma|mul|add)_f32", l))
    assert packed <= 40, f"{packed} packed fp32 instructions: is -fno-slp-vectorize still applied to cmcd_kernels.hip?"


def test_the_9_tile_net_keeps_the_next_fragments_in_flight(traj_asm):
    """r05 (config 4 on one GPU 2.315 -> 2.003 ms): the layer-2 fragments of input tile ti + 1 are requested while tile ti's
    matrix instructions run.  Left alone the machine scheduler sinks every `ds_read_b128` to just in front of the four matrix
    instructions that use it, each behind a full `s_waitcnt lgkmcnt(0)`: 81 waits inside the matrix chain of one evaluation."""
    body, tail = _kernel_whole(traj_asm, "_ZN4cmcd11traj_kernelILi2ELi0ELi2ELi9ELb1ELb0EEEvNS_8TrajArgsE")
    loop_start = max(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
    loop = body[loop_start:]
    m = [i for i, l in enumerate(loop) if "v_mfma_f32_16x16x4" in l]
    assert len(m) >= 324                                            # 9 x 9 x 4 per evaluation
    chain = loop[m[0]:m[323] + 1]
    full_waits = sum(1 for l in chain if re.match(r"\s*s_waitcnt\s+lgkmcnt\(0\)", l))
    assert full_waits <= 12, f"{full_waits} full LDS waits inside the matrix chain: the fragment reads were sunk to their uses"
    assert not any("ScratchSize: " in l and "ScratchSize: 0" not in l for l in tail)


def test_the_132_wide_net_runs_without_its_twelve_zero_neurons(traj_asm):
    """r05 (config 4 on one GPU 1.93 -> 1.77 ms): eval_net_tail4 — per evaluation 8 x 8 x 4 + 8 `16x16x4` steps (the 4 real inputs
    of the ninth tile are ONE k-step) and 33 `4x4x1` steps for its 4 real outputs, instead of 324 `16x16x4`."""
    body, tail = _kernel_whole(traj_asm, "_ZN4cmcd11traj_kernelILi2ELi0ELi2ELi9ELb1ELb1EEEvNS_8TrajArgsE")
    loop_start = max(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
    loop = body[loop_start:]
    big = sum("v_mfma_f32_16x16x4" in l for l in loop)
    small = sum("v_mfma_f32_4x4x1" in l for l in loop)
    assert big == 264 and small == 33, (big, small)
    assert not any("ScratchSize: " in l and "ScratchSize: 0" not in l for l in tail)


@pytest.fixture(scope="module")
def wide8_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "cmcd_coop_wide.hip")


def test_funnel_dealt_coordinates_kernel_structure(wide8_asm):
    """coop_wide8_kernel<geffner, 10, T = 4> (BASELINE configs[1]): no scratch, one loop per role with TWO barriers per
    evaluation (r04's form had three), layer 2 on 32 `4x4x1` matrix instructions with row-broadcast activations, packed
    fp32 in layers 1 / 3, the schedule row as the hand-issued scalar load."""
    body, tail = _kernel_whole(wide8_asm, "_ZN4cmcd17coop_wide8_kernelILi0ELi10ELi4EEEvNS_8TrajArgsE")
    assert any("ScratchSize: 0" in l for l in tail)
    # one bridge loop per role (MLP, TGT, RNG, ACC; the compiler rotates and peels them, so barriers are counted for the whole
    # function): 2 prologue + 2 per evaluation (once or twice in the text of a peeled loop) + 1 hand-over per role — the
    # three-barrier form of r04 would need at least 4 x (2 + 3 + 1) with every loop emitted once
    assert sum("Loop Header: Depth=1" in l for l in body) >= 4
    nbar = sum(1 for l in body if l.strip() == "s_barrier")
    assert 4 * 5 <= nbar <= 4 * 7 + 2, nbar
    mf = [l for l in body if "v_mfma_f32_4x4x1" in l]
    assert len(mf) >= 32 and all("blgp:" in l for l in mf)
    assert sum(1 for l in body if re.match(r"\s*v_pk_(fma|mul)_f32", l)) >= 15
    assert sum(1 for l in body if "s_load_dwordx4" in l) >= 2


def test_wide_lgcp_gemm_loop_has_no_vector_address_arithmetic(tmp_path_factory):
    """r05 (N = 600: 17.4 -> 15.8 ms): the operands of lgcp_wide_gemm_kernel come through buffer descriptors with the chunk's
    position in a scalar offset — on gfx950 every VALU instruction of the loop adds to the matrix time of its SIMD."""
    lines = _asm(tmp_path_factory, "cmcd_lgcp_wide.hip")
    body, tail = _kernel_whole(lines, "_ZN4cmcd21lgcp_wide_gemm_kernelENS_8WideArgsE")
    assert any("ScratchSize: 0" in l for l in tail)
    # the contraction loop exists once per (operand shift or not) x (column blocks of the tile: 4, or 3 / 2 / 1 in the last column
    # tile of a layer, r05): every copy is free of vector address arithmetic and keeps its loads on buffer descriptors
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l]
    loops = []
    for h in heads:
        end = next(i for i in range(h, len(body)) if "s_cbranch" in body[i])
        seg = body[h:end]
        n_mfma = sum("v_mfma_f32_32x32x2" in l for l in seg)
        if n_mfma:
            loops.append((n_mfma, seg))
    assert sorted(n for n, _ in loops) == [12, 12, 24, 24, 36, 36, 48, 48], [n for n, _ in loops]
    for n_mfma, seg in loops:
        width = {48: "dwordx4", 36: "dwordx3", 24: "dwordx2", 12: "dword"}[n_mfma]
        n_w = sum(bool(re.search(r"buffer_load_%s\b" % width, l)) for l in seg)
        n_a = sum("buffer_load_dwordx4" in l for l in seg)
        # 3 chunks x (4 weight rows + 1 row of activations, always 16 bytes)
        assert (n_w == 15) if n_mfma == 48 else (n_w == 12 and n_a == 3), (n_mfma, width, n_w, n_a)
        assert not any(re.match(r"\s*(v_lshl_add_u64|v_mad_i64_i32|global_load)", l) for l in seg), "vector addresses are back in the loop"
    # of the two full-width loops one subtracts the operand shift, the other has no VALU arithmetic between its matrix instructions
    subs = sorted(sum(1 for l in seg if re.match(r"\s*v_sub(rev)?_f32", l)) for n, seg in loops if n == 48)
    assert subs[0] == 0 and subs[1] >= 12, subs


# ---------------------------------------------------------------------------------------------- 2nd-order mode (cmcd_uha.hip)
@pytest.fixture(scope="module")
def uha_asm(tmp_path_factory):
    from cmcd_amd import build
    return _asm(tmp_path_factory, "cmcd_uha.hip", build.EXTRA_FLAGS.get("cmcd_uha.hip", []))


# many_gmm (2), dds (1), D = 2, T = 4 on 8-particle tiles: the kernel bench.py's `second_order` line runs
UHA_HALF = "_ZN4cmcd15uha_coop_kernelILi2ELi1ELi2ELi4ELb1ELb0EEEvNS_8TrajArgsE"
# funnel (1), geffner (0), D = 10, T = 5 on 16-particle tiles: the instance that spilled 344 bytes per lane (and ran 21 000
# cycles per bridge) while the roles were branches of one loop body
UHA_FUNNEL = "_ZN4cmcd15uha_coop_kernelILi1ELi0ELi10ELi5ELb0ELb0EEEvNS_8TrajArgsE"
# r05: the same net on 8-particle tiles — state dealt over two waves, four MLP waves and a light tail wave for the 4 real
# neurons of the padded fifth tile
UHA_FUNNEL_TAIL = "_ZN4cmcd15uha_coop_kernelILi1ELi0ELi10ELi5ELb1ELb1EEEvNS_8TrajArgsE"


def _kernel_whole(lines, mangled_prefix):
    """Like _kernel, for kernels whose roles return separately (several s_endpgm): up to the function-end label."""
    start = next(i for i, l in enumerate(lines) if l.startswith(mangled_prefix + ":"))
    end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
    tail = next(i for i in range(end, len(lines)) if "ScratchSize" in lines[i])
    return lines[start:end + 1], lines[end:tail + 1]


def test_second_order_kernel_runs_one_loop_per_role_with_five_barriers_per_bridge(uha_asm):
    body, _ = _kernel_whole(uha_asm, UHA_HALF)
    barriers = [i for i, l in enumerate(body) if l.strip() == "s_barrier"]
    # one staging barrier, two prologue barriers and five per bridge IN EACH of the three roles' own code (as branches of one
    # shared loop body the whole kernel held nine)
    assert len(barriers) >= 1 + 3 * (2 + 5), len(barriers)
    # the barriers of the bridge loops wait for LDS only (the next bridge's bias row and the schedule scalars are in flight
    # across them): at most the staging barrier in front of the loops drains vmcnt
    drains = [body[i - 1] for i in barriers if "vmcnt(0)" in body[i - 1]]
    assert len(drains) <= 1, drains
    # and every role has a loop of its own: at least three depth-1 loops are followed by barriers before the next one starts
    heads = [i for i, l in enumerate(body) if "Loop Header: Depth=1" in l] + [len(body)]
    with_barriers = sum(any(h < b < heads[k + 1] for b in barriers) for k, h in enumerate(heads[:-1]))
    assert with_barriers >= 3, with_barriers


def test_second_order_kernel_on_8_particle_tiles_broadcasts_the_layer2_operand(uha_asm):
    body, _ = _kernel_whole(uha_asm, UHA_HALF)
    mfma = [l for l in body if "v_mfma_f32_4x4x1_16b_f32" in l]
    assert len(mfma) >= 64          # 32 per pass, two passes per bridge
    assert all(re.search(r"blgp:[4-7]", l) for l in mfma), "a 4x4x1 matrix instruction without the row broadcast"


def test_second_order_funnel_on_8_particle_tiles_runs_the_tail_as_one_short_pass(uha_asm):
    body, _ = _kernel_whole(uha_asm, UHA_FUNNEL_TAIL)
    mfma = [l for l in body if "v_mfma_f32_4x4x1_16b_f32" in l]
    own = [l for l in mfma if re.search(r"blgp:[4-7]", l)]
    tail = [l for l in mfma if not re.search(r"blgp:", l)]
    # per pass: 40 broadcast steps of an MLP wave's own tile; the tail wave's 12 steps multiply every block by its own slice
    assert len(own) == 80 and len(tail) == 24, (len(own), len(tail))
    # the layer-3 partials leave reduce-scattered: two LDS stores per pass, not one per output pair
    loop = body[body.index(own[0]):]
    stores = [l for l in loop if l.strip().startswith("ds_write_b32")]
    first_pass = loop[:loop.index(own[40])]
    assert sum(l.strip().startswith("ds_write_b32") for l in first_pass) <= 3, "layer-3 partials stored per pair again"
    assert stores


@pytest.mark.parametrize("kern", [UHA_HALF, UHA_FUNNEL, UHA_FUNNEL_TAIL],
                         ids=["named_shape_8_tiles", "funnel_16_tiles", "funnel_8_tiles_tail"])
def test_second_order_kernels_do_not_spill(uha_asm, kern):
    _, meta = _kernel_whole(uha_asm, kern)
    scratch = next(l for l in meta if "ScratchSize" in l)
    assert re.search(r"ScratchSize:\s*0\b", scratch), scratch


# ------------------------------------------------------------------------------------------ r04: the d = 1600 GEMM bodies
@pytest.fixture(scope="module")
def lgcp_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "cmcd_lgcp.hip")


@pytest.fixture(scope="module")
def wide_asm(tmp_path_factory):
    return _asm(tmp_path_factory, "cmcd_lgcp_wide.hip")


def _scratch(tail):
    return int(next(l for l in tail if "ScratchSize" in l).split(":")[1].split()[0])


def _vgprs(tail):
    return int(next(l for l in tail if "NumVgprs:" in l or "; NumVgprs" in l).split(":")[1].split()[0])


def test_wide_gemm_keeps_three_chunks_in_flight(wide_asm):
    """cmcd_lgcp_wide.hip: the loop body is unconditional and fenced by sched_barriers — the machine scheduler had sunk every
    load to its use, and a branch around an issue made the wait-count merge put `s_waitcnt vmcnt(0)` at the loop head."""
    body, tail = _kernel(wide_asm, "_ZN4cmcd21lgcp_wide_gemm_kernelENS_8WideArgsE")
    assert _scratch(tail) == 0
    start = next(i for i, l in enumerate(body) if "Loop Header: Depth=1" in l)
    end = next(i for i in range(start, len(body)) if "s_cbranch" in body[i])
    loop = body[start:end]
    mfma = [l for l in loop if "v_mfma_f32_32x32x2_f32" in l]
    loads = [l for l in loop if "buffer_load_dwordx4" in l]        # r05: through buffer descriptors (scalar chunk offsets)
    waits = [int(re.search(r"vmcnt\((\d+)\)", l).group(1)) for l in loop if "s_waitcnt vmcnt" in l]
    assert len(mfma) == 48 and len(loads) == 15, (len(mfma), len(loads))     # 3 chunks x (16 matrix instructions, 5 loads)
    assert waits and min(waits) >= 10, waits          # the oldest chunk only: two younger ones (10 loads) stay in flight


def test_no_split_k_gemm_issues_all_its_loads_before_the_first_matrix_instruction(lgcp_asm):
    """cmcd_lgcp.hip, lgcp_nsk_kernel: 26 sixteen-byte loads per lane in flight before the first wait (the scheduler left to
    itself keeps 43 registers and pays 26 round trips); the activation instance fits two workgroups per CU without spilling."""
    body, tail = _kernel_whole(lgcp_asm, "_ZN4cmcd15lgcp_nsk_kernelILi0ELb0ELi1EEEvNS_7NskArgsE")
    assert _scratch(tail) == 0
    assert _vgprs(tail) <= 128
    first = next(i for i, l in enumerate(body) if "v_mfma_f32_16x16x4_f32" in l)
    assert sum("global_load_dwordx4" in l for l in body[:first]) >= 26
    assert sum("v_mfma_f32_16x16x4_f32" in l for l in body) == 52
    # the 17 .. 20-particle form runs its extra rows on 4x4x1 against the same weight registers
    body_m, tail_m = _kernel_whole(lgcp_asm, "_ZN4cmcd15lgcp_nsk_kernelILi0ELb1ELi1EEEvNS_7NskArgsE")
    assert _scratch(tail_m) == 0
    assert sum("v_mfma_f32_4x4x1_16b_f32" in l for l in body_m) == 52 and sum("v_mfma_f32_16x16x4_f32" in l for l in body_m) == 52
    # the state-update instance waits for its OWN operands first: a vmcnt >= 26 in front of the first matrix instruction
    body_s, tail_s = _kernel_whole(lgcp_asm, "_ZN4cmcd15lgcp_nsk_kernelILi1ELb0ELi1EEEvNS_7NskArgsE")
    assert _scratch(tail_s) == 0
    first_s = next(i for i, l in enumerate(body_s) if "v_mfma_f32_16x16x4_f32" in l)
    pre = [int(m.group(1)) for l in body_s[:first_s] for m in [re.search(r"vmcnt\((\d+)\)", l)] if m]
    assert any(v >= 26 for v in pre), pre


# ------------------------------------------------------------------------------------------ r04: the gradient kernels
@pytest.fixture(scope="module")
def grad_asm(tmp_path_factory):
    from cmcd_amd import build
    return _asm(tmp_path_factory, "cmcd_grad.hip", build.EXTRA_FLAGS.get("cmcd_grad.hip", []))


def _total_vgprs(tail):
    return int(next(l for l in tail if "TotalNumVgprs:" in l).split(":")[1].split()[0])


# many_gmm (2), dds (1), D = 2, T = 4, 4 waves, W2 streamed, reparameterised / VarGrad, work items and whole chains
NARROW_GRAD = ["_ZN4cmcd11grad_kernelILi2ELi1ELi2ELi4ELi4ELb1ELb%dELb%dEEEvNS_8GradArgsE" % (b, i) for b in (0, 1) for i in (0, 1)]


@pytest.mark.parametrize("kern", NARROW_GRAD, ids=["vargrad_chains", "vargrad_items", "reparam_chains", "reparam_items"])
def test_narrow_gradient_instances_fit_two_workgroups_per_cu(grad_asm, kern):
    """r04: the narrow 2-d instances stream W2 from L2 and stage tile-locally so that two workgroups share a CU (69 KB of LDS
    each, requested at launch) — which only happens if a wave also stays within half the register file, without scratch
    (244.7 against 318.7 us for the named shape's reparameterised sweep, profiles/r04_grad_two_workgroups_per_cu.txt)."""
    body, tail = _kernel(grad_asm, kern)
    assert _scratch(tail) == 0
    assert _total_vgprs(tail) <= 256, "VGPRs + AGPRs past half the register file: one wave per SIMD"
    # the staged weight tiles of the LDS-resident form are gone: no ds_read of W2 fragments ahead of the layer-2 products means
    # the fragments come through global loads
    assert sum("global_load_dwordx4" in l for l in body) >= 8


def test_gradient_table_sums_have_a_store_path(grad_asm):
    """r04: the bias-row / schedule sums over tiles are per-tile slot stores + a fixed-order reduction launch; the float
    atomics of rounds 1 - 3 survive as the fallback for slot tables past 1 GB only (both paths are in the instance)."""
    body, _ = _kernel(grad_asm, NARROW_GRAD[3])
    assert any("global_store_dword" in l for l in body)
    assert any(l.startswith("_ZN4cmcd22grad_det_reduce_kernelENS_7DetArgsE") for l in grad_asm)
    red, tail = _kernel(grad_asm, "_ZN4cmcd22grad_det_reduce_kernelENS_7DetArgsE")
    assert not any("atomic" in l for l in red)
    assert _scratch(tail) == 0


def test_second_order_sweep_has_no_float_atomics(uha_asm):
    """r04: the 2nd-order sweep's shared-table sums are slot stores only (cmcd_uha.hip: UhaGradArgs::det)."""
    body, tail = _kernel_whole(uha_asm, "_ZN4cmcd15uha_grad_kernelILi2ELi1ELi2ELi4ELi4ELb1ELb0EEEvNS_11UhaGradArgsE")
    assert not any("global_atomic_add_f32" in l for l in body)
    scratch = next(l for l in tail if "ScratchSize" in l)
    assert re.search(r"ScratchSize:\s*0\b", scratch), scratch


@pytest.mark.parametrize("kern", ["_ZN4cmcd15lgcp_nsk_kernelILi3ELb1ELi1EEEvNS_7NskArgsE", "_ZN4cmcd15lgcp_nsk_kernelILi3ELb0ELi2EEEvNS_7NskArgsE"],
                         ids=["merged_one_round", "two_rounds"])
def test_reverse_sweep_gemm_consumer_sums_without_atomics(lgcp_asm, kern):
    """r04: the backward activation consumer of the d = 1600 reverse sweep (lgcp_nsk_kernel<3, ..>) — its column sums over the
    pass's particles have ONE writer per column (the tile's workgroup, fixed order through LDS): no atomics, no scratch."""
    body, tail = _kernel(lgcp_asm, kern)
    assert _scratch(tail) == 0
    assert not any("atomic" in l for l in body)
    assert sum("s_barrier" in l for l in body) >= 2      # cross-wave product sum, then the cross-wave column sums
