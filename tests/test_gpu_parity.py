"""HIP path vs the CPU oracle on identical seeds and parameters (through the C ABI)."""
import numpy as np
import pytest
import torch

from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import synthetic
from oracle import cmcd_oracle as orc

from helpers import compare_losses, run_oracle

pytestmark = pytest.mark.gpu

CASES = [
    ("gmm_n300_k8", 300, {}),
    ("funnel_n300_k64", 300, {}),
    ("many_gmm_n2000_k256_dds", 256, {}),
    ("many_gmm_n2000_k256_dds", 100, dict(nbridges=16)),          # ragged: 100 = 6 tiles + 4
    ("many_gmm_var_n16000_k256", 128, dict(nbridges=64)),
    ("many_gmm_n2000_k256_dds", 64, dict(nbridges=8, eps_schedule="linear", init_eps=0.05)),
    ("gmm_n300_k8", 1, {}),                                       # a single particle
    ("many_gmm_n2000_k256_dds", 9, dict(nbridges=4)),             # one 8-particle tile + 1 (twin-column tiling, r02 lane order)
    ("many_gmm_var_n16000_k256", 2041, dict(nbridges=4)),         # the 132-wide net at the top of the 8-particle range, ragged
    ("funnel_n300_k64", 40, dict(emb_dim=122, nbridges=6)),       # d = 10 on the 132-wide net: the 13-wave cooperative instance
]


# 2 = cooperative with the library's tile choice (8-particle tiles up to 2048 particles), 3 = 16-particle tiles forced
@pytest.fixture(params=[1, 2, 3], ids=["wave_per_tile", "cooperative", "cooperative_16"])
def variant(request, monkeypatch):
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", request.param)
    return request.param


@pytest.mark.parametrize("name,n,over", CASES)
def test_bound_matches_oracle(hip_lib, param_set, variant, name, n, over):
    b = synthetic.build(name, device="cuda", **over)
    seeds = synthetic.parity_seeds(n)
    fn = mcdbm.compute_bound_var if "var" in b["cfg"]["boundmode"] else mcdbm.compute_bound
    val, (losses, z) = fn(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"],
                          b["target"], eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64)
    rep = compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"{name} n={n}", K=b["params_fixed"][1])
    print(name, n, over, rep)
    # the scalar the reference returns
    lh = losses.double().cpu().numpy()
    if "var" in b["cfg"]["boundmode"]:
        want = np.clip(np.var(lh), -1e7, 1e7)
    else:
        want = np.mean(lh)
    if np.isfinite(want):
        assert abs(float(val) - want) <= 1e-5 * max(1.0, abs(want))
    else:
        assert not np.isfinite(float(val))


@pytest.mark.parametrize("form", [4, 5], ids=["dealt_coordinates", "narrow_form"])
@pytest.mark.parametrize("n,over", [
    (300, {}),                                                    # BASELINE configs[1] itself
    (13, dict(nbridges=9)),                                       # ragged: one full tile + 5; odd bridge count
    (1, dict(nbridges=2)),
    (64, dict(emb_dim=20, nbridges=12)),                          # 30-wide net: the two-MLP-wave instance
    (64, dict(nn_arch="dds", nbridges=12)),                       # PISGRADNet on the funnel
    (96, dict(boundmode="MCD_CAIS_var_sn", nbridges=12, grad_clipping=True, init_sigma=3.0)),   # both clips (1e2) active
    (96, dict(boundmode="MCD_ULA_sn", nbridges=12)),
    (2048, dict(nbridges=3)),                                     # the top of the 8-particle range: 256 workgroups
])
def test_funnel_on_8_particle_tiles(hip_lib, param_set, monkeypatch, n, over, form):
    """d = 10 on 8-particle tiles: kernel variant 4 = coop_wide8_kernel (cmcd_coop_wide.hip: every per-coordinate job dealt to
    the lanes of its particle; what a funnel batch of <= 2048 particles runs on), 5 = the same batch on coop_kernel's 8-particle
    instance (the r04 form, kept for A / B)."""
    from cmcd_amd import _lib
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", form)
    b = synthetic.build("funnel_n300_k64", device="cuda", **over)
    seeds = synthetic.parity_seeds(n)
    fn = mcdbm.compute_bound_var if "var" in b["cfg"]["boundmode"] else mcdbm.compute_bound
    val, (losses, z) = fn(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                          eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    assert _lib.last_kernel_name() == ("coop_wide8_kernel<8-particle tiles>" if form == 4 else "coop_kernel<8-particle tiles>")
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64)
    rep = compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"funnel n={n} {over} form={form}",
                         K=b["params_fixed"][1])
    print(n, over, form, rep)
    # determinism and batch-composition invariance of the new kernel: same bits on a repeat, and for a particle launched in
    # another tile / column
    if form == 4 and n >= 13:
        l2 = fn(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])[1][0]
        assert torch.equal(l2, losses)
        if "var" not in b["cfg"]["boundmode"]:
            perm = np.random.default_rng(0).permutation(n)
            lp, zp = fn(torch.from_numpy(seeds[perm]).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                        eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])[1]
            assert torch.equal(lp.cpu(), losses.cpu()[perm]) and torch.equal(zp.cpu(), z.cpu()[perm])


@pytest.mark.parametrize("n,k,form", [(20, 8, 0), (5, 3, 0), (40, 2, 0), (20, 128, 0), (600, 16, 0), (17, 3, 0), (16, 2, 0),
                                      (20, 8, 3), (40, 2, 3), (20, 128, 3)])
def test_lgcp_matches_oracle(hip_lib, param_set, monkeypatch, n, k, form):
    """d = 1600 (config 5): per-bridge GEMM path.  40 particles = two passes of 32 rows; (20, 128) is the
    configuration's own size (BASELINE.json configs[4]), 600 particles the evaluation batch of the reference's lgcp
    runs (n_samples 500-600 per seed group, /root/reference/README.md:63; the wide-batch form).  form 0 = the library's
    choice (r04: the no-split-K GEMM on packed operands for <= 32-row passes; 16 / 17 particles = one row half / one row
    into the second), form 3 = the split-K launch sequence (still the recompute of the reverse sweep and the fallback)."""
    import os
    from cmcd_amd import _lib
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", form)
    counts = np.load(os.path.join(os.path.dirname(__file__), "golden", "lgcp_bin_counts.npy"))
    b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts, nbridges=k)
    seeds = synthetic.parity_seeds(n)
    val, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                           b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                           grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64, lgcp_counts=counts)
    rep = compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"lgcp n={n} k={k}", K=k)
    print("lgcp", n, k, rep, "mean loss", float(val), l_ref.mean())
    assert abs(float(val) - losses.double().mean().item()) <= 1e-5 * abs(l_ref.mean())


@pytest.mark.parametrize("mode,n,k,over", [
    ("MCD_CAIS_sn", 130, 3, {}),                                   # 4 full row tiles + 2 rows
    ("MCD_CAIS_sn", 33, 2, dict(grad_clipping=True)),              # one row past a tile; clip on grad log p
    ("MCD_CAIS_var_sn", 70, 2, dict(grad_clipping=True)),          # clip at 1e2 on both gradients (mcd_cais_var.py:33-40)
    ("MCD_ULA_sn", 40, 3, {}),                                     # network in the backward kernel only (mcd_over_orig.py)
    ("MCD_ULA", 40, 3, {}),                                        # no network: one launch per evaluation
    ("MCD_CAIS_sn", 257, 2, dict(emb_dim=12)),                     # width 1612: other padding of rows / columns
    ("MCD_CAIS_sn", 96, 16, dict(eps_schedule="cos_sq", init_eps=1e-3)),
    # grids of MORE workgroups than the chip holds at once (832 / 455 tiles): the row tiles of late column tiles start after
    # early ones have finished.  r04: MCD_ULA updated the state in place while it was the launch's own operand — invisible on
    # one-wave grids, a 2.4-nat ELBO shift on the reference's 15 000-particle evaluation (tests/test_gpu_reference_tables.py)
    ("MCD_ULA", 2048, 3, dict(init_eps=2e-4)),
    ("MCD_CAIS_sn", 1100, 2, {}),
    # r05: on many-round grids (>= 512 tiles per launch) the last column tile of a layer runs only the column blocks that hold
    # weights (3 of 4 for the 1620-wide layers, 2 of 4 for the 1600-wide ones; 1612 wide: 3) and the XCD ranges are cut by cost
    ("MCD_CAIS_sn", 1400, 2, {}),
    ("MCD_CAIS_var_sn", 1320, 2, dict(emb_dim=12, grad_clipping=True)),
])
def test_lgcp_wide_batch_path_matches_oracle(hip_lib, param_set, monkeypatch, mode, n, k, over):
    """The wide-batch form of the d = 1600 path (cmcd_lgcp_wide.hip: whole-batch launches of a 32 x 128-tile fp32 GEMM body,
    taken by forward-only calls of >= 225 particles (kLgcpWideMin: the crossover measured at K = 128) — the reference's evaluation batches, /root/reference/src/opt.py:167-197)
    pinned here at small sizes through the kernel-variant hook, for every mode flag of its state update, against the
    float64 oracle AND against the 32-row launch sequence on the same seeds."""
    from cmcd_amd import _lib
    from helpers import lgcp_counts_fixture, oracle_target
    counts = lgcp_counts_fixture()
    b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts, nbridges=k, boundmode=mode, **over)
    dim, K, _, spec = b["params_fixed"]
    if mode == "MCD_ULA":    # the reference keeps no network for this mode: params_fixed[3] is None
        flat, unflatten, fixed = mcdbm.initialize(dim=dim, nbridges=K, vdparams=None, eps=b["cfg"]["init_eps"],
                                                  trainable=("eps",), mode="MCD_ULA", device="cuda")
        b = dict(b, params_flat=flat, unflatten=unflatten, params_fixed=fixed)
    seeds = synthetic.parity_seeds(n)
    fn = mcdbm.compute_bound_var if "var" in mode else mcdbm.compute_bound
    out = {}
    for variant in (2, 1, 3):       # wide batch | 32-row passes (no-split-K GEMM) | 32-row passes (split-K GEMM)
        monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", variant)
        val, (losses, z) = fn(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"],
                              b["target"], eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
        torch.cuda.synchronize()
        out[variant] = (float(val), losses.cpu().numpy(), z.cpu().numpy(), _lib.last_kernel_name())
    assert out[2][3].startswith("lgcp wide-batch") and out[1][3].startswith("lgcp launch sequence"), (out[2][3], out[1][3])
    if mode == "MCD_ULA":
        train, notrain = b["unflatten"](b["params_flat"].cpu())
        allp = {**train, **notrain}
        f = lambda t: np.asarray(t.numpy(), np.float64)
        p = {"vd": {kk: f(v) for kk, v in allp["vd"].items()}, "eps": f(allp["eps"]), "mgridref_y": f(allp["mgridref_y"]),
             "gridref_x": f(allp["gridref_x"]), "target_x": f(allp["target_x"])}
        l_ref, z_ref = orc.compute_log_elbo_batch(seeds, p, dim, K, mode, "dds", oracle_target(b["cfg"], counts), dtype=np.float64)
    elif mode == "MCD_ULA_sn":
        p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
        l_ref, z_ref = orc.compute_log_elbo_batch(seeds, p, dim, K, mode, spec.arch, oracle_target(b["cfg"], counts), dtype=np.float64)
    else:
        l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64, lgcp_counts=counts)
    rep = compare_losses(out[2][1], l_ref, out[2][2], z_ref, tag=f"lgcp wide {mode} n={n} k={k}", K=k)
    print("lgcp wide", mode, n, k, over, rep)
    # the two forms of the path sum the contractions in different orders: float32 rounding apart, nothing else
    np.testing.assert_allclose(out[2][1], out[1][1], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(out[2][2], out[1][2], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(out[3][1], out[1][1], rtol=2e-5, atol=2e-3)
    np.testing.assert_allclose(out[3][2], out[1][2], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name,mode,n,over", [
    ("gmm_n300_k8", "MCD_ULA", 300, {}),
    ("gmm_n300_k8", "MCD_ULA_sn", 300, {}),
    ("many_gmm_n2000_k256_dds", "MCD_ULA_sn", 200, dict(nbridges=32, init_eps=0.3, init_sigma=15.0)),
    ("many_gmm_n2000_k256_dds", "MCD_ULA", 200, dict(nbridges=32, init_eps=0.3, init_sigma=15.0)),
    ("funnel_n300_k64", "MCD_ULA_sn", 100, dict(nbridges=16)),
    ("many_gmm_var_n16000_k256", "MCD_ULA_sn", 70, dict(nbridges=6)),     # the 132-wide net (eval_net_tail4 on the wave-per-tile form)
])
def test_sibling_overdamped_modes_match_oracle(hip_lib, param_set, variant, name, mode, n, over):
    """config.boundmode = MCD_ULA / MCD_ULA_sn (reference mcd_over_orig.py) on the same kernels."""
    if mode == "MCD_ULA" and variant >= 2:
        pytest.skip("MCD_ULA has no network: the cooperative (MLP-split) kernel does not apply")
    b = synthetic.build(name, device="cuda", boundmode=mode, **over)
    dim, K, _, spec = b["params_fixed"]
    if mode == "MCD_ULA":    # the reference keeps no network for this mode: params_fixed[3] is None
        flat, unflatten, fixed = mcdbm.initialize(dim=dim, nbridges=K, vdparams=None, eps=b["cfg"]["init_eps"],
                                                  trainable=("eps",), mode="MCD_ULA", device="cuda")
        assert fixed[3] is None
        b = dict(b, params_flat=flat, unflatten=unflatten, params_fixed=fixed)
    seeds = synthetic.parity_seeds(n)
    _, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                         b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                         grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    train, notrain = b["unflatten"](b["params_flat"].cpu())
    if mode == "MCD_ULA":
        allp = {**train, **notrain}
        f = lambda t: np.asarray(t.numpy(), np.float64)
        p = {"vd": {k: f(v) for k, v in allp["vd"].items()}, "eps": f(allp["eps"]), "mgridref_y": f(allp["mgridref_y"]),
             "gridref_x": f(allp["gridref_x"]), "target_x": f(allp["target_x"])}
        arch = "dds"
    else:
        p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
        arch = spec.arch
    from helpers import oracle_target
    l_ref, z_ref = orc.compute_log_elbo_batch(seeds, p, dim, K, mode, arch, oracle_target(b["cfg"]), dtype=np.float64)
    rep = compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"{name} {mode}", K=K)
    print(name, mode, rep)


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("n_mixes", [7, 17, 64])
def test_many_gmm_with_other_mixture_sizes(hip_lib, monkeypatch, n_mixes, variant):
    """config.n_mixes != 40 takes the generic component loop (the 40-mode fast path keeps squared distances — in the
    cooperative kernel also the means — in registers); 64 is the library's maximum."""
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", variant)
    b = synthetic.build("many_gmm_n2000_k256_dds", device="cuda", nbridges=24, n_mixes=n_mixes, init_sigma=20.0)
    seeds = synthetic.parity_seeds(333)
    mean, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                            b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                            grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64)
    compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"n_mixes={n_mixes}", K=24)


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("model,emb_dim", [("gmm", 5), ("many_gmm", 40), ("many_gmm", 70), ("funnel", 30)])
def test_network_widths_between_the_instances_run_zero_padded(hip_lib, param_set, monkeypatch, model, emb_dim, variant):
    """config.emb_dim is free in the reference (README: --config.emb_dim 40).  Widths without an instance of their
    own (here 7, 42, 72 and 40) run on the next larger one with zero-padded weights: same numbers."""
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", variant)
    name = {"gmm": "gmm_n300_k8", "many_gmm": "many_gmm_var_n16000_k256", "funnel": "funnel_n300_k64"}[model]
    b = synthetic.build(name, device="cuda", emb_dim=emb_dim, nbridges=12, boundmode="MCD_CAIS_sn")
    seeds = synthetic.parity_seeds(200)
    mean, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                            b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                            grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64)
    compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"{model} emb_dim={emb_dim}", K=12)


@pytest.mark.parametrize("variant", [2, 3, 4])
@pytest.mark.parametrize("emb_dim", [128, 130, 131])
def test_gmm_target_on_the_132_wide_net(hip_lib, param_set, monkeypatch, emb_dim, variant):
    """The 2-d gmm target with config 4's net width (emb_dim 130 -> 132 hidden units: eight tiles + 4 neurons; 128 -> 130: + 2;
    131 -> 133: the general nine-tile form): the wave-per-tile kernel's tail form (eval_net_tail4, r05) has its own gmm instance."""
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", variant)
    b = synthetic.build("gmm_n300_k8", device="cuda", emb_dim=emb_dim, nbridges=7)
    seeds = synthetic.parity_seeds(77)
    mean, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                            b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                            grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64)
    compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"gmm emb_dim={emb_dim} variant={variant}", K=7)


@pytest.mark.parametrize("emb_dim", [127, 128, 129, 130, 131, 136])
def test_widths_around_the_132_wide_net_on_the_cooperative_kernels(hip_lib, param_set, monkeypatch, emb_dim, variant):
    """129 ... 132 hidden units (emb_dim 127 ... 130, d = 2) run the ninth MLP wave of the 8-particle tiling in its
    4-neuron form (eight contraction slices, cmcd_common.h coop_tail4), with 1 ... 4 real neurons; 133 and 138 units run it
    in the general form; the 16-particle tiling has only the general form.  Same numbers as the oracle either way."""
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", variant)
    b = synthetic.build("many_gmm_var_n16000_k256", device="cuda", emb_dim=emb_dim, nbridges=9, boundmode="MCD_CAIS_sn")
    seeds = synthetic.parity_seeds(203)            # ragged last tile on both tilings
    mean, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                            b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                            grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64)
    compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"many_gmm emb_dim={emb_dim} variant={variant}", K=9)


@pytest.mark.parametrize("tag", ["gmm_k8", "funnel_k64", "many_gmm_dds_k256", "many_gmm_var_k32", "dense_gmm_k8",
                                 "dense_funnel_k64", "dense_many_gmm_dds_k256", "dense_many_gmm_var_k32"])
def test_bound_matches_committed_golden_vectors(hip_lib, variant, tag):
    """The HIP path against tests/golden/oracle_*.npz (float64 restatement, tools/make_golden.py) — no oracle code runs."""
    from test_golden import load_case
    g, name, over = load_case(tag)
    b = synthetic.build(name, device="cuda", **over)
    fn = mcdbm.compute_bound_var if "var" in b["cfg"]["boundmode"] else mcdbm.compute_bound
    _, (losses, z) = fn(torch.from_numpy(g["seeds"]).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"],
                        b["target"], eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    compare_losses(losses.cpu().numpy(), g["loss"], z.cpu().numpy(), g["z"], tag=f"golden {tag}", K=b["params_fixed"][1])
