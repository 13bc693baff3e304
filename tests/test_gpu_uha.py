"""`config.boundmode = "MCD_CAIS_UHA_sn"` (2nd-order CMCD, /root/reference/src/mcd_under_lp_a_cais.py:6-115): the HIP path
against the float64 restatement on identical seeds and parameters, through the C ABI; its key chain bit for bit."""
import numpy as np
import pytest
import torch

from cmcd_amd import _lib
from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import synthetic
from oracle import prng

from helpers import compare_losses, run_oracle
from test_gpu_prng import ulp_distance

pytestmark = pytest.mark.gpu
MODE = "MCD_CAIS_UHA_sn"

FWD_CASES = [
    ("gmm_n300_k8", 300, {}),                                                              # geffner 2*2+20 = 24: 2 tiles
    ("funnel_n300_k64", 300, dict(init_eps=0.05, init_gamma=4.0)),                         # geffner 2*10+48 = 68: 5 tiles
    ("many_gmm_n2000_k256_dds", 256, dict(init_eps=0.2, init_gamma=2.0, init_sigma=15.0)),  # dds, first layer [68, 64]
    ("many_gmm_n2000_k256_dds", 100, dict(nbridges=16, init_eps=0.2, init_sigma=15.0)),    # ragged: 6 tiles + 4
    ("many_gmm_var_n16000_k256", 128, dict(nbridges=32, init_eps=0.1, init_gamma=3.0)),    # geffner 2*2+130 = 134: 9 tiles
    ("many_gmm_n2000_k256_dds", 2000, dict(init_eps=0.2, init_gamma=2.0)),                 # sigma_0 = 60: floored particles (+inf)
    ("gmm_n300_k8", 1, {}),                                                                # a single particle
    ("gmm_n300_k8", 200, dict(emb_dim=40, nbridges=12)),                                   # width 44 zero-padded to 64
    ("funnel_n300_k64", 40, dict(emb_dim=100, nbridges=6, init_eps=0.05)),                 # d = 10 on 120 -> 144 wide
    ("gmm_n300_k8", 1500, dict(nbridges=3, nn_arch="dds")),                                # many tiles, 4-wave workgroups
    ("funnel_n300_k64", 64, dict(nbridges=5, nn_arch="dds", init_eps=0.05)),               # d = 10, dds first layer [84, 64]
    ("gmm_n300_k8", 33, dict(nbridges=1)),                                                 # a single bridge
    ("funnel_n300_k64", 77, dict(emb_dim=56, nbridges=7, init_eps=0.05)),                  # 76 of 80 wide: five MLP waves, no tail
]


# 1 = one wave per tile (uha_traj_kernel); CU-cooperative (uha_coop_kernel: T MLP waves + state / target wave + RNG wave) on
# 3 = 16-particle tiles, 4 = 8-particle tiles (4x4x1 matrix instructions, 8 lanes per particle on the state wave)
KERNEL_NAMES = {1: "uha_traj_kernel", 3: "uha_coop_kernel<16-particle tiles>", 4: "uha_coop_kernel<8-particle tiles>"}


@pytest.fixture(params=[1, 3, 4], ids=["wave_per_tile", "cooperative16", "cooperative8"])
def variant(request, monkeypatch):
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", request.param)
    return request.param


# The oracle's answer for a case does not depend on the kernel form or the gradient path under test: computed once per
# (case, parameter set) and reused across the variants (the float64 restatement of a 2000 x 256 batch takes ~6 s, the
# autograd sweep of a 40-bridge case ~8 s — a minute of the suite otherwise).
_ORACLE_FWD, _ORACLE_GRAD = {}, {}


def _case_key(param_set, name, n, over):
    return (param_set, name, n, tuple(sorted(over.items())))


def _skip_without_instance(variant, name, over):
    if variant != 4 or over.get("nn_arch") == "dds" or name.endswith("_dds"):
        return
    dim = 10 if name.startswith("funnel") else 2
    emb = over.get("emb_dim", {"gmm": 20, "fun": 48, "man": 130}[name[:3]])
    if dim == 10 and 2 * dim + emb > 80:
        pytest.skip("the 9-tile nets on d = 10 have no 8-particle-tile instance")


@pytest.mark.parametrize("n,over", [(300, dict(init_eps=0.05, init_gamma=4.0)), (77, dict(nbridges=9, init_eps=0.05)),
                                    (5, dict(nbridges=3))])
def test_funnel_on_8_particle_tiles_with_and_without_the_tail(hip_lib, param_set, monkeypatch, n, over):
    """r05: the funnel's 8-particle form deals the state over two waves and gives the last tile's 4 real neurons to a light
    tail wave beside four MLP waves (form 4); form 5 keeps five MLP waves (the A / B partner).  Both against the restatement."""
    name = "funnel_n300_k64"
    b = synthetic.build(name, device="cuda", boundmode=MODE, **over)
    seeds = synthetic.parity_seeds(n)
    key = _case_key(param_set, name, n, over)
    if key not in _ORACLE_FWD:
        _ORACLE_FWD[key] = run_oracle(b, seeds, dtype=np.float64)
    l_ref, z_ref = _ORACLE_FWD[key]
    got = {}
    for form in (4, 5):
        monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", form)
        mean, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                                b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                                grad_clipping=b["grad_clipping"])
        torch.cuda.synchronize()
        assert _lib.last_kernel_name() == KERNEL_NAMES[4]
        compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"UHA funnel form {form} n={n}",
                       K=b["params_fixed"][1])
        got[form] = losses.cpu().numpy()
    # same key chain, same arithmetic up to the summation order of layer 2 / 3: the two forms stay close to each other too
    fin = np.isfinite(got[4]) & np.isfinite(got[5])
    assert np.allclose(got[4][fin], got[5][fin], rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("name,n,over", FWD_CASES)
def test_bound_matches_oracle(hip_lib, param_set, variant, name, n, over):
    _skip_without_instance(variant, name, over)
    b = synthetic.build(name, device="cuda", boundmode=MODE, **over)
    seeds = synthetic.parity_seeds(n)
    mean, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                            b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                            grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    key = _case_key(param_set, name, n, over)
    if key not in _ORACLE_FWD:
        _ORACLE_FWD[key] = run_oracle(b, seeds, dtype=np.float64)
    l_ref, z_ref = _ORACLE_FWD[key]
    rep = compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"UHA {name} n={n}", K=b["params_fixed"][1])
    print(name, n, over, rep)
    assert _lib.last_kernel_name() == KERNEL_NAMES[variant]
    want = np.mean(losses.double().cpu().numpy())
    if np.isfinite(want):
        assert abs(float(mean) - want) <= 1e-5 * max(1.0, abs(want))
    else:
        assert not np.isfinite(float(mean))


def test_descriptor_schedule_and_clip_are_ignored(hip_lib):
    """The function body fixes cos^2 and the 1e2 clip (:33-40,23-30): whatever the partial args say, same numbers."""
    seeds = torch.from_numpy(synthetic.parity_seeds(96)).cuda()
    b = synthetic.build("gmm_n300_k8", device="cuda", boundmode=MODE)
    ref = None
    for sched, clip in (("", False), ("linear", True), ("cos_sq", False)):
        _, (l, _) = mcdbm.compute_bound(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                                        eps_schedule=sched, grad_clipping=clip)
        ref = l if ref is None else ref
        assert torch.equal(l, ref)


def oracle_chain_uha(seeds, dim, K):
    """-> bits uint32 [K+2, N, dim], gen keys uint32 [K+1, N, 2], deviates float32 [K+2, N, dim]; stage 0 = z_0,
    stage 1 = rho_0 (mcd_under_lp_a_cais.py:92-93), stage i + 2 = bridge i (:55-56)."""
    k0 = prng.prng_key(seeds)
    a, b = prng.split(k0)
    bits, keys, dev = [prng.random_bits(a, dim)], [], [prng.normal(a, dim)]
    c, _ = prng.split(b)
    r, gp = prng.split(c)
    bits.append(prng.random_bits(r, dim))
    dev.append(prng.normal(r, dim))
    _, gen = prng.split(gp)
    keys.append(gen)
    for _ in range(K):
        g, h = prng.split(gen)
        bits.append(prng.random_bits(g, dim))
        dev.append(prng.normal(g, dim))
        _, gen = prng.split(h)
        keys.append(gen)
    return np.stack(bits), np.stack(keys), np.stack(dev)


@pytest.mark.parametrize("name,n,K", [("many_gmm_n2000_k256_dds", 203, 40), ("funnel_n300_k64", 77, 9),
                                      ("gmm_n300_k8", 2000, 64)])
def test_key_chain_and_deviates_are_bit_exact(hip_lib, variant, name, n, K):
    b = synthetic.build(name, device="cuda", boundmode=MODE, nbridges=K, init_eps=0.02, dense=True)
    dim = b["params_fixed"][0]
    seeds = np.random.default_rng(3).integers(1, 10 ** 6, n).astype(np.int32)
    seeds[:2] = (1, 999999)
    bits = torch.zeros(K + 2, n, dim, dtype=torch.int32, device="cuda")
    keys = torch.zeros(K + 1, n, 2, dtype=torch.int32, device="cuda")
    noise = torch.zeros(K + 2, n, dim, dtype=torch.float32, device="cuda")
    _lib.check(hip_lib.cmcd_debug_capture_noise(bits.data_ptr(), keys.data_ptr(), noise.data_ptr()))
    mcdbm.bound_forward(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    torch.cuda.synchronize()
    rb, rk, rd = oracle_chain_uha(seeds, dim, K)
    assert np.array_equal(keys.cpu().numpy().view(np.uint32), rk), "Threefry split chain differs from jax's"
    assert np.array_equal(bits.cpu().numpy().view(np.uint32), rb), "random_bits of the normal draws differ from jax's"
    d = ulp_distance(noise.cpu().numpy(), rd)
    assert d.max() <= 4, f"deviates differ by up to {d.max()} ulp"


# ---------------------------------------------------------------------------------------------- gradient
GRAD_CASES = [
    ("gmm_n300_k8", 128, {}),                                                               # geffner 24 (2 tiles)
    ("gmm_n300_k8", 64, dict(nn_arch="dds", init_gamma=3.0)),                               # dds on d = 2
    ("many_gmm_n2000_k256_dds", 96, dict(nbridges=8, init_eps=0.2, init_gamma=2.0, init_sigma=15.0)),
    ("many_gmm_n2000_k256_dds", 40, dict(nbridges=24, init_eps=0.1, init_gamma=3.0, init_sigma=15.0)),   # longer chain
    ("funnel_n300_k64", 70, dict(nbridges=6, init_eps=0.05, init_gamma=4.0)),               # d = 10, geffner 68 (5 tiles)
    ("funnel_n300_k64", 40, dict(nbridges=5, nn_arch="dds", init_eps=0.05)),                # d = 10: two input tiles
    ("funnel_n300_k64", 40, dict(nbridges=4, emb_dim=20, init_eps=0.05)),                   # width 40 -> 64
    ("many_gmm_var_n16000_k256", 80, dict(nbridges=5, init_eps=0.1, init_gamma=3.0)),       # geffner 134 (9 tiles)
    ("gmm_n300_k8", 50, dict(emb_dim=40, nbridges=5)),                                      # width 44 -> 64
    ("gmm_n300_k8", 300, dict(nbridges=3)),                                                 # 19 tiles: five workgroups, ragged
    ("gmm_n300_k8", 20, dict(nbridges=1, nn_arch="dds")),                                   # a single bridge
    ("gmm_n300_k8", 70, dict(nbridges=40, init_eps=0.05)),                                  # 41 points: ten chunks of work items
    ("funnel_n300_k64", 50, dict(nbridges=19, init_eps=0.03, init_gamma=4.0)),              # d = 10 in chunks (330-float items)
]


@pytest.mark.parametrize("item", [0, 1], ids=["whole_chain", "work_items"])
@pytest.mark.parametrize("name,n,over", GRAD_CASES)
def test_reparameterised_gradient_matches_autograd(hip_lib, param_set, variant, monkeypatch, name, n, over, item):
    _skip_without_instance(variant, name, over)
    if variant != 3 and GRAD_CASES.index((name, n, over)) not in (0, 2, 4, 7):
        pytest.skip("the wave-per-tile and 8-particle forwards keep the trajectory for four representative cases (suite time)")
    if item and variant != 3:
        pytest.skip("the work-item path (Jacobian launch, scan, chunked sweep) reads the same kept trajectory whatever forward "
                    "wrote it")
    # whole_chain: one sweep over every chain; work_items: cmcd_uha.hip's small-batch path (both pinned, not auto-selected)
    monkeypatch.setenv("CMCD_GRAD_ITEM", str(item))
    """jax.grad(compute_bound, 1) (/root/reference/src/main.py:174-176) through mcd_under_lp_a_cais.py:42-88: every leaf of
    params_flat (network, eps, gamma, q, mgridref_y) against torch-autograd through the float64 restatement."""
    from test_gpu_grad import _compare, oracle_grad_flat
    b = synthetic.build(name, device="cuda", boundmode=MODE, **over)
    seeds = synthetic.parity_seeds(n)
    grad, (losses, z) = mcdbm.compute_bound_grad(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                                 b["params_fixed"], b["target"], eps_schedule=b["eps_schedule"],
                                                 grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    key = _case_key(param_set, name, n, over)
    if key not in _ORACLE_GRAD:
        _ORACLE_GRAD[key] = oracle_grad_flat(b, seeds)
    val, l_ref, g_ref = _ORACLE_GRAD[key]
    assert np.isfinite(l_ref).all(), "pick a case without +inf particles for the gradient check"
    np.testing.assert_allclose(losses.double().cpu().numpy(), l_ref, rtol=2e-3, atol=2e-3)
    _compare(name, over, b["unflatten"], grad.double().cpu(), g_ref)
    assert float(g_ref[b["unflatten"].offset("gamma")].abs()) > 0     # the friction is a trained leaf in this mode


@pytest.mark.parametrize("item", [0, 1], ids=["whole_chain", "work_items"])
@pytest.mark.parametrize("name,n,over", [("many_gmm_n2000_k256_dds", 500, dict(nbridges=16, init_sigma=15.0)),
                                         ("gmm_n300_k8", 300, dict()),
                                         ("funnel_n300_k64", 100, dict(nbridges=8))])
def test_repeated_gradient_calls_are_bitwise_identical(hip_lib, monkeypatch, name, n, over, item):
    """The sums over particles behind d bias-table / d beta / d eps go through one slot per (tile, bridge / point) and a
    fixed-order reduction (cmcd_uha.hip: UhaGradArgs::det, uha_det_reduce_kernel); round 3 used float atomics on the shared
    tables, so the same call returned gradients that differed in the last bits from run to run."""
    monkeypatch.setenv("CMCD_GRAD_ITEM", str(item))
    b = synthetic.build(name, device="cuda", boundmode=MODE, **over)
    seeds = torch.from_numpy(synthetic.parity_seeds(n)).cuda()
    first = None
    noise = torch.randn(1 << 20, device="cuda")
    for rep in range(20):
        if rep % 3 == 1:
            noise = noise * 1.0001   # an unrelated launch in between
        grad, (losses, _) = mcdbm.compute_bound_grad(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
        g, l = grad.clone(), losses.clone()
        if first is None:
            first = (g, l)
            assert torch.isfinite(g).all()
        else:
            assert torch.equal(g, first[0]), (rep, float((g - first[0]).abs().max()))
            assert torch.equal(l, first[1])


def test_gradient_shards_add_up(hip_lib):
    """Multi-GPU contract: shards called with the global particle count sum to the single-call gradient."""
    b = synthetic.build("gmm_n300_k8", device="cuda", boundmode=MODE, dense=True)
    seeds = torch.from_numpy(synthetic.parity_seeds(200)).cuda()
    args = (b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    g_all, _ = mcdbm.compute_bound_grad(seeds, *args)
    g_a, _ = mcdbm.compute_bound_grad(seeds[:120], *args, n_total=200)
    g_b, _ = mcdbm.compute_bound_grad(seeds[120:], *args, n_total=200)
    torch.testing.assert_close(g_a + g_b, g_all, rtol=2e-4, atol=2e-6 * float(g_all.abs().max()))


def test_training_lowers_the_loss(hip_lib):
    """opt.run on the 2nd-order mode: Adam on the HIP gradient lowers the mean loss on fresh seeds."""
    from cmcd_amd import opt
    import types
    b = synthetic.build("funnel_n300_k64", device="cuda", boundmode=MODE, nbridges=8, init_eps=0.05, init_gamma=4.0)
    grad_and_loss, loss_fn = mcdbm.make_grad_and_loss(MODE)
    ev = torch.from_numpy(synthetic.throughput_seeds(2000, stream=9)).cuda()
    before = float(loss_fn(ev, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])[0])
    trainable = ("eta", "gamma", "eps", "vd", "mgridref_y")
    _, out, _ = opt.run(types.SimpleNamespace(N=300), 1e-3, 600, b["params_flat"].clone(), b["unflatten"], b["params_fixed"],
                        b["target"], grad_and_loss, trainable, 0)
    after = float(loss_fn(ev, out, b["unflatten"], b["params_fixed"], b["target"])[0])
    print("UHA funnel K=8 mean loss", before, "->", after)
    assert after < before - 0.2


# ---------------------------------------------------------------------------------------------- lgcp (d = 1600)
@pytest.mark.parametrize("n,k", [(20, 8), (5, 3), (40, 2), (20, 32), (20, 128)])
def test_lgcp_matches_oracle(hip_lib, param_set, n, k):
    """d = 1600, geffner net on concat(z, rho): width 2 * 1600 + 20 = 3220 (the reference's lgcp runs of this mode,
    /root/reference/src/notebooks/plotting_rebuttal.ipynb:3538-3548).  40 particles = two passes of the 32-row GEMM;
    (20, 128) is the configuration's own size (BASELINE.json configs[4] on this mode; /root/reference/src/mcd_under_lp_a_cais.py:42-88)."""
    from helpers import lgcp_counts_fixture
    counts = lgcp_counts_fixture()
    b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts, boundmode=MODE, nbridges=k, init_eps=0.02,
                        init_gamma=5.0)
    assert b["params_fixed"][3].width == 3220
    seeds = synthetic.parity_seeds(n)
    val, (losses, z) = mcdbm.compute_bound(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                           b["params_fixed"], b["target"])
    torch.cuda.synchronize()
    l_ref, z_ref = run_oracle(b, seeds, dtype=np.float64, lgcp_counts=counts)
    rep = compare_losses(losses.cpu().numpy(), l_ref, z.cpu().numpy(), z_ref, tag=f"UHA lgcp n={n} k={k}", K=k)
    print("UHA lgcp", n, k, rep, "mean loss", float(val), l_ref.mean())
    assert abs(float(val) - losses.double().mean().item()) <= 1e-5 * abs(l_ref.mean())


@pytest.mark.parametrize("n,k,reps", [(20, 128, 100), (40, 8, 30)])
def test_lgcp_repeat_calls_are_bitwise_identical(hip_lib, n, k, reps):
    """The 2nd-order launch sequences on d = 1600 at the configuration's size, the same call `reps` times: forward (8 GEMM
    launches per bridge) and the reverse sweep must return bit-identical results every time.  Every launch carries the
    split-K ticket protocol of cmcd_lgcp.hip (K-slice workgroups of a column block count arrivals on a device counter, the
    last one sums the slabs in fixed order and runs the fused consumer) — this sequence has more seams per bridge than the
    overdamped one stressed in tests/test_gpu_fullsize.py: a missing release / acquire would show as a stale slab, i.e. a
    run-to-run difference.  (40, 8) runs two concurrent passes on side streams."""
    from helpers import lgcp_counts_fixture
    b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=lgcp_counts_fixture(), boundmode=MODE, nbridges=k, N=n,
                        dense=True, init_eps=0.02, init_gamma=5.0)
    seeds = torch.from_numpy(synthetic.throughput_seeds(n, stream=4)).cuda()
    args = (b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    l0, z0, s0 = mcdbm.bound_forward(seeds, *args)
    torch.cuda.synchronize()
    assert torch.isfinite(l0).all()
    for r in range(reps):
        l, z, st = mcdbm.bound_forward(seeds, *args)
        assert torch.equal(l, l0) and torch.equal(z, z0) and torch.equal(st, s0), f"forward repeat {r} differs"
    g0, (lg0, _) = mcdbm.compute_bound_grad(seeds, *args)
    torch.cuda.synchronize()
    assert torch.equal(lg0, l0) and torch.isfinite(g0).all()
    for r in range(max(reps // 5, 6)):
        g, (lg, _) = mcdbm.compute_bound_grad(seeds, *args)
        assert torch.equal(lg, l0) and torch.equal(g, g0), f"gradient repeat {r} differs"
    # batch-composition invariance: a particle's loss does not depend on its row / pass.  Bitwise, except for passes of 17 .. 20
    # particles (r04): there rows 16 .. 19 share a workgroup with rows 0 .. 15 and run on v_mfma_f32_4x4x1 against the same weight
    # registers — the same products, summed per k quarter first — so a particle that moves across row 16 changes in the last
    # bits of every product (1e-7 relative, a few 1e-6 after 128 chaotic bridges); contamination from another row would be O(1)
    perm = torch.from_numpy(np.random.default_rng(1).permutation(n)).cuda()
    lp, _, _ = mcdbm.bound_forward(seeds[perm], *args)
    if 16 < n % 32 <= 20:
        torch.testing.assert_close(lp, l0[perm], rtol=5e-5, atol=1e-3)
    else:
        assert torch.equal(lp, l0[perm])


@pytest.mark.parametrize("n,K", [(5, 3), (37, 2)])
def test_lgcp_gradient_matches_autograd(hip_lib, param_set, n, K):
    """d = 1600: the reverse launch sequence (both network evaluations of every bridge recomputed and back-propagated,
    deferred A^T B parameter contractions over 2 K n rows) vs autograd through the float64 restatement.  n = 37 spans two
    passes of 32 particles."""
    from helpers import lgcp_counts_fixture
    from oracle import cmcd_oracle_torch as ot
    counts = lgcp_counts_fixture()
    b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=counts, boundmode=MODE, nbridges=K, N=n, init_eps=0.02,
                        init_gamma=5.0)
    seeds = synthetic.parity_seeds(n)
    grad, (losses, z) = mcdbm.compute_bound_grad(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"],
                                                 b["params_fixed"], b["target"])
    torch.cuda.synchronize()
    dim, _, mode, spec = b["params_fixed"]
    p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
    val, l_ref, z_ref, g = ot.bound_and_grad(seeds, p, dim, K, mode, spec.arch, ot.make_logp_lgcp(counts))
    np.testing.assert_allclose(losses.cpu().numpy(), l_ref, rtol=2e-4, atol=0.5)
    un = b["unflatten"]
    gh = grad.double().cpu().numpy()

    def leaf(*path):
        off, shape = un.layout[(0,) + path] if (0,) + path in un.layout else un.layout[(1,) + path]
        return gh[off:off + max(1, int(np.prod(shape)))].reshape(shape)
    (w1, b1), (w2, b2), (w3, b3) = [(("sn", "nn", i, 0), ("sn", "nn", i, 1)) for i in range(3)]
    checks = {"vd.mean": (leaf("vd", "mean"), g["vd"]["mean"]), "vd.logdiag": (leaf("vd", "logdiag"), g["vd"]["logdiag"]),
              "eps": (leaf("eps"), g["eps"]), "gamma": (leaf("gamma"), g["gamma"]),
              "mgridref_y": (leaf("mgridref_y"), g["mgridref_y"]),
              "W1": (leaf(*w1), g["sn"]["W1"]), "b1": (leaf(*b1), g["sn"]["b1"]), "W2": (leaf(*w2), g["sn"]["W2"]),
              "b2": (leaf(*b2), g["sn"]["b2"]), "W3": (leaf(*w3), g["sn"]["W3"]), "b3": (leaf(*b3), g["sn"]["b3"]),
              "emb": (leaf("sn", "emb"), g["sn"]["emb"]), "factor_sn": (leaf("sn", "factor_sn"), g["sn"]["factor_sn"])}
    worst = {}
    for name, (a, r) in checks.items():
        r = np.asarray(r, np.float64).reshape(a.shape)
        scale = max(np.abs(r).max(), 1e-12)
        worst[name] = (float(np.abs(a - r).max() / scale), float(scale))
    print({k: "%.1e (|ref| %.1e)" % v for k, v in worst.items()})
    bad = {k: v for k, v in worst.items() if v[0] > 5e-3 and v[1] > 1e-9}
    assert not bad, bad
