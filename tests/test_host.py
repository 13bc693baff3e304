"""Host-side mirror of the reference interface: initialize / unflatten / load_model / error behaviour."""
import numpy as np
import pytest
import torch

from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import model_handler, synthetic
from cmcd_amd import variationaldist as vd


def test_initialize_signature_and_params_fixed():
    flat, unflatten, fixed = mcdbm.initialize(
        dim=2, nbridges=256, vdparams=vd.initialize(2, 60.0), eta=0.0, eps=1.0,
        trainable=("eta", "gamma", "mgridref_y"), mode="MCD_CAIS_sn", emb_dim=20, nlayers=3, nn_arch="dds",
        device="cpu")
    dim, K, mode, spec = fixed
    assert (dim, K, mode) == (2, 256, "MCD_CAIS_sn") and spec.arch == "dds" and spec.width == 64
    hash(fixed)                                              # static_argnums-style hashable
    train, notrain = unflatten(flat)
    assert set(train) == {"eta", "gamma", "mgridref_y", "sn"}
    assert set(notrain) == {"vd", "eps", "gridref_x", "target_x"}
    # dds: 64 + (128*64+64) + (64*64+64) + (66*64+64) + (64*64+64) + (64*2+2) = 21058 floats
    n_sn = sum(v.numel() for m in train["sn"].values() for v in m.values())
    assert n_sn == 21058
    assert train["mgridref_y"].shape == (33,) and notrain["gridref_x"].shape == (34,)
    assert notrain["target_x"].shape == (256,)
    assert flat.numel() == unflatten.size == 21058 + 33 + 2 + 4 + 1 + 34 + 256
    # LinearZero / timestep_phase start at zero (nn_dds.py:101-103,189-190)
    assert train["sn"]["drift_net/~/linear_zero"]["w"].abs().sum() == 0
    assert torch.allclose(notrain["vd"]["logdiag"], torch.full((2,), float(np.log(60.0))))


def test_flat_layout_follows_ravel_pytree_order():
    flat, unflatten, _ = mcdbm.initialize(dim=2, nbridges=8, eps=0.01, trainable=("eps", "vd"),
                                          mode="MCD_CAIS_sn", emb_dim=20, nn_arch="geffner", device="cpu")
    o = unflatten.offset
    # params_train first, keys sorted: eps < sn < vd; inside sn: emb < factor_sn < nn (W1,b1,W2,b2,W3,b3)
    assert o("eps") == 0 and o("sn", "emb") == 1
    assert o("sn", "factor_sn") == 1 + 8 * 20
    assert o("sn", "nn", 0, 0) == o("sn", "factor_sn") + 1
    assert o("sn", "nn", 0, 1) == o("sn", "nn", 0, 0) + 22 * 22
    assert o("sn", "nn", 2, 1) == o("sn", "nn", 2, 0) + 22 * 2
    assert o("vd", "logdiag") < o("vd", "mean") < o("eta")          # notrain after train
    train, _ = unflatten(flat)
    train["eps"].fill_(0.5)                                          # views alias params_flat
    assert flat[0] == 0.5
    assert float(train["sn"]["factor_sn"]) == 0.0                    # nn.py:63


def test_ngrid_clamps_to_nbridges():
    _, unflatten, _ = mcdbm.initialize(dim=2, nbridges=8, mode="MCD_CAIS_sn", nn_arch="geffner", emb_dim=4,
                                       device="cpu")
    assert unflatten.shape("mgridref_y") == (9,)                     # min(32, K) + 1


def test_load_model_routing():
    t, d, _ = model_handler.load_model("many_gmm")
    assert (t.name, d, t.n_mixes) == ("many_gmm", 2, 40)
    c = t.consts_on("cpu").numpy()
    assert c.shape == (81,) and abs(c[0] - np.log1p(np.exp(0.1))) < 1e-7
    np.testing.assert_array_equal(c[1:3], np.float32([-15.758228, 18.116531]))
    assert model_handler.load_model("gmm")[0].name == "gmm"
    assert model_handler.load_model("funnel")[1] == 10
    with pytest.raises(NotImplementedError):
        model_handler.load_model("lorenz")
    with pytest.raises(NotImplementedError):
        t(torch.zeros(2))


def test_no_cpu_fallback_and_plugin_errors():
    b = synthetic.build("gmm_n300_k8", device="cpu")
    seeds = torch.arange(1, 9, dtype=torch.int32)
    with pytest.raises(RuntimeError, match="ROCm device only"):
        mcdbm.compute_bound(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], b["target"])
    dim, K, _, spec = b["params_fixed"]
    with pytest.raises(NotImplementedError, match="Mode not implemented."):
        mcdbm.compute_bound(seeds, b["params_flat"], b["unflatten"], (dim, K, "MCD_U_a-lp", spec), b["target"])
    with pytest.raises(TypeError):
        mcdbm.compute_bound(seeds, b["params_flat"], b["unflatten"], b["params_fixed"], lambda z: z.sum())
    with pytest.raises(NotImplementedError):
        mcdbm.initialize(dim=2, nbridges=8, mode="MCD_U_a-lp-sn", device="cpu")     # the other momentum modes stay out
    # 2nd-order CMCD is a plugin value: its network is built with rho_dim = dim (mcdboundingmachine.py:82-98)
    _, un, fixed = mcdbm.initialize(dim=2, nbridges=8, mode="MCD_CAIS_UHA_sn", nn_arch="geffner", emb_dim=20, device="cpu")
    assert fixed[3].rho_dim == 2 and un.shape("sn", "nn", 0, 0) == (24, 24)


def test_synthetic_configs_resolve():
    for name, cfg in synthetic.CONFIGS.items():
        if cfg["model"] == "lgcp":
            continue
        b = synthetic.build(name, device="cpu")
        p = synthetic.oracle_params(b["unflatten"], b["params_flat"])
        assert p["sn"] and b["params_fixed"][1] == cfg["nbridges"]


def test_params_pickle_round_trip(tmp_path):
    """utils.save_params / load_params: the merged `params` dict the reference pickles (main.py:283-296)."""
    import pickle
    from cmcd_amd import utils
    for arch in ("dds", "geffner"):
        flat, unflatten, _ = mcdbm.initialize(dim=2, nbridges=6, eps=0.3, trainable=("eps", "vd"), mode="MCD_CAIS_sn",
                                              nn_arch=arch, emb_dim=12, device="cpu")
        flat = flat + torch.arange(flat.numel(), dtype=torch.float32) * 1e-3      # make every leaf distinct
        path = tmp_path / f"params_{arch}.pkl"
        utils.save_params(path, flat, unflatten)
        d = pickle.load(open(path, "rb"))
        assert {"vd", "eps", "sn", "mgridref_y", "gridref_x", "target_x"} <= set(d)
        assert d["vd"]["mean"].shape == (2,) and isinstance(d["eps"], np.ndarray)
        back = utils.load_params(path, torch.zeros_like(flat), unflatten)
        assert torch.equal(back, flat)
        d["vd"]["mean"] = np.zeros(3)
        with pytest.raises(ValueError):
            utils.load_params(d, flat, unflatten)


def test_main_flag_surface_follows_the_reference():
    """--config.x value / --config.x=value / --config.flag / --noconfig.flag (configs/base.py + absl)."""
    from cmcd_amd import main as drv
    c = drv.parse_flags(["--config.model", "many_gmm", "--config.boundmode=MCD_CAIS_var_sn", "--config.N", "300",
                         "--config.nbridges=128", "--noconfig.pretrain_mfvi", "--config.init_sigma", "10",
                         "--config.grad_clipping", "--config.init_eps", "0.65", "--config.emb_dim", "40",
                         "--noconfig.train_eps", "--noconfig.train_vi"], drv.get_config())
    assert (c.model, c.boundmode, c.N, c.nbridges, c.pretrain_mfvi, c.init_sigma, c.grad_clipping, c.init_eps, c.emb_dim,
            c.train_eps, c.train_vi, c.train_betas) == ("many_gmm", "MCD_CAIS_var_sn", 300, 128, False, 10.0, True, 0.65,
                                                        40, False, False, True)
    d = drv.get_config()
    assert (d.boundmode, d.N, d.nbridges, d.lr, d.mfvi_lr, d.iters, d.n_samples, d.n_input_dist_seeds, d.nn_arch) == (
        "UHA", 5, 8, 1e-4, 0.01, 150000, 500, 30, "geffner")
    with pytest.raises(SystemExit):
        drv.parse_flags(["--config.nonsense", "1"], drv.get_config())
    with pytest.raises(SystemExit):
        drv.parse_flags(["--noconfig.N"], drv.get_config())


def test_main_applies_the_tuned_lr_and_eps_tables():
    from cmcd_amd import main as drv
    c = drv.setup_config(drv.parse_flags(["--config.model", "funnel", "--config.nbridges", "32"], drv.get_config()))
    assert (c.init_eps, c.lr) == (0.1, 0.005)
    c = drv.setup_config(drv.parse_flags(["--config.model", "lgcp", "--config.boundmode", "MCD_CAIS_sn"], drv.get_config()))
    assert c.lr == 1e-4
    c = drv.setup_config(drv.parse_flags(["--config.model", "many_gmm", "--config.lr", "0.01"], drv.get_config()))
    assert c.lr == 0.01
    c = drv.setup_config(drv.parse_flags(["--config.model", "funnel", "--config.nbridges", "7"], drv.get_config()))
    assert c.lr == 1e-4                                                     # KeyError branch: flags stand


def test_target_samplers_and_the_sinkhorn_metric():
    """load_model's third return value (exact samplers of the tractable targets) and utils.W2_distance: a cloud is
    closer to a second draw of its own law than to a shifted copy; identical clouds are at (almost) zero."""
    import types
    from cmcd_amd import utils
    from cmcd_amd.model_handler import load_model
    for name, dim in (("gmm", 2), ("funnel", 10), ("many_gmm", 2)):
        _, d, sampler = load_model(name, types.SimpleNamespace())
        x = sampler(1, 400) if name != "many_gmm" else sampler(1, (400,))
        assert x.shape == (400, dim) and np.isfinite(x).all()
    _, _, sampler = load_model("gmm", types.SimpleNamespace())
    a, b = sampler(1, 300), sampler(2, 300)
    near = utils.W2_distance(a, b)
    far = utils.W2_distance(a, b + np.array([4.0, 0.0], np.float32))
    same = utils.W2_distance(a, a)
    assert 0.0 <= same < near < far
    out = utils.calculate_W2_distances(torch.from_numpy(np.concatenate([a, b])), torch.from_numpy(np.concatenate([b, a])),
                                       torch.from_numpy(np.concatenate([a, b])[::-1].copy()), 300, 2, 200)
    assert set(out) == {"w2_dist", "w2_dist_std", "self_w2_dist", "self_w2_dist_std"}


def test_cli_takes_the_reference_readme_command_lines_verbatim():
    """Every flag of the reference README's replicate commands parses (alpha, wandb.name, the single-dash
    `-config.init_sigma`), nested fields land in the nested namespace, fixed fields are refused when changed."""
    from cmcd_amd import main as cli
    argv = ["--config.boundmode", "MCD_CAIS_sn", "--config.model", "funnel", "--config.N", "300", "--config.alpha", "0.05",
            "--config.emb_dim", "48", "--config.init_eps", "0.1", "-config.init_sigma", "1", "--config.iters", "11000",
            "--noconfig.pretrain_mfvi", "--config.train_vi", "--noconfig.train_eps", "--config.wandb.name",
            "funnel replicate w/ cos_sq", "--config.lr", "0.01", "--config.n_samples", "2000", "--config.eps_schedule", "cos_sq"]
    c = cli.parse_flags(argv, cli.get_config())
    assert (c.model, c.N, c.emb_dim, c.init_sigma, c.iters, c.pretrain_mfvi, c.train_eps) == ("funnel", 300, 48, 1.0, 11000, False, False)
    assert c.wandb.name == "funnel replicate w/ cos_sq" and c.alpha == 0.05 and c.n_samples == 2000
    cli.check_fixed_fields(c)
    c = cli.parse_flags(["--noconfig.wandb.log", "--config.use_whitened"], cli.get_config())
    assert c.wandb.log is False
    with pytest.raises(NotImplementedError):
        cli.check_fixed_fields(c)
    c = cli.parse_flags(["--config.nn_arch", "dds", "--config.fully_connected_units", "[128, 128]"], cli.get_config())
    with pytest.raises(NotImplementedError):
        cli.check_fixed_fields(c)
    for bad in (["--config.wandb", "x"], ["--config.wandb.nope", "x"], ["--config.nope", "1"]):
        with pytest.raises(SystemExit):
            cli.parse_flags(bad, cli.get_config())


def test_unflatten_with_a_zero_size_leaf_in_the_middle():
    """nbridges = 0 makes target_x an empty leaf (linspace(0, 1, 2)[1:-1]); with the default trainable it is not the
    last one.  jax's ravel_pytree handles that; so must this one."""
    from cmcd_amd import mcdboundingmachine as mcdbm
    flat, unflatten, fixed = mcdbm.initialize(dim=2, nbridges=0, mode="MCD_ULA", device="cpu")
    train, notrain = unflatten(flat)
    assert notrain["target_x"].shape == (0,)
    assert sum(int(np.prod(s)) if s else 1 for _, s in unflatten.layout.values()) == flat.numel()
    assert float(train["eps"]) == pytest.approx(0.01) and notrain["vd"]["mean"].shape == (2,)


def test_bench_self_launch_builds_the_launcher_command(monkeypatch):
    """`python3 bench.py --gpus N` (N > 1, no launcher in the environment) starts torch.distributed.run as a CHILD process
    with the same argv, on 127.0.0.1 and a free port, and refuses clearly when the node shows fewer than N devices —
    all before anything of the parent touches the GPU (no GPU here: the child is intercepted)."""
    import importlib.util
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "2"])
    monkeypatch.delenv("CMCD_BENCH_SHARED_GPU", raising=False)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 1)
    assert bench.self_launch(4) == 2 and "cmd" not in seen                   # 1 device visible: refused, nothing started
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    assert bench.self_launch(4) == 7                                         # the child's return code comes back
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-7:] == [os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "2"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # the plain command on this GPU-less box: exit code 2 and the reason, not an assertion error from a rank
    monkeypatch.undo()
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "CMCD_BENCH_SHARED_GPU")})
    if torch.cuda.device_count() < 2:
        assert out.returncode == 2 and "GPU(s) visible" in out.stderr
