"""Deterministic synthetic inputs for the five BASELINE.json configurations.

Used by bench.py, __graft_entry__.smoke() and the tests so that the HIP path and the oracle see
byte-identical parameters.  Nothing here reads /root/reference.
"""
import math
import types

import numpy as np
import torch

from . import mcdboundingmachine as mcdbm
from . import variationaldist as vd
from .model_handler import load_model

# Resolved flag values of the BASELINE.json configs.
CONFIGS = {
    "gmm_n300_k8": dict(model="gmm", boundmode="MCD_CAIS_sn", N=300, nbridges=8, nn_arch="geffner",
                        emb_dim=20, init_eps=0.01, eps_schedule="", grad_clipping=False, init_sigma=1.0),
    "funnel_n300_k64": dict(model="funnel", boundmode="MCD_CAIS_sn", N=300, nbridges=64, nn_arch="geffner",
                            emb_dim=48, init_eps=0.1, eps_schedule="cos_sq", grad_clipping=False,
                            init_sigma=1.0),
    "many_gmm_n2000_k256_dds": dict(model="many_gmm", boundmode="MCD_CAIS_sn", N=2000, nbridges=256,
                                    nn_arch="dds", emb_dim=20, init_eps=1.0, eps_schedule="cos_sq",
                                    grad_clipping=True, init_sigma=60.0),
    "many_gmm_var_n16000_k256": dict(model="many_gmm", boundmode="MCD_CAIS_var_sn", N=16000, nbridges=256,
                                     nn_arch="geffner", emb_dim=130, init_eps=0.65, eps_schedule="",
                                     grad_clipping=True, init_sigma=15.0),
    "lgcp_n20_k128": dict(model="lgcp", boundmode="MCD_CAIS_sn", N=20, nbridges=128, nn_arch="geffner",
                          emb_dim=20, init_eps=1e-5, eps_schedule="", grad_clipping=False, init_sigma=0.5),
}
NORTH_STAR = "many_gmm_n2000_k256_dds"


def parity_seeds(n):
    return np.arange(1, n + 1, dtype=np.int32)


def throughput_seeds(n, stream=0):
    return np.random.default_rng(stream).integers(1, 10 ** 6, n).astype(np.int32)


# Parameter set used when build() is called without `dense=`: False = the measurement inputs of SURVEY.md section 8d
# (biases, timestep_phase and the q mean zero, one sigma, uniform mgridref_y); True = every leaf non-trivial.  The parity
# tests switch it through the `param_set` fixture of tests/conftest.py.
DENSE_DEFAULT = False


def _fill_synthetic_sn(sn, arch, rng, dense=False):
    """W ~ N(0, 1/fan_in), b = 0, last layer x0.1, factor_sn = 0.1, emb ~ 0.05 N(0,1): a network
    that actually moves the particles (the reference's zero-init last layer / factor_sn = 0 would
    hide every MLP bug).  `dense`: additionally every bias ~ N(0, 0.1) (the output layer's too),
    timestep_phase ~ U(0, 2 pi) and factor_sn = 0.17, so that a leaf the kernels dropped or folded
    wrongly changes the result (/root/reference/src/nn_dds.py:111-127,155-164, src/nn.py:45-70)."""
    def mat(fan_in, fan_out, scale=1.0):
        return torch.from_numpy((rng.standard_normal((fan_in, fan_out)) * scale / math.sqrt(fan_in)).astype(np.float32))

    def bias(b):
        if dense:
            b.copy_(torch.from_numpy((0.1 * rng.standard_normal(tuple(b.shape))).astype(np.float32)))
        else:
            b.zero_()

    if arch == "geffner":
        for li, (w, b) in enumerate(sn["nn"]):
            w.copy_(mat(w.shape[0], w.shape[1], 0.1 if li == 2 else 1.0))
            bias(b)
        sn["emb"].copy_(torch.from_numpy((0.05 * rng.standard_normal(tuple(sn["emb"].shape))).astype(np.float32)))
        sn["factor_sn"].fill_(0.17 if dense else 0.1)
    else:
        ph = sn["drift_net"]["timestep_phase"]
        if dense:
            ph.copy_(torch.from_numpy(rng.uniform(0.0, 2.0 * math.pi, tuple(ph.shape)).astype(np.float32)))
        else:
            ph.zero_()
        for name in ("linear", "linear_1", "linear_2", "linear_3", "linear_zero"):
            mod = sn["drift_net/~/" + name]
            mod["w"].copy_(mat(mod["w"].shape[0], mod["w"].shape[1], 0.1 if name == "linear_zero" else 1.0))
            bias(mod["b"])


def build(config_name=None, device=None, lgcp_counts=None, dense=None, **overrides):
    """-> dict(cfg, params_flat, unflatten, params_fixed, target, eps_schedule, grad_clipping).

    dense=True: the all-leaves-non-trivial parameter set — biases, timestep_phase and factor_sn as in
    `_fill_synthetic_sn`, q with a non-zero mean and a per-dimension logdiag (/root/reference/src/vardist/
    diag_gauss.py:26-62) and random positive mgridref_y, i.e. a non-uniform beta grid (/root/reference/src/
    mcdboundingmachine.py:146-149)."""
    if dense is None:
        dense = DENSE_DEFAULT
    cfg = dict(CONFIGS[config_name or NORTH_STAR])
    cfg.update(overrides)
    info = types.SimpleNamespace(**cfg)
    if cfg["model"] == "lgcp":
        from .lgcp import load_model_lgcp
        target, dim = load_model_lgcp("lgcp", info, flat_bin_counts=lgcp_counts)
    else:
        target, dim, _ = load_model(cfg["model"], info)
    vdparams = vd.initialize(dim, init_sigma=cfg["init_sigma"])
    if cfg["model"] == "lgcp":
        # stand-in for the MFVI-pretrained q: mean = mu_0
        vdparams["mean"] += math.log(126.0) - 0.955
    mgridref_y = None
    if dense:
        rq = np.random.default_rng(7)
        spread = 0.05 if cfg["model"] == "lgcp" else 0.2 * cfg["init_sigma"]
        vdparams["mean"] += torch.from_numpy((spread * rq.standard_normal(dim)).astype(np.float32))
        vdparams["logdiag"] += torch.from_numpy((0.15 * rq.standard_normal(dim)).astype(np.float32))
        if cfg["nbridges"] >= 1:
            mgridref_y = rq.uniform(0.5, 1.5, min(32, cfg["nbridges"]) + 1).astype(np.float32)
    flat, unflatten, fixed = mcdbm.initialize(
        dim=dim, nbridges=cfg["nbridges"], vdparams=vdparams, eta=0.0, eps=cfg["init_eps"], mgridref_y=mgridref_y,
        gamma=cfg.get("init_gamma", 10.0),
        trainable=("eta", "gamma", "eps", "vd", "mgridref_y"), mode=cfg["boundmode"],
        emb_dim=cfg["emb_dim"], nlayers=3, nn_arch=cfg["nn_arch"], device="cpu")
    train, _ = unflatten(flat)
    if "sn" in train:   # MCD_ULA keeps no network
        _fill_synthetic_sn(train["sn"], cfg["nn_arch"], np.random.default_rng(1), dense=dense)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    flat = flat.to(device)
    return dict(cfg=cfg, params_flat=flat, unflatten=unflatten, params_fixed=fixed, target=target,
                eps_schedule=cfg["eps_schedule"], grad_clipping=cfg["grad_clipping"])


def oracle_params(unflatten, params_flat):
    """The parameter dict layout oracle/cmcd_oracle.py documents, as float64 NumPy (tests only
    *pass* this to the oracle; the product never imports it)."""
    train, notrain = unflatten(params_flat.detach().cpu())
    allp = {**train, **notrain}
    f = lambda t: np.asarray(t.numpy(), np.float64)
    out = {"vd": {k: f(v) for k, v in allp["vd"].items()}, "eps": f(allp["eps"]), "gamma": f(allp["gamma"]),
           "mgridref_y": f(allp["mgridref_y"]), "gridref_x": f(allp["gridref_x"]), "target_x": f(allp["target_x"])}
    sn = allp["sn"]
    if "nn" in sn:
        (w1, b1), (w2, b2), (w3, b3) = sn["nn"]
        out["sn"] = {"emb": f(sn["emb"]), "factor_sn": f(sn["factor_sn"]), "W1": f(w1), "b1": f(b1),
                     "W2": f(w2), "b2": f(b2), "W3": f(w3), "b3": f(b3)}
    else:
        m = lambda n: sn["drift_net/~/" + n]
        out["sn"] = {"timestep_phase": f(sn["drift_net"]["timestep_phase"]),
                     "t_w1": f(m("linear")["w"]), "t_b1": f(m("linear")["b"]),
                     "t_w2": f(m("linear_1")["w"]), "t_b2": f(m("linear_1")["b"]),
                     "s_w1": f(m("linear_2")["w"]), "s_b1": f(m("linear_2")["b"]),
                     "s_w2": f(m("linear_3")["w"]), "s_b2": f(m("linear_3")["b"]),
                     "s_w3": f(m("linear_zero")["w"]), "s_b3": f(m("linear_zero")["b"])}
    return out
