"""Shared test plumbing: run the oracle on the same synthetic inputs as the HIP path."""
import numpy as np

from cmcd_amd import synthetic
from oracle import cmcd_oracle as orc
from oracle import targets as otg


def oracle_target(cfg, lgcp_counts=None):
    m = cfg["model"]
    if m == "gmm":
        return otg.Gmm()
    if m == "funnel":
        return otg.Funnel(10)
    if m == "many_gmm":
        return otg.ManyGmm(n_mixes=int(cfg.get("n_mixes", 40)), loc_scaling=float(cfg.get("loc_scaling", 40.0)))
    if m == "lgcp":
        return otg.Lgcp(lgcp_counts)
    raise KeyError(m)


def run_oracle(built, seeds, dtype=np.float64, reuse=True, lgcp_counts=None):
    cfg = built["cfg"]
    dim, K, mode, spec = built["params_fixed"]
    p = synthetic.oracle_params(built["unflatten"], built["params_flat"])
    return orc.compute_log_elbo_batch(
        np.asarray(seeds), p, dim, K, mode, spec.arch, oracle_target(cfg, lgcp_counts),
        eps_schedule=cfg["eps_schedule"], grad_clipping=cfg["grad_clipping"], dtype=dtype, reuse=reuse)


def compare_losses(l_hip, l_ref, z_hip, z_ref, tag="", *, K, rel_max=None, z_max=None):
    """Parity bar of SURVEY.md section 8c / BASELINE.md section 2 (float32 path vs float64 oracle) on all three outputs of
    compute_bound, `(mean, (losses, z))` (/root/reference/src/mcdboundingmachine.py:183-205):
      * identical set of +inf particles; no NaN;
      * batch mean and ln Z within 1e-3 (absolute, relative above 1);
      * per-particle loss: p99 of the relative error <= 5e-3, and the WORST particle <= 1e-3 for chains of K <= 32 bridges,
        <= 0.2 for longer ones (float32 round-off is amplified along a chaotic chain: 0.13 seen at K = 256);
      * z_K (finite particles): p99 of |z - z_ref| <= 1e-3 max(1, p99 |z_ref|), and for K <= 32 every element within 1e-3
        (scaled the same way).
    `K` = bridges of the chain (0 for the mean-field bound).  `rel_max` / `z_max` override the worst-particle bounds for a
    case that needs a looser one: every such case is listed in DESIGN.md section 2 with its measured value."""
    l_hip = np.asarray(l_hip, np.float64)
    l_ref = np.asarray(l_ref, np.float64)
    z_hip = np.asarray(z_hip, np.float64).reshape(len(l_ref), -1)
    z_ref = np.asarray(z_ref, np.float64).reshape(len(l_ref), -1)
    assert not np.isnan(l_hip).any(), f"{tag}: NaN loss"
    inf_h, inf_r = np.isinf(l_hip), np.isinf(l_ref)
    assert np.array_equal(inf_h, inf_r), f"{tag}: +inf particle sets differ: {np.flatnonzero(inf_h != inf_r)}"
    f = ~inf_r
    rel = np.abs(l_hip[f] - l_ref[f]) / np.maximum(1.0, np.abs(l_ref[f]))
    mean_err = abs(l_hip[f].mean() - l_ref[f].mean())
    lnz_err = abs(orc.ln_z(l_hip) - orc.ln_z(l_ref))
    assert not np.isnan(z_hip[f]).any(), f"{tag}: NaN in z of a finite particle"
    zerr = np.abs(z_hip - z_ref)[f]
    z_scale = max(1.0, float(np.quantile(np.abs(z_ref[f]), 0.99)))
    short = K <= 32
    rel_bound = rel_max if rel_max is not None else (1e-3 if short else 0.2)
    report = dict(n=len(l_ref), n_inf=int(inf_r.sum()), K=K, mean_err=mean_err, lnz_err=lnz_err,
                  rel_p50=float(np.median(rel)), rel_p99=float(np.quantile(rel, 0.99)), rel_max=float(rel.max()),
                  z_p99=float(np.quantile(zerr, 0.99)), z_max=float(zerr.max()), z_scale=z_scale, rel_bound=rel_bound)
    assert mean_err <= 1e-3 * max(1.0, abs(l_ref[f].mean())), f"{tag}: {report}"
    assert lnz_err <= 1e-3 * max(1.0, abs(orc.ln_z(l_ref))), f"{tag}: {report}"
    assert report["rel_p99"] <= 5e-3, f"{tag}: {report}"
    assert report["rel_max"] <= rel_bound, f"{tag}: worst particle: {report}"
    assert report["z_p99"] <= 1e-3 * z_scale, f"{tag}: z: {report}"
    if short or z_max is not None:
        assert report["z_max"] <= (z_max if z_max is not None else 1e-3 * z_scale), f"{tag}: worst z element: {report}"
    return report


def run_c_oracle(built, seeds):
    """The plain-C restatement (oracle/cmcd_oracle.c) on the same inputs; float32, reference-faithful."""
    from cmcd_amd import _lib as hip_abi
    from cmcd_amd import mcdboundingmachine as mcdbm
    from oracle import c_oracle
    cfg = built["cfg"]
    dim, K, mode, spec = built["params_fixed"]
    un = built["unflatten"]
    desc = hip_abi.Desc(dim=dim, nbridges=K, mode=hip_abi.MODE[mode], arch=hip_abi.ARCH[spec.arch],
                        emb_dim=spec.emb_dim, target=built["target"].target_id,
                        eps_schedule=hip_abi.EPS_SCHEDULE.get(cfg["eps_schedule"], 0),
                        grad_clipping=int(bool(cfg["grad_clipping"])), ngrid=un.shape("mgridref_y")[0] - 1,
                        reserved=0)
    lay = mcdbm._layout(un, spec)
    consts = built["target"].consts_on("cpu")
    return c_oracle.bound(desc, lay, np.asarray(seeds), built["params_flat"].detach().cpu().numpy(),
                          None if consts is None else consts.numpy())


def lgcp_counts_fixture():
    """The 40x40 bin counts of the Finnish pines point set (tests/golden/lgcp_bin_counts.npy: data)."""
    import os
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "lgcp_bin_counts.npy"))
