"""Generates tests/golden/*: run in the BUILD container (the only place /root/reference exists).

  * lgcp_bin_counts.npy — the 40x40 histogram of the Finnish-pines point set, derived from the
    public data file /root/reference/pines.csv with the binning rule of
    /root/reference/src/cp_utils.py:16-42 (data, not source; the CSV itself is not copied).
  * oracle_<case>.npz   — outputs of THIS repo's float64 oracle on the synthetic inputs of
    cmcd_amd.synthetic (seeds 1..n).  The reference cannot be imported here (no jax), so these
    are regression pins of the restatement, not reference outputs: "parity unpinned".
  * prng_kat.npz        — the public jax.random known answers + derived streams.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from cmcd_amd import synthetic  # noqa: E402
from oracle import prng, targets  # noqa: E402
from helpers import run_oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = {
    "gmm_k8": ("gmm_n300_k8", 48, {}),
    "funnel_k64": ("funnel_n300_k64", 32, {}),
    "many_gmm_dds_k256": ("many_gmm_n2000_k256_dds", 64, {}),
    "many_gmm_var_k32": ("many_gmm_var_n16000_k256", 32, dict(nbridges=32)),
    # every parameter leaf non-trivial (synthetic.build(dense=True)): biases, timestep_phase, q mean / per-dimension
    # logdiag, non-uniform mgridref_y, factor_sn
    "dense_gmm_k8": ("gmm_n300_k8", 48, dict(dense=True)),
    "dense_funnel_k64": ("funnel_n300_k64", 32, dict(dense=True)),
    "dense_many_gmm_dds_k256": ("many_gmm_n2000_k256_dds", 64, dict(dense=True)),
    "dense_many_gmm_var_k32": ("many_gmm_var_n16000_k256", 32, dict(nbridges=32, dense=True)),
}


def main():
    os.makedirs(GOLD, exist_ok=True)
    pines = "/root/reference/pines.csv"
    if os.path.exists(pines):
        pts = np.genfromtxt(pines, delimiter=",")
        counts = targets.lgcp_bin_counts(pts, 40).astype(np.int16)
        np.save(os.path.join(GOLD, "lgcp_bin_counts.npy"), counts)
        print("lgcp counts:", counts.sum(), "points in", (counts > 0).sum(), "bins")
    for tag, (name, n, over) in CASES.items():
        b = synthetic.build(name, device="cpu", **over)
        seeds = synthetic.parity_seeds(n)
        loss, z = run_oracle(b, seeds, dtype=np.float64)
        np.savez_compressed(os.path.join(GOLD, f"oracle_{tag}.npz"), seeds=seeds, loss=loss, z=z,
                            config=name, overrides=repr(over))
        print(tag, "mean loss", loss[np.isfinite(loss)].mean(), "n_inf", np.isinf(loss).sum())
    k0 = prng.prng_key(np.array(0))
    a, b = prng.split(k0)
    eps0, eps = prng.particle_noise(np.arange(1, 5), 3, 4)
    np.savez(os.path.join(GOLD, "prng_kat.npz"),
             split0=np.stack([a, b]), normal0=prng.normal(k0, 1), normal_sub=prng.normal(b, 1),
             normal42=prng.normal(prng.prng_key(np.array(42)), 1),
             many_gmm_means=targets.many_gmm_means(), chain_eps0=eps0, chain_eps=eps)


if __name__ == "__main__":
    main()
