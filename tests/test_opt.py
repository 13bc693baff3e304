"""Host-side optimiser pieces (reference opt.py:14-35): clip + Adam arithmetic and the projection."""
import numpy as np
import torch

from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import opt


def test_clip_adam_matches_hand_computation():
    o = opt.create_optimizer(0.1)
    p = torch.zeros(3)
    st = o.init(p)
    g = torch.tensor([1.0, -10.0, 0.0])
    upd, st = o.update(g, st, p)
    gc = np.array([1.0, -5.0, 0.0])                       # optax.clip(5.0)
    mu, nu = 0.1 * gc, 0.001 * gc ** 2
    want = -0.1 * (mu / 0.1) / (np.sqrt(nu / 0.001) + 1e-8)
    np.testing.assert_allclose(upd.numpy(), want, rtol=1e-6)
    upd2, st = o.update(g, st, p)
    assert st["count"] == 2 and np.all(np.sign(upd2.numpy()) == np.sign(want))


def test_project_clamps_like_the_reference():
    flat, unflatten, _ = mcdbm.initialize(dim=2, nbridges=8, eps=0.7, eta=2.0, gamma=-1.0,
                                          trainable=("eps", "eta", "gamma", "mgridref_y"), mode="MCD_CAIS_sn",
                                          nn_arch="geffner", emb_dim=4, device="cpu")
    train, _ = unflatten(flat)
    train["mgridref_y"][0] = -3.0
    opt.project(flat, unflatten, ("eps", "eta", "gamma", "mgridref_y"))
    train, _ = unflatten(flat)
    assert float(train["eps"]) == 0.5 and abs(float(train["eta"]) - 0.99) < 1e-7 and abs(float(train["gamma"]) - 0.001) < 1e-9
    assert abs(float(train["mgridref_y"][0]) - 0.001) < 1e-9 and float(train["mgridref_y"][1]) == 1.0
