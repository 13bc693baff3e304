"""The integer PRNG path of the HIP kernels, bit for bit.

`cmcd_debug_capture_noise` makes the next `cmcd_bound_forward` write, from inside the trajectory kernel that runs, the
random words it turns into deviates, the chain key entering every bridge and the deviates themselves.  They must equal
oracle/prng.py — which tests/test_oracle_prng.py pins to jax's published known answers — exactly (uint32) and, for the
deviates, to 4 ulp (the kernels evaluate XLA's -log1p(-fl(u u)) of Giles' erfinv through one v_log_f32; written as
(1 - u)(1 + u) — more accurate, but not what the reference computes — the tails were 90 ulp away).  Covers the three kernel forms
(wave per tile, cooperative on 16- and on 8-particle tiles: three separate implementations of the key chain), d = 2
(one Threefry block per normal draw) and d = 10 (five blocks dealt to the rows of a wave, odd pad counter unused).
Reference sites: /root/reference/src/mcdboundingmachine.py:151-162, /root/reference/src/mcd_cais.py:66,87,94,
/root/reference/src/mcd_utils.py:14-16."""
import ctypes as C

import numpy as np
import pytest
import torch

from cmcd_amd import _lib
from cmcd_amd import mcdboundingmachine as mcdbm
from cmcd_amd import synthetic
from oracle import prng

pytestmark = pytest.mark.gpu


def oracle_chain(seeds, dim, K):
    """-> bits uint32 [K+1, N, dim], gen keys uint32 [K+1, N, 2], deviates float32 [K+1, N, dim] (stage 0 = z_0)."""
    k0 = prng.prng_key(seeds)
    a, b = prng.split(k0)
    bits, keys, dev = [prng.random_bits(a, dim)], [], [prng.normal(a, dim)]
    c, _ = prng.split(b)
    _, gen = prng.split(c)
    keys.append(gen)
    for _ in range(K):
        g, h = prng.split(gen)
        bits.append(prng.random_bits(g, dim))
        dev.append(prng.normal(g, dim))
        _, gen = prng.split(h)
        keys.append(gen)
    return np.stack(bits), np.stack(keys), np.stack(dev)


def ulp_distance(a, b):
    ia, ib = a.view(np.int32).astype(np.int64), b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2 ** 31) - ia, ia)     # order-preserving map of float32 bit patterns
    ib = np.where(ib < 0, np.int64(-2 ** 31) - ib, ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("variant", [1, 3, 4], ids=["wave_per_tile", "cooperative_16", "cooperative_8"])
@pytest.mark.parametrize("name,n,K", [("many_gmm_n2000_k256_dds", 203, 40), ("funnel_n300_k64", 77, 9),
                                      ("gmm_n300_k8", 33, 8), ("many_gmm_n2000_k256_dds", 2000, 256),
                                      ("many_gmm_var_n16000_k256", 100, 12)])   # 132-wide net: the 12-wave instance (variant 4)
def test_key_chain_and_deviates_are_bit_exact(hip_lib, monkeypatch, variant, name, n, K):
    monkeypatch.setattr(mcdbm, "KERNEL_VARIANT", variant)
    b = synthetic.build(name, device="cuda", nbridges=K, dense=True)
    dim = b["params_fixed"][0]
    # seeds over the whole range opt.py:94 draws from, including both ends
    seeds = np.random.default_rng(3).integers(1, 10 ** 6, n).astype(np.int32)
    seeds[:2] = (1, 999999)
    bits = torch.zeros(K + 1, n, dim, dtype=torch.int32, device="cuda")
    keys = torch.zeros(K + 1, n, 2, dtype=torch.int32, device="cuda")
    noise = torch.zeros(K + 1, n, dim, dtype=torch.float32, device="cuda")
    _lib.check(hip_lib.cmcd_debug_capture_noise(bits.data_ptr(), keys.data_ptr(), noise.data_ptr()))
    losses, z, _ = mcdbm.bound_forward(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"],
                                       b["target"], eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    rb, rk, rd = oracle_chain(seeds, dim, K)
    got_bits = bits.cpu().numpy().view(np.uint32)
    got_keys = keys.cpu().numpy().view(np.uint32)
    assert np.array_equal(got_keys, rk), "Threefry split chain differs from jax's"
    assert np.array_equal(got_bits, rb), "random_bits of the normal draws differ from jax's"
    d = ulp_distance(noise.cpu().numpy(), rd)
    assert d.max() <= 4, f"deviates differ by up to {d.max()} ulp"
    print(name, variant, "ulp distance histogram", np.bincount(d.ravel().astype(np.int64), minlength=3)[:3])
    # the capture is one-shot: a second call must leave the buffers alone
    bits.zero_()
    mcdbm.bound_forward(torch.from_numpy(seeds).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"], b["target"],
                        eps_schedule=b["eps_schedule"], grad_clipping=b["grad_clipping"])
    torch.cuda.synchronize()
    assert int(bits.abs().max()) == 0


def test_capture_is_refused_on_the_lgcp_launch_sequence(hip_lib):
    from helpers import lgcp_counts_fixture
    b = synthetic.build("lgcp_n20_k128", device="cuda", lgcp_counts=lgcp_counts_fixture(), nbridges=2)
    buf = torch.zeros(3 * 4 * 1600, dtype=torch.int32, device="cuda")
    fbuf = torch.zeros(3 * 4 * 1600, dtype=torch.float32, device="cuda")
    _lib.check(hip_lib.cmcd_debug_capture_noise(buf.data_ptr(), None, fbuf.data_ptr()))
    with pytest.raises(NotImplementedError):
        mcdbm.bound_forward(torch.arange(1, 5, dtype=torch.int32).cuda(), b["params_flat"], b["unflatten"],
                            b["params_fixed"], b["target"])
    # ... and is disarmed by the refused call
    mcdbm.bound_forward(torch.arange(1, 5, dtype=torch.int32).cuda(), b["params_flat"], b["unflatten"], b["params_fixed"],
                        b["target"])
    torch.cuda.synchronize()
    assert int(buf.abs().max()) == 0
