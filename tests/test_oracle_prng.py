"""Pins the oracle's PRNG (and the product's host copy) to the public jax.random known answers."""
import os

import numpy as np

from cmcd_amd import prng as host_prng
from oracle import prng

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_split_known_answer():
    a, b = prng.split(prng.prng_key(np.array(0)))
    assert a.tolist() == [4146024105, 967050713]
    assert b.tolist() == [2718843009, 1272950319]


def test_normal_known_answers():
    k0 = prng.prng_key(np.array(0))
    _, sub = prng.split(k0)
    assert prng.normal(k0, 1)[0] == np.float32(-0.20584226)
    assert prng.normal(sub, 1)[0] == np.float32(-1.2515389)
    assert prng.normal(prng.prng_key(np.array(42)), 1)[0] == np.float32(-0.18471177)


def test_many_gmm_means_first_rows():
    from oracle.targets import many_gmm_means
    m = many_gmm_means()
    assert m.shape == (40, 2)
    np.testing.assert_array_equal(m[0], np.float32([-15.758228, 18.116531]))
    np.testing.assert_array_equal(m[1], np.float32([-34.80889, 21.225481]))


def test_host_prng_matches_oracle_bits():
    for seed in (0, 1, 12345, 999999):
        for n in (1, 2, 3, 10, 80):
            np.testing.assert_array_equal(host_prng.random_bits(seed, n),
                                          prng.random_bits(prng.prng_key(np.array(seed)), n))
    np.testing.assert_array_equal(host_prng.uniform(0, (40, 2), -1.0, 1.0),
                                  prng.uniform(prng.prng_key(np.array(0)), (40, 2), -1.0, 1.0))


def test_odd_length_bits_padding():
    # counters are zero-padded to even length; the first n outputs must not depend on the pad slot
    key = prng.prng_key(np.array(7))
    b3 = prng.random_bits(key, 3)
    x0, x1 = prng.threefry2x32(key[0], key[1], np.uint32([0, 1]), np.uint32([2, 0]))
    assert b3.tolist() == [int(x0[0]), int(x0[1]), int(x1[0])]


def test_golden_prng_fixture():
    g = np.load(os.path.join(GOLD, "prng_kat.npz"))
    k0 = prng.prng_key(np.array(0))
    a, b = prng.split(k0)
    np.testing.assert_array_equal(g["split0"], np.stack([a, b]))
    eps0, eps = prng.particle_noise(np.arange(1, 5), 3, 4)
    np.testing.assert_array_equal(g["chain_eps0"], eps0)
    np.testing.assert_array_equal(g["chain_eps"], eps)


def test_key_chain_structure():
    """The chain of mcdboundingmachine.py:151-162 / mcd_cais.py:66,87,94 written out longhand."""
    seed, d, K = np.array([5]), 2, 3
    eps0, eps = prng.particle_noise(seed, d, K)
    k = prng.prng_key(seed)
    rng_key, gen = prng.split(k)
    np.testing.assert_array_equal(eps0, prng.normal(rng_key, d))
    rng_key, gen = prng.split(gen)          # key handed to evolve is rng_key
    gen = rng_key
    rng_key, gen = prng.split(gen)          # mcd_cais.py:94
    for i in range(K):
        rng_key, gen = prng.split(gen)      # :66
        np.testing.assert_array_equal(eps[:, i], prng.normal(rng_key, d))
        rng_key, gen = prng.split(gen)      # :87


def test_normal_statistics():
    x = prng.normal(prng.prng_key(np.arange(1, 2001)), 10).astype(np.float64)
    assert abs(x.mean()) < 0.03 and abs(x.std() - 1.0) < 0.03
