"""Oracle restatement of the JAX PRNG contract used by the reference hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  PARITY UNPINNED by the
reference itself; pinned by the public JAX known-answer constants in
tests/test_oracle_prng.py.

The reference draws every random number through ``jax.random`` with the default
"threefry2x32" implementation in its original (non-partitionable) layout:
  * ``jax.random.PRNGKey(seed)``                /root/reference/src/mcdboundingmachine.py:151
  * ``jax.random.split``                        /root/reference/src/mcdboundingmachine.py:153,162
                                                /root/reference/src/mcd_cais.py:66,87,94
  * ``jax.random.normal(key, (d,))``            /root/reference/src/mcd_utils.py:15
                                                /root/reference/src/vardist/diag_gauss.py:54
  * ``jax.random.uniform(key, (40,2), -1, 1)``  /root/reference/src/model_handler.py:256-260
jax itself is a third-party dependency absent from /root/reference (and
unpinned in /root/reference/env.yaml); what follows restates its published
algorithm (Salmon et al. Threefry-2x32, 20 rounds; Giles' single-precision
erfinv as emitted by XLA).  All arithmetic here is uint32 / float32 exactly as
in jax, so outputs are bit-reproducible.
"""
import numpy as np

_U32 = np.uint32
_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return (x << _U32(r)) | (x >> _U32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32, 20 rounds.  All args uint32 arrays (broadcastable)."""
    k0 = np.asarray(k0, _U32)
    k1 = np.asarray(k1, _U32)
    x0 = np.asarray(x0, _U32).copy()
    x1 = np.asarray(x1, _U32).copy()
    with np.errstate(over="ignore"):
        ks = (k0, k1, k0 ^ k1 ^ _U32(0x1BD11BDA))
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for g in range(1, 6):
            for r in _ROT[(g - 1) % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r)
                x1 = x1 ^ x0
            x0 = x0 + ks[g % 3]
            x1 = x1 + ks[(g + 1) % 3] + _U32(g)
    return x0, x1


def prng_key(seed):
    """PRNGKey(seed) for int32 seeds -> uint32[..., 2] = (0, seed)."""
    seed = np.asarray(seed)
    out = np.zeros(seed.shape + (2,), _U32)
    out[..., 1] = seed.astype(np.int64).astype(_U32)
    return out


def random_bits(key, n):
    """jax ``_threefry_random_bits`` (original layout): key uint32[...,2] -> uint32[..., n].

    Counters 0..n-1, zero-padded to even length 2h; block j encrypts
    (ctr[j], ctr[h+j]) and yields (out[j], out[h+j]).
    """
    key = np.asarray(key, _U32)
    h = (n + 1) // 2
    ctr = np.zeros(2 * h, _U32)
    ctr[:n] = np.arange(n, dtype=_U32)
    lo, hi = threefry2x32(key[..., 0:1], key[..., 1:2], ctr[:h], ctr[h:])
    return np.concatenate([lo, hi], axis=-1)[..., :n]


def split(key):
    """jax.random.split(key) -> (first, second), each uint32[..., 2]."""
    b = random_bits(key, 4)
    return b[..., 0:2], b[..., 2:4]


def _bits_to_unit_float(bits):
    """uint32 -> float32 in [0, 1): mantissa trick of jax.random.uniform."""
    f = ((bits >> _U32(9)) | _U32(0x3F800000)).view(np.float32)
    return f - np.float32(1.0)


def uniform(key, shape, minval=0.0, maxval=1.0):
    n = int(np.prod(shape))
    u = _bits_to_unit_float(random_bits(key, n))
    lo, hi = np.float32(minval), np.float32(maxval)
    u = u * (hi - lo) + lo
    u = np.maximum(lo, u)
    return u.reshape(key.shape[:-1] + tuple(shape))


# Giles, "Approximating the erfinv function" (single precision), the polynomial
# XLA emits for f32 erf_inv.  Horner, highest degree first.
_ERFINV_CENTRAL = [2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06,
                   0.00021858087, -0.00125372503, -0.00417768164, 0.246640727, 1.50140941]
_ERFINV_TAIL = [-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844,
                0.00573950773, -0.0076224613, 0.00943887047, 1.00167406, 2.83297682]


def erfinv_f32(x):
    x = np.asarray(x, np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = -np.log1p((-x * x).astype(np.float32)).astype(np.float32)
        wc = w - np.float32(2.5)
        wt = np.sqrt(w).astype(np.float32) - np.float32(3.0)
        pc = np.full_like(x, np.float32(_ERFINV_CENTRAL[0]))
        pt = np.full_like(x, np.float32(_ERFINV_TAIL[0]))
        for c in _ERFINV_CENTRAL[1:]:
            pc = (np.float32(c) + pc * wc).astype(np.float32)
        for c in _ERFINV_TAIL[1:]:
            pt = (np.float32(c) + pt * wt).astype(np.float32)
        p = np.where(w < np.float32(5.0), pc, pt)
        return (p * x).astype(np.float32)


def normal(key, d):
    """jax.random.normal(key, (d,)) for float32 -> float32[..., d]."""
    bits = random_bits(key, d)
    u = _bits_to_unit_float(bits)
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    hi = np.float32(1.0)
    u = u * (hi - lo) + lo
    u = np.maximum(lo, u)
    return (np.float32(np.sqrt(2.0)) * erfinv_f32(u)).astype(np.float32)


def particle_noise(seeds, dim, nbridges):
    """Key chain of one particle (vectorised over seeds).

    /root/reference/src/mcdboundingmachine.py:151-162 and
    /root/reference/src/mcd_cais.py:66,87,94.  Returns
    (eps0 float32[N, dim], eps float32[N, nbridges, dim]).
    """
    k0 = prng_key(seeds)
    a, b = split(k0)                      # rng_key, rng_key_gen
    eps0 = normal(a, dim)                 # vd.sample_rep
    c, _ = split(b)                       # key handed to evolve
    _, gen = split(c)                     # mcd_cais.py:94
    out = np.zeros((k0.shape[0], nbridges, dim), np.float32)
    for i in range(nbridges):
        g, h = split(gen)                 # mcd_cais.py:66
        out[:, i, :] = normal(g, dim)
        _, gen = split(h)                 # mcd_cais.py:87
    return eps0, out


def particle_noise_uha(seeds, dim, nbridges):
    """Key chain of one particle under ``MCD_CAIS_UHA_sn`` (vectorised over seeds).

    /root/reference/src/mcdboundingmachine.py:151-162 hands ``C = first(split(B))`` to evolve;
    /root/reference/src/mcd_under_lp_a_cais.py:92-93 draws the initial momentum from
    ``first(split(C))``, :100 takes ``gen = second(split(second(split(C))))``, and every step
    draws its momentum-refresh noise from ``first(split(gen))`` (:55) and moves on to
    ``second(split(second(split(gen))))`` (:84).  Returns
    (eps0 float32[N, dim], rho0 float32[N, dim], eps float32[N, nbridges, dim]).
    """
    k0 = prng_key(seeds)
    a, b = split(k0)                      # rng_key, rng_key_gen
    eps0 = normal(a, dim)                 # vd.sample_rep
    c, _ = split(b)                       # key handed to evolve
    r, gen = split(c)                     # mcd_under_lp_a_cais.py:92
    rho0 = normal(r, dim)                 # :93
    _, gen = split(gen)                   # :100
    out = np.zeros((k0.shape[0], nbridges, dim), np.float32)
    for i in range(nbridges):
        g, h = split(gen)                 # :55
        out[:, i, :] = normal(g, dim)     # :56 sample_kernel
        _, gen = split(h)                 # :84
    return eps0, rho0, out
