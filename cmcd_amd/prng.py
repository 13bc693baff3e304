"""Host-side Threefry-2x32 (the jax.random generator) — product code needs it only for target
constants that the reference draws from a PRNG (many_gmm means,
/root/reference/src/model_handler.py:255-261).  The device kernels carry their own copy."""
import numpy as np

_ROT = (13, 15, 26, 6, 17, 29, 16, 24)
_M = 0xFFFFFFFF


def _threefry(k0, k1, x0, x1):
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0 = (x0 + ks[0]) & _M
    x1 = (x1 + ks[1]) & _M
    for g in range(5):
        for r in _ROT[4 * (g % 2):4 * (g % 2) + 4]:
            x0 = (x0 + x1) & _M
            x1 = ((x1 << r) | (x1 >> (32 - r))) & _M
            x1 ^= x0
        x0 = (x0 + ks[(g + 1) % 3]) & _M
        x1 = (x1 + ks[(g + 2) % 3] + g + 1) & _M
    return x0, x1


def random_bits(seed, n):
    """bits of PRNGKey(seed) for n draws, jax's original (non-partitionable) counter layout."""
    h = (n + 1) // 2
    ctr = list(range(n)) + [0] * (2 * h - n)
    out = [0] * (2 * h)
    for j in range(h):
        out[j], out[h + j] = _threefry(0, seed & _M, ctr[j], ctr[h + j])
    return np.array(out[:n], np.uint32)


def uniform(seed, shape, minval, maxval):
    """jax.random.uniform(PRNGKey(seed), shape, minval=minval, maxval=maxval) in float32."""
    n = int(np.prod(shape))
    bits = random_bits(seed, n)
    u = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    lo, hi = np.float32(minval), np.float32(maxval)
    return np.maximum(lo, u * (hi - lo) + lo).reshape(shape)
