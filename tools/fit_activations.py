"""Derives the polynomial behind the device GELU:  erfc(s/sqrt2) = 2^(-s*R(s)),  s = |x| in [0, SMAX].
gelu(x) = max(x,0) - 0.5*|x|*erfc(|x|/sqrt2).  Weighted least squares on Chebyshev nodes (weight =
sensitivity of the final value to an error in R), then the fp32 Horner evaluation is checked
against float64 scipy.  Prints C++ coefficients (highest degree first)."""
import numpy as np
from scipy.special import erfc, erf

SMAX = 6.0
LOG2E = 1.4426950408889634

def target_R(s):
    # R(s) = -log2(erfc(s/sqrt2))/s ; limit s->0: sqrt(2/pi)*log2e
    with np.errstate(divide="ignore", invalid="ignore"):
        from scipy.special import log_ndtr
        q = -(np.log(2.0) + log_ndtr(-s))          # -ln erfc(s/sqrt2) = -ln(2 Phi(-s))
        r = q * LOG2E / s
    r[s == 0] = np.sqrt(2 / np.pi) * LOG2E
    return r

def fit(deg, iters=40):
    n = 4000
    k = np.arange(n)
    s = 0.5 * SMAX * (1 - np.cos(np.pi * (k + 0.5) / n))
    R = target_R(s)
    e = erfc(s / np.sqrt(2))
    sens = 0.5 * np.maximum(s, 1.0) * e * np.log(2.0) * s      # d gelu / dR  (abs error weight)
    w = sens.copy() + 1e-12
    x = 2 * s / SMAX - 1
    V = np.polynomial.chebyshev.chebvander(x, deg)
    for _ in range(iters):                                       # Lawson reweighting -> minimax-ish
        c, *_ = np.linalg.lstsq(V * w[:, None], R * w, rcond=None)
        err = np.abs((V @ c - R) * sens)
        w = w * (0.5 + err / err.max())
        w /= w.max()
    p = np.polynomial.chebyshev.cheb2poly(c)
    # convert from x to s: x = 2s/SMAX - 1
    P = np.polynomial.Polynomial(p)(np.polynomial.Polynomial([-1.0, 2.0 / SMAX]))
    return P.coef  # ascending in s

def gelu_f32(x, coef):
    x = x.astype(np.float32)
    s = np.minimum(np.abs(x), np.float32(SMAX))
    r = np.full_like(s, np.float32(coef[-1]))
    for c in coef[-2::-1]:
        r = (r * s + np.float32(c)).astype(np.float32)           # (fma on device; mul+add here)
    e = np.exp2((-(s * r)).astype(np.float32)).astype(np.float32)
    return (np.maximum(x, np.float32(0)) - np.float32(0.5) * np.abs(x) * e).astype(np.float32)

if __name__ == "__main__":
    xs = np.concatenate([np.linspace(-12, 12, 2000001), np.random.default_rng(0).normal(0, 1.5, 1000000)])
    ref = xs * 0.5 * (1 + erf(xs / np.sqrt(2)))
    for deg in (4, 5, 6, 7, 8, 9):   # the forward kernels use 5, the gradient kernels (value + derivative) 9
        coef = fit(deg)
        g = gelu_f32(xs, coef).astype(np.float64)
        xr = xs.astype(np.float32).astype(np.float64)
        ref32 = xr * 0.5 * (1 + erf(xr / np.sqrt(2)))
        abs_err = np.abs(g - ref32)
        rel = abs_err / np.maximum(np.abs(ref32), 1e-30)
        print(deg, "max abs err %.3e  max err/max(1,|x|) %.3e  max rel(|x|<4) %.3e" % (
            abs_err.max(), (abs_err / np.maximum(1, np.abs(xr))).max(), rel[np.abs(xr) < 4].max()))
        print("   coef (highest first):", ", ".join("%.9ef" % c for c in coef[::-1]))
