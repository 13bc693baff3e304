// Device-side building blocks of the CMCD trajectory kernel (gfx950 / CDNA4 only).
// Counter-based PRNG (the jax.random contract), transcendental helpers, targets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmcd_hip.h"

namespace cmcd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr float kHalfLog2Pi = 0.91893853320467274178f;
constexpr float kLog2Pi = 1.8378770664093453f;

// ---------------------------------------------------------------------------------------------
// Threefry-2x32, 20 rounds: jax.random's default generator, reached from the reference through
// jax.random.split / normal (/root/reference/src/mcd_cais.py:66-67,87,94;
// /root/reference/src/mcdboundingmachine.py:151-162).  Integer arithmetic: bit-exact.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rotl32(uint32_t x, int r) { return __builtin_rotateleft32(x, r); }

__device__ __forceinline__ void threefry2x32(uint32_t k0, uint32_t k1, uint32_t& x0, uint32_t& x1) {
  const uint32_t k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  x0 += k0;
  x1 += k1;
#define CMCD_TF_ROUND(r) \
  x0 += x1;              \
  x1 = rotl32(x1, r);    \
  x1 ^= x0;
  CMCD_TF_ROUND(13) CMCD_TF_ROUND(15) CMCD_TF_ROUND(26) CMCD_TF_ROUND(6)
  x0 += k1; x1 += k2 + 1u;
  CMCD_TF_ROUND(17) CMCD_TF_ROUND(29) CMCD_TF_ROUND(16) CMCD_TF_ROUND(24)
  x0 += k2; x1 += k0 + 2u;
  CMCD_TF_ROUND(13) CMCD_TF_ROUND(15) CMCD_TF_ROUND(26) CMCD_TF_ROUND(6)
  x0 += k0; x1 += k1 + 3u;
  CMCD_TF_ROUND(17) CMCD_TF_ROUND(29) CMCD_TF_ROUND(16) CMCD_TF_ROUND(24)
  x0 += k1; x1 += k2 + 4u;
  CMCD_TF_ROUND(13) CMCD_TF_ROUND(15) CMCD_TF_ROUND(26) CMCD_TF_ROUND(6)
  x0 += k2; x1 += k0 + 5u;
#undef CMCD_TF_ROUND
}

// XLA's f32 erf_inv (Giles' single-precision polynomial), as used by jax.random.normal.
__device__ __forceinline__ float erfinv_giles(float x) {
  float w = -log1pf(-x * x);
  float p;
  if (w < 5.0f) {
    w = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, w, 3.43273939e-07f);
    p = fmaf(p, w, -3.5233877e-06f);
    p = fmaf(p, w, -4.39150654e-06f);
    p = fmaf(p, w, 0.00021858087f);
    p = fmaf(p, w, -0.00125372503f);
    p = fmaf(p, w, -0.00417768164f);
    p = fmaf(p, w, 0.246640727f);
    p = fmaf(p, w, 1.50140941f);
  } else {
    w = sqrtf(w) - 3.0f;
    p = -0.000200214257f;
    p = fmaf(p, w, 0.000100950558f);
    p = fmaf(p, w, 0.00134934322f);
    p = fmaf(p, w, -0.00367342844f);
    p = fmaf(p, w, 0.00573950773f);
    p = fmaf(p, w, -0.0076224613f);
    p = fmaf(p, w, 0.00943887047f);
    p = fmaf(p, w, 1.00167406f);
    p = fmaf(p, w, 2.83297682f);
  }
  return p * x;
}

// uint32 random bits -> one N(0,1) float, jax.random.normal's recipe for float32.
__device__ __forceinline__ float bits_to_normal(uint32_t b) {
  const float lo = -0.99999994f;  // nextafter(-1, 0)
  float u = __uint_as_float((b >> 9) | 0x3F800000u) - 1.0f;
  u = fmaxf(lo, u * 2.0f + lo);   // (hi - lo) rounds to 2.0f in float32
  return 1.41421356237309504880f * erfinv_giles(u);
}

// ---------------------------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------------------------
// gelu(x) = x/2 (1 + erf(x / sqrt 2))   /root/reference/src/nn_dds.py:167-176
// Evaluated as max(x,0) - |x|/2 * erfc(|x|/sqrt 2) with erfc(s/sqrt 2) = 2^(-s R(s)), R a degree-9
// polynomial (tools/fit_activations.py): one v_exp_f32, no branch, no cancellation for x < 0.
// Max error vs float64: 2.4e-7 absolute (= 1/2 ulp of the result), 8e-8 * max(1,|x|).
__device__ __forceinline__ float gelu_fast(float x) {
  const float ax = fabsf(x);
  const float s = fminf(ax, 6.0f);
  float r = 5.626459558e-08f;
  r = fmaf(r, s, -1.389874702e-06f);
  r = fmaf(r, s, 1.521236383e-05f);
  r = fmaf(r, s, -9.455732447e-05f);
  r = fmaf(r, s, 3.240720773e-04f);
  r = fmaf(r, s, -6.315276129e-05f);
  r = fmaf(r, s, -6.896958595e-03f);
  r = fmaf(r, s, 5.242151140e-02f);
  r = fmaf(r, s, 4.592238824e-01f);
  r = fmaf(r, s, 1.151104120e+00f);
  const float e = __builtin_amdgcn_exp2f(-(s * r));
  return fmaf(-0.5f * ax, e, fmaxf(x, 0.0f));
}
// reference-grade version (ocml erff), used by the prep kernel's time coder
__device__ __forceinline__ float gelu_exact(float x) {
  return x * 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
}
// stax Softplus = logaddexp(x, 0)        /root/reference/src/nn.py:46
// max(x,0) + ln2 * log2(1 + 2^(-|x| log2 e)): v_exp_f32 + v_log_f32, absolute error < 1.5e-7.
__device__ __forceinline__ float softplus(float x) {
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(x));
  return fmaf(0.69314718055994530942f, __builtin_amdgcn_logf(1.0f + e), fmaxf(x, 0.0f));
}

// lanes l, l^16, l^32, l^48 hold the four k-slices of one particle: butterfly over them
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}

// ---------------------------------------------------------------------------------------------
// Targets: log p(z) and its gradient in closed form (the reference uses jax.grad,
// /root/reference/src/mcd_cais.py:24-30).  `tc` points at the target constants in LDS.
// `g` = lane >> 4 (which quarter of the particle's work this lane owns).
// ---------------------------------------------------------------------------------------------
template <int TARGET, int D>
struct Target;

// many_gmm: /root/reference/src/model_handler.py:245-281.  tc = {inv_scale, logc, n_mixes(bits),
// pad, means[n_mixes][2]}; components are dealt round-robin to the 4 lanes of a particle.
template <>
struct Target<CMCD_TARGET_MANY_GMM, 2> {
  static constexpr int kLdsHeader = 4;
  __device__ static __forceinline__ void eval(const float (&z)[2], int g, const float* tc, float& logp,
                                              float (&grad)[2]) {
    const float inv_s = tc[0], logc = tc[1];
    const int nm = __float_as_int(tc[2]);
    const float2* mu = reinterpret_cast<const float2*>(tc + kLdsHeader);
    constexpr int kMaxPer = 16;  // up to 64 mixtures
    float lg[kMaxPer], dx[kMaxPer], dy[kMaxPer];
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < kMaxPer; ++q) {
      const int k = g + 4 * q;
      if (4 * q < nm) {  // wave-uniform trip bound
        const float2 mk = mu[k < nm ? k : 0];
        dx[q] = (z[0] - mk.x) * inv_s;
        dy[q] = (z[1] - mk.y) * inv_s;
        lg[q] = (k < nm) ? fmaf(-0.5f, dx[q] * dx[q] + dy[q] * dy[q], logc) : -INFINITY;
        m = fmaxf(m, lg[q]);
      }
    }
    m = group_max(m);
    float s = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int q = 0; q < kMaxPer; ++q) {
      if (4 * q < nm) {
        const float e = __expf(lg[q] - m);  // exp(-inf) = 0 for padded slots
        s += e;
        sx = fmaf(e, dx[q], sx);
        sy = fmaf(e, dy[q], sy);
      }
    }
    s = group_sum(s);
    sx = group_sum(sx);
    sy = group_sum(sy);
    const float lp = m + logf(s);
    const bool valid = lp > -1e4f;  // model_handler.py:279-280
    const float sc = -inv_s / s;
    logp = valid ? lp : -INFINITY;
    grad[0] = valid ? sx * sc : 0.f;
    grad[1] = valid ? sy * sc : 0.f;
  }
};

// gmm: /root/reference/src/model_handler.py:157-200 (3 components, symmetrised by flip).
template <>
struct Target<CMCD_TARGET_GMM, 2> {
  static constexpr int kLdsHeader = 0;
  __device__ static __forceinline__ void raw(float x, float y, float& f, float& gx, float& gy) {
    // Sigma^-1 of diag(0.7, 0.05) and of [[1,.95],[.95,1]];  logc = -log 2pi - sum log diag(chol) + log(1/3)
    constexpr float pa00 = 1.0f / 0.7f, pa11 = 20.0f;
    constexpr float pc00 = 10.256410256410257f, pc01 = -9.743589743589743f;
    constexpr float lca = -1.2602857463310935f;  // components a, b
    constexpr float lcc = -1.7725379045882876f;  // component c
    float d0 = x - 3.0f, d1 = y;
    const float pa0 = pa00 * d0, pa1 = pa11 * d1;
    const float la = fmaf(-0.5f, d0 * pa0 + d1 * pa1, lca);
    d0 = x + 2.5f;
    const float pb0 = pa00 * d0, pb1 = pa11 * d1;
    const float lb = fmaf(-0.5f, d0 * pb0 + d1 * pb1, lca);
    d0 = x - 2.0f;
    d1 = y - 3.0f;
    const float pc0 = pc00 * d0 + pc01 * d1, pc1 = pc01 * d0 + pc00 * d1;
    const float lc = fmaf(-0.5f, d0 * pc0 + d1 * pc1, lcc);
    const float m = fmaxf(la, fmaxf(lb, lc));
    const float ea = expf(la - m), eb = expf(lb - m), ec = expf(lc - m);
    const float s = ea + eb + ec;
    f = m + logf(s);
    const float is = -1.0f / s;
    gx = (ea * pa0 + eb * pb0 + ec * pc0) * is;
    gy = (ea * pa1 + eb * pb1 + ec * pc1) * is;
  }
  __device__ static __forceinline__ void eval(const float (&z)[2], int, const float*, float& logp,
                                              float (&grad)[2]) {
    float fa, gax, gay, fb, gbx, gby;
    raw(z[0], z[1], fa, gax, gay);
    raw(z[1], z[0], fb, gbx, gby);  // log_density(flip(x)), model_handler.py:192-195
    const float m = fmaxf(fa, fb);
    const float lse = m + logf(expf(fa - m) + expf(fb - m));
    logp = lse - 0.69314718055994530942f;
    const float wa = expf(fa - lse), wb = expf(fb - lse);
    grad[0] = wa * gax + wb * gby;  // un-flip the second gradient
    grad[1] = wa * gay + wb * gbx;
  }
};

// funnel: /root/reference/src/model_handler.py:124-143 (scale of v hard-coded 3.0).
template <int D>
struct Target<CMCD_TARGET_FUNNEL, D> {
  static constexpr int kLdsHeader = 0;
  __device__ static __forceinline__ void eval(const float (&z)[D], int, const float*, float& logp,
                                              float (&grad)[D]) {
    const float v = z[0];
    float ss = 0.f;
#pragma unroll
    for (int j = 1; j < D; ++j) ss = fmaf(z[j], z[j], ss);
    const float emv = expf(-v);
    constexpr float c0 = -0.5f * kLog2Pi - 1.0986122886681098f;  // -log sqrt(2pi) - log 3
    constexpr float c1 = -0.5f * (D - 1) * kLog2Pi;
    logp = c0 - v * v / 18.0f + c1 - 0.5f * (D - 1) * v - 0.5f * emv * ss;
    grad[0] = -v / 9.0f - 0.5f * (D - 1) + 0.5f * emv * ss;
#pragma unroll
    for (int j = 1; j < D; ++j) grad[j] = -z[j] * emv;
  }
};

}  // namespace cmcd
