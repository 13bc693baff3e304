// Device-side building blocks of the CMCD trajectory kernel (gfx950 / CDNA4 only).
// Counter-based PRNG (the jax.random contract), transcendental helpers, targets.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "cmcd_hip.h"

namespace cmcd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
  // w = -log1p(-x * x) with x * x ROUNDED to float32 first, as XLA evaluates it: for |x| near 1 the rounding of x^2
  // moves 1 - x^2 by up to 2e-4 relative, i.e. the deviate by up to ~90 ulp against the mathematically better
  // (1 - x)(1 + x) — the reference's numbers carry that rounding, so this does too (the subtraction 1 - t is exact
  // for t >= 1/2 and costs < 0.3 ulp of the deviate below).  One v_log_f32; measured against the float32 restatement
  // of XLA's formula over all 2^23 mantissas: <= 4 ulp (tests/test_gpu_prng.py holds every captured deviate to that).
  float om;
  {
#pragma clang fp contract(off)   // hipcc contracts 1 - x * x into fma(-x, x, 1) — exactly the rounding XLA does NOT skip
    const float t = x * x;
    om = 1.0f - t;
  }
  const float w = -0.69314718055994530942f * __builtin_amdgcn_logf(om);
  // central branch (w < 5, |x| < 0.9966: 99.7 % of the lanes) for every lane; the tail branch — an IEEE square root
  // (~17 instructions) and a second degree-8 polynomial — only in waves where some lane needs it (a scalar branch on the
  // ballot: ~20 % of the waves).  As straight-line code the compiler evaluated both for every lane and selected: ~45
  // instead of ~18 instructions per deviate, 110 of the wave-per-tile kernel's ~980 per tile-evaluation (ISA reading r02).
  float p;
  {
    const float v = w - 2.5f;
    p = 2.81022636e-08f;
    p = fmaf(p, v, 3.43273939e-07f);
    p = fmaf(p, v, -3.5233877e-06f);
    p = fmaf(p, v, -4.39150654e-06f);
    p = fmaf(p, v, 0.00021858087f);
    p = fmaf(p, v, -0.00125372503f);
    p = fmaf(p, v, -0.00417768164f);
    p = fmaf(p, v, 0.246640727f);
    p = fmaf(p, v, 1.50140941f);
  }
  const bool tail = !(w < 5.0f);
  if (__builtin_amdgcn_ballot_w64(tail) != 0) {
    float wt = w;
    asm volatile("; erfinv tail" : "+v"(wt));   // the tail's input passes through a volatile statement: not speculatable,
                                                 // so this stays a branch (the compiler flattened it back otherwise)
    const float v = sqrtf(wt) - 3.0f;
    float t = -0.000200214257f;
    t = fmaf(t, v, 0.000100950558f);
    t = fmaf(t, v, 0.00134934322f);
    t = fmaf(t, v, -0.00367342844f);
    t = fmaf(t, v, 0.00573950773f);
    t = fmaf(t, v, -0.0076224613f);
    t = fmaf(t, v, 0.00943887047f);
    t = fmaf(t, v, 1.00167406f);
    t = fmaf(t, v, 2.83297682f);
    p = tail ? t : p;
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
// Evaluated as max(x,0) - |x|/2 * erfc(|x|/sqrt 2) with erfc(s/sqrt 2) = 2^(-s R(s)), R a polynomial
// (tools/fit_activations.py): one v_exp_f32, no branch, no cancellation for x < 0.
__device__ __forceinline__ float gelu_fast(float x) {
  // degree 5 (r02; tools/fit_activations.py): max abs error 3.9e-7 against float64 (rms 9e-8) where the degree-9 fit of
  // r01 reaches 2.4e-7 — the float32 rounding of the result itself — for four FMAs fewer on the one function that is a
  // third of the forward kernels' VALU instructions.  The parity bar is 1e-3 on the batch statistics, observed 1e-5.
  const float ax = fabsf(x);
  const float s = fminf(ax, 6.0f);
  float r = -2.386156740e-05f;
  r = fmaf(r, s, 6.893407597e-04f);
  r = fmaf(r, s, -7.823501478e-03f);
  r = fmaf(r, s, 5.302766042e-02f);
  r = fmaf(r, s, 4.590415202e-01f);
  r = fmaf(r, s, 1.151121845e+00f);
  // erfc / 2 = 2^(-s R - 1): the halving rides in the exponent (one fma instead of a multiply there and another on |x|)
  const float he = __builtin_amdgcn_exp2f(fmaf(-s, r, -1.0f));
  return fmaf(-ax, he, fmaxf(x, 0.0f));
}
// d gelu / dx = Phi(x) + x phi(x), Phi from the same erfc polynomial as gelu_fast (r02: degree 5 here too — the
// derivative's max abs error is 2.5e-7 against 8e-8 with the degree-9 fit, both at the float32 rounding of a quantity of
// order one, and the value the gradient kernels recompute is now bit for bit the forward kernels' activation).
__device__ __forceinline__ float gelu_erfc_half(float ax) {   // erfc(|x| / sqrt 2) / 2
  const float s = fminf(ax, 6.0f);
  float r = -2.386156740e-05f;
  r = fmaf(r, s, 6.893407597e-04f);
  r = fmaf(r, s, -7.823501478e-03f);
  r = fmaf(r, s, 5.302766042e-02f);
  r = fmaf(r, s, 4.590415202e-01f);
  r = fmaf(r, s, 1.151121845e+00f);
  return __builtin_amdgcn_exp2f(fmaf(-s, r, -1.0f));
}
__device__ __forceinline__ float gelu_grad_fast(float x) {
  const float he = gelu_erfc_half(fabsf(x));
  const float cdf = x >= 0.f ? 1.0f - he : he;
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  return fmaf(x, pdf, cdf);
}
// gelu and its derivative from ONE evaluation of the erfc polynomial and exponential (the gradient kernels need
// both at every hidden unit): 19 instructions (24 with the degree-9 fit of r01)
__device__ __forceinline__ float gelu_fast_both(float x, float& dg) {
  const float ax = fabsf(x);
  const float he = gelu_erfc_half(ax);
  const float cdf = x >= 0.f ? 1.0f - he : he;
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  dg = fmaf(x, pdf, cdf);
  return fmaf(-ax, he, fmaxf(x, 0.0f));
}
// softplus and its derivative (sigmoid) from one exponential
__device__ __forceinline__ float softplus_both(float x, float& dsp) {
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(x));
  const float inv = __builtin_amdgcn_rcpf(1.0f + e);
  dsp = x >= 0.f ? inv : e * inv;
  return fmaf(0.69314718055994530942f, __builtin_amdgcn_logf(1.0f + e), fmaxf(x, 0.0f));
}
// d softplus / dx = sigmoid(x), branch-free and overflow-free
__device__ __forceinline__ float sigmoid_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * fabsf(x));
  const float inv = __builtin_amdgcn_rcpf(1.0f + e);
  return x >= 0.f ? inv : e * inv;
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

// Cross-row exchange with the gfx950 row-swap instructions (VALU, no LDS round trip).  Measured
// semantics (tools/probes/permlane_probe.hip), rows = 16-lane groups of the wave:
//   v_permlane32_swap(a, b) -> r0 = rows [a0 a1 b0 b1], r1 = rows [a2 a3 b2 b3]
//   v_permlane16_swap(a, b) -> r0 = rows [a0 b0 a2 b2], r1 = rows [a1 b1 a3 b3]
__device__ __forceinline__ void swap32(uint32_t a, uint32_t b, uint32_t& r0, uint32_t& r1) {
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  r0 = r[0];
  r1 = r[1];
}
__device__ __forceinline__ void swap16(uint32_t a, uint32_t b, uint32_t& r0, uint32_t& r1) {
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  r0 = r[0];
  r1 = r[1];
}
// lanes l, l^16, l^32, l^48 hold the four k-slices of one particle: butterfly over them
__device__ __forceinline__ float group_sum(float v) {
  uint32_t a, b;
  swap32(__float_as_uint(v), __float_as_uint(v), a, b);
  v = __uint_as_float(a) + __uint_as_float(b);
  swap16(__float_as_uint(v), __float_as_uint(v), a, b);
  return __uint_as_float(a) + __uint_as_float(b);
}
__device__ __forceinline__ float group_sum_swap(float v) { return group_sum(v); }
__device__ __forceinline__ float group_max(float v) {
  uint32_t a, b;
  swap32(__float_as_uint(v), __float_as_uint(v), a, b);
  v = fmaxf(__uint_as_float(a), __uint_as_float(b));
  swap16(__float_as_uint(v), __float_as_uint(v), a, b);
  return fmaxf(__uint_as_float(a), __uint_as_float(b));
}
// xor-8 exchange inside each 16-lane row (DPP row_ror:8)
__device__ __forceinline__ float xor8(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));
}
// rotation by 4 inside each 16-lane row (DPP row_ror:4)
__device__ __forceinline__ float ror4(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));
}
// sum / max over the LP lanes that share a particle: LP = 4 -> lanes {c, c+16, c+32, c+48};
// LP = 8 -> lanes {c8 + 8 s, s < 8}; LP = 16 -> lanes {c4 + 4 s, s < 16}
template <int LP>
__device__ __forceinline__ float part_sum(float v) {
  if (LP >= 8) v += xor8(v);
  if (LP == 16) v += ror4(v);
  return group_sum(v);
}
template <int LP>
__device__ __forceinline__ float part_max(float v) {
  if (LP >= 8) v = fmaxf(v, xor8(v));
  if (LP == 16) v = fmaxf(v, ror4(v));
  return group_max(v);
}
// minimum of a NON-NEGATIVE float (a squared distance, possibly +inf) over the LP lanes of a particle, taken on the bit
// patterns as unsigned integers: same order, same result as fminf (a NaN pattern is larger than +inf and is ignored the
// same way), but no canonicalising v_max x, x in front of every step (llvm.minnum quiets its inputs: 3 instructions per
// butterfly stage instead of 1) and the DPP moves fold into the v_min_u32.
template <int LP>
__device__ __forceinline__ float part_min_nonneg(float v) {
  uint32_t u = __float_as_uint(v);
  if (LP >= 8) u = min(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xf, 0xf, false));
  if (LP == 16) u = min(u, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)u, 0x124, 0xf, 0xf, false));
  uint32_t a, b;
  swap32(u, u, a, b);
  u = min(a, b);
  swap16(u, u, a, b);
  return __uint_as_float(min(a, b));
}
// sum over the 16 lanes of a row (the 16 particles of a tile, for a value held per particle), DPP only:
// quad butterflies then row rotations by 4 and 8.  Every lane of the row receives the total.
__device__ __forceinline__ float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xf, 0xf, false));  // row_ror:8
  return v;
}
// every lane receives x of row 0 (r0) and of row 1 (r1) at its own column
__device__ __forceinline__ void rows01(uint32_t x, uint32_t& r0, uint32_t& r1) {
  uint32_t a, b;
  swap32(x, x, a, b);   // a = rows [x0 x1 x0 x1]
  swap16(a, a, r0, r1); // r0 = [x0 x0 x0 x0], r1 = [x1 x1 x1 x1]
}
// every lane receives x of all four rows at its own column
__device__ __forceinline__ void rows0123(uint32_t x, uint32_t (&r)[4]) {
  uint32_t a, b;
  swap32(x, x, a, b);   // a = [x0 x1 x0 x1], b = [x2 x3 x2 x3]
  swap16(a, a, r[0], r[1]);
  swap16(b, b, r[2], r[3]);
}

// ---------------------------------------------------------------------------------------------
// Targets: log p(z) and its gradient in closed form (the reference uses jax.grad,
// /root/reference/src/mcd_cais.py:24-30).  `tc` points at the target constants in LDS.
// `g` = lane >> 4 (which quarter of the particle's work this lane owns).
// ---------------------------------------------------------------------------------------------
template <int TARGET, int D>
struct Target;

// many_gmm: /root/reference/src/model_handler.py:245-281.  tc = {inv_scale, c2, n_mixes(bits), c0,
// means[n_mixes][2]} with logit_k (in log2 units) = c0 + c2 * |z - mu_k|^2.  Components are dealt
// round-robin to the 4 lanes of a particle; evaluation is split in two halves so that a
// cooperative kernel can put a workgroup barrier between them:
//   pass1: squared distances (kept in registers) and their minimum  -> the log-sum-exp shift
//   pass2: exp2, weighted sums, combine over the 4 lanes, log p and its gradient.
template <>
struct Target<CMCD_TARGET_MANY_GMM, 2> {
  static constexpr int kLdsHeader = 4;
  static constexpr int kFastMix = 40;  // register-resident fast path: n_mixes == 40
  struct State {
    float d2[10], dx[10], dy[10], dmin;
  };
  // `sub` in [0, LP): which share of the components this lane owns
  // LP = 16 does not divide 40: the last slot of lanes sub >= 8 is empty (d2 = +inf -> weight exp2(-inf) = 0)
  template <int LP>
  __device__ static __forceinline__ void pass1(const float (&z)[2], int sub, const float* tc, State& st) {
    constexpr int kQ = (kFastMix + LP - 1) / LP;
    constexpr bool kRagged = kFastMix % LP != 0;
    const int nm = __float_as_int(tc[2]);
    const float2* mu = reinterpret_cast<const float2*>(tc + kLdsHeader);
    float dmin = INFINITY;
    if (nm == kFastMix) {
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const bool has = !kRagged || sub + LP * q < kFastMix;
        const float2 mk = has ? mu[sub + LP * q] : float2{0.f, 0.f};
        st.dx[q] = z[0] - mk.x;
        st.dy[q] = z[1] - mk.y;
        st.d2[q] = has ? fmaf(st.dx[q], st.dx[q], st.dy[q] * st.dy[q]) : INFINITY;
        dmin = fminf(dmin, st.d2[q]);
      }
    } else {
      for (int k = sub; k < nm; k += LP) {
        const float2 mk = mu[k];
        const float dx = z[0] - mk.x, dy = z[1] - mk.y;
        dmin = fminf(dmin, fmaf(dx, dx, dy * dy));
      }
    }
    st.dmin = dmin;
  }
  // The cooperative kernel's target waves own the same components for the whole launch: their means stay in
  // registers (loaded once), so the distance pass has no LDS read on the per-bridge critical path.
  struct Means {
    float mx[10], my[10];
    bool fast;
    float inv_s, c2, c0;   // the header, for pass1f / pass2f
  };
  static constexpr bool kHasFast = true;
  __device__ static __forceinline__ bool is_fast(const Means& m) { return m.fast; }
  template <int LP>
  __device__ static __forceinline__ void load_means(int sub, const float* tc, Means& m) {
    constexpr int kQ = (kFastMix + LP - 1) / LP;
    const float2* mu = reinterpret_cast<const float2*>(tc + kLdsHeader);
    m.fast = __float_as_int(tc[2]) == kFastMix;
    m.inv_s = tc[0]; m.c2 = tc[1]; m.c0 = tc[3];
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      const float2 mk = (m.fast && sub + LP * q < kFastMix) ? mu[sub + LP * q] : float2{0.f, 0.f};
      m.mx[q] = mk.x;
      m.my[q] = mk.y;
    }
  }
  template <int LP>
  __device__ static __forceinline__ void pass1r(const float (&z)[2], int sub, const float* tc, const Means& m, State& st) {
    constexpr int kQ = (kFastMix + LP - 1) / LP;
    constexpr bool kRagged = kFastMix % LP != 0;
    if (!m.fast) { pass1<LP>(z, sub, tc, st); return; }
    float dmin = INFINITY;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      st.dx[q] = z[0] - m.mx[q];
      st.dy[q] = z[1] - m.my[q];
      const float d2 = fmaf(st.dx[q], st.dx[q], st.dy[q] * st.dy[q]);
      st.d2[q] = (!kRagged || sub + LP * q < kFastMix) ? d2 : INFINITY;
      dmin = fminf(dmin, st.d2[q]);
    }
    st.dmin = dmin;
  }
  // The cooperative kernel's target waves when n_mixes == 40 (decided once per launch): no per-bridge read of the
  // header, no mixture-size branches; the same arithmetic as pass1r / pass2 (min over non-negative distances taken on
  // the bit patterns: identical value).
  // EARLY (r04, the 2nd-order forward: 314 -> 306 us; the overdamped cooperative kernel measured 182.6 -> 183.8 with it and
  // keeps the r03 split): the exponents' scale and the reduction of the extremum move to the distance pass, so that the
  // exponential pass opens with the exponentials.  d2 then holds c2 d^2 <= 0 (sign bit set: the unsigned minimum of the bit
  // patterns is the value of smallest magnitude, i.e. the maximum).
  template <int LP, bool EARLY = false>
  __device__ static __forceinline__ void pass1f(const float (&z)[2], int sub, const Means& m, State& st) {
    constexpr int kQ = (kFastMix + LP - 1) / LP;
    constexpr bool kRagged = kFastMix % LP != 0;
    float dmin = INFINITY;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      st.dx[q] = z[0] - m.mx[q];
      st.dy[q] = z[1] - m.my[q];
      const float d2 = fmaf(st.dx[q], st.dx[q], st.dy[q] * st.dy[q]);
      st.d2[q] = (!kRagged || sub + LP * q < kFastMix) ? d2 : INFINITY;
      if constexpr (EARLY) {
        st.d2[q] *= m.c2;
        dmin = q == 0 ? st.d2[q] : __uint_as_float(min(__float_as_uint(dmin), __float_as_uint(st.d2[q])));
      } else {
        dmin = q == 0 ? st.d2[q] : fminf(dmin, st.d2[q]);
      }
    }
    if constexpr (EARLY) st.dmin = part_min_nonneg<LP>(dmin);
    else st.dmin = dmin;
  }
  template <int LP, bool EARLY = false>
  __device__ static __forceinline__ void pass2f(const float (&z)[2], int sub, const Means& m, const State& st,
                                                float& logp, float (&grad)[2]) {
    constexpr int kQ = (kFastMix + LP - 1) / LP;
    const float inv_s = m.inv_s, c2 = m.c2, c0 = m.c0;
    const float dmin = EARLY ? st.dmin : part_min_nonneg<LP>(st.dmin);
    float s = 0.f, sx = 0.f, sy = 0.f;
#pragma unroll
    for (int q = 0; q < kQ; ++q) {
      const float e = __builtin_amdgcn_exp2f(EARLY ? st.d2[q] - dmin : c2 * (st.d2[q] - dmin));
      s += e;
      sx = fmaf(e, st.dx[q], sx);
      sy = fmaf(e, st.dy[q], sy);
    }
    s = part_sum<LP>(s);
    sx = part_sum<LP>(sx);
    sy = part_sum<LP>(sy);
    const float lp = 0.69314718055994530942f * ((EARLY ? dmin + c0 : fmaf(c2, dmin, c0)) + __builtin_amdgcn_logf(s));
    const bool valid = lp > -1e4f;  // model_handler.py:279-280
    const float sc = -(inv_s * inv_s) * __builtin_amdgcn_rcpf(s);
    logp = valid ? lp : -INFINITY;
    grad[0] = valid ? sx * sc : 0.f;
    grad[1] = valid ? sy * sc : 0.f;
  }
  template <int LP>
  __device__ static __forceinline__ void pass2(const float (&z)[2], int sub, const float* tc, const State& st,
                                               float& logp, float (&grad)[2]) {
    constexpr int kQ = (kFastMix + LP - 1) / LP;
    const float inv_s = tc[0], c2 = tc[1], c0 = tc[3];
    const int nm = __float_as_int(tc[2]);
    const float2* mu = reinterpret_cast<const float2*>(tc + kLdsHeader);
    const float dmin = part_min_nonneg<LP>(st.dmin);
    float s = 0.f, sx = 0.f, sy = 0.f;
    if (nm == kFastMix) {
#pragma unroll
      for (int q = 0; q < kQ; ++q) {
        const float e = __builtin_amdgcn_exp2f(c2 * (st.d2[q] - dmin));
        s += e;
        sx = fmaf(e, st.dx[q], sx);
        sy = fmaf(e, st.dy[q], sy);
      }
    } else {
      for (int k = sub; k < nm; k += LP) {
        const float2 mk = mu[k];
        const float dx = z[0] - mk.x, dy = z[1] - mk.y;
        const float e = __builtin_amdgcn_exp2f(c2 * (fmaf(dx, dx, dy * dy) - dmin));
        s += e;
        sx = fmaf(e, dx, sx);
        sy = fmaf(e, dy, sy);
      }
    }
    s = part_sum<LP>(s);
    sx = part_sum<LP>(sx);
    sy = part_sum<LP>(sy);
    // log p = ln2 * (c0 + c2 dmin + log2 s)
    const float lp = 0.69314718055994530942f * (fmaf(c2, dmin, c0) + __builtin_amdgcn_logf(s));
    const bool valid = lp > -1e4f;  // model_handler.py:279-280
    const float sc = -(inv_s * inv_s) * __builtin_amdgcn_rcpf(s);
    logp = valid ? lp : -INFINITY;
    grad[0] = valid ? sx * sc : 0.f;
    grad[1] = valid ? sy * sc : 0.f;
  }
  __device__ static __forceinline__ void eval(const float (&z)[2], int g, const float* tc, float& logp,
                                              float (&grad)[2]) {
    State st;
    pass1<4>(z, g, tc, st);
    pass2<4>(z, g, tc, st, logp, grad);
  }
  // Second order (reparameterised gradient): with responsibilities r_k and d_k = z - mu_k,
  //   grad = -(1/s^2) sum r d,   Hessian = -I/s^2 + (1/s^4) (sum r d d^T - (sum r d)(sum r d)^T);  H = {h00, h01, h11}
  static constexpr int HN = 3;
  __device__ static __forceinline__ void eval_hess(const float (&z)[2], int g, const float* tc, float& logp,
                                                   float (&grad)[2], float (&H)[3]) {
    const float inv_s = tc[0], c2 = tc[1], c0 = tc[3];
    const int nm = __float_as_int(tc[2]);
    const float2* mu = reinterpret_cast<const float2*>(tc + kLdsHeader);
    float dmin = INFINITY;
    for (int k = g; k < nm; k += 4) {
      const float2 mk = mu[k];
      const float dx = z[0] - mk.x, dy = z[1] - mk.y;
      dmin = fminf(dmin, fmaf(dx, dx, dy * dy));
    }
    dmin = -group_max(-dmin);
    float s = 0.f, sx = 0.f, sy = 0.f, sxx = 0.f, sxy = 0.f, syy = 0.f;
    for (int k = g; k < nm; k += 4) {
      const float2 mk = mu[k];
      const float dx = z[0] - mk.x, dy = z[1] - mk.y;
      const float e = __builtin_amdgcn_exp2f(c2 * (fmaf(dx, dx, dy * dy) - dmin));
      s += e;
      sx = fmaf(e, dx, sx);
      sy = fmaf(e, dy, sy);
      sxx = fmaf(e * dx, dx, sxx);
      sxy = fmaf(e * dx, dy, sxy);
      syy = fmaf(e * dy, dy, syy);
    }
    s = group_sum(s); sx = group_sum(sx); sy = group_sum(sy);
    sxx = group_sum(sxx); sxy = group_sum(sxy); syy = group_sum(syy);
    const float lp = 0.69314718055994530942f * (fmaf(c2, dmin, c0) + __builtin_amdgcn_logf(s));
    const bool valid = lp > -1e4f;
    const float rs = 1.0f / s, i2 = inv_s * inv_s;
    const float mx = sx * rs, my = sy * rs;
    logp = valid ? lp : -INFINITY;
    grad[0] = valid ? -i2 * mx : 0.f;
    grad[1] = valid ? -i2 * my : 0.f;
    H[0] = valid ? fmaf(i2 * i2, fmaf(sxx, rs, -mx * mx), -i2) : 0.f;
    H[1] = valid ? i2 * i2 * fmaf(sxy, rs, -mx * my) : 0.f;
    H[2] = valid ? fmaf(i2 * i2, fmaf(syy, rs, -my * my), -i2) : 0.f;
  }
  __device__ static __forceinline__ void hvp(const float (&H)[3], const float (&)[2], const float (&v)[2], float (&hv)[2]) {
    hv[0] = H[0] * v[0] + H[1] * v[1];
    hv[1] = H[1] * v[0] + H[2] * v[1];
  }
};

// exp / log on the hardware transcendentals (v_exp_f32 / v_log_f32, 1 ulp): exp(x) = 2^(x log2 e), log(x) = ln 2 log2(x).
// The gmm target evaluates ten exponentials and three logarithms per evaluation; with the library routines (range
// reduction, ~8 - 10 instructions each) its two target waves were the long pole of that configuration's bridge.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(1.44269504088896340736f * x); }
__device__ __forceinline__ float log_fast(float x) { return 0.69314718055994530942f * __builtin_amdgcn_logf(x); }

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
    const float ea = exp_fast(la - m), eb = exp_fast(lb - m), ec = exp_fast(lc - m);
    const float s = ea + eb + ec;
    f = m + log_fast(s);
    const float is = -1.0f / s;
    gx = (ea * pa0 + eb * pb0 + ec * pc0) * is;
    gy = (ea * pa1 + eb * pb1 + ec * pc1) * is;
  }
  struct State {};
  struct Means {};
  template <int LP>
  __device__ static __forceinline__ void load_means(int, const float*, Means&) {}
  static constexpr bool kHasFast = false;
  __device__ static __forceinline__ bool is_fast(const Means&) { return false; }
  template <int LP, bool EARLY = false>
  __device__ static __forceinline__ void pass1f(const float (&)[2], int, const Means&, State&) {}
  template <int LP, bool EARLY = false>
  __device__ static __forceinline__ void pass2f(const float (&)[2], int, const Means&, const State&, float&, float (&)[2]) {}
  template <int LP>
  __device__ static __forceinline__ void pass1r(const float (&)[2], int, const float*, const Means&, State&) {}
  template <int LP>
  __device__ static __forceinline__ void pass1(const float (&)[2], int, const float*, State&) {}
  template <int LP>
  __device__ static __forceinline__ void pass2(const float (&z)[2], int g, const float* tc, const State&,
                                               float& logp, float (&grad)[2]) {
    eval(z, g, tc, logp, grad);
  }
  __device__ static __forceinline__ void eval(const float (&z)[2], int, const float*, float& logp,
                                              float (&grad)[2]) {
    float fa, gax, gay, fb, gbx, gby;
    raw(z[0], z[1], fa, gax, gay);
    raw(z[1], z[0], fb, gbx, gby);  // log_density(flip(x)), model_handler.py:192-195
    const float m = fmaxf(fa, fb);
    const float lse = m + log_fast(exp_fast(fa - m) + exp_fast(fb - m));
    logp = lse - 0.69314718055994530942f;
    const float wa = exp_fast(fa - lse), wb = exp_fast(fb - lse);
    grad[0] = wa * gax + wb * gby;  // un-flip the second gradient
    grad[1] = wa * gay + wb * gbx;
  }
  // Second order: per half  M = sum_k r_k (-P_k + q_k q_k^T), q_k = -P_k (x - mu_k);  Hessian = sum_halves w M - grad grad^T
  __device__ static __forceinline__ void raw2(float x, float y, float& f, float& gx, float& gy, float (&M)[3]) {
    constexpr float pa00 = 1.0f / 0.7f, pa11 = 20.0f;
    constexpr float pc00 = 10.256410256410257f, pc01 = -9.743589743589743f;
    constexpr float lca = -1.2602857463310935f, lcc = -1.7725379045882876f;
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
    const float ea = exp_fast(la - m), eb = exp_fast(lb - m), ec = exp_fast(lc - m);
    const float s = ea + eb + ec;
    f = m + log_fast(s);
    const float rs = 1.0f / s;
    const float ra = ea * rs, rb = eb * rs, rc = ec * rs;
    gx = -(ra * pa0 + rb * pb0 + rc * pc0);
    gy = -(ra * pa1 + rb * pb1 + rc * pc1);
    M[0] = ra * (pa0 * pa0 - pa00) + rb * (pb0 * pb0 - pa00) + rc * (pc0 * pc0 - pc00);
    M[1] = ra * (pa0 * pa1) + rb * (pb0 * pb1) + rc * (pc0 * pc1 - pc01);
    M[2] = ra * (pa1 * pa1 - pa11) + rb * (pb1 * pb1 - pa11) + rc * (pc1 * pc1 - pc00);
  }
  static constexpr int HN = 3;
  __device__ static __forceinline__ void eval_hess(const float (&z)[2], int, const float*, float& logp,
                                                   float (&grad)[2], float (&H)[3]) {
    float fa, gax, gay, fb, gbx, gby, Ma[3], Mb[3];
    raw2(z[0], z[1], fa, gax, gay, Ma);
    raw2(z[1], z[0], fb, gbx, gby, Mb);
    const float m = fmaxf(fa, fb);
    const float lse = m + log_fast(exp_fast(fa - m) + exp_fast(fb - m));
    logp = lse - 0.69314718055994530942f;
    const float wa = exp_fast(fa - lse), wb = exp_fast(fb - lse);
    grad[0] = wa * gax + wb * gby;
    grad[1] = wa * gay + wb * gbx;
    H[0] = wa * Ma[0] + wb * Mb[2] - grad[0] * grad[0];
    H[1] = wa * Ma[1] + wb * Mb[1] - grad[0] * grad[1];
    H[2] = wa * Ma[2] + wb * Mb[0] - grad[1] * grad[1];
  }
  __device__ static __forceinline__ void hvp(const float (&H)[3], const float (&)[2], const float (&v)[2], float (&hv)[2]) {
    hv[0] = H[0] * v[0] + H[1] * v[1];
    hv[1] = H[1] * v[0] + H[2] * v[1];
  }
};

// funnel: /root/reference/src/model_handler.py:124-143 (scale of v hard-coded 3.0).
template <int D>
struct Target<CMCD_TARGET_FUNNEL, D> {
  static constexpr int kLdsHeader = 0;
  struct State {};
  struct Means {};
  template <int LP>
  __device__ static __forceinline__ void load_means(int, const float*, Means&) {}
  static constexpr bool kHasFast = false;
  __device__ static __forceinline__ bool is_fast(const Means&) { return false; }
  template <int LP, bool EARLY = false>
  __device__ static __forceinline__ void pass1f(const float (&)[D], int, const Means&, State&) {}
  template <int LP, bool EARLY = false>
  __device__ static __forceinline__ void pass2f(const float (&)[D], int, const Means&, const State&, float&, float (&)[D]) {}
  template <int LP>
  __device__ static __forceinline__ void pass1r(const float (&)[D], int, const float*, const Means&, State&) {}
  template <int LP>
  __device__ static __forceinline__ void pass1(const float (&)[D], int, const float*, State&) {}
  template <int LP>
  __device__ static __forceinline__ void pass2(const float (&z)[D], int g, const float* tc, const State&,
                                               float& logp, float (&grad)[D]) {
    eval(z, g, tc, logp, grad);
  }
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
  // Second order: d2/dv2 = -1/9 - e^{-v} ss / 2,  d2/dv dz_j = z_j e^{-v},  d2/dz_j2 = -e^{-v};  H = {e^{-v}, ss}
  static constexpr int HN = 2;
  __device__ static __forceinline__ void eval_hess(const float (&z)[D], int g, const float* tc, float& logp,
                                                   float (&grad)[D], float (&H)[2]) {
    eval(z, g, tc, logp, grad);
    float ss = 0.f;
#pragma unroll
    for (int j = 1; j < D; ++j) ss = fmaf(z[j], z[j], ss);
    H[0] = expf(-z[0]);
    H[1] = ss;
  }
  __device__ static __forceinline__ void hvp(const float (&H)[2], const float (&z)[D], const float (&v)[D], float (&hv)[D]) {
    const float emv = H[0];
    float zv = 0.f;
#pragma unroll
    for (int j = 1; j < D; ++j) zv = fmaf(z[j], v[j], zv);
    hv[0] = (-1.0f / 9.0f - 0.5f * emv * H[1]) * v[0] + emv * zv;
#pragma unroll
    for (int j = 1; j < D; ++j) hv[j] = emv * (z[j] * v[0] - v[j]);
  }
};

// The merge of the per-workgroup statistics records {count, sum, sum of squares, max, sum exp(. - max)} by ONE wave, in the
// order of finalize_kernel (cmcd_kernels.hip) — 256 virtual threads, each a contiguous chunk of records, then the halving
// tree — so that the fused form returns the same five doubles bit for bit.  Lane l plays virtual threads l, l + 64, l + 128,
// l + 192.  `partials` is read with agent-scope loads: the records come from other workgroups of the same launch.
__device__ __forceinline__ void wave_merge_stats(const double* partials, int n_rec, double* out, int lane) {
  auto ld = [&](int64_t i) { return __hip_atomic_load(partials + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  const int per = (n_rec + 255) / 256;
  double mv[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = lane + 64 * q, lo = t * per, hi = min(n_rec, lo + per);
    double m = -INFINITY;
    for (int i = lo; i < hi; ++i) m = fmax(m, ld((int64_t)i * CMCD_NSTATS + 3));
    mv[q] = m;
  }
  double m = fmax(fmax(mv[0], mv[2]), fmax(mv[1], mv[3]));     // max is order-free
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_xor(m, o));
  const double M = m;
  double acc[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int t = lane + 64 * q, lo = t * per, hi = min(n_rec, lo + per);
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[q][k] = 0.0;
    for (int i = lo; i < hi; ++i) {
      const int64_t b = (int64_t)i * CMCD_NSTATS;
      const double p0 = ld(b), p1 = ld(b + 1), p2 = ld(b + 2), p3 = ld(b + 3), p4 = ld(b + 4);
      acc[q][0] += p0;
      acc[q][1] += p1;
      acc[q][2] += p2;
      acc[q][3] += (p3 > -INFINITY && M < INFINITY) ? p4 * exp(p3 - M) : (p3 == M ? p4 : 0.0);
    }
  }
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // tree steps s = 128 (t += t + 128), s = 64 (t += t + 64): inside the lane; s = 32 .. 1: t += t + s across lanes
    const double a0 = acc[0][k] + acc[2][k], a1 = acc[1][k] + acc[3][k];
    double x = a0 + a1;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double y = __shfl_down(x, o);
      x = lane < o ? x + y : x;
    }
    v[k] = x;
  }
  if (lane == 0) {
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = M; out[4] = v[3];
  }
}

}  // namespace cmcd
