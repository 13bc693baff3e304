// MCD_CAIS_UHA_sn — 2nd-order (underdamped) CMCD on gfx950: the trajectory kernel.
//
// Reference: /root/reference/src/mcd_under_lp_a_cais.py:6-115 (`evolve_underdamped_lp_a_cais`) reached through
// /root/reference/src/mcd_utils.py:174-188 from compute_log_elbo (/root/reference/src/mcdboundingmachine.py:126-179);
// network built with rho_dim = dim (/root/reference/src/mcdboundingmachine.py:82-98, src/nn.py:42-43,
// src/nn_dds.py:55-56).  Per particle the state is (z, rho); bridge i does
//
//   uf   = -(beta_i clip(grad log p(z), +-1e2) + (1 - beta_i) grad log q(z))            :23-30,46
//   eps  = eps0 cos^2(((i / K + 0.008) / 1.008) pi / 2),  eta = gamma eps                :33-40,48,50
//   m_f  = rho (1 - eta) - 2 eta s([z; rho], i)                                          :51-54
//   rho' = m_f + sqrt(2 eta) n_i                                                         :56-59
//   rho''= rho' - eps uf / 2;  z' = z + eps rho'';  rho_new = rho'' - eps ub(z') / 2     :62-67  (leap-frog)
//   m_b  = rho' (1 - eta) + 2 eta s([z; rho'], i)                                        :77-80  (old z, same index i)
//   w   += log N(rho; m_b, sqrt(2 eta)) - log N(rho'; m_f, sqrt(2 eta))                  :83-88
//
// with w_0 = -log q(z_0) - log N(rho_0; 0, 1), rho_0 ~ N(0, I) (:92-97) and the closing terms log N(rho_K; 0, 1)
// (:112) + log p(z_K) (mcdboundingmachine.py:178).  The function body fixes the cos^2 schedule and the 1e2 clip (it has
// no eps_schedule / grad_clipping arguments); cmcd_desc.eps_schedule / grad_clipping are ignored for this mode.
//
// Mapping = the wave-per-tile trajectory kernel's (cmcd_kernels.hip): one wave owns 16 particles for all K bridges,
// lane (g, c) holds particle c and the hidden units {16 t + 4 g + r}; layer 2 on v_mfma_f32_16x16x4_f32 with the
// A fragments streamed from LDS.  What differs from the overdamped kernel:
//   * two network evaluations per bridge that cannot be shared with the next bridge (different momentum, same z and
//     same index): the first layer's z / time part (bias row + z W1[:d]) is formed once per bridge, each evaluation
//     adds its momentum's W1[d:2d] part;
//   * ONE target-gradient evaluation per bridge: grad log p(z') closes bridge i (ub) and opens bridge i + 1 (uf);
//   * the key chain has one more split and one more `normal` in front of the loop (the initial momentum).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

// d standard normals from key (ka, kb): jax.random.normal(key, (D,)) — block j encrypts (j, Hh + j); the blocks are dealt
// to the four rows of the wave.  `stage`: row of the capture buffers (tests), or -1.
template <int D>
__device__ __forceinline__ void draw_normal(uint32_t ka, uint32_t kb, int g, float (&nz)[2 * ((D + 1) / 2)],
                                            const TrajArgs& a, int64_t stage, int64_t p, bool valid) {
  constexpr int Hh = (D + 1) / 2;
#pragma unroll
  for (int j0 = 0; j0 < Hh; j0 += 4) {
    const int j = j0 + g;
    uint32_t y0 = j, y1 = (Hh + j < D) ? Hh + j : 0;
    threefry2x32(ka, kb, y0, y1);
    if (a.dbg_bits && valid && j < Hh) {
      const int64_t o = (stage * a.n + p) * D;
      a.dbg_bits[o + j] = y0;
      a.dbg_noise[o + j] = bits_to_normal(y0);
      if (Hh + j < D) {
        a.dbg_bits[o + Hh + j] = y1;
        a.dbg_noise[o + Hh + j] = bits_to_normal(y1);
      }
    }
    uint32_t r0[4], r1[4];
    rows0123(__float_as_uint(bits_to_normal(y0)), r0);
    rows0123(__float_as_uint(bits_to_normal(y1)), r1);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (j0 + q < Hh) {
        nz[j0 + q] = __uint_as_float(r0[q]);
        nz[Hh + j0 + q] = __uint_as_float(r1[q]);
      }
  }
}

// first-layer pre-activation shared by the two evaluations of a bridge: bias row (time path folded by the prep
// launch) + z W1[:d]
template <int D, int T>
__device__ __forceinline__ void net_pre_z(const float (&z)[D], const float* __restrict__ brow, const float* lds_w1z,
                                          int g, f32x4 (&pre)[T]) {
  constexpr int HP = 16 * T;
  asm volatile("" ::: "memory");  // keep the LDS-resident weights streaming (no LICM into VGPRs)
#pragma unroll
  for (int t = 0; t < T; ++t) {
    pre[t] = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
    for (int j = 0; j < D; ++j) pre[t] += z[j] * *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
  }
}

// s([z; rho], i) for the 16 particles of this wave, given the shared part of the first layer.
//   dds     (nn_dds.py:159-162): h1 = gelu(W1^T [z; rho; tau] + b1); h2 = gelu(W2^T h1 + b2); clip(W3^T h2 + b3, +-1e4)
//   geffner (nn.py:45-52,66-70): u = [z; rho; emb]; u += softplus(u W1 + b1); u += softplus(u W2 + b2); factor (u W3 + b3)
template <int ARCH, int D, int T>
__device__ __forceinline__ void net_eval_rho(const f32x4 (&prez)[T], const float (&z)[D], const float (&rho)[D],
                                             const float* __restrict__ urow, const float* lds_w2,
                                             const float* lds_w1z, const float* lds_b2, const float* lds_w3t,
                                             const float* lds_b3, int lane, float (&s)[D]) {
  constexpr int HP = 16 * T;
  const int g = lane >> 4;
  asm volatile("" ::: "memory");
  f32x4 h[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f32x4 pre = prez[t];
#pragma unroll
    for (int j = 0; j < D; ++j) pre += rho[j] * *reinterpret_cast<const f32x4*>(lds_w1z + (D + j) * HP + 16 * t + 4 * g);
    if (ARCH == CMCD_ARCH_DDS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = gelu_fast(pre[r]);
    } else {
      f32x4 u = *reinterpret_cast<const f32x4*>(urow + 16 * t + 4 * g);
      if (16 * t < 2 * D) {  // the first 2 D entries of u are [z; rho] themselves
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nidx = 16 * t + 4 * g + r;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            if (j >= 16 * t && j < 16 * t + 16) u[r] = (nidx == j) ? z[j] : u[r];
            if (D + j >= 16 * t && D + j < 16 * t + 16) u[r] = (nidx == D + j) ? rho[j] : u[r];
          }
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = u[r] + softplus(pre[r]);
    }
  }
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = *reinterpret_cast<const f32x4*>(lds_b2 + 16 * t + 4 * g);
#pragma unroll
  for (int ti = 0; ti < T; ++ti) {
    asm volatile("" ::: "memory");
    f32x4 af[T];
#pragma unroll
    for (int to = 0; to < T; ++to) af[to] = *reinterpret_cast<const f32x4*>(lds_w2 + ((ti * T + to) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int to = 0; to < T; ++to) acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[to][r], h[ti][r], acc[to], 0, 0, 0);
    }
  }
  float part[D];
#pragma unroll
  for (int j = 0; j < D; ++j) part[j] = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f32x4 h2;
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[r] = (ARCH == CMCD_ARCH_DDS) ? gelu_fast(acc[t][r]) : h[t][r] + softplus(acc[t][r]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(lds_w3t + j * HP + 16 * t + 4 * g);
      part[j] += h2[0] * wv[0] + h2[1] * wv[1] + h2[2] * wv[2] + h2[3] * wv[3];
    }
  }
  const float factor = lds_b3[15];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float o = group_sum(part[j]) + lds_b3[j];
    s[j] = (ARCH == CMCD_ARCH_DDS) ? fminf(fmaxf(o, -1e4f), 1e4f) : o * factor;
  }
}

// Kept trajectory (a.traj, for the reverse sweep): rows [0, K] = z_0..z_K, rows [K+1, 2K+1] = rho_0..rho_K,
// rows [2K+2, 3K+1] = rho'_0..rho'_{K-1}; each row [n][D].
template <int TARGET, int ARCH, int D, int T>
__global__ __launch_bounds__(512, (T > 4 || D > 4) ? 2 : 4) void uha_traj_kernel(TrajArgs a) {
  constexpr int HP = 16 * T;
  constexpr int Hh = (D + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_w2 = lds;                     // HP*HP
  float* lds_w1z = lds_w2 + HP * HP;       // 2D*HP   rows [0, D) = z part, [D, 2D) = rho part
  float* lds_w3t = lds_w1z + 2 * D * HP;   // D*HP
  float* lds_b2 = lds_w3t + D * HP;        // HP
  float* lds_b3 = lds_b2 + HP;             // 16
  float* lds_tgt = lds_b3 + 16;            // tgt_floats
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.ws + a.w.w1z);
    f32x4* dst = reinterpret_cast<f32x4*>(lds_w1z);
    for (int i = threadIdx.x; i < 2 * D * HP / 4; i += blockDim.x) dst[i] = src[i];
    src = reinterpret_cast<const f32x4*>(a.ws + a.w.w2);
    dst = reinterpret_cast<f32x4*>(lds_w2);
    for (int i = threadIdx.x; i < HP * HP / 4; i += blockDim.x) dst[i] = src[i];
    src = reinterpret_cast<const f32x4*>(a.ws + a.w.w3t);
    dst = reinterpret_cast<f32x4*>(lds_w3t);
    for (int i = threadIdx.x; i < D * HP / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i < HP; i += blockDim.x) lds_b2[i] = a.ws[a.w.b2 + i];
    for (int i = threadIdx.x; i < 16; i += blockDim.x) lds_b3[i] = a.ws[a.w.b3 + i];
    for (int i = threadIdx.x; i < a.w.tgt_floats; i += blockDim.x) lds_tgt[i] = a.ws[a.w.tgt + i];
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wave * 16 >= a.n) return;  // whole wave out of range (after the only barrier)
  const int64_t p = wave * 16 + c;
  const bool valid = p < a.n;
  const int32_t seed = a.seeds[valid ? p : a.n - 1];
  const int K = a.K;
  const bool keep = a.traj && valid && g == 0;
  float* tz = a.traj;
  float* trho = a.traj ? a.traj + (int64_t)(K + 1) * a.n * D : nullptr;
  float* trhop = a.traj ? a.traj + (int64_t)(2 * K + 2) * a.n * D : nullptr;

  float qmean[D], qstd[D], qiv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qmean[j] = a.params[a.lay.vd_mean + j];
    qstd[j] = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (qstd[j] * qstd[j]);
  }
  const float gamma = a.params[a.lay.gamma];

  // ---- key chain (mcdboundingmachine.py:151-162; mcd_under_lp_a_cais.py:92-93,100): lane g computes block (g & 1) of a split
  const int gb = g & 1;
  uint32_t x0, x1, k0 = 0u, k1 = (uint32_t)seed;  // PRNGKey(seed) = (0, seed)
  float z[D], rho[D];
  {
    x0 = gb; x1 = 2 + gb;
    threefry2x32(k0, k1, x0, x1);  // split(PRNGKey(seed)) -> A = (out0, out1), B = (out2, out3)
    uint32_t a0, a1, b0, b1;
    rows01(x0, a0, a1);
    rows01(x1, b0, b1);
    float nz[2 * Hh];
    draw_normal<D>(a0, a1, g, nz, a, 0, p, valid);   // z0 = mean + std * normal(A)      diag_gauss.py:49-62
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = qstd[j] * nz[j] + qmean[j];
    x0 = gb; x1 = 2 + gb;
    threefry2x32(b0, b1, x0, x1);  // C = first(split(B)): the key handed to evolve
    uint32_t c0, c1;
    rows01(x0, c0, c1);
    x0 = gb; x1 = 2 + gb;
    threefry2x32(c0, c1, x0, x1);  // (R, G') = split(C)                                mcd_under_lp_a_cais.py:92
    uint32_t r0, r1, p0, p1;
    rows01(x0, r0, r1);
    rows01(x1, p0, p1);
    draw_normal<D>(r0, r1, g, nz, a, 1, p, valid);   // rho_0 = normal(R, (d,))             :93
#pragma unroll
    for (int j = 0; j < D; ++j) rho[j] = nz[j];
    x0 = gb; x1 = 2 + gb;
    threefry2x32(p0, p1, x0, x1);  // gen_0 = second(split(G'))                         :100
    rows01(x1, k0, k1);
    if (a.dbg_keys && valid && g == 0) {
      a.dbg_keys[p * 2] = k0;
      a.dbg_keys[p * 2 + 1] = k1;
    }
  }
  if (keep) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      tz[p * D + j] = z[j];
      trho[p * D + j] = rho[j];
    }
  }

  // w = -log q(z0) - log N(rho_0; 0, 1)                  mcdboundingmachine.py:157, mcd_under_lp_a_cais.py:96-97
  float w = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float dz = z[j] - qmean[j];
    w -= -(dz * dz) / (2.0f * qstd[j] * qstd[j]) - logf(qstd[j]) - kHalfLog2Pi;
  }
  {
    float l0 = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) l0 += -(rho[j] * rho[j]) * 0.5f - kHalfLog2Pi;
    w -= l0;
  }

  const float* bias1 = a.ws + a.w.bias1;
  const float* utab = a.ws + a.w.utab;
  constexpr float clipv = 1e2f;   // gradU(z, beta, clip=1e2), stable=True      mcd_under_lp_a_cais.py:23-30,42,46

  float gp[D], gq[D], logp;
  Target<TARGET, D>::eval(z, g, lds_tgt, logp, gp);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
    gq[j] = -(z[j] - qmean[j]) * qiv[j];
  }

  for (int i = 0; i < K; ++i) {
    const float beta = a.ws[a.w.beta + i], eps = a.ws[a.w.eps + i];
    const float eta = gamma * eps;                    // :50
    const float sig = sqrtf(2.0f * eta);              // :56
    const float inv2s2 = 1.0f / (2.0f * sig * sig), cst = logf(sig) + kHalfLog2Pi;
    const float ome = 1.0f - eta;

    // ---- noise: (G, H) = split(gen); n_i = normal(G, (d,)); gen = second(split(H))       :55,84
    float nz[2 * Hh];
    {
      x0 = gb; x1 = 2 + gb;
      threefry2x32(k0, k1, x0, x1);
      uint32_t g0, g1, h0, h1;
      rows01(x0, g0, g1);
      rows01(x1, h0, h1);
      x0 = gb; x1 = 2 + gb;
      threefry2x32(h0, h1, x0, x1);
      rows01(x1, k0, k1);
      if (a.dbg_keys && valid && g == 0) {
        a.dbg_keys[((int64_t)(i + 1) * a.n + p) * 2] = k0;
        a.dbg_keys[((int64_t)(i + 1) * a.n + p) * 2 + 1] = k1;
      }
      draw_normal<D>(g0, g1, g, nz, a, i + 2, p, valid);
    }

    f32x4 prez[T];
    net_pre_z<D, T>(z, bias1 + (int64_t)i * HP, lds_w1z, g, prez);
    float s1[D], s2[D], rhop[D];
    net_eval_rho<ARCH, D, T>(prez, z, rho, utab + (int64_t)i * HP, lds_w2, lds_w1z, lds_b2, lds_w3t, lds_b3, lane, s1);
    float fk_lp = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float mf = rho[j] * ome - 2.0f * eta * s1[j];           // :52-54
      rhop[j] = mf + sig * nz[j];                                   // :58-59 sample_kernel
      const float df = rhop[j] - mf;
      fk_lp += -(df * df) * inv2s2 - cst;                           // log_prob_kernel(rho', m_f, scale)   :83
    }
    net_eval_rho<ARCH, D, T>(prez, z, rhop, utab + (int64_t)i * HP, lds_w2, lds_w1z, lds_b2, lds_w3t, lds_b3, lane, s2);
    float bk_lp = 0.f, rpp[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float mb = rhop[j] * ome + 2.0f * eta * s2[j];          // :77-80
      const float db = rho[j] - mb;
      bk_lp += -(db * db) * inv2s2 - cst;                           // log_prob_kernel(rho, m_b, scale)    :84
      const float uf = -1.0f * (beta * gp[j] + (1.0f - beta) * gq[j]);
      rpp[j] = rhop[j] - eps * uf / 2.0f;                           // :62
      z[j] = z[j] + eps * rpp[j];                                   // :63
    }
    w += bk_lp - fk_lp;                                             // :88
    Target<TARGET, D>::eval(z, g, lds_tgt, logp, gp);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
      gq[j] = -(z[j] - qmean[j]) * qiv[j];
      const float ub = -1.0f * (beta * gp[j] + (1.0f - beta) * gq[j]);   // :65 (same beta_i)
      rho[j] = rpp[j] - eps * ub / 2.0f;                            // :67
    }
    if (keep) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        tz[((int64_t)(i + 1) * a.n + p) * D + j] = z[j];
        trho[((int64_t)(i + 1) * a.n + p) * D + j] = rho[j];
        trhop[((int64_t)i * a.n + p) * D + j] = rhop[j];
      }
    }
  }
  {
    float lK = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) lK += -(rho[j] * rho[j]) * 0.5f - kHalfLog2Pi;
    w += lK;     // + log N(rho_K; 0, 1)   :112
  }
  w += logp;     // + log p(z_K)           mcdboundingmachine.py:178
  const float loss = -w;

  if (valid && g == 0) {
    a.out_loss[p] = loss;
#pragma unroll
    for (int j = 0; j < D; ++j) a.out_z[p * D + j] = z[j];
  }

  // ---- per-wave statistics over lanes 0..15 (g == 0), fixed butterfly order -> deterministic
  const bool use = valid && g == 0;
  double cnt = (use && isfinite(loss)) ? 1.0 : 0.0;
  double sm = use ? (double)loss : 0.0;
  double sq = use ? (double)loss * (double)loss : 0.0;
  double mx = use ? -(double)loss : -INFINITY;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    cnt += __shfl_xor(cnt, o);
    sm += __shfl_xor(sm, o);
    sq += __shfl_xor(sq, o);
    mx = fmax(mx, __shfl_xor(mx, o));
  }
  double ex = (use && mx > -INFINITY && mx < INFINITY) ? exp(-(double)loss - mx) : 0.0;
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) ex += __shfl_xor(ex, o);
  if (lane == 0) {
    double* o = a.partials + wave * CMCD_NSTATS;
    o[0] = cnt; o[1] = sm; o[2] = sq; o[3] = mx; o[4] = ex;
  }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
typedef void (*uha_fn)(TrajArgs);

template <int TARGET, int ARCH, int D>
static uha_fn uha_pick_T(int T) {
  switch (T) {
    case 2: return uha_traj_kernel<TARGET, ARCH, D, 2>;
    case 4: return uha_traj_kernel<TARGET, ARCH, D, 4>;
    case 5: return uha_traj_kernel<TARGET, ARCH, D, 5>;
    case 9: return uha_traj_kernel<TARGET, ARCH, D, 9>;
    default: return nullptr;
  }
}

static uha_fn uha_pick(const cmcd_desc& d, int T) {
  if (d.arch == CMCD_ARCH_DDS) {
    if (T != 4) return nullptr;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return uha_traj_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2) return uha_traj_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return uha_traj_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_DDS, 10, 4>;
    return nullptr;
  }
  if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return uha_pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2>(T);
  if (d.target == CMCD_TARGET_GMM && d.dim == 2) return uha_pick_T<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2>(T);
  if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return uha_pick_T<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10>(T);
  return nullptr;
}

bool uha_available(const cmcd_desc& d, int T) { return uha_pick(d, T) != nullptr; }

int64_t uha_traj_floats(const cmcd_desc& d, int64_t n) { return (int64_t)(3 * d.nbridges + 2) * n * d.dim; }

int uha_forward_launch(const cmcd_desc& d, const TrajArgs& ta, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const WsLayout& w = ta.w;
  uha_fn fn = uha_pick(d, w.T);
  if (!fn) return CMCD_ERR_UNSUPPORTED;
  const int64_t tiles = w.n_waves;
  const size_t lds_bytes = size_t(w.HP * w.HP + 3 * d.dim * w.HP + w.HP + 16 + w.tgt_floats) * 4;
  if (lds_bytes > 160 * 1024) return CMCD_ERR_UNSUPPORTED;
  // waves per workgroup as in the overdamped wave-per-tile kernel: one wave per workgroup until every SIMD has one
  const int64_t per_cu = (160 * 1024) / (int64_t)lds_bytes;
  int nw = tiles <= 1024 ? 1 : (tiles <= 8192 ? 4 : 8);
  if (per_cu < 2 && tiles > 256) nw = tiles <= 1024 ? 4 : 8;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_bytes) != hipSuccess)
    return CMCD_ERR_HIP;
  const unsigned blocks = unsigned((tiles + nw - 1) / nw);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * nw), lds_bytes, stream, ta);
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

}  // namespace cmcd
