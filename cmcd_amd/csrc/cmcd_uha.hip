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

#include <map>
#include <mutex>
#include <utility>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

// The opt-in for dynamic LDS once per (function, device) and size, not on every call (the work-item gradient path launches
// three kernels with opted-in LDS per gradient; the attribute call costs host microseconds each time).
static bool ensure_dynamic_lds(const void* fn, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> raised;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  std::lock_guard<std::mutex> lock(mu);
  auto& have = raised[std::make_pair(fn, dev)];
  if (have >= bytes) return true;
  if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  have = bytes;
  return true;
}

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
    {   // r05: the per-bridge tables into this XCD's L2, one touch per 128-byte line (cmcd_coop.hip: behind the prep launch every XCD's copy is gone)
      float warm = 0.f;
      const int64_t wt0 = a.w.sched;
      const int64_t wt1 = (ARCH == CMCD_ARCH_GEFFNER ? a.w.utab : a.w.bias1) + (int64_t)(a.K + 1) * (16 * T);
      const int64_t per_xcd = (gridDim.x + 7) >> 3, rank = blockIdx.x >> 3;     // dealt to the workgroups that share an XCD
      for (int64_t wi = wt0 + 32 * (rank * blockDim.x + threadIdx.x); wi < wt1; wi += 32 * per_xcd * blockDim.x) warm += a.ws[wi];
      asm volatile("" ::"v"(warm));
    }
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

// Diagnostic build only (-DCMCD_STAMPS, tools/probes/uha_stamp_probe.py): per-wave cycle totals of each phase of
// workgroup 0 of uha_coop_kernel.  Never compiled into the product.
#ifdef CMCD_STAMPS
__device__ unsigned long long g_uha_stamps[16][16];
#define USTAMP(slot)                                                                     \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    st_acc[slot] += t_ - st_last;                                                        \
    st_last = t_;                                                                        \
  } while (0)
#else
#define USTAMP(slot)
#endif

// workgroup barrier that publishes LDS writes but does not drain outstanding global loads / stores (the next bridge's
// bias row is in flight across the first barrier of every bridge; __syncthreads() would wait for it there)
__device__ __forceinline__ void uha_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ------------------------------------------------------------------------------------------
// uha_coop_kernel — the CU-cooperative form for batches that cannot fill the chip with one wave per tile (the named
// shape: 125 tiles on a 256-long chain).  One workgroup of T + 2 waves per 16-particle tile:
//   waves 0..T-1  MLP: wave v owns hidden units 16 v .. 16 v + 15 — its W1 rows, its W2 A-fragments (resident in VGPRs for
//                 the whole launch), its slice of the layer-3 dot product;
//   wave T        STATE + TGT: (z, rho, w) of the 16 particles, the leap-frog, both kernels' log-densities, grad log p(z')
//                 (distance pass / exponential pass in the two intervals of the bridge's SECOND network evaluation);
//   wave T + 1    RNG: jax's key chain one bridge ahead, bits -> deviates.
// A bridge is two passes of the MLP waves — s([z; rho], i), then s([z; rho'], i) — each: layer 1 + activation -> LDS (MFMA B
// layout) | barrier | layer 2 on the matrix cores from LDS, activation, layer-3 partials -> LDS | barrier | the state wave
// combines the partials and publishes the next network input | barrier.
// Every role runs its OWN loop with the same barrier sequence (raw s_barrier counts arrivals, not program counters): as
// branches of one shared loop body (r03 first form) every role's loop-carried registers were live in every wave — the
// d = 10 instances spilled 270 - 990 bytes per lane and ran 21 000 cycles per bridge (tools/probes/uha_stamp_probe.py).
// The key chain of bridge i + 1 is laid over the bridge's intervals so that no auxiliary interval is longer than the MLP
// waves' own:  pass 0, matrix interval: (G, H) = split(gen) | pass 1, matrix interval: ONE Threefry pass whose rows 0, 1
// encrypt split(H) (-> gen') and rows 2, 3 the first two blocks of normal(G) | next bridge's first interval: bits ->
// deviates; for d > 4 the remaining blocks of normal(G) ride in the two intervals in between, split at a key injection.
// Same arithmetic as uha_traj_kernel up to the summation order of the layer-3 partials (per wave, then over waves).
// ------------------------------------------------------------------------------------------
// Threefry-2x32 in two resumable parts (8 + 12 rounds): the same 20 rounds and key injections as threefry2x32
// (cmcd_device.h), cut at the second key injection.
struct TfState { uint32_t x0, x1, k0, k1, k2; };
#define UHA_TF_ROUND(r) \
  t.x0 += t.x1;         \
  t.x1 = rotl32(t.x1, r); \
  t.x1 ^= t.x0;
__device__ __forceinline__ void tf_part1(TfState& t, uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) {
  t.k0 = k0; t.k1 = k1; t.k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  t.x0 = c0 + k0; t.x1 = c1 + k1;
  UHA_TF_ROUND(13) UHA_TF_ROUND(15) UHA_TF_ROUND(26) UHA_TF_ROUND(6)
  t.x0 += t.k1; t.x1 += t.k2 + 1u;
  UHA_TF_ROUND(17) UHA_TF_ROUND(29) UHA_TF_ROUND(16) UHA_TF_ROUND(24)
  t.x0 += t.k2; t.x1 += t.k0 + 2u;
}
__device__ __forceinline__ void tf_part2(TfState& t) {
  UHA_TF_ROUND(13) UHA_TF_ROUND(15) UHA_TF_ROUND(26) UHA_TF_ROUND(6)
  t.x0 += t.k0; t.x1 += t.k1 + 3u;
  UHA_TF_ROUND(17) UHA_TF_ROUND(29) UHA_TF_ROUND(16) UHA_TF_ROUND(24)
  t.x0 += t.k1; t.x1 += t.k2 + 4u;
  UHA_TF_ROUND(13) UHA_TF_ROUND(15) UHA_TF_ROUND(26) UHA_TF_ROUND(6)
  t.x0 += t.k2; t.x1 += t.k0 + 5u;
}
#undef UHA_TF_ROUND

// The launch's first draws on the columns layout of the state wave (g = row, c = column): (A, B) = split(PRNGKey(seed)),
// z_0 = mean + std normal(A), then the two further splits that give rho_0 = normal(R) and the chain's first key gen_0
// (mcd_under_lp_a_cais.py: the prologue of the bound) — every lane ends with all d coordinates of its particle.
template <int D>
__device__ __forceinline__ void uha_first_draws(const TrajArgs& a, int64_t p, bool valid, int g, int gb,
                                                const float (&qmean)[D], const float (&qstd)[D], float (&z)[D],
                                                float (&rho)[D], uint32_t& k0, uint32_t& k1) {
  constexpr int Hh = (D + 1) / 2;
  const int32_t seed = a.seeds[valid ? p : a.n - 1];
  k0 = 0u; k1 = (uint32_t)seed;
  uint32_t x0 = gb, x1 = 2 + gb;
  threefry2x32(k0, k1, x0, x1);
  uint32_t a0, a1, b0, b1;
  rows01(x0, a0, a1);
  rows01(x1, b0, b1);
  float nz[2 * Hh];
  draw_normal<D>(a0, a1, g, nz, a, 0, p, valid);
#pragma unroll
  for (int j = 0; j < D; ++j) z[j] = qstd[j] * nz[j] + qmean[j];
  x0 = gb; x1 = 2 + gb;
  threefry2x32(b0, b1, x0, x1);
  uint32_t c0, c1;
  rows01(x0, c0, c1);
  x0 = gb; x1 = 2 + gb;
  threefry2x32(c0, c1, x0, x1);
  uint32_t r0, r1, p0, p1;
  rows01(x0, r0, r1);
  rows01(x1, p0, p1);
  draw_normal<D>(r0, r1, g, nz, a, 1, p, valid);
#pragma unroll
  for (int j = 0; j < D; ++j) rho[j] = nz[j];
  x0 = gb; x1 = 2 + gb;
  threefry2x32(p0, p1, x0, x1);
  rows01(x1, k0, k1);
}

// waves of one uha_coop_kernel workgroup: T MLP + state + RNG; the funnel on 8-particle tiles deals its state over TWO waves
// (16 lanes per particle, lane = coordinate)
constexpr bool uha_coop_dealt(int target, bool half) { return half && target == CMCD_TARGET_FUNNEL; }
constexpr int uha_coop_waves(int target, int T, bool half, bool tail = false) {
  return T + 2 + (uha_coop_dealt(target, half) ? 1 : 0);      // (tail: T - 1 MLP waves + the tail wave)
}
// TAIL (r05): the last tile of the padded width holds at most 4 real neurons (the funnel's geffner net: 2 x 10 + 48 = 68 of
// 80) — a whole MLP wave for them doubled up with another MLP wave on one SIMD and was the pole of both matrix intervals
// (profiles/r05_uha_funnel_stamps_dealt.txt: 1 728 / 1 932 cycles against 1 140 / 1 240 for a wave with a SIMD of its own).
// With TAIL the workgroup runs T - 1 MLP waves and a light TAIL wave for the 4 neurons: ONE more 4x4x1 pass whose 16 blocks
// are 2 particle groups x 8 contraction slices (12 steps instead of 40), a reduce-scatter over the slices, one activation
// per lane; W3 of the 4 neurons is applied by the state waves.  (First form: the tail on the last MLP wave itself, 0.127 ms;
// on its own wave with its own layer-3 reduce-scatter 0.123; as below 0.117 — profiles/r05_uha_funnel_forward_times.txt.)
__host__ __device__ constexpr bool uha_coop_tail_fits(int T, int real_width) { return T >= 2 && real_width <= 16 * (T - 1) + 4; }

// reduce-scatter over the four rows of the wave (cmcd_coop_wide.hip has the same pair):
//   uha_rs32(a, b): rows 0, 1 <- a(row r) + a(row r + 2);  rows 2, 3 <- b(row r - 2) + b(row r)
//   uha_rs16(p, q): rows 0, 2 <- p(row r) + p(row r + 1);  rows 1, 3 <- q(row r - 1) + q(row r)
__device__ __forceinline__ float uha_rs32(float x, float y) {
  uint32_t r0, r1;
  swap32(__float_as_uint(x), __float_as_uint(y), r0, r1);
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float uha_rs16(float x, float y) {
  uint32_t r0, r1;
  swap16(__float_as_uint(x), __float_as_uint(y), r0, r1);
  return __uint_as_float(r0) + __uint_as_float(r1);
}

// HALF: 8 particles per tile (batches of <= 2048 particles: twice the workgroups, on CUs that would otherwise idle).  The
// state and RNG waves keep 16 columns — columns c and c + 8 carry the SAME particle (same seed: identical values in both)
// — so a particle has 8 lanes for its target gradient; the MLP waves give every particle 8 lanes instead of 4 (lane (qi,
// pg, kh, ng) = particle 4 pg + qi, neurons 4 ng + 2 kh + {0, 1} of the wave's 16) and run layer 2 on `4x4x1` (16 blocks of
// 4 neurons x 4 particles, the activations broadcast from one 16-lane row by the instruction): half the matrix time and
// half the element-wise work per lane — the overdamped kernel's form (cmcd_coop.hip), same operand table (w2q).
template <int TARGET, int ARCH, int D, int T, bool HALF, bool TAIL = false>
__global__ __launch_bounds__(64 * uha_coop_waves(TARGET, T, HALF, TAIL)) void uha_coop_kernel(TrajArgs a) {
  constexpr int HP = 16 * T;
  constexpr bool DEALT = uha_coop_dealt(TARGET, HALF);
  static_assert(!TAIL || DEALT, "the 4-neuron tail rides on the dealt form");
  constexpr int TM = TAIL ? T - 1 : T;        // MLP waves
  constexpr bool TAILW = TAIL;                // the tail on wave TM + 3 (its 4 second-layer activations: row TM of `part`)
  constexpr int DIN = 2 * D;
  constexpr int Hh = (D + 1) / 2;
  constexpr int PPT = HALF ? 8 : 16;          // particles per tile
  constexpr int LP = HALF ? 8 : 4;            // lanes per particle on the state wave
  constexpr int HQP = HP + 4;                 // HALF: pitch of the [particle][neuron] activation buffer (bank spread)
  constexpr int NQ = HP / 2;                  // HALF: 4x4x1 steps per pass (two contraction halves side by side)
  constexpr int RSA = ((HP / 2 + 15) / 16) * 4;   // HALF: activations one row of an MLP wave fetches
  static_assert(!HALF || D % 2 == 0, "8-particle tiles pair the outputs");
  constexpr bool GEF = ARCH == CMCD_ARCH_GEFFNER;
  constexpr bool RNG_C = Hh > 2;              // normal(G) needs more than the two blocks that ride beside split(H)
  static_assert(Hh <= 6, "normal(G): at most 2 + 4 Threefry blocks per draw");
  using Tg = Target<TARGET, D>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* hbuf = lds;                          // [T][64 lanes][4]   layer-1 activations, MFMA B order
  float* part = hbuf + T * 256;               // [T][16][D]         layer-3 partials per MLP wave
  float* xin = part + T * 16 * D;             // [16][2 D]          network input [z; rho] / [z; rho']
  float* nzb = xin + 16 * DIN;                // [2][16][D]         deviates of bridge i (parity i & 1)
  uint32_t* keyb = reinterpret_cast<uint32_t*>(nzb + 2 * 16 * D);   // [16][2]  gen_0 handed from the state wave to the RNG wave
  float* lds_tgt = nzb + 2 * 16 * D + 32;     // tgt_floats
  for (int i = threadIdx.x; i < a.w.tgt_floats; i += blockDim.x) lds_tgt[i] = a.ws[a.w.tgt + i];
  if constexpr (TAIL)                         // the pad neurons of the last tile are never written: zero weights, finite activations
    for (int i = threadIdx.x; i < 8 * HQP + 32; i += blockDim.x) hbuf[i] = 0.f;   // (+ 16 zeros the tail's idle slices read)
  {   // r05: the per-bridge tables into this XCD's L2, one touch per 128-byte line (cmcd_coop.hip: behind the prep launch every XCD's copy is gone)
      float warm = 0.f;
      const int64_t wt0 = a.w.sched;
      const int64_t wt1 = (ARCH == CMCD_ARCH_GEFFNER ? a.w.utab : a.w.bias1) + (int64_t)(a.K + 1) * (16 * T);
      const int64_t per_xcd = (gridDim.x + 7) >> 3, rank = blockIdx.x >> 3;     // dealt to the workgroups that share an XCD
      for (int64_t wi = wt0 + 32 * (rank * blockDim.x + threadIdx.x); wi < wt1; wi += 32 * per_xcd * blockDim.x) warm += a.ws[wi];
      asm volatile("" ::"v"(warm));
    }

  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t tile = blockIdx.x;
  const int cp = HALF ? (c & 7) : c;                 // particle column
  const int64_t p = tile * PPT + cp;
  const bool valid = p < a.n;
  const bool own = g == 0 && (!HALF || c < 8);       // the lane that writes its particle's exchange rows and outputs
  const int sub = HALF ? (lane >> 3) : g;            // state wave: which share of the target's terms this lane owns
  const int K = a.K;
  const int gb = g & 1;
  __syncthreads();                                   // target constants staged

#ifdef CMCD_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
#define USTAMP_START() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory")
#define USTAMP_END()                                                  \
  if (blockIdx.x == 0 && lane == 0)                                   \
    for (int k = 0; k < 16; ++k) g_uha_stamps[wv][k] = st_acc[k]
#else
#define USTAMP_START()
#define USTAMP_END()
#endif

  // =========================================================================================== MLP waves
  if (wv < TM) {
    if constexpr (HALF) {
      const int ng = g, kh = (lane >> 3) & 1;
      const int nb = 16 * wv + 4 * ng + 2 * kh;        // first of this lane's two hidden units
      float w1[DIN][2], w3[D][2], aq[NQ], b2p[2];
#pragma unroll
      for (int j = 0; j < DIN; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r) w1[j][r] = a.ws[a.w.w1z + j * HP + nb + r];
#pragma unroll
      for (int j = 0; j < D; ++j)
#pragma unroll
        for (int r = 0; r < 2; ++r) w3[j][r] = a.ws[a.w.w3t + j * HP + nb + r];
#pragma unroll
      for (int q = 0; q < NQ; ++q) aq[q] = a.ws[a.w.w2q + (int64_t)(wv * NQ + q) * 64 + lane];
#pragma unroll
      for (int r = 0; r < 2; ++r) b2p[r] = a.ws[a.w.b2 + nb + r];
      const float* bias1 = a.ws + a.w.bias1;
      const float* utab = a.ws + a.w.utab;
      float2 brow_n = *reinterpret_cast<const float2*>(bias1 + nb), urow_n = {0.f, 0.f};
      if (GEF) urow_n = *reinterpret_cast<const float2*>(utab + nb);
      float* const my_h = hbuf + cp * HQP + nb;
      const float* const rd_h = hbuf + cp * HQP + (HP / 2) * kh + RSA * ng;
      // layer-3 partials leave the wave reduce-scattered (d > 2): lane (kh, ng) ends with outputs j0 and j0 + 8
      const int j0 = 4 * (ng & 1) + 2 * (ng >> 1) + kh;
      float* const my_p = part + (wv * 16 + cp) * D + j0;
      uha_lds_barrier();                             // gen_0 and the first network input published
      uha_lds_barrier();                             // deviates of bridge 0 published
      USTAMP_START();
      for (int i = 0; i < K; ++i) {
        const float2 brow = brow_n, urow = urow_n;
        if (i + 1 < K) {
          brow_n = *reinterpret_cast<const float2*>(bias1 + (int64_t)(i + 1) * HP + nb);
          if (GEF) urow_n = *reinterpret_cast<const float2*>(utab + (int64_t)(i + 1) * HP + nb);
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          // -------------------------------------------------------------- interval 1
          float h[2];
          {
            float x[DIN];
#pragma unroll
            for (int j = 0; j < DIN; ++j) x[j] = xin[cp * DIN + j];
            float pre[2] = {brow.x, brow.y};
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
              for (int j = 0; j < DIN; ++j) pre[r] = fmaf(x[j], w1[j][r], pre[r]);
            if (!GEF) {
              h[0] = gelu_fast(pre[0]); h[1] = gelu_fast(pre[1]);
            } else {
              float u[2] = {urow.x, urow.y};
#pragma unroll
              for (int r = 0; r < 2; ++r) {
                if (nb + r < DIN) u[r] = xin[cp * DIN + nb + r];   // the first 2 D entries of u are [z; rho] themselves
                h[r] = u[r] + softplus(pre[r]);
              }
            }
            *reinterpret_cast<float2*>(my_h) = float2{h[0], h[1]};
          }
          USTAMP(pass * 6 + 0);
          uha_lds_barrier();
          USTAMP(pass * 6 + 1);
          // -------------------------------------------------------------- interval 2
          {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            f32x4 hb[RSA / 4];
#pragma unroll
            for (int q = 0; q < RSA / 4; ++q) hb[q] = *reinterpret_cast<const f32x4*>(rd_h + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int sq = 0; sq < NQ; ++sq) {
              const int row = sq / RSA, t = sq % RSA;
              const float bv = hb[t / 4][t % 4];
              f32x4& ac = (sq & 1) ? acc1 : acc;
              if (row == 0) ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 4);
              else if (row == 1) ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 5);
              else if (row == 2) ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 6);
              else ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 7);
            }
            acc += acc1;
            // the two halves of the contraction sit in lanes l and l ^ 8; lane kh keeps neurons 2 kh + {0, 1} of its group
            float av[2], h2[2];
            av[0] = ((kh ? acc[2] : acc[0]) + xor8(kh ? acc[0] : acc[2])) + b2p[0];
            av[1] = ((kh ? acc[3] : acc[1]) + xor8(kh ? acc[1] : acc[3])) + b2p[1];
#pragma unroll
            for (int r = 0; r < 2; ++r) h2[r] = GEF ? h[r] + softplus(av[r]) : gelu_fast(av[r]);
            if constexpr (D <= 2) {
              // outputs in pairs (j, j + 1): lane kh = 0 collects both halves of output j, lane kh = 1 both of output j + 1
#pragma unroll
              for (int j = 0; j < D; j += 2) {
                const float p0 = h2[0] * w3[j][0] + h2[1] * w3[j][1];
                const float p1 = h2[0] * w3[j + 1][0] + h2[1] * w3[j + 1][1];
                float pj = (kh ? p1 : p0) + xor8(kh ? p0 : p1);
                pj = group_sum(pj);
                if (ng == 0) part[(wv * 16 + cp) * D + j + kh] = pj;
              }
            } else {
              // r05: a reduce-scatter instead of d / 2 full reductions — the pairs first (lane kh keeps output 2 m + kh),
              // then the row pairs (rows 0, 1 keep m even, rows 2, 3 m odd), then the rows (even rows keep the lower of their
              // two): lane (kh, ng) ends with outputs j0 = 4 (ng & 1) + 2 (ng >> 1) + kh and j0 + 8
              static_assert(D <= 16, "two values per lane after the reduce-scatter");
              constexpr int M = D / 2, M2 = (M + 1) / 2;
              float pj[D];
#pragma unroll
              for (int j = 0; j < D; ++j) pj[j] = h2[0] * w3[j][0] + h2[1] * w3[j][1];
              float qm[2 * M2], rm[4];
#pragma unroll
              for (int m = 0; m < 2 * M2; ++m)
                qm[m] = m < M ? (kh ? pj[2 * m + 1] : pj[2 * m]) + xor8(kh ? pj[2 * m] : pj[2 * m + 1]) : 0.f;
#pragma unroll
              for (int ii = 0; ii < 4; ++ii) rm[ii] = ii < M2 ? uha_rs32(qm[2 * ii], qm[2 * ii + 1]) : 0.f;
              const float t0 = uha_rs16(rm[0], rm[1]);
              my_p[0] = t0;
              if constexpr (D > 8) {
                const float t1 = uha_rs16(rm[2], rm[3]);
                if (j0 + 8 < D) my_p[8] = t1;
              }
            }
          }
          USTAMP(pass * 6 + 2);
          uha_lds_barrier();
          USTAMP(pass * 6 + 3);
          if (pass == 0) {
            uha_lds_barrier();                       // the state wave forms rho' from the partials
            USTAMP(pass * 6 + 5);
          }
        }
      }
      USTAMP_END();
      if constexpr (DEALT) uha_lds_barrier();          // the state waves' hand-over of the tile's losses
      return;
    }
    float w1[DIN][4], w3[D][4];
    f32x4 afrag[T];
    const int nb = 16 * wv + 4 * g;                  // first of this lane's four hidden units
#pragma unroll
    for (int j = 0; j < DIN; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) w1[j][r] = a.ws[a.w.w1z + j * HP + nb + r];
#pragma unroll
    for (int j = 0; j < D; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) w3[j][r] = a.ws[a.w.w3t + j * HP + nb + r];
#pragma unroll
    for (int ti = 0; ti < T; ++ti) afrag[ti] = *reinterpret_cast<const f32x4*>(a.ws + a.w.w2 + ((ti * T + wv) * 64 + lane) * 4);
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.ws + a.w.b2 + nb);
    const float* bias1 = a.ws + a.w.bias1;
    const float* utab = a.ws + a.w.utab;
    f32x4 brow_n = *reinterpret_cast<const f32x4*>(bias1 + nb), urow_n = {0.f, 0.f, 0.f, 0.f};
    if (GEF) urow_n = *reinterpret_cast<const f32x4*>(utab + nb);
    uha_lds_barrier();                               // gen_0 and the first network input published
    uha_lds_barrier();                               // deviates of bridge 0 under way
    USTAMP_START();
    for (int i = 0; i < K; ++i) {
      const f32x4 brow = brow_n, urow = urow_n;
      if (i + 1 < K) {                               // the next bridge's rows arrive while this bridge runs
        brow_n = *reinterpret_cast<const f32x4*>(bias1 + (int64_t)(i + 1) * HP + nb);
        if (GEF) urow_n = *reinterpret_cast<const f32x4*>(utab + (int64_t)(i + 1) * HP + nb);
      }
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        // ---------------------------------------------------------------- interval 1
        float h[4];
        {
          float x[DIN];
#pragma unroll
          for (int j = 0; j < DIN; ++j) x[j] = xin[c * DIN + j];
          float pre[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            pre[r] = brow[r];
#pragma unroll
            for (int j = 0; j < DIN; ++j) pre[r] = fmaf(x[j], w1[j][r], pre[r]);
          }
          if (!GEF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) h[r] = gelu_fast(pre[r]);
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float u = urow[r];
              if (nb + r < DIN) u = xin[c * DIN + nb + r];      // the first 2 D entries of u are [z; rho] themselves
              h[r] = u + softplus(pre[r]);
            }
          }
          *reinterpret_cast<f32x4*>(hbuf + (wv * 64 + lane) * 4) = f32x4{h[0], h[1], h[2], h[3]};
        }
        USTAMP(pass * 6 + 0);
        uha_lds_barrier();
        USTAMP(pass * 6 + 1);
        // ---------------------------------------------------------------- interval 2
        {
          // two accumulator chains (even / odd input tiles): a 16x16x4 that reads its predecessor's result waits for it
          f32x4 acc = b2v, acc1 = {0.f, 0.f, 0.f, 0.f};
          f32x4 hbv[T];
#pragma unroll
          for (int ti = 0; ti < T; ++ti) hbv[ti] = *reinterpret_cast<const f32x4*>(hbuf + (ti * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int ti = 0; ti < T; ++ti) {
              if (ti & 1) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ti][r], hbv[ti][r], acc1, 0, 0, 0);
              else acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ti][r], hbv[ti][r], acc, 0, 0, 0);
            }
          }
          acc += acc1;
          float h2[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) h2[r] = GEF ? h[r] + softplus(acc[r]) : gelu_fast(acc[r]);
#pragma unroll
          for (int j = 0; j < D; ++j) {
            float pj = h2[0] * w3[j][0] + h2[1] * w3[j][1] + h2[2] * w3[j][2] + h2[3] * w3[j][3];
            pj = group_sum(pj);
            if (g == 0) part[(wv * 16 + c) * D + j] = pj;
          }
        }
        USTAMP(pass * 6 + 2);
        uha_lds_barrier();
        USTAMP(pass * 6 + 3);
        if (pass == 0) {
          uha_lds_barrier();                         // the state wave forms rho' from the partials
          USTAMP(pass * 6 + 5);
        }
      }
    }
    USTAMP_END();
    return;
  }

  // =========================================================================================== RNG wave
  if (wv == TM + 1) {
    uha_lds_barrier();                               // gen_0 published
    uint32_t k0 = keyb[2 * c], k1 = keyb[2 * c + 1];
    uint32_t sg0 = 0, sg1 = 0, sh0 = 0, sh1 = 0;     // (G, H) = split(gen)
    uint32_t bb0 = 0, bb1 = 0, cb0 = 0, cb1 = 0;     // random words of this row's block of normal(G): pass B / pass C
    TfState tb, tc;
    // A: (G, H) = split(gen)
    auto pass_a = [&]() {
      uint32_t x0 = gb, x1 = 2 + gb;
      threefry2x32(k0, k1, x0, x1);
      rows01(x0, sg0, sg1);
      rows01(x1, sh0, sh1);
    };
    // B (in two parts): rows 0, 1: gen' = second(split(H));  rows 2, 3: blocks 0, 1 of normal(G) — block j encrypts (j, Hh + j)
    auto pass_b1 = [&]() {
      const int j = g - 2;
      tf_part1(tb, g < 2 ? sh0 : sg0, g < 2 ? sh1 : sg1, g < 2 ? (uint32_t)gb : (uint32_t)j,
               g < 2 ? (uint32_t)(2 + gb) : (uint32_t)((Hh + j < D) ? Hh + j : 0));
    };
    auto pass_b2 = [&](int ib) {
      tf_part2(tb);
      bb0 = tb.x0; bb1 = tb.x1;
      rows01(tb.x1, k0, k1);
      if (a.dbg_keys && valid && g == 0) {
        a.dbg_keys[((int64_t)(ib + 1) * a.n + p) * 2] = k0;
        a.dbg_keys[((int64_t)(ib + 1) * a.n + p) * 2 + 1] = k1;
      }
    };
    // one random word -> deviate `idx` of bridge ib
    auto put = [&](int ib, int idx, uint32_t y) {
      const float nv = bits_to_normal(y);
      nzb[((ib & 1) * 16 + cp) * D + idx] = nv;      // (HALF: the twin columns write the same value)
      if (a.dbg_bits && valid) {
        const int64_t o = ((int64_t)(ib + 2) * a.n + p) * D + idx;
        a.dbg_bits[o] = y;
        a.dbg_noise[o] = nv;
      }
    };
    // the words of pass B -> deviates, by the rows that encrypted them; d <= 2 (one block): its second word moves to the
    // idle row 3, so each row converts one word
    auto publish_b = [&](int ib) {
      if (Hh == 1) {
        uint32_t r0, r1;
        swap16(bb0, bb1, r0, r1);                    // r0 = rows [bb0(0) bb1(0) bb0(2) bb1(2)]
        if (g >= 2 && g - 2 < D) put(ib, g - 2, r0);
      } else if (g >= 2) {
        const int j = g - 2;
        if constexpr (HALF) {                        // r05: the twin columns hold the same two words — one conversion each
          const int idx = c >= 8 ? Hh + j : j;
          if (idx < D) put(ib, idx, c >= 8 ? bb1 : bb0);
        } else {
          put(ib, j, bb0);
          if (Hh + j < D) put(ib, Hh + j, bb1);
        }
      }
    };
    auto pass_c1 = [&]() {                           // C: blocks 2 .. 5 of normal(G) on rows 0 .. 3
      const int j = 2 + g;
      tf_part1(tc, sg0, sg1, (uint32_t)j, (uint32_t)((Hh + j < D) ? Hh + j : 0));
    };
    auto pass_c2 = [&]() { tf_part2(tc); cb0 = tc.x0; cb1 = tc.x1; };
    auto publish_c = [&](int ib) {
      const int j = 2 + g;
      if (j < Hh) {
        if constexpr (HALF) {
          const int idx = c >= 8 ? Hh + j : j;
          if (idx < D) put(ib, idx, c >= 8 ? cb1 : cb0);
        } else {
          put(ib, j, cb0);
          if (Hh + j < D) put(ib, Hh + j, cb1);
        }
      }
    };
    pass_a();
    if (RNG_C) { pass_c1(); pass_c2(); }
    pass_b1();
    pass_b2(0);
    publish_b(0);
    if (RNG_C) publish_c(0);
    uha_lds_barrier();                               // deviates of bridge 0 published
    USTAMP_START();
    for (int i = 0; i < K; ++i) {
      const bool more = i + 1 < K;
      if (i > 0) {                                   // words -> deviates of bridge i (read behind the next barrier)
        publish_b(i);
        if (RNG_C) publish_c(i);
      }
      USTAMP(0); uha_lds_barrier(); USTAMP(1);       // pass 0, barrier 1
      if (more) pass_a();
      USTAMP(2); uha_lds_barrier(); USTAMP(3);
      if (RNG_C && more) pass_c1();
      USTAMP(4); uha_lds_barrier(); USTAMP(5);
      if (more) pass_b1();
      USTAMP(6); uha_lds_barrier(); USTAMP(7);       // pass 1, barrier 1
      if (more) pass_b2(i + 1);
      if (RNG_C && more) pass_c2();
      USTAMP(8); uha_lds_barrier(); USTAMP(9);
    }
    USTAMP_END();
    if constexpr (DEALT) uha_lds_barrier();
    return;
  }

  // =========================================================================================== tail wave (TAILW)
  // The last tile's 4 real neurons on a wave of their own (wave TM + 3: the SIMD of the last MLP wave, which no auxiliary wave
  // shares): lane (qi, pg, kh, ng) = particle 4 pg + qi, layer 1 of neuron NX + ng, layer 2 as ONE 4x4x1 pass whose 16 blocks are 2
  // particle groups x 8 contraction slices (kh, ng) — the slice an MLP lane of the same (kh, ng) fetches for its own tile —
  // reduce-scattered over the slices back to neuron ng; the 4 activations go to the state waves (row TM of `part`), which apply W3.
  if constexpr (TAILW) {
    if (wv == TM + 3) {
      // (a raised issue priority for this short chain beside MLP wave TM - 1's long one: 4 218 -> 4 096 stamped cycles per
      //  bridge, but 0.1164 -> 0.1169 ms per call on the product build: not kept)
      const int ng = g, kh = (lane >> 3) & 1;
      constexpr int NX = 16 * TM;
      float w1x[DIN], aqx[RSA];
#pragma unroll
      for (int j = 0; j < DIN; ++j) w1x[j] = a.ws[a.w.w1z + j * HP + NX + ng];
#pragma unroll
      for (int t = 0; t < RSA; ++t)
        aqx[t] = RSA * ng + t < NQ ? a.ws[a.w.w2q + (int64_t)(TM * NQ + RSA * ng + t) * 64 + (lane & 15)] : 0.f;
      const float b2x = a.ws[a.w.b2 + NX + ng];
      const float* bias1 = a.ws + a.w.bias1;
      const float* utab = a.ws + a.w.utab;
      float brx_n = bias1[NX + ng], urx_n = GEF ? utab[NX + ng] : 0.f;
      const float* const rd_h = hbuf + cp * HQP + (HP / 2) * kh + RSA * ng;
      // row 3's slice has NQ - 3 RSA real inputs; what lies behind them belongs to the NEXT particle: zero weights, but it
      // must not be another particle's inf / nan — those quads come from 16 zeros behind the buffer
      const float* const rd_h2 = (RSA * ng + 4 >= NQ) ? hbuf + 8 * HQP + 16 - 4 : rd_h;
      static_assert(NQ - 3 * RSA == 4, "row 3 of the tail keeps one quad of real inputs");
      float* const my_h2 = part + TM * 16 * D + cp * 4 + ng;   // [8][4] second-layer activations of the 4 neurons -> the state waves
      uha_lds_barrier();                             // gen_0 and the first network input published
      uha_lds_barrier();                             // deviates of bridge 0 published
      USTAMP_START();
      for (int i = 0; i < K; ++i) {
        const float brx = brx_n, urx = urx_n;
        if (i + 1 < K) {
          brx_n = bias1[(int64_t)(i + 1) * HP + NX + ng];
          if (GEF) urx_n = utab[(int64_t)(i + 1) * HP + NX + ng];
        }
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          float hx;
          {
            float px = brx;
#pragma unroll
            for (int j = 0; j < DIN; ++j) px = fmaf(xin[cp * DIN + j], w1x[j], px);
            hx = GEF ? urx + softplus(px) : gelu_fast(px);     // (NX >= 2 D: u comes from the table)
            hbuf[cp * HQP + NX + ng] = hx;
          }
          USTAMP(pass * 6 + 0);
          uha_lds_barrier();
          USTAMP(pass * 6 + 1);
          {
            f32x4 hb[RSA / 4];
#pragma unroll
            for (int q = 0; q < RSA / 4; ++q) hb[q] = *reinterpret_cast<const f32x4*>((q == 0 ? rd_h : rd_h2) + 4 * q);
            f32x4 ax = {0.f, 0.f, 0.f, 0.f}, ax1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < RSA; ++t) {
              f32x4& ac = (t & 1) ? ax1 : ax;
              ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aqx[t], hb[t / 4][t % 4], ac, 0, 0, 0);
            }
            ax += ax1;
            // 8 slices (kh, ng) -> neuron ng on every lane of the row: rows first, then the two halves of the row
            const float u01 = uha_rs16(ax[0], ax[1]), u23 = uha_rs16(ax[2], ax[3]);
            float avx = uha_rs32(u01, u23);
            avx += xor8(avx);
            avx += b2x;
            const float h2x = GEF ? hx + softplus(avx) : gelu_fast(avx);
            // the 4 neurons' share of layer 3 is taken by the state waves (4 products per coordinate beside their sum of the
            // MLP waves' partials): a reduce-scatter of ten outputs here was the longest chain of the interval
            if (kh == 0) *my_h2 = h2x;
          }
          USTAMP(pass * 6 + 2);
          uha_lds_barrier();
          USTAMP(pass * 6 + 3);
          if (pass == 0) {
            uha_lds_barrier();
            USTAMP(pass * 6 + 5);
          }
        }
      }
      USTAMP_END();
      uha_lds_barrier();                             // the state waves' hand-over of the tile's losses
      return;
    }
  }

  // =========================================================================================== state waves, dealt (funnel, 8-particle tiles)
  // r05: the columns layout below keeps all d coordinates of a particle in each of its 8 lanes — for d = 10 the two
  // readings of the network output (5 waves x 10 partials per lane) and the ten-fold element-wise step made this wave the
  // pole of pass 0's first interval and of the combine (profiles/r05_uha_funnel_stamps_before.txt: 1 202 and 980 of 6 700
  // cycles per bridge, every MLP wave waiting).  Here a particle has 16 lanes on one of TWO waves and lane `sub` owns
  // coordinate `sub` (the overdamped funnel's layout, cmcd_coop_wide.hip): one partial per MLP wave, one deviate, one step;
  // the log-weight is kept as a per-coordinate share and summed over the coordinates once, at the end.  The funnel's
  // gradient needs z_0 and the sum of squares of the rest: two 16-lane sums per point.
  if constexpr (DEALT) {
    float* tz = a.traj;
    float* trho = a.traj ? a.traj + (int64_t)(K + 1) * a.n * D : nullptr;
    float* trhop = a.traj ? a.traj + (int64_t)(2 * K + 2) * a.n * D : nullptr;
    if (wv == TM) {                                   // the first draws, once, on the columns layout (all 8 particles)
      float qmean[D], qstd[D], z[D], rho[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        qmean[j] = a.params[a.lay.vd_mean + j];
        qstd[j] = expf(a.params[a.lay.vd_logdiag + j]);
      }
      uint32_t k0, k1;
      uha_first_draws<D>(a, p, valid, g, gb, qmean, qstd, z, rho, k0, k1);
      if (g == 0) { keyb[2 * c] = k0; keyb[2 * c + 1] = k1; }
      if (a.dbg_keys && valid && own) {
        a.dbg_keys[p * 2] = k0;
        a.dbg_keys[p * 2 + 1] = k1;
      }
      if (a.traj && valid && own) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          tz[p * D + j] = z[j];
          trho[p * D + j] = rho[j];
        }
      }
      if (own) {
#pragma unroll
        for (int j = 0; j < D; ++j) { xin[cp * DIN + j] = z[j]; xin[cp * DIN + D + j] = rho[j]; }
      }
    }
    uha_lds_barrier();                               // gen_0 and the first network input published
    const int c4 = lane & 3, sub = lane >> 2;        // 4 particles per wave, 16 lanes each (part_sum<16>'s lanes)
    const int pc = 4 * (wv == TM ? 0 : 1) + c4;
    const int64_t pd = tile * 8 + pc;
    const bool vld = pd < a.n;
    const int j = sub < D ? sub : 0;                 // lanes sub >= d idle along on coordinate 0 and publish nothing
    const bool act = sub < D;
    const bool keepd = a.traj && vld && act;
    const float qm = a.params[a.lay.vd_mean + j];
    const float qs = expf(a.params[a.lay.vd_logdiag + j]);
    const float qiv = 1.0f / (qs * qs);
    const float gamma = a.params[a.lay.gamma];
    const float factor = a.ws[a.w.b3 + 15], b3j = a.ws[a.w.b3 + j];
    float zj = xin[pc * DIN + j], rj = xin[pc * DIN + D + j];
    float wl;                                        // this coordinate's share of w
    {
      const float dz = zj - qm;
      wl = (dz * dz) / (2.0f * qs * qs) + logf(qs) + kHalfLog2Pi;     // -log q(z_0)
      wl += (rj * rj) * 0.5f + kHalfLog2Pi;                           // -log N(rho_0; 0, I)
    }
    // funnel (/root/reference/src/model_handler.py:124-143; Target<CMCD_TARGET_FUNNEL>::eval, cmcd_device.h):
    float v = 0.f, ss = 0.f, gp = 0.f, gq = 0.f, lp = 0.f;
    auto dist = [&]() {
      v = part_sum<16>(sub == 0 ? zj : 0.f);
      ss = part_sum<16>((act && j >= 1) ? zj * zj : 0.f);
    };
    auto grad = [&]() {
      const float emv = __builtin_amdgcn_exp2f(-1.44269504088896340736f * v);
      constexpr float c0 = -0.5f * kLog2Pi - 1.0986122886681098f, c1 = -0.5f * (D - 1) * kLog2Pi;
      const float hes = 0.5f * emv * ss;
      const float g0 = fmaf(v, -1.0f / 9.0f, hes - 0.5f * (D - 1));
      gp = __builtin_amdgcn_fmed3f(j == 0 ? g0 : -zj * emv, -1e2f, 1e2f);
      lp = fmaf(v * v, -1.0f / 18.0f, c0 + c1) - 0.5f * (D - 1) * v - hes;
    };
    dist();
    grad();
    gq = -(zj - qm) * qiv;
    struct Sc { float beta, eps, eta, sig, inv2s2, cst, ome; };
    auto scalars = [&](float beta, float eps) {
      Sc q;
      q.beta = beta; q.eps = eps;
      q.eta = gamma * eps; q.sig = sqrtf(2.0f * q.eta);
      q.inv2s2 = 1.0f / (2.0f * q.sig * q.sig); q.cst = logf(q.sig) + kHalfLog2Pi; q.ome = 1.0f - q.eta;
      return q;
    };
    Sc sc_n = scalars(a.ws[a.w.beta], a.ws[a.w.eps]);
    float cl_rome = 0.f, cl_rho = 0.f, cl_eta = 0.f, cl_inv2s2 = 0.f, cl_cst = 0.f, cl_fk = 0.f;
    const float* const rd_p = part + pc * D + j;
    f32x4 w3s = {0.f, 0.f, 0.f, 0.f};                // TAILW: W3 rows of the tail's 4 neurons, this lane's coordinate
    if constexpr (TAILW) {
#pragma unroll
      for (int e = 0; e < 4; ++e) w3s[e] = a.ws[a.w.w3t + j * HP + 16 * TM + e];
    }
    const f32x4* const rd_t = reinterpret_cast<const f32x4*>(part + TM * 16 * D + pc * 4);
    auto net_out = [&]() {
      float o = b3j;
#pragma unroll
      for (int u = 0; u < TM; ++u) o += rd_p[u * 16 * D];
      if constexpr (TAILW) {
        const f32x4 ht = *rd_t;
        o += (ht[0] * w3s[0] + ht[1] * w3s[1]) + (ht[2] * w3s[2] + ht[3] * w3s[3]);
      }
      return GEF ? o * factor : fminf(fmaxf(o, -1e4f), 1e4f);
    };
    auto close_bridge = [&]() {
      const float s = net_out();
      const float mb = cl_rome + 2.0f * cl_eta * s;
      const float db = cl_rho - mb;
      wl += (-(db * db) * cl_inv2s2 - cl_cst) - cl_fk;
    };
    uha_lds_barrier();                               // deviates of bridge 0 published
    USTAMP_START();
    for (int i = 0; i < K; ++i) {
      const Sc q = sc_n;
      float beta_n = 0.f, eps_n = 0.f;
      if (i + 1 < K) { beta_n = a.ws[a.w.beta + i + 1]; eps_n = a.ws[a.w.eps + i + 1]; }
      if (i > 0) close_bridge();                     // beside the first layer of pass 0
      USTAMP(0); uha_lds_barrier(); USTAMP(1);
      // pass 0, matrix interval: everything of rho' but the network
      const float nzv = nzb[((i & 1) * 16 + pc) * D + j];
      const float rome = rj * q.ome;
      const float snz = q.sig * nzv;
      const float uf = -1.0f * (q.beta * gp + (1.0f - q.beta) * gq);
      const float huf = q.eps * uf / 2.0f;
      if (i + 1 < K) sc_n = scalars(beta_n, eps_n);
      USTAMP(2); uha_lds_barrier(); USTAMP(3);
      // pass 0: rho' -> the second evaluation's input
      const float s0 = net_out();
      const float mf = rome - 2.0f * q.eta * s0;
      const float rhop = mf + snz;
      if (act) xin[pc * DIN + D + j] = rhop;
      USTAMP(4); uha_lds_barrier(); USTAMP(5);
      // pass 1, first layer: forward log-density, half step, z', the sums of grad log p(z')
      const float df = rhop - mf;
      const float rpp = rhop - huf;
      cl_fk = -(df * df) * q.inv2s2 - q.cst;
      cl_rho = rj;
      cl_rome = rhop * q.ome;
      cl_eta = q.eta; cl_inv2s2 = q.inv2s2; cl_cst = q.cst;
      zj = zj + q.eps * rpp;
      gq = -(zj - qm) * qiv;
      dist();
      USTAMP(6); uha_lds_barrier(); USTAMP(7);
      // pass 1, matrix interval: grad log p(z'), second half step -> the NEXT bridge's input
      grad();
      const float ub = -1.0f * (q.beta * gp + (1.0f - q.beta) * gq);
      rj = rpp - q.eps * ub / 2.0f;
      if (act) { xin[pc * DIN + j] = zj; xin[pc * DIN + D + j] = rj; }
      if (keepd) {
        tz[((int64_t)(i + 1) * a.n + pd) * D + j] = zj;
        trho[((int64_t)(i + 1) * a.n + pd) * D + j] = rj;
        trhop[((int64_t)i * a.n + pd) * D + j] = rhop;
      }
      USTAMP(8); uha_lds_barrier(); USTAMP(9);
    }
    if (K > 0) close_bridge();
    USTAMP_END();
    wl += -(rj * rj) * 0.5f - kHalfLog2Pi;
    const float loss = -(part_sum<16>(act ? wl : 0.f) + lp);
    float* const lossb = reinterpret_cast<float*>(keyb);     // gen_0's slots, long read
    if (sub == 0) lossb[pc] = loss;
    if (vld && sub == 0) a.out_loss[pd] = loss;
    if (vld && act) a.out_z[pd * D + j] = zj;
    uha_lds_barrier();                               // the tile's 8 losses on the first state wave
    if (wv == TM) {
      const int l8 = lane & 7;
      const bool use = lane < 8 && tile * 8 + l8 < a.n;
      const float ls = lossb[l8];
      double cnt = (use && isfinite(ls)) ? 1.0 : 0.0;
      double sm = use ? (double)ls : 0.0;
      double sq = use ? (double)ls * (double)ls : 0.0;
      double mx = use ? -(double)ls : -INFINITY;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        cnt += __shfl_xor(cnt, o);
        sm += __shfl_xor(sm, o);
        sq += __shfl_xor(sq, o);
        mx = fmax(mx, __shfl_xor(mx, o));
      }
      double ex = (use && mx > -INFINITY && mx < INFINITY) ? exp(-(double)ls - mx) : 0.0;
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) ex += __shfl_xor(ex, o);
      if (lane == 0) {
        double* o = a.partials + tile * CMCD_NSTATS;
        o[0] = cnt; o[1] = sm; o[2] = sq; o[3] = mx; o[4] = ex;
      }
    }
    return;
  }

  // =========================================================================================== state wave
  // (equal issue priorities: one or two s_setprio levels for this wave and / or the RNG wave measured 0.444 - 0.466 ms
  //  against 0.426 at the named shape on 16-particle tiles, 0.336 - 0.354 against 0.323 on 8-particle tiles; two extra TARGET
  //  waves taking grad log p(z') off this wave on 8-particle tiles: 0.343 against 0.3225 ms — the exponential pass is a
  //  dependent chain that runs no faster on 16 lanes per particle, profiles/r03_uha_target_wave_pair_rejected.txt)
  // q, gamma, initial draws (the prologue of uha_traj_kernel)
  float qmean[D], qstd[D], qiv[D], z[D], rho[D], gp[D], gq[D];
  float w = 0.f, logp = 0.f;
  float* tz = a.traj;
  float* trho = a.traj ? a.traj + (int64_t)(K + 1) * a.n * D : nullptr;
  float* trhop = a.traj ? a.traj + (int64_t)(2 * K + 2) * a.n * D : nullptr;
  const bool keep = a.traj && valid && own;
  constexpr float clipv = 1e2f;
  const float gamma = a.params[a.lay.gamma];
  const float factor = a.ws[a.w.b3 + 15];
  float b3r[D];                                       // output bias, resident (a load per pass would sit on the bridge's critical path)
#pragma unroll
  for (int j = 0; j < D; ++j) b3r[j] = a.ws[a.w.b3 + j];
  {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      qmean[j] = a.params[a.lay.vd_mean + j];
      qstd[j] = expf(a.params[a.lay.vd_logdiag + j]);
      qiv[j] = 1.0f / (qstd[j] * qstd[j]);
    }
    uint32_t k0, k1;
    uha_first_draws<D>(a, p, valid, g, gb, qmean, qstd, z, rho, k0, k1);
    if (g == 0) { keyb[2 * c] = k0; keyb[2 * c + 1] = k1; }
    if (a.dbg_keys && valid && own) {
      a.dbg_keys[p * 2] = k0;
      a.dbg_keys[p * 2 + 1] = k1;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float dz = z[j] - qmean[j];
      w -= -(dz * dz) / (2.0f * qstd[j] * qstd[j]) - logf(qstd[j]) - kHalfLog2Pi;
    }
    float l0 = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) l0 += -(rho[j] * rho[j]) * 0.5f - kHalfLog2Pi;
    w -= l0;
    if (keep) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        tz[p * D + j] = z[j];
        trho[p * D + j] = rho[j];
      }
    }
    if (own) {
#pragma unroll
      for (int j = 0; j < D; ++j) { xin[cp * DIN + j] = z[j]; xin[cp * DIN + D + j] = rho[j]; }
    }
  }
  typename Tg::Means means;
  Tg::template load_means<LP>(sub, lds_tgt, means);
  {
    typename Tg::State t0;
    Tg::template pass1r<LP>(z, sub, lds_tgt, means, t0);
    Tg::template pass2<LP>(z, sub, lds_tgt, t0, logp, gp);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
      gq[j] = -(z[j] - qmean[j]) * qiv[j];
    }
  }
  const bool tfast = Tg::kHasFast && Tg::is_fast(means);   // wave-uniform: the register-resident form of the target (many_gmm, 40 modes)
  // the scalars of a bridge (mcd_under_lp_a_cais.py: eta_aux = gamma eps; the refresh's scale sqrt(2 eta))
  struct Sc { float beta, eps, eta, sig, inv2s2, cst, ome; };
  auto scalars = [&](float beta, float eps) {
    Sc q;
    q.beta = beta; q.eps = eps;
    q.eta = gamma * eps; q.sig = sqrtf(2.0f * q.eta);
    q.inv2s2 = 1.0f / (2.0f * q.sig * q.sig); q.cst = logf(q.sig) + kHalfLog2Pi; q.ome = 1.0f - q.eta;
    return q;
  };
  Sc sc_n = scalars(a.ws[a.w.beta], a.ws[a.w.eps]);
  // what the backward kernel of a bridge still needs once the next bridge's input is out: closed one interval later,
  // beside the MLP waves' first layer (its network evaluation feeds the log-weight only, not the state)
  float cl_rome[D], cl_rho[D], cl_eta = 0.f, cl_inv2s2 = 0.f, cl_cst = 0.f, cl_fk = 0.f;
  auto net_out = [&](float (&s)[D]) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float o = b3r[j];
#pragma unroll
      for (int v = 0; v < T; ++v) o += part[(v * 16 + cp) * D + j];
      s[j] = GEF ? o * factor : fminf(fmaxf(o, -1e4f), 1e4f);
    }
  };
  auto close_bridge = [&]() {
    float s[D];
    net_out(s);
    float bk_lp = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float mb = cl_rome[j] + 2.0f * cl_eta * s[j];
      const float db = cl_rho[j] - mb;
      bk_lp += -(db * db) * cl_inv2s2 - cl_cst;
    }
    w += bk_lp - cl_fk;
  };
  uha_lds_barrier();                                 // gen_0 and the first network input published
  uha_lds_barrier();
  USTAMP_START();
  for (int i = 0; i < K; ++i) {
    const Sc q = sc_n;
    float beta_n = 0.f, eps_n = 0.f;
    if (i + 1 < K) { beta_n = a.ws[a.w.beta + i + 1]; eps_n = a.ws[a.w.eps + i + 1]; }   // used one interval on
    // ------------------------------------------------------------------ beside the first layer of pass 0
    if (i > 0) close_bridge();
    USTAMP(0); uha_lds_barrier(); USTAMP(1);
    // ------------------------------------------------------------------ pass 0, matrix interval: everything of rho' but the network
    float rome[D], snz[D], huf[D], mf[D], rhop[D], rpp[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float nzv = nzb[((i & 1) * 16 + cp) * D + j];
      rome[j] = rho[j] * q.ome;
      snz[j] = q.sig * nzv;
      const float uf = -1.0f * (q.beta * gp[j] + (1.0f - q.beta) * gq[j]);
      huf[j] = q.eps * uf / 2.0f;
    }
    if (i + 1 < K) sc_n = scalars(beta_n, eps_n);
    USTAMP(2); uha_lds_barrier(); USTAMP(3);
    // ------------------------------------------------------------------ pass 0: rho' -> the second evaluation's input
    {
      float s[D];
      net_out(s);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        mf[j] = rome[j] - 2.0f * q.eta * s[j];
        rhop[j] = mf[j] + snz[j];
      }
      if (own) {
#pragma unroll
        for (int j = 0; j < D; ++j) xin[cp * DIN + D + j] = rhop[j];     // [z; rho']: the z half stays
      }
    }
    USTAMP(4); uha_lds_barrier(); USTAMP(5);
    // ------------------------------------------------------------------ pass 1, first layer: forward log-density, half step, z',
    //                                                                    distance pass of grad log p(z')
    float fk_lp = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float df = rhop[j] - mf[j];
      fk_lp += -(df * df) * q.inv2s2 - q.cst;
      rpp[j] = rhop[j] - huf[j];
      cl_rho[j] = rho[j];
      cl_rome[j] = rhop[j] * q.ome;
      z[j] = z[j] + q.eps * rpp[j];
      gq[j] = -(z[j] - qmean[j]) * qiv[j];
    }
    cl_eta = q.eta; cl_inv2s2 = q.inv2s2; cl_cst = q.cst; cl_fk = fk_lp;
    typename Tg::State tst;
    if (tfast) Tg::template pass1f<LP, true>(z, sub, means, tst);
    else Tg::template pass1r<LP>(z, sub, lds_tgt, means, tst);
    USTAMP(6); uha_lds_barrier(); USTAMP(7);
    // ------------------------------------------------------------------ pass 1, matrix interval: exponential pass, second half
    //                                                                    step -> the NEXT bridge's input (it does not need s)
    if (tfast) Tg::template pass2f<LP, true>(z, sub, means, tst, logp, gp);
    else Tg::template pass2<LP>(z, sub, lds_tgt, tst, logp, gp);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
      const float ub = -1.0f * (q.beta * gp[j] + (1.0f - q.beta) * gq[j]);
      rho[j] = rpp[j] - q.eps * ub / 2.0f;
    }
    if (own) {
#pragma unroll
      for (int j = 0; j < D; ++j) { xin[cp * DIN + j] = z[j]; xin[cp * DIN + D + j] = rho[j]; }
    }
    if (keep) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        tz[((int64_t)(i + 1) * a.n + p) * D + j] = z[j];
        trho[((int64_t)(i + 1) * a.n + p) * D + j] = rho[j];
        trhop[((int64_t)i * a.n + p) * D + j] = rhop[j];
      }
    }
    USTAMP(8); uha_lds_barrier(); USTAMP(9);
  }
  if (K > 0) close_bridge();
  USTAMP_END();
  {
    float lK = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) lK += -(rho[j] * rho[j]) * 0.5f - kHalfLog2Pi;
    w += lK;
  }
  w += logp;
  const float loss = -w;
  if (valid && own) {
    a.out_loss[p] = loss;
#pragma unroll
    for (int j = 0; j < D; ++j) a.out_z[p * D + j] = z[j];
  }
  const bool use = valid && own;
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
    double* o = a.partials + tile * CMCD_NSTATS;
    o[0] = cnt; o[1] = sm; o[2] = sq; o[3] = mx; o[4] = ex;
  }
}
#undef USTAMP_START
#undef USTAMP_END

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

template <int TARGET, int ARCH, int D, bool HALF>
static uha_fn uha_coop_pick_T(int T) {
  switch (T) {
    case 2: return uha_coop_kernel<TARGET, ARCH, D, 2, HALF>;
    case 4: return uha_coop_kernel<TARGET, ARCH, D, 4, HALF>;
    case 5: return uha_coop_kernel<TARGET, ARCH, D, 5, HALF>;
    case 9:
      // (d = 10: 72 resident 4x4x1 operands beside a 20-wide first layer do not fit three waves per SIMD)
      if constexpr (HALF && D > 4) return nullptr;
      else return uha_coop_kernel<TARGET, ARCH, D, 9, HALF>;
    default: return nullptr;
  }
}

template <bool HALF>
static uha_fn uha_coop_pick_h(const cmcd_desc& d, int T) {
  if (d.arch == CMCD_ARCH_DDS) {
    if (T != 4) return nullptr;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return uha_coop_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4, HALF>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2) return uha_coop_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4, HALF>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return uha_coop_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_DDS, 10, 4, HALF>;
    return nullptr;
  }
  if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return uha_coop_pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, HALF>(T);
  if (d.target == CMCD_TARGET_GMM && d.dim == 2) return uha_coop_pick_T<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, HALF>(T);
  if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return uha_coop_pick_T<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10, HALF>(T);
  return nullptr;
}
static uha_fn uha_coop_pick(const cmcd_desc& d, int T, bool half) {
  return half ? uha_coop_pick_h<true>(d, T) : uha_coop_pick_h<false>(d, T);
}
// the 8-particle instance whose last MLP wave takes a 4-neuron tail along (uha_coop_tail_fits): the funnel's geffner net
static bool uha_coop_has_tail(const cmcd_desc& d, int T) {
  return d.arch == CMCD_ARCH_GEFFNER && d.target == CMCD_TARGET_FUNNEL && d.dim == 10 && T == 5 &&
         uha_coop_tail_fits(T, net_in_dim(d) + d.emb_dim);
}

bool uha_available(const cmcd_desc& d, int T) { return uha_pick(d, T) != nullptr; }

// tiles up to which the cooperative form is preferred (MI355X, profiles/r03_uha_variant_crossover.txt)
#ifdef CMCD_STAMPS
extern "C" int cmcd_debug_read_uha_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_uha_stamps), sizeof(unsigned long long) * 16 * 16);
}
#endif
static int uha_coop_max_tiles(const cmcd_desc& d) { (void)d; return 1024; }
static thread_local const char* g_uha_kernel_name = "uha_traj_kernel";
const char* uha_last_kernel_name() { return g_uha_kernel_name; }

int64_t uha_traj_floats(const cmcd_desc& d, int64_t n) { return (int64_t)(3 * d.nbridges + 2) * n * d.dim; }

int uha_forward_launch(const cmcd_desc& d, const TrajArgs& ta, void* stream_, int* n_records) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const WsLayout& w = ta.w;
  uha_fn fn = uha_pick(d, w.T);
  if (!fn) return CMCD_ERR_UNSUPPORTED;
  const int64_t tiles = w.n_waves;
  *n_records = w.n_waves;
  // kernel variant (desc.reserved, as for the overdamped kernels): 0 auto, 1 wave per tile, 2 cooperative, 3 cooperative on
  // 16-particle tiles, 4 cooperative on 8-particle tiles (auto: while those still get a CU each, n <= 8 x 256)
  const bool forced = d.reserved >= 2 && d.reserved <= 5;
  const bool half_ok = uha_coop_pick(d, w.T, true) != nullptr;
  if ((d.reserved == 4 || d.reserved == 5) && !half_ok) return CMCD_ERR_UNSUPPORTED;
  const bool half = d.reserved == 4 || d.reserved == 5 || (d.reserved != 3 && half_ok && ta.n <= 8 * 256);
  uha_fn cfn = uha_coop_pick(d, w.T, half);
  if (forced && !cfn) return CMCD_ERR_UNSUPPORTED;
  if (cfn && (forced || (d.reserved != 1 && tiles <= uha_coop_max_tiles(d)))) {
    const size_t cl = size_t(w.T * 256 + w.T * 16 * d.dim + 16 * 2 * d.dim + 2 * 16 * d.dim + 32 + w.tgt_floats) * 4;
    const int64_t wgs = half ? (ta.n + 7) / 8 : tiles;
    g_uha_kernel_name = half ? "uha_coop_kernel<8-particle tiles>" : "uha_coop_kernel<16-particle tiles>";
    *n_records = (int)wgs;
    // (desc.reserved 5, as for the overdamped kernels: the 8-particle form WITHOUT the r05 tail, for A / B)
    const bool tail = half && d.reserved != 5 && uha_coop_has_tail(d, w.T);
    if (tail) cfn = uha_coop_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10, 5, true, true>;
    hipLaunchKernelGGL(cfn, dim3((unsigned)wgs), dim3(64 * uha_coop_waves(d.target, w.T, half, tail)), cl, stream, ta);
    return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
  }
  g_uha_kernel_name = "uha_traj_kernel";
  const size_t lds_bytes = size_t(w.HP * w.HP + 3 * d.dim * w.HP + w.HP + 16 + w.tgt_floats) * 4;
  if (lds_bytes > 160 * 1024) return CMCD_ERR_UNSUPPORTED;
  // waves per workgroup as in the overdamped wave-per-tile kernel: one wave per workgroup until every SIMD has one
  const int64_t per_cu = (160 * 1024) / (int64_t)lds_bytes;
  int nw = tiles <= 1024 ? 1 : (tiles <= 8192 ? 4 : 8);
  if (per_cu < 2 && tiles > 256) nw = tiles <= 1024 ? 4 : 8;
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(fn), lds_bytes)) return CMCD_ERR_HIP;
  const unsigned blocks = unsigned((tiles + nw - 1) / nw);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * nw), lds_bytes, stream, ta);
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

// =============================================================================================================
// The reparameterised gradient of MCD_CAIS_UHA_sn: d / d params_flat of L = omega sum_n loss_n, i.e. what
// jax.grad(compute_bound, 1) (/root/reference/src/main.py:174-176) back-propagates through
// /root/reference/src/mcd_under_lp_a_cais.py:42-88 — no stop_gradient anywhere in this mode.
//
// The forward launch keeps (z_e, rho_e) for e = 0..K and rho'_i for i = 0..K-1.  One wave per 16-particle tile walks the
// bridges in reverse with the adjoints lz = dL/dz_e, lr = dL/drho_e in registers.  With r = rho - m_b,
// g_b = dL/dm_b = -omega r / (2 eta) (w_i = -|r|^2 / (4 eta) + |n_i|^2 / 2: the two log sigma terms cancel and n_i is
// the raw deviate):
//
//   point e (one target gradient + Hessian at z_e serves ub of bridge e-1 and uf of bridge e):
//     adj_ub = -eps_{e-1}/2 lr,  adj_uf = -eps_e/2 arpp_e           (arpp_e = dL/drho''_e, carried from bridge e)
//     a_gp = -beta_{e-1} adj_ub - beta_e adj_uf,  a_gq likewise with 1 - beta
//     lz += H_p(z_e) (clipmask . a_gp) - a_gq / std_q^2  [- omega grad log p(z_K) at e = K, + omega grad log q(z_0) at e = 0]
//   bridge i = e-1:
//     arpp = lr + eps lz;   arp = arpp + (1 - eta) g_b;   cot(s2) = 2 eta g_b;   (dz, drho') = J_s([z; rho'])^T cot
//     lz += dz;  arp += drho';   cot(s1) = -2 eta arp;   (dz, drho) = J_s([z; rho])^T cot
//     lz += dz;  lr = (1 - eta) arp - g_b + drho
//     dL/d eta = (2 s2 - rho').g_b - omega |r|^2 / (4 eta^2) + ((rho' - m_f) / (2 eta) - rho - 2 s1).arp
//     dL/d eps_i = rho''.lz - (ub.lr + uf.arpp) / 2 + gamma dL/d eta;   dL/d gamma += eps_i dL/d eta
//     dL/d beta_i = -(gpc - gq)(z_{i+1}).adj_ub - (gpc - gq)(z_i).adj_uf
//
// Mapping and parameter contractions as in cmcd_grad.hip: 4-wave workgroups, wave q on its own tile; every contraction
// over particles is an fp32 MFMA outer product of [feature][particle] tiles staged in LDS (XOR-swizzled); the dW2 / dW3
// accumulator row tiles are dealt to the waves; bias-like sums (first-layer bias-table row, residual row, b2) are DPP row
// sums.  Per-workgroup slabs + a fixed-order reduce write grad_flat; the particle-independent tails (time coder /
// embedding table / schedules) are cmcd_grad.hip's, fed with the same S / S2 / beta / eps tables.
// =============================================================================================================
struct UhaGradArgs {
  const float* params;
  const float* ws;        // forward workspace (prep tables + packed weights)
  const float* traj;      // [3K+2][n][D] as left by uha_traj_kernel
  float* gtab;            // zeroed: S [(K+1)][HP], S2 [(K+1)][HP], gbeta [K4], geps [K4]
  float* slabs;
  cmcd_layout lay;
  WsLayout w;
  int64_t n;
  int32_t K, nquads;
  float omega;
  int64_t o_S, o_S2, o_gbeta, o_geps, slab_stride;
  // small-batch path (work items over chain chunks; 0 / nullptr: one sweep over the whole chain)
  int32_t nchunks = 1, chunk_len = 1 << 30;
  const float* xbuf = nullptr;   // [K+1][3 D][n]  adjoint state (lz, lr, arpp) entering point e   (uha_scan_kernel)
  float* jac = nullptr;          // [K][3 D^2 + 3 D][n]  Jacobian launch output
  float* xdump = nullptr;        // tests: the sweep writes the state it carries in the xbuf layout
  // Jacobian launch: the accumulation tables and the output vector, zeroed on its way (no memset launches on that path)
  float* zero_a = nullptr;
  int64_t n_a = 0;
  float* zero_b = nullptr;
  int64_t n_b = 0;
  // fixed-order accumulation (r04, as GradArgs::det of cmcd_grad.hip): one slot per (tile, bridge, evaluation) for the bias-table
  // rows and per (tile, point) for the schedule gradients — every one written by exactly one wave, with plain stores — summed
  // over tiles by uha_det_reduce_kernel.  (r03: float atomics on the shared tables; a training seed did not reproduce, and
  // the stores are cheaper: 592 -> 525 us for the named shape's sweep.)  The table holds a slot for every tile of every quad.
  // per tile: S [2 evaluations][K][HP] | S2 [2][K][HP] (geffner) | [(K+1)][8] = {d beta_e, d eps_e, d beta_{e-1}, d eps_{e-1}, d eps_{e-1} (kernel side)}
  float* det = nullptr;
};

__device__ __forceinline__ int uha_sw(int f, int p) { return f * 16 + (p ^ (f & 15)); }

template <int TARGET, int ARCH, int D, int T, int NW, bool WGLOBAL, bool JAC = false>
// (r04) the d = 2 Jacobian launch at three waves per SIMD (168 registers, 12 spilled): 337 -> 314 us at the named shape.  The sweep
// at two (241 registers, so that the second workgroup a CU's LDS has room for is really resident) measured 606 against 592 us
// with one: on gfx950 an fp32 MFMA and the VALU work of the SIMD's other wave do not overlap (tools/probes/mfma_valu_probe.hip),
// so a second wave only fills the stalls, and these are not what bounds the sweep (CHANGELOG.md (DESIGN r04 section 7d), "the floor").
#ifndef UHA_SWEEP_WAVES
#define UHA_SWEEP_WAVES 1
#endif
__global__ __launch_bounds__(64 * NW, (D == 2 && JAC) ? 3 : ((D == 2 && WGLOBAL && T <= 4) ? UHA_SWEEP_WAVES : 1)) void uha_grad_kernel(UhaGradArgs a) {
  constexpr int HP = 16 * T;
  constexpr int DIN = 2 * D;
  constexpr int S_JAC = 3 * D * D + 3 * D;     // floats per (point, particle) of the Jacobian launch, see uha_scan_kernel
  constexpr int UNR_D = D <= 4 ? D : 1;        // Jacobian launch: loops over output dimensions unrolled for the 2-d targets only
  constexpr int XT = (DIN + 15) / 16;          // 16-row tiles of the staged network input
  constexpr bool GEF = ARCH == CMCD_ARCH_GEFFNER;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WLDS = WGLOBAL ? 0 : 2 * HP * HP;
  float* lds_w2 = lds;                     // HP*HP   forward A fragments   (absent when WGLOBAL)
  float* lds_w2t = lds + HP * HP;          // HP*HP   A fragments of W2^T
  float* lds_w1z = lds + WLDS;             // DIN*HP
  float* lds_w3t = lds_w1z + DIN * HP;     // D*HP
  float* lds_b2 = lds_w3t + D * HP;        // HP
  float* lds_b3 = lds_b2 + HP;             // 16
  float* lds_tgt = lds_b3 + 16;            // tgt_floats
  float* stage = lds_tgt + a.w.tgt_floats;
  // per wave: u1T, u2T, da2T [HP][16] and doT [16][16] (read by every wave of the workgroup); da1T, du1T [16][16] and
  // xT [XT*16][16] (this wave only); accZ1 [DIN][HP], accB2 [HP] (wave-private sums over evaluations)
  constexpr int OFF_DOT = 3 * HP * 16;
  constexpr int OFF_DA1 = OFF_DOT + 256;
  constexpr int OFF_DU1 = OFF_DA1 + 256;
  constexpr int OFF_X = OFF_DU1 + 256;
  constexpr int OFF_ACC = OFF_X + XT * 256;
  constexpr int STG = OFF_ACC + (DIN + 1) * HP;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.ws + a.w.w1z);
    f32x4* dst = reinterpret_cast<f32x4*>(lds_w1z);
    for (int i = threadIdx.x; i < DIN * HP / 4; i += blockDim.x) dst[i] = src[i];
    if (!WGLOBAL) {
      src = reinterpret_cast<const f32x4*>(a.ws + a.w.w2);   // w2 and w2t are adjacent in the workspace
      dst = reinterpret_cast<f32x4*>(lds_w2);
      for (int i = threadIdx.x; i < 2 * HP * HP / 4; i += blockDim.x) dst[i] = src[i];
    }
    src = reinterpret_cast<const f32x4*>(a.ws + a.w.w3t);
    dst = reinterpret_cast<f32x4*>(lds_w3t);
    for (int i = threadIdx.x; i < D * HP / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i < HP; i += blockDim.x) lds_b2[i] = a.ws[a.w.b2 + i];
    for (int i = threadIdx.x; i < 16; i += blockDim.x) lds_b3[i] = a.ws[a.w.b3 + i];
    for (int i = threadIdx.x; i < a.w.tgt_floats; i += blockDim.x) lds_tgt[i] = a.ws[a.w.tgt + i];
  }
  __syncthreads();

  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  float* my = stage + wv * STG;
  float* u1T = my;
  float* u2T = my + HP * 16;
  float* da2T = my + 2 * HP * 16;
  float* doT = my + OFF_DOT;
  float* da1T = my + OFF_DA1;
  float* du1T = my + OFF_DU1;
  float* xT = my + OFF_X;
  float* accZ1 = my + OFF_ACC;             // [DIN][HP]
  float* accB2 = accZ1 + DIN * HP;         // [HP]
  if constexpr (!JAC) {                    // (the Jacobian launch stages nothing: it is launched without the staging area)
    for (int i = lane; i < (DIN + 1) * HP; i += 64) accZ1[i] = 0.f;
  } else {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < a.n_a; i += nth) a.zero_a[i] = 0.f;
    for (int64_t i = tid; i < a.n_b; i += nth) a.zero_b[i] = 0.f;
  }
  const int K = a.K;
  const float factor = lds_b3[15];
  int wb[4], rb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    wb[r] = (4 * g + r) * 16 + (c ^ (4 * g + r));   // write base of register r:  feature 16 t + 4 g + r, particle c
    rb[r] = c * 16 + ((4 * r + g) ^ c);             // read base of k-step r:     feature 16 t + c, particle 4 r + g
  }
  const float* w2f = WGLOBAL ? a.ws + a.w.w2 : lds_w2;
  const float* w2tf = WGLOBAL ? a.ws + a.w.w2t : lds_w2t;

  float qmean[D], qiv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qmean[j] = a.params[a.lay.vd_mean + j];
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (sd * sd);
  }
  const float gamma = a.params[a.lay.gamma];
  constexpr float clipv = 1e2f;
  const float* bias1 = a.ws + a.w.bias1;
  const float* utab = a.ws + a.w.utab;
  const float* tz = a.traj;
  const float* trho = a.traj + (int64_t)(K + 1) * a.n * D;
  const float* trhop = a.traj + (int64_t)(2 * K + 2) * a.n * D;

  // persistent accumulators (C layout: lane (g, c), register r <-> row 16 tile + 4 g + r, column 16 tile' + c)
  constexpr int OWN = (T + NW - 1) / NW;   // dW2 / dW3 row tiles owned by this wave: ti = wv + NW k < T
  f32x4 gW2[OWN][T], gW3[OWN];
#pragma unroll
  for (int k = 0; k < OWN; ++k) {
    gW3[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t) gW2[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  float gfac = 0.f, ggam = 0.f, gmu[D], glam[D], gb3[D];
#pragma unroll
  for (int j = 0; j < D; ++j) { gmu[j] = 0.f; glam[j] = 0.f; gb3[j] = 0.f; }
  // the ones row of the staged network input: its product with d a1 is the sum of d a1 over the tile's particles
  constexpr int XT0 = DIN / 16, G0 = (DIN % 16) / 4, R0 = DIN % 4;
  static_assert(DIN % 16 != 0, "the staged network input needs a spare row for the ones");

  // Work items: MODE_SWEEP — a quad of tiles x a chunk of the chain (nchunks = 1: the whole chain, adjoints carried from e = K;
  // else the adjoint state entering the chunk's first point comes from `xbuf`, written by uha_scan_kernel);  MODE_JAC — a quad
  // x ONE point / bridge pair: no parameter contraction, the pieces of the affine adjoint map are written to `jac`.
  const int64_t nwork = JAC ? (int64_t)a.nquads * K : (int64_t)a.nquads * a.nchunks;
  for (int64_t work = blockIdx.x; work < nwork; work += gridDim.x) {
    const int64_t quad = JAC ? work / K : work / a.nchunks;
    const int ch = JAC ? 0 : (int)(work % a.nchunks);
    const int e_hi = JAC ? K - (int)(work % K) : K - ch * a.chunk_len;
    const int e_lo = JAC ? e_hi : (e_hi - a.chunk_len + 1 > 0 ? e_hi - a.chunk_len + 1 : 0);
    const int64_t tile = quad * NW + wv;
    const int64_t p = tile * 16 + c;
    const bool valid = p < a.n;
    const int64_t pc = valid ? p : a.n - 1;
    const float om = valid ? a.omega : 0.f;
    // this tile's slots (the table holds one per tile of every quad: a tile past the batch writes its zeros like any other).
    // Per-lane (vector-register) pointers: the stores below then carry compile-time offsets only, and
    // nothing of the slot addressing sits in scalar registers (the sweep is at its SGPR limit: 41 spilled with scalar bases)
    const int64_t det_stride = (int64_t)K * HP * (GEF ? 4 : 2) + (int64_t)(K + 1) * 8;
    float* dlane = a.det + tile * det_stride;
    float* dbe = a.det + tile * det_stride + (det_stride - (int64_t)(K + 1) * 8);
    asm volatile("" : "+v"(dlane), "+v"(dbe));
    // schedule-gradient contribution `which` (0: bridge e, forward side; 2: bridge e - 1; 4: bridge e - 1, kernel side) of point e
    auto emit_be = [&](int e, int which, float tb, float te) {
      if (which < 4) { dbe[8 * e + which] = tb; dbe[8 * e + which + 1] = te; }
      else dbe[8 * e + 4] = te;
    };

    float lz[D], lr[D], arpp_c[D];   // dL/dz_e, dL/drho_e; dL/drho''_e of the bridge already walked (its uf side is pending)
    float znext[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      lz[j] = 0.f;
      lr[j] = om * trho[((int64_t)K * a.n + pc) * D + j];    // + log N(rho_K; 0, 1):  dL/drho_K = omega rho_K
      arpp_c[j] = 0.f;
      znext[j] = 0.f;
    }
    if (!JAC && e_hi < K) {          // the adjoint state entering point e_hi (uha_scan_kernel)
      const float* xs = a.xbuf + (int64_t)e_hi * 3 * D * a.n + pc;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        lz[j] = valid ? xs[(int64_t)j * a.n] : 0.f;
        lr[j] = valid ? xs[(int64_t)(D + j) * a.n] : 0.f;
        arpp_c[j] = valid ? xs[(int64_t)(2 * D + j) * a.n] : 0.f;
      }
    }

    for (int e = e_hi; e >= e_lo; --e) {
      asm volatile("" ::: "memory");   // LDS contents are loop-invariant: keep the weights in LDS, not in hoisted registers
      if (!JAC && a.xdump && valid && g == 0) {   // tests: the adjoint state entering point e, as the sweep carries it
        float* xs = a.xdump + (int64_t)e * 3 * D * a.n + p;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          xs[(int64_t)j * a.n] = lz[j];
          xs[(int64_t)(D + j) * a.n] = lr[j];
          xs[(int64_t)(2 * D + j) * a.n] = arpp_c[j];
        }
      }
      // ---------------------------------------------------------------- point e
      float z[D], gp[D], gq[D], gpraw[D], logp;
      constexpr int HN = Target<TARGET, D>::HN;
      float hs[HN];
#pragma unroll
      for (int j = 0; j < D; ++j) z[j] = tz[((int64_t)e * a.n + pc) * D + j];
      Target<TARGET, D>::eval_hess(z, g, lds_tgt, logp, gp, hs);
      float a_gp[D], a_gq[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        gq[j] = -(z[j] - qmean[j]) * qiv[j];
        gpraw[j] = gp[j];
        gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
        a_gp[j] = 0.f;
        a_gq[j] = 0.f;
      }
      float* jrow = nullptr;           // JAC: this particle's item (e), entries strided by n
      if constexpr (JAC) {
        jrow = a.jac + (int64_t)(e - 1) * S_JAC * a.n + pc;
        // P = H_p(z_e) diag(clip mask), column k;  c0 = the direct term of point e (e = K: - omega grad log p(z_K))
#pragma unroll UNR_D
        for (int k = 0; k < D; ++k) {
          float v[D], hv[D];
#pragma unroll
          for (int j = 0; j < D; ++j) v[j] = (j == k && fabsf(gpraw[j]) < clipv) ? 1.f : 0.f;
          Target<TARGET, D>::hvp(hs, z, v, hv);
          if (valid && g == 0) {
#pragma unroll
            for (int j = 0; j < D; ++j) jrow[(int64_t)(j * D + k) * a.n] = hv[j];
          }
        }
        if (valid && g == 0) {
#pragma unroll
          for (int j = 0; j < D; ++j) jrow[(int64_t)(3 * D * D + j) * a.n] = e == K ? -om * gpraw[j] : 0.f;
        }
      } else {
        if (e >= 1) {   // ub of bridge e - 1:  rho_e = rho'' - eps ub / 2
          const float be = a.ws[a.w.beta + e - 1], ee = a.ws[a.w.eps + e - 1];
          float sb = 0.f, se = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const float adj = -0.5f * ee * lr[j];
            const float ub = -1.0f * (be * gp[j] + (1.0f - be) * gq[j]);
            a_gp[j] -= be * adj;
            a_gq[j] -= (1.0f - be) * adj;
            sb -= (gp[j] - gq[j]) * adj;
            se -= 0.5f * ub * lr[j];
          }
          const float tb = row_sum16(sb), te = row_sum16(se);
          if (lane == 0) emit_be(e, 2, tb, te);
        }
        if (e <= K - 1) {   // uf of bridge e:  rho'' = rho' - eps uf / 2
          const float be = a.ws[a.w.beta + e], ee = a.ws[a.w.eps + e];
          float sb = 0.f, se = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const float adj = -0.5f * ee * arpp_c[j];
            const float uf = -1.0f * (be * gp[j] + (1.0f - be) * gq[j]);
            a_gp[j] -= be * adj;
            a_gq[j] -= (1.0f - be) * adj;
            sb -= (gp[j] - gq[j]) * adj;
            se -= 0.5f * uf * arpp_c[j];
          }
          const float tb = row_sum16(sb), te = row_sum16(se);
          if (lane == 0) emit_be(e, 0, tb, te);
        }
        {
          float v[D], hv[D];
#pragma unroll
          for (int j = 0; j < D; ++j) {
            if (e == K) lz[j] -= om * gpraw[j];                    // + log p(z_K), unclipped   mcdboundingmachine.py:178
            if (e == 0) lz[j] += om * gq[j];                       // - log q(z_0) in w  ->  + omega log q in L
            v[j] = fabsf(gpraw[j]) < clipv ? a_gp[j] : 0.f;        // jnp.clip passes no gradient outside its bounds
            gmu[j] += a_gq[j] * qiv[j];
            glam[j] += a_gq[j] * (-2.0f * gq[j]);
          }
          Target<TARGET, D>::hvp(hs, z, v, hv);
#pragma unroll
          for (int j = 0; j < D; ++j) lz[j] += hv[j] - a_gq[j] * qiv[j];   // H_q = -diag(1 / std^2)
        }
        if (e == 0) {   // z_0 = mean + std e0 and the explicit parameters of log q(z_0)
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const float dz = z[j] - qmean[j];
            gmu[j] += lz[j] - om * gq[j];
            glam[j] += lz[j] * dz + om * (dz * dz * qiv[j] - 1.0f);
          }
          break;
        }
      }
#pragma unroll
      for (int j = 0; j < D; ++j) znext[j] = z[j];

      // ---------------------------------------------------------------- bridge i = e - 1
      const int i = e - 1;
      const float eps = a.ws[a.w.eps + i];
      const float eta = gamma * eps, ome = 1.0f - eta, inv2eta = 0.5f / eta;
      float rho[D], rhop[D], arp[D], gb[D], lrn[D];
      float geta = 0.f, gepsd = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        z[j] = tz[((int64_t)i * a.n + pc) * D + j];
        rho[j] = trho[((int64_t)i * a.n + pc) * D + j];
        rhop[j] = trhop[((int64_t)i * a.n + pc) * D + j];
        arp[j] = lr[j] + eps * lz[j];                            // dL/drho'' = dL/drho'
        arpp_c[j] = arp[j];
        gepsd += ((znext[j] - z[j]) / eps) * lz[j];              // z' = z + eps rho''
        gb[j] = 0.f;
        lrn[j] = 0.f;
      }
      const int64_t erow = i;
      const float* brow = bias1 + erow * HP;
      float karp[D], dz2[D];           // JAC: the inhomogeneous part of dL/drho', and J_s2^T (2 eta g_b) on z
#pragma unroll
      for (int j = 0; j < D; ++j) { karp[j] = 0.f; dz2[j] = 0.f; }

      for (int pass = 0; pass < 2; ++pass) {   // pass 0: s2 = s([z; rho'], i) (backward kernel); pass 1: s1 = s([z; rho], i)
        asm volatile("" ::: "memory");
        float rin[D];
#pragma unroll
        for (int j = 0; j < D; ++j) rin[j] = pass == 0 ? rhop[j] : rho[j];
        float* srow = dlane + ((int64_t)pass * K + erow) * HP;   // this evaluation's row of the tile's slots
        // ------------------------------------------------------------ forward (keeps the activation derivatives)
        constexpr bool KEEP_A1 = T <= 4;
        f32x4 a1[KEEP_A1 ? T : 1], u1[T], a2[T];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          f32x4 pre = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
          for (int j = 0; j < D; ++j) {
            pre += z[j] * *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
            pre += rin[j] * *reinterpret_cast<const f32x4*>(lds_w1z + (D + j) * HP + 16 * t + 4 * g);
          }
          f32x4 dact = {0.f, 0.f, 0.f, 0.f};
          if (!GEF) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float dv = 0.f;
              u1[t][r] = KEEP_A1 ? gelu_fast_both(pre[r], dv) : gelu_fast(pre[r]);
              dact[r] = dv;
            }
          } else {
            f32x4 u = *reinterpret_cast<const f32x4*>(utab + erow * HP + 16 * t + 4 * g);
            if (16 * t < DIN) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int nidx = 16 * t + 4 * g + r;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                  if (j >= 16 * t && j < 16 * t + 16) u[r] = (nidx == j) ? z[j] : u[r];
                  if (D + j >= 16 * t && D + j < 16 * t + 16) u[r] = (nidx == D + j) ? rin[j] : u[r];
                }
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float dv = 0.f;
              u1[t][r] = u[r] + (KEEP_A1 ? softplus_both(pre[r], dv) : softplus(pre[r]));
              dact[r] = dv;
            }
          }
          if (KEEP_A1) a1[KEEP_A1 ? t : 0] = dact;
          if constexpr (!JAC) {
#pragma unroll
            for (int r = 0; r < 4; ++r) u1T[wb[r] + 256 * t] = u1[t][r];
          }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) a2[t] = *reinterpret_cast<const f32x4*>(lds_b2 + 16 * t + 4 * g);
        {
          f32x4 afn[T];
          {
            int lofs = lane * 4;
            asm volatile("" : "+v"(lofs));
#pragma unroll
            for (int to = 0; to < T; ++to) afn[to] = *reinterpret_cast<const f32x4*>(w2f + to * 256 + lofs);
          }
#pragma unroll
          for (int ti = 0; ti < T; ++ti) {
            asm volatile("" ::: "memory");
            int lofs = lane * 4;
            asm volatile("" : "+v"(lofs));
            f32x4 afc[T];
#pragma unroll
            for (int to = 0; to < T; ++to) afc[to] = afn[to];
            if (ti + 1 < T) {
#pragma unroll
              for (int to = 0; to < T; ++to) afn[to] = *reinterpret_cast<const f32x4*>(w2f + ((ti + 1) * T + to) * 256 + lofs);
            }
            if (WGLOBAL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int to = 0; to < T; ++to) {
#pragma unroll
              for (int r = 0; r < 4; ++r) a2[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(afc[to][r], u1[ti][r], a2[to], 0, 0, 0);
            }
          }
        }
        float opre[D], sn[D];
        {
          float part[D];
#pragma unroll
          for (int j = 0; j < D; ++j) part[j] = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            f32x4 u2t;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float dact2;
              u2t[r] = GEF ? u1[t][r] + softplus_both(a2[t][r], dact2) : gelu_fast_both(a2[t][r], dact2);
              a2[t][r] = dact2;
              if constexpr (!JAC) u2T[wb[r] + 256 * t] = u2t[r];
            }
#pragma unroll
            for (int j = 0; j < D; ++j) {
              const f32x4 wv4 = *reinterpret_cast<const f32x4*>(lds_w3t + j * HP + 16 * t + 4 * g);
              part[j] += u2t[0] * wv4[0] + u2t[1] * wv4[1] + u2t[2] * wv4[2] + u2t[3] * wv4[3];
            }
          }
#pragma unroll
          for (int j = 0; j < D; ++j) {
            opre[j] = group_sum(part[j]) + lds_b3[j];
            sn[j] = GEF ? opre[j] * factor : fminf(fmaxf(opre[j], -1e4f), 1e4f);
          }
        }
        // ------------------------------------------------------------ MLP backward: (dzo, dro) = J_s([z; rin])^T cot; ACC: with
        // the parameter contractions of this evaluation (staged rows, bias sums, this wave's dW1 outer products)
        auto backward = [&](const float (&cot)[D], auto acc_tag, float (&dzo)[D], float (&dro)[D]) {
          constexpr bool ACC = decltype(acc_tag)::value;
          asm volatile("" ::: "memory");
          float dob[D];
#pragma unroll
          for (int j = 0; j < D; ++j) {
            if (GEF) {
              dob[j] = cot[j] * factor;
              if (ACC && g == 0) gfac += cot[j] * opre[j];
            } else {
              dob[j] = fabsf(opre[j]) < 1e4f ? cot[j] : 0.f;
            }
            if (ACC && g == 0) gb3[j] += dob[j];
          }
          f32x4 d2[T], d1[T];
          float b2sel[T];
#pragma unroll
          for (int t = 0; t < T; ++t) b2sel[t] = 0.f;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            f32x4 du2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < D; ++j) du2 += dob[j] * *reinterpret_cast<const f32x4*>(lds_w3t + j * HP + 16 * t + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              d2[t][r] = du2[r] * a2[t][r];
              if constexpr (ACC) {
                da2T[wb[r] + 256 * t] = d2[t][r];
                const float s = row_sum16(d2[t][r]);                  // db2[n] += sum over the tile's particles: every lane has
                b2sel[t] = (c & 3) == r ? s : b2sel[t];               // it, lane c < 4 keeps hidden unit 16 t + 4 g + c
              }
            }
            d1[t] = GEF ? du2 : f32x4{0.f, 0.f, 0.f, 0.f};           // (geffner: d u2 rides on through the residual path)
          }
          if constexpr (ACC) {   // db2: ONE exec region per evaluation (r03: a branch + a single-lane update per hidden-unit register)
            if (c < 4) {
#pragma unroll
              for (int t = 0; t < T; ++t) accB2[16 * t + 4 * g + c] += b2sel[t];
            }
          }
          {
            f32x4 atn[T];
            {
              int lofs = lane * 4;
              asm volatile("" : "+v"(lofs));
#pragma unroll
              for (int tk = 0; tk < T; ++tk) atn[tk] = *reinterpret_cast<const f32x4*>(w2tf + (tk * T) * 256 + lofs);
            }
#pragma unroll
            for (int tn = 0; tn < T; ++tn) {
              asm volatile("" ::: "memory");
              int lofs = lane * 4;
              asm volatile("" : "+v"(lofs));
              f32x4 atc[T];
#pragma unroll
              for (int tk = 0; tk < T; ++tk) atc[tk] = atn[tk];
              if (tn + 1 < T) {
#pragma unroll
                for (int tk = 0; tk < T; ++tk) atn[tk] = *reinterpret_cast<const f32x4*>(w2tf + (tk * T + tn + 1) * 256 + lofs);
              }
              if (WGLOBAL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int tk = 0; tk < T; ++tk) {
#pragma unroll
                for (int r = 0; r < 4; ++r) d1[tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(atc[tk][r], d2[tn][r], d1[tk], 0, 0, 0);
              }
            }
          }
          float xa_own[XT][4];
          if constexpr (ACC) {
            // the two small tiles: network input [z; rin] and d o
#pragma unroll
            for (int xt = 0; xt < XT; ++xt) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int f = 16 * xt + 4 * g + r;
                float xv = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) {
                  xv = (f == j) ? z[j] : xv;
                  xv = (f == D + j) ? rin[j] : xv;
                }
                if (f == DIN) xv = 1.0f;
                xT[wb[r] + 256 * xt] = xv;
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int f = 4 * g + r;
              float dv = 0.f;
#pragma unroll
              for (int j = 0; j < D; ++j) dv = (f == j) ? dob[j] : dv;
              doT[wb[r]] = dv;
            }
#pragma unroll
            for (int xt = 0; xt < XT; ++xt)
#pragma unroll
              for (int s = 0; s < 4; ++s) xa_own[xt][s] = xT[rb[s] + 256 * xt];
          }
          float jpart[DIN];   // J_s^T cot, this lane's share of the hidden units
#pragma unroll
          for (int j = 0; j < DIN; ++j) jpart[j] = 0.f;
          float s1row[T], s2sel[T];   // d / d bias-table row i (d a1 sums) and the residual row (d u1 sums), stored below
#pragma unroll
          for (int t = 0; t < T; ++t) { s1row[t] = 0.f; s2sel[t] = 0.f; }
#pragma unroll
          for (int t = 0; t < T; ++t) {
            f32x4 pre1;
            if (KEEP_A1) {
              pre1 = a1[KEEP_A1 ? t : 0];
            } else {
              pre1 = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
              for (int j = 0; j < D; ++j) {
                pre1 += z[j] * *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
                pre1 += rin[j] * *reinterpret_cast<const f32x4*>(lds_w1z + (D + j) * HP + 16 * t + 4 * g);
              }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (GEF) {
                if constexpr (ACC) {
                  du1T[wb[r]] = d1[t][r];
                  const float s2v = row_sum16(d1[t][r]);              // residual row: sum_p d u1  (-> d emb): every lane has it
                  s2sel[t] = (c & 3) == r ? s2v : s2sel[t];           // lane c < 4 keeps hidden unit 16 t + 4 g + c
                }
                if (16 * t < DIN) {                                 // residual path of the first block: d x_j += d u1_j
#pragma unroll
                  for (int j = 0; j < DIN; ++j)
                    if (j >= 16 * t && j < 16 * t + 16) jpart[j] += (16 * t + 4 * g + r == j) ? d1[t][r] : 0.f;
                }
              }
              d1[t][r] *= KEEP_A1 ? pre1[r] : (GEF ? sigmoid_fast(pre1[r]) : gelu_grad_fast(pre1[r]));
              if constexpr (ACC) da1T[wb[r]] = d1[t][r];
            }
#pragma unroll
            for (int j = 0; j < DIN; ++j) {
              const f32x4 wv4 = *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
              jpart[j] += d1[t][0] * wv4[0] + d1[t][1] * wv4[1] + d1[t][2] * wv4[2] + d1[t][3] * wv4[3];
            }
            if constexpr (ACC) {
              // this wave's own outer products for tile t (same wave: the LDS queue is in order, no barrier): dW1[:2d]
#pragma unroll
              for (int xt = 0; xt < XT; ++xt) {
                f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa_own[xt][s4], da1T[rb[s4]], sacc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int row = 16 * xt + 4 * g + r;
                  if (row < DIN) accZ1[row * HP + 16 * t + c] += sacc[r];
                }
                if (xt == XT0) s1row[t] = sacc[R0];                   // lanes g == G0: sum_p d a1[16 t + c][p] (the ones row)
              }
              asm volatile("" ::: "memory");   // the next tile's stores stay behind these reads
            }
          }
          if constexpr (ACC) {
            // this evaluation's rows of the tile's slots: ONE exec region each (r04 first cut: a branch + a 4-lane store per
            // hidden-unit register), 64-byte stores from the quarter wave that holds the ones row
            if (g == G0) {
              float* q = srow + c;
#pragma unroll
              for (int t = 0; t < T; ++t) q[16 * t] = s1row[t];
            }
            if (GEF && c < 4) {
              float* q = srow + 2 * (int64_t)K * HP + 4 * g + c;
#pragma unroll
              for (int t = 0; t < T; ++t) q[16 * t] = s2sel[t];
            }
          }
#pragma unroll
          for (int j = 0; j < D; ++j) {
            dzo[j] = group_sum(jpart[j]);
            dro[j] = group_sum(jpart[D + j]);
          }
        };
        if constexpr (JAC) {
          if (pass == 0) {
            // the second evaluation feeds the log-weight only: everything behind it is the INHOMOGENEOUS part of the map
            float cot[D], dr2[D];
#pragma unroll
            for (int j = 0; j < D; ++j) {
              const float mb = rhop[j] * ome + 2.0f * eta * sn[j];
              const float r = rho[j] - mb;
              gb[j] = -om * r * inv2eta;
              cot[j] = 2.0f * eta * gb[j];
            }
            backward(cot, std::false_type{}, dz2, dr2);
#pragma unroll
            for (int j = 0; j < D; ++j) karp[j] = ome * gb[j] + dr2[j];
          } else {
            // J_s1 row by row (cotangent e_k): JzT[j][k] = d s_k / d z_j, JrT[j][k] = d s_k / d rho_j;  then
            // cz = dz2 - 2 eta JzT karp,  cr = (1 - eta) karp - 2 eta JrT karp - g_b
            float cz[D], cr[D];
#pragma unroll
            for (int j = 0; j < D; ++j) { cz[j] = dz2[j]; cr[j] = ome * karp[j] - gb[j]; }
#pragma unroll UNR_D
            for (int k = 0; k < D; ++k) {               // (d = 10: a rolled loop, one copy of the backward pass)
              float cot[D], dzk[D], drk[D], kk = 0.f;
#pragma unroll
              for (int j = 0; j < D; ++j) {
                cot[j] = j == k ? 1.f : 0.f;
                kk = j == k ? karp[j] : kk;
              }
              backward(cot, std::false_type{}, dzk, drk);
#pragma unroll
              for (int j = 0; j < D; ++j) {
                cz[j] -= 2.0f * eta * dzk[j] * kk;
                cr[j] -= 2.0f * eta * drk[j] * kk;
                if (valid && g == 0) {
                  jrow[(int64_t)(D * D + j * D + k) * a.n] = dzk[j];
                  jrow[(int64_t)(2 * D * D + j * D + k) * a.n] = drk[j];
                }
              }
            }
            if (valid && g == 0) {
#pragma unroll
              for (int j = 0; j < D; ++j) {
                jrow[(int64_t)(3 * D * D + D + j) * a.n] = cz[j];
                jrow[(int64_t)(3 * D * D + 2 * D + j) * a.n] = cr[j];
              }
            }
          }
        } else {
          // ------------------------------------------------------------ cotangent of this evaluation
          float cot[D];
          if (pass == 0) {
            float r2 = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
              const float mb = rhop[j] * ome + 2.0f * eta * sn[j];
              const float r = rho[j] - mb;
              gb[j] = -om * r * inv2eta;                            // dL/dm_b
              arp[j] += ome * gb[j];
              cot[j] = 2.0f * eta * gb[j];
              geta += (2.0f * sn[j] - rhop[j]) * gb[j];
              r2 += r * r;
            }
            geta -= om * r2 * inv2eta * inv2eta;
          } else {
#pragma unroll
            for (int j = 0; j < D; ++j) {
              const float mf = rho[j] * ome - 2.0f * eta * sn[j];
              cot[j] = -2.0f * eta * arp[j];
              geta += ((rhop[j] - mf) * inv2eta - rho[j] - 2.0f * sn[j]) * arp[j];   // sigma n / sigma^2 = (rho' - m_f) / (2 eta)
              lrn[j] = ome * arp[j] - gb[j];
            }
          }
          float dzo[D], dro[D];
          backward(cot, std::true_type{}, dzo, dro);
#pragma unroll
          for (int j = 0; j < D; ++j) {
            lz[j] += dzo[j];
            if (pass == 0) arp[j] += dro[j]; else lrn[j] += dro[j];
          }
          __syncthreads();
          // ---------------------------------------------------------- outer products over particles (all tiles of the workgroup)
#pragma unroll
          for (int k = 0; k < OWN; ++k) {
            const int ti = wv + NW * k;
            if (ti < T) {
#pragma unroll
              for (int q = 0; q < NW; ++q) {
                const float* base = stage + q * STG;
                float xa[4], x2[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                  xa[s] = base[rb[s] + 256 * ti];                 // u1T
                  x2[s] = base[HP * 16 + rb[s] + 256 * ti];       // u2T
                }
#pragma unroll
                for (int to = 0; to < T; ++to)
#pragma unroll
                  for (int s = 0; s < 4; ++s)
                    gW2[k][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], base[2 * HP * 16 + rb[s] + 256 * to], gW2[k][to], 0, 0, 0);
#pragma unroll
                for (int s = 0; s < 4; ++s) gW3[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2[s], base[OFF_DOT + rb[s]], gW3[k], 0, 0, 0);
              }
            }
          }
          __syncthreads();
        }
      }
      if constexpr (!JAC) {
#pragma unroll
        for (int j = 0; j < D; ++j) lr[j] = lrn[j];
        ggam += eps * geta;
        const float te = row_sum16(gepsd + gamma * geta);
        if (lane == 0) emit_be(e, 4, 0.f, te);
      }
    }
  }
  if constexpr (JAC) return;

  // ---------------------------------------------------------------- write the workgroup slab
  // layout: dW2 [HP][HP] | dW3 [HP][16] | per wave: dW1x [DIN][HP], db2 [HP], scalars [64]: gfac, ggam, gmu[D], glam[D], gb3[D]
  float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
#pragma unroll
  for (int k = 0; k < OWN; ++k) {
    const int ti = wv + NW * k;
    if (ti < T) {
#pragma unroll
      for (int to = 0; to < T; ++to)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(16 * ti + 4 * g + r) * HP + 16 * to + c] = gW2[k][to][r];
#pragma unroll
      for (int r = 0; r < 4; ++r) slab[HP * HP + (16 * ti + 4 * g + r) * 16 + c] = gW3[k][r];
    }
  }
  float* pw = slab + HP * HP + HP * 16 + wv * ((DIN + 1) * HP + 64);
  for (int i = lane; i < (DIN + 1) * HP; i += 64) pw[i] = accZ1[i];   // dW1x rows, then db2
  {
    float* sc = pw + (DIN + 1) * HP;
    const float tf = row_sum16(gfac), tg = row_sum16(g == 0 ? ggam : 0.f);
    if (lane == 0) { sc[0] = tf; sc[1] = tg; }
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float tm = row_sum16(gmu[j]), tl = row_sum16(glam[j]), t3 = row_sum16(gb3[j]);
      if (lane == 0) { sc[2 + j] = tm; sc[2 + D + j] = tl; sc[2 + 2 * D + j] = t3; }
    }
  }
}

// fixed-order sum of the slabs into grad_flat: one thread per output entry
struct UhaReduceArgs {
  const float* slabs;
  float* grad;
  cmcd_layout lay;
  int64_t slab_stride;
  int32_t nslabs, nw, HP, D, wid, arch;
};

#ifndef UHA_RED_GROUPS
#define UHA_RED_GROUPS 32
#endif
constexpr int kRedG = UHA_RED_GROUPS;
__global__ __launch_bounds__(32 * kRedG) void uha_reduce_kernel(UhaReduceArgs a) {
  // block = 32 outputs x kRedG groups of slabs: thread (o, gr) sums its share of the terms in slab order with eight loads in
  // flight, the partial sums are added in group order — a fixed order whatever the launch (the work-item path writes up
  // to 512 slabs: one thread per output with one dependent load per slab took 0.32 ms; r04: 32 groups instead of 8 — the
  // per-wave outputs have 2048 terms, 32 dependent rounds of eight loads per thread were the launch's 30 us)
  __shared__ float red[kRedG][32];
  const int HP = a.HP, D = a.D, DIN = 2 * D, wid = a.wid;
  const bool dds = a.arch == CMCD_ARCH_DDS;
  const int64_t o_w1 = dds ? a.lay.d_sw1 : a.lay.g_w1, o_w2 = dds ? a.lay.d_sw2 : a.lay.g_w2;
  const int64_t o_b2 = dds ? a.lay.d_sb2 : a.lay.g_b2, o_w3 = dds ? a.lay.d_sw3 : a.lay.g_w3;
  const int64_t o_b3 = dds ? a.lay.d_sb3 : a.lay.g_b3;
  const int ol = threadIdx.x & 31, gr = threadIdx.x >> 5;
  int64_t i = (int64_t)blockIdx.x * 32 + ol;
  const int64_t per = (int64_t)(DIN + 1) * HP + 64, base = (int64_t)HP * HP + HP * 16;
  int64_t dst = -1, off = 0;
  bool perwave = false;
  if (i < (int64_t)wid * wid) {                                   // dW2[k][n]
    dst = o_w2 + i; off = (i / wid) * HP + (i % wid);
  } else if ((i -= (int64_t)wid * wid) < (int64_t)wid * D) {      // dW3[n][j]
    dst = o_w3 + i; off = (int64_t)HP * HP + (i / D) * 16 + (i % D);
  } else if ((i -= (int64_t)wid * D) < (int64_t)DIN * wid) {      // dW1[j][n], j < 2 D ([z; rho] rows)
    dst = o_w1 + i; off = (i / wid) * HP + (i % wid); perwave = true;
  } else if ((i -= (int64_t)DIN * wid) < wid) {                   // db2
    dst = o_b2 + i; off = (int64_t)DIN * HP + i; perwave = true;
  } else if ((i -= wid) < D) {                                    // db3
    dst = o_b3 + i; off = (int64_t)(DIN + 1) * HP + 2 + 2 * D + i; perwave = true;
  } else if ((i -= D) < D) {                                      // d vd.mean
    dst = a.lay.vd_mean + i; off = (int64_t)(DIN + 1) * HP + 2 + i; perwave = true;
  } else if ((i -= D) < D) {                                      // d vd.logdiag
    dst = a.lay.vd_logdiag + i; off = (int64_t)(DIN + 1) * HP + 2 + D + i; perwave = true;
  } else if ((i -= D) < 1) {                                      // d gamma
    dst = a.lay.gamma; off = (int64_t)(DIN + 1) * HP + 1; perwave = true;
  } else if ((i -= 1) < 1 && !dds) {                              // d factor_sn
    dst = a.lay.g_factor; off = (int64_t)(DIN + 1) * HP; perwave = true;
  }
  float v = 0.f;
  if (dst >= 0) {
    const int64_t nterms = perwave ? (int64_t)a.nslabs * a.nw : a.nslabs;
    auto term = [&](int64_t t) -> int64_t {
      return perwave ? (t / a.nw) * a.slab_stride + base + (t % a.nw) * per + off : t * a.slab_stride + off;
    };
    const int64_t chunk = (nterms + kRedG - 1) / kRedG;
    int64_t t = gr * chunk;
    const int64_t tend = t + chunk < nterms ? t + chunk : nterms;
    for (; t + 8 <= tend; t += 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = a.slabs[term(t + u)];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; t < tend; ++t) v += a.slabs[term(t)];
  }
  red[gr][ol] = v;
  __syncthreads();
  if (gr == 0 && dst >= 0) {
    float tot = red[0][ol];
#pragma unroll
    for (int q = 1; q < kRedG; ++q) tot += red[q][ol];
    a.grad[dst] = tot;
  }
}

// The small-batch path's second launch: the adjoint recursion of the reverse sweep is AFFINE in the state
// X_e = (lz, lr, arpp) entering point e, and its coefficients depend on the kept trajectory only (uha_grad_kernel<.., JAC>
// writes them per (point, particle): P = H_p(z_e) diag(clip mask) [d][d], JzT / JrT = the rows of J_s([z; rho], i)
// [d][d] each, c0, cz, cr [d] — the second evaluation and everything behind it sit in cz / cr):
//   u = beta_{e-1} eps_{e-1} lr / 2 + beta_e eps_e arpp / 2,   v = the same with 1 - beta
//   lz+ = lz + P u - v / std_q^2 + c0;   a = lr + eps_{e-1} lz+
//   lz' = lz+ - 2 eta JzT a + cz;   lr' = (1 - eta) a - 2 eta JrT a + cr;   arpp' = a          (eta = gamma eps_{e-1})
// One thread per particle walks e = K .. 1 and stores X_e for the work items of the third launch.
struct UhaScanArgs {
  const float* params;
  const float* ws;
  const float* traj;
  const float* jac;
  float* xbuf;
  cmcd_layout lay;
  WsLayout w;
  int64_t n;
  int32_t K;
  float omega;
};

template <int D>
__global__ __launch_bounds__(64) void uha_scan_kernel(UhaScanArgs a) {
  constexpr int S = 3 * D * D + 3 * D;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n) return;
  const int K = a.K;
  const float gamma = a.params[a.lay.gamma];
  float qiv[D], lz[D], lr[D], ar[D];
  const float* trho = a.traj + (int64_t)(K + 1) * a.n * D;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (sd * sd);
    lz[j] = 0.f;
    lr[j] = a.omega * trho[((int64_t)K * a.n + p) * D + j];
    ar[j] = 0.f;
  }
  // one step of the recursion with the item's coefficients read through `co(s)`
  auto step = [&](int e, auto co) {
    float* xs = a.xbuf + (int64_t)e * 3 * D * a.n + p;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      xs[(int64_t)j * a.n] = lz[j];
      xs[(int64_t)(D + j) * a.n] = lr[j];
      xs[(int64_t)(2 * D + j) * a.n] = ar[j];
    }
    const float eps_i = a.ws[a.w.eps + e - 1], be_i = a.ws[a.w.beta + e - 1];
    const float eps_e = e <= K - 1 ? a.ws[a.w.eps + e] : 0.f, be_e = e <= K - 1 ? a.ws[a.w.beta + e] : 0.f;
    const float eta = gamma * eps_i, ome = 1.0f - eta;
    float u[D], lzp[D], arh[D];
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = be_i * (0.5f * eps_i * lr[k]) + be_e * (0.5f * eps_e * ar[k]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float acc = lz[j] + co(3 * D * D + j);
#pragma unroll
      for (int k = 0; k < D; ++k) acc += co(j * D + k) * u[k];
      const float tub = 0.5f * eps_i * lr[j], tuf = 0.5f * eps_e * ar[j];
      acc -= qiv[j] * ((1.0f - be_i) * tub + (1.0f - be_e) * tuf);
      lzp[j] = acc;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) arh[j] = lr[j] + eps_i * lzp[j];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float az = lzp[j] + co(3 * D * D + D + j), al = ome * arh[j] + co(3 * D * D + 2 * D + j);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        az -= 2.0f * eta * co(D * D + j * D + k) * arh[k];
        al -= 2.0f * eta * co(2 * D * D + j * D + k) * arh[k];
      }
      lz[j] = az;
      lr[j] = al;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) ar[j] = arh[j];
  };
  if constexpr (D <= 4) {
    // the coefficients do not depend on the state: items e - 1 .. e - PF are requested before item e is consumed
    constexpr int PF = 4;
    float ring[PF][S], it[S];
    auto fetch = [&](int e, float (&dst)[S]) {      // item of point e (clamped: a request below 1 re-reads item 1)
      const float* jr = a.jac + (int64_t)((e >= 1 ? e : 1) - 1) * S * a.n + p;
#pragma unroll
      for (int s = 0; s < S; ++s) dst[s] = jr[(int64_t)s * a.n];
    };
#pragma unroll
    for (int q = 0; q < PF; ++q) fetch(K - q, ring[q]);
    for (int e0 = K; e0 >= 1; e0 -= PF) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int e = e0 - q;
        if (e < 1) break;
#pragma unroll
        for (int s = 0; s < S; ++s) it[s] = ring[q][s];
        fetch(e - PF, ring[q]);
        step(e, [&](int s) { return it[s]; });
      }
    }
  } else {
    // d = 10: 330 coefficients per item — read where they are used (each once), not staged in registers
    for (int e = K; e >= 1; --e) {
      const float* jr = a.jac + (int64_t)(e - 1) * S * a.n + p;
      step(e, [&](int s) { return jr[(int64_t)s * a.n]; });
    }
  }
  float* xs = a.xbuf + p;                          // X_0
#pragma unroll
  for (int j = 0; j < D; ++j) {
    xs[(int64_t)j * a.n] = lz[j];
    xs[(int64_t)(D + j) * a.n] = lr[j];
    xs[(int64_t)(2 * D + j) * a.n] = ar[j];
  }
}

// d = 10: the scan with one LANE per state row — lane (particle, j) owns row j of the three d x d products and the j-th
// components of (lz, lr, arpp); the vectors every row needs (lr, arpp, then a) pass through LDS inside the wave (one
// single-wave workgroup holds 64 / d particles).  33 coefficient loads and ~35 FMAs per lane and step instead of 330 and
// ~700 on one thread per particle: funnel, N = 300, K = 64: 0.42 -> see profiles/r03_uha_grad_work_items.txt.
template <int D>
__global__ __launch_bounds__(64) void uha_scan_rows_kernel(UhaScanArgs a) {
  constexpr int S = 3 * D * D + 3 * D, PPB = 64 / D;
  __shared__ float sh[2][64 + D];                 // (+ D: the idle lanes behind the last whole particle read their own slots)
  const int lane = threadIdx.x, pl = lane / D, j = lane % D;
  const int64_t p0 = (int64_t)blockIdx.x * PPB + pl;
  const bool act = pl < PPB && p0 < a.n;
  const int64_t p = act ? p0 : a.n - 1;
  const int K = a.K;
  const float gamma = a.params[a.lay.gamma];
  const float* trho = a.traj + (int64_t)(K + 1) * a.n * D;
  const float sd = expf(a.params[a.lay.vd_logdiag + j]);
  const float qiv = 1.0f / (sd * sd);
  float lz = 0.f, lr = a.omega * trho[((int64_t)K * a.n + p) * D + j], ar = 0.f;
  const int base = pl * D;                       // this particle's slots in the exchange rows
  for (int e = K; e >= 1; --e) {
    const float* jr = a.jac + (int64_t)(e - 1) * S * a.n + p;
    float rp[D], rz[D], rr[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
      rp[k] = jr[(int64_t)(j * D + k) * a.n];
      rz[k] = jr[(int64_t)(D * D + j * D + k) * a.n];
      rr[k] = jr[(int64_t)(2 * D * D + j * D + k) * a.n];
    }
    const float c0 = jr[(int64_t)(3 * D * D + j) * a.n], cz = jr[(int64_t)(3 * D * D + D + j) * a.n];
    const float cr = jr[(int64_t)(3 * D * D + 2 * D + j) * a.n];
    if (act) {
      float* xs = a.xbuf + (int64_t)e * 3 * D * a.n + p;
      xs[(int64_t)j * a.n] = lz;
      xs[(int64_t)(D + j) * a.n] = lr;
      xs[(int64_t)(2 * D + j) * a.n] = ar;
    }
    const float eps_i = a.ws[a.w.eps + e - 1], be_i = a.ws[a.w.beta + e - 1];
    const float eps_e = e <= K - 1 ? a.ws[a.w.eps + e] : 0.f, be_e = e <= K - 1 ? a.ws[a.w.beta + e] : 0.f;
    const float eta = gamma * eps_i, ome = 1.0f - eta;
    const float tub = 0.5f * eps_i * lr, tuf = 0.5f * eps_e * ar;
    sh[0][lane] = be_i * tub + be_e * tuf;         // u_j
    __syncthreads();
    float acc = lz + c0;
#pragma unroll
    for (int k = 0; k < D; ++k) acc += rp[k] * sh[0][base + k];
    acc -= qiv * ((1.0f - be_i) * tub + (1.0f - be_e) * tuf);
    const float arh = lr + eps_i * acc;
    sh[1][lane] = arh;
    __syncthreads();
    float az = acc + cz, al = ome * arh + cr;
#pragma unroll
    for (int k = 0; k < D; ++k) {
      const float ak = sh[1][base + k];
      az -= 2.0f * eta * rz[k] * ak;
      al -= 2.0f * eta * rr[k] * ak;
    }
    lz = az; lr = al; ar = arh;
  }
  if (act) {
    float* xs = a.xbuf + p;                        // X_0
    xs[(int64_t)j * a.n] = lz;
    xs[(int64_t)(D + j) * a.n] = lr;
    xs[(int64_t)(2 * D + j) * a.n] = ar;
  }
}

// The scan in two parallel launches (2-d targets): the work items of the third launch need X only where their chunks START,
// and a chunk's composite map is affine too.  uha_compose_kernel: thread (particle, chunk, v) runs the recursion over the
// chunk from the unit vector e_v with the inhomogeneous terms switched off (v < 3 d: column v of the chunk's matrix) or from
// zero with them on (v = 3 d: its offset); uha_chain_kernel: one thread per particle applies the nchunks composite maps in
// turn and stores X at the chunk starts.  256 dependent steps of 2000 threads (0.16 ms) become 17 steps of 224 000 threads
// and 16 steps of 2000.
struct UhaComposeArgs {
  const float* params;
  const float* ws;
  const float* traj;
  const float* jac;
  float* cbuf;       // [nchunks][3 d + 1][3 d][n]
  float* xbuf;
  cmcd_layout lay;
  WsLayout w;
  int64_t n;
  int32_t K, nchunks, chunk_len;
  float omega;
};

template <int D>
__global__ __launch_bounds__(256) void uha_compose_kernel(UhaComposeArgs a) {
  constexpr int S = 3 * D * D + 3 * D, NV = 3 * D + 1;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t p = tid % a.n;
  const int64_t rest = tid / a.n;
  const int v = (int)(rest % NV), ch = (int)(rest / NV);
  if (ch >= a.nchunks) return;
  const int K = a.K;
  const int e_hi = K - ch * a.chunk_len;
  const int e_lo = e_hi - a.chunk_len + 1 > 0 ? e_hi - a.chunk_len + 1 : 0;
  const float gamma = a.params[a.lay.gamma];
  const bool inh = v == 3 * D;
  float qiv[D], lz[D], lr[D], ar[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (sd * sd);
    lz[j] = v == j ? 1.f : 0.f;
    lr[j] = v == D + j ? 1.f : 0.f;
    ar[j] = v == 2 * D + j ? 1.f : 0.f;
  }
  // iteration e (point e, bridge e - 1) for e = e_hi .. max(e_lo, 1): the state entering the NEXT chunk's first point
  for (int e = e_hi; e >= e_lo && e >= 1; --e) {
    const float* jr = a.jac + (int64_t)(e - 1) * S * a.n + p;
    float it[S];
#pragma unroll
    for (int s = 0; s < S; ++s) it[s] = jr[(int64_t)s * a.n];
    const float eps_i = a.ws[a.w.eps + e - 1], be_i = a.ws[a.w.beta + e - 1];
    const float eps_e = e <= K - 1 ? a.ws[a.w.eps + e] : 0.f, be_e = e <= K - 1 ? a.ws[a.w.beta + e] : 0.f;
    const float eta = gamma * eps_i, ome = 1.0f - eta;
    float u[D], lzp[D], arh[D];
#pragma unroll
    for (int k = 0; k < D; ++k) u[k] = be_i * (0.5f * eps_i * lr[k]) + be_e * (0.5f * eps_e * ar[k]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float acc = lz[j] + (inh ? it[3 * D * D + j] : 0.f);
#pragma unroll
      for (int k = 0; k < D; ++k) acc += it[j * D + k] * u[k];
      const float tub = 0.5f * eps_i * lr[j], tuf = 0.5f * eps_e * ar[j];
      acc -= qiv[j] * ((1.0f - be_i) * tub + (1.0f - be_e) * tuf);
      lzp[j] = acc;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) arh[j] = lr[j] + eps_i * lzp[j];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      float az = lzp[j] + (inh ? it[3 * D * D + D + j] : 0.f), al = ome * arh[j] + (inh ? it[3 * D * D + 2 * D + j] : 0.f);
#pragma unroll
      for (int k = 0; k < D; ++k) {
        az -= 2.0f * eta * it[D * D + j * D + k] * arh[k];
        al -= 2.0f * eta * it[2 * D * D + j * D + k] * arh[k];
      }
      lz[j] = az;
      lr[j] = al;
    }
#pragma unroll
    for (int j = 0; j < D; ++j) ar[j] = arh[j];
  }
  float* cb = a.cbuf + ((int64_t)(ch * NV + v) * 3 * D) * a.n + p;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    cb[(int64_t)j * a.n] = lz[j];
    cb[(int64_t)(D + j) * a.n] = lr[j];
    cb[(int64_t)(2 * D + j) * a.n] = ar[j];
  }
}

template <int D>
__global__ __launch_bounds__(64) void uha_chain_kernel(UhaComposeArgs a) {
  constexpr int NV = 3 * D + 1, N3 = 3 * D;
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n) return;
  const int K = a.K;
  const float* trho = a.traj + (int64_t)(K + 1) * a.n * D;
  float x[N3];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    x[j] = 0.f;
    x[D + j] = a.omega * trho[((int64_t)K * a.n + p) * D + j];
    x[2 * D + j] = 0.f;
  }
  for (int ch = 0; ch < a.nchunks; ++ch) {
    const int e_hi = K - ch * a.chunk_len;
    float* xs = a.xbuf + (int64_t)e_hi * N3 * a.n + p;
#pragma unroll
    for (int j = 0; j < N3; ++j) xs[(int64_t)j * a.n] = x[j];
    const float* cb = a.cbuf + ((int64_t)ch * NV * N3) * a.n + p;
    float y[N3];
#pragma unroll
    for (int j = 0; j < N3; ++j) y[j] = cb[((int64_t)N3 * N3 + j) * a.n];          // the offset (v = 3 d)
#pragma unroll
    for (int v = 0; v < N3; ++v)
#pragma unroll
      for (int j = 0; j < N3; ++j) y[j] += cb[((int64_t)v * N3 + j) * a.n] * x[v];
#pragma unroll
    for (int j = 0; j < N3; ++j) x[j] = y[j];
  }
}

typedef void (*uha_grad_fn)(UhaGradArgs);
constexpr int kUhaNW = 4;

template <int TARGET, int ARCH, int D>
static uha_grad_fn uha_grad_pick_T(int T) {
  switch (T) {
    case 2: return uha_grad_kernel<TARGET, ARCH, D, 2, kUhaNW, false>;
    case 4: return uha_grad_kernel<TARGET, ARCH, D, 4, kUhaNW, false>;
    case 5: return uha_grad_kernel<TARGET, ARCH, D, 5, kUhaNW, true>;
    case 9: return uha_grad_kernel<TARGET, ARCH, D, 9, kUhaNW, true>;
    default: return nullptr;
  }
}

static uha_grad_fn uha_grad_pick(const cmcd_desc& d, int T) {
  if (d.arch == CMCD_ARCH_DDS) {
    if (T != 4) return nullptr;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return uha_grad_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4, kUhaNW, false>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2) return uha_grad_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4, kUhaNW, false>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return uha_grad_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_DDS, 10, 4, kUhaNW, false>;
    return nullptr;
  }
  if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return uha_grad_pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2>(T);
  if (d.target == CMCD_TARGET_GMM && d.dim == 2) return uha_grad_pick_T<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2>(T);
  if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return uha_grad_pick_T<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10>(T);
  return nullptr;
}

// the Jacobian launch of the small-batch path (2-d targets: the per-item map is 3 d^2 + 3 d = 18 floats)
template <int TARGET, int ARCH>
static uha_grad_fn uha_jac_pick_T(int T) {
  switch (T) {
    case 2: return uha_grad_kernel<TARGET, ARCH, 2, 2, kUhaNW, false, true>;
    case 4: return uha_grad_kernel<TARGET, ARCH, 2, 4, kUhaNW, false, true>;
    case 5: return uha_grad_kernel<TARGET, ARCH, 2, 5, kUhaNW, true, true>;
    case 9: return uha_grad_kernel<TARGET, ARCH, 2, 9, kUhaNW, true, true>;
    default: return nullptr;
  }
}
template <int ARCH>
static uha_grad_fn uha_jac_pick_funnel(int T) {
  switch (T) {
    case 2: return uha_grad_kernel<CMCD_TARGET_FUNNEL, ARCH, 10, 2, kUhaNW, false, true>;
    case 4: return uha_grad_kernel<CMCD_TARGET_FUNNEL, ARCH, 10, 4, kUhaNW, false, true>;
    case 5: return uha_grad_kernel<CMCD_TARGET_FUNNEL, ARCH, 10, 5, kUhaNW, true, true>;
    default: return nullptr;          // (the 9-tile instance of d = 10 spills 3.6 KB per lane already: whole chains only)
  }
}
static uha_grad_fn uha_jac_pick(const cmcd_desc& d, int T) {
  if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) {
    if (d.arch == CMCD_ARCH_DDS) return T == 4 ? uha_jac_pick_funnel<CMCD_ARCH_DDS>(4) : nullptr;
    return uha_jac_pick_funnel<CMCD_ARCH_GEFFNER>(T);
  }
  if (d.dim != 2) return nullptr;
  if (d.arch == CMCD_ARCH_DDS) {
    if (T != 4) return nullptr;
    if (d.target == CMCD_TARGET_MANY_GMM) return uha_grad_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4, kUhaNW, false, true>;
    if (d.target == CMCD_TARGET_GMM) return uha_grad_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4, kUhaNW, false, true>;
    return nullptr;
  }
  if (d.target == CMCD_TARGET_MANY_GMM) return uha_jac_pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER>(T);
  if (d.target == CMCD_TARGET_GMM) return uha_jac_pick_T<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER>(T);
  return nullptr;
}
// The sweep over work items streams the W2 / W2^T fragments from L2 (WGLOBAL) also for the narrow nets: without the 32 KB of
// LDS-resident weights a workgroup needs 73 KB, two share a CU, and the second wave per SIMD covers the first one's latencies.
template <int TARGET, int ARCH>
static uha_grad_fn uha_item_pick_T(int T) {
  switch (T) {
    case 2: return uha_grad_kernel<TARGET, ARCH, 2, 2, kUhaNW, true>;
    case 4: return uha_grad_kernel<TARGET, ARCH, 2, 4, kUhaNW, true>;
    case 5: return uha_grad_kernel<TARGET, ARCH, 2, 5, kUhaNW, true>;
    case 9: return uha_grad_kernel<TARGET, ARCH, 2, 9, kUhaNW, true>;
    default: return nullptr;
  }
}
static uha_grad_fn uha_item_pick(const cmcd_desc& d, int T) {
  if (d.dim != 2) return nullptr;
  if (d.arch == CMCD_ARCH_DDS) {
    if (T != 4) return nullptr;
    if (d.target == CMCD_TARGET_MANY_GMM) return uha_grad_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4, kUhaNW, true>;
    if (d.target == CMCD_TARGET_GMM) return uha_grad_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4, kUhaNW, true>;
    return nullptr;
  }
  if (d.target == CMCD_TARGET_MANY_GMM) return uha_item_pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER>(T);
  if (d.target == CMCD_TARGET_GMM) return uha_item_pick_T<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER>(T);
  return nullptr;
}
constexpr int kUhaSlabs = 512;                     // workgroup slabs of the gradient workspace

// Small-batch path: while whole-chain waves cannot fill the chip (quads of tiles < CUs) and the chain is long enough to cut.
// cmcd_debug_grad_item(0 / 1) pins it, as for the overdamped modes.
constexpr int64_t kUhaItemMaxN = 8192;             // the item buffers are part of the workspace up to this batch size ...
constexpr int64_t kUhaItemMaxFloats = int64_t(1) << 28;   // ... and up to 1 GB (d = 10: 330 floats per point and particle)
static int64_t uha_item_floats_raw(const cmcd_desc& d, int64_t n) {
  const int64_t D = d.dim, K = d.nbridges;
  const int64_t compose = D == 2 ? ((K + 1) / 4 + 1) * (3 * D + 1) * 3 * D * n : 0;   // composite maps of the chunks
  return (K + 1) * 3 * D * n + K * (3 * D * D + 3 * D) * n + compose;
}
static bool uha_item_capable(const cmcd_desc& d, int T, int64_t n) {
  return uha_jac_pick(d, T) != nullptr && n <= kUhaItemMaxN && d.nbridges >= 2 && uha_item_floats_raw(d, n) <= kUhaItemMaxFloats;
}
static bool uha_item_mode(const cmcd_desc& d, int T, int64_t n) {
  if (!uha_item_capable(d, T, n)) return false;
  const int ov = get_grad_item_override();
  if (ov >= 0) return ov != 0;
  // measured on MI355X (profiles/r03_uha_grad_work_items.txt): faster at every batch up to 8192 particles (5.03 vs 6.31 ms
  // there; 16384: 12.0 vs 8.5) and down to K = 8 (N = 300: 0.117 vs 0.135 ms)
  return n <= 8192 && d.nbridges >= 4;
}
static int64_t uha_item_floats(const cmcd_desc& d, int T, int64_t n) {
  return uha_item_capable(d, T, n) ? uha_item_floats_raw(d, n) : 0;
}

bool uha_grad_available(const cmcd_desc& d, int T) { return uha_grad_pick(d, T) != nullptr; }

static void uha_grad_offsets(const cmcd_desc& d, int HP, int64_t& o_S, int64_t& o_S2, int64_t& o_gbeta, int64_t& o_geps,
                             int64_t& total) {
  const int64_t K = d.nbridges, K4 = (K + 3) & ~int64_t(3);
  int64_t o = 0;
  o_S = o; o += (K + 1) * HP;
  o_S2 = o; o += (K + 1) * HP;
  o_gbeta = o; o += K4;
  o_geps = o; o += K4;
  total = o;
}
static int64_t uha_slab_floats(const cmcd_desc& d, int HP) {
  return (int64_t)HP * HP + HP * 16 + kUhaNW * ((int64_t)(2 * d.dim + 1) * HP + 64);
}
static int64_t uha_dds_tail_floats(const cmcd_desc& d) { return d.arch == CMCD_ARCH_DDS ? (int64_t)(d.nbridges + 1) * 448 : 0; }

// ---- fixed-order sums of the per-(tile, bridge / point) slots (UhaGradArgs::det) into the tables the tails read
struct UhaDetArgs {
  const float* det;
  float* gtab;
  int64_t o_S, o_S2, o_gbeta, o_geps, tile_stride, obe, ntiles;
  int32_t K, HP, gef;
};
// A block = 64 consecutive outputs x 4 tile classes (thread (q, l): tiles q, q + 4, ... of output l in that order, eight loads
// in flight; the four partial sums are added in the order q = 0 .. 3).  Outputs: S[i][col] (and S2) for the K bridges, then
// d beta_i / d eps_i from the three contributions a bridge receives (point i's forward side, point i + 1's two).
__global__ __launch_bounds__(256) void uha_det_reduce_kernel(UhaDetArgs a) {
  __shared__ float part[4][64];
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t nS = (int64_t)a.K * a.HP, nSS = a.gef ? 2 * nS : nS;
  const int64_t o = (int64_t)blockIdx.x * 64 + l;
  int64_t src[3] = {-1, -1, -1};
  float* dst = nullptr;
  if (o < nSS) {   // the bridge's two evaluations (s2 first, as the sweep walks them)
    src[0] = o < nS ? o : 2 * nS + (o - nS);
    src[1] = src[0] + nS;
    dst = a.gtab + (o < nS ? a.o_S + o : a.o_S2 + (o - nS));
  } else if (o < nSS + 2 * (int64_t)a.K) {
    const int64_t oo = o - nSS;
    const int i = (int)(oo >> 1), w = (int)(oo & 1);   // w = 0: d beta_i, 1: d eps_i
    src[0] = a.obe + 8 * (int64_t)i + w;
    src[1] = a.obe + 8 * (int64_t)(i + 1) + 2 + w;
    if (w) src[2] = a.obe + 8 * (int64_t)(i + 1) + 4;
    dst = a.gtab + (w ? a.o_geps : a.o_gbeta) + i;
  }
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (src[k] < 0) continue;
    const float* base = a.det + src[k];
    int64_t t = q;
    for (; t + 4 * 7 < a.ntiles; t += 4 * 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = base[(t + 4 * u) * a.tile_stride];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; t < a.ntiles; t += 4) v += base[t * a.tile_stride];
  }
  part[q][l] = v;
  __syncthreads();
  if (q == 0 && dst) *dst = ((part[0][l] + part[1][l]) + part[2][l]) + part[3][l];
}
static int64_t uha_det_tile_floats(const cmcd_desc& d, int HP) {
  const int64_t K = d.nbridges;
  return K * HP * (d.arch == CMCD_ARCH_GEFFNER ? 4 : 2) + (K + 1) * 8;
}
static int64_t uha_det_floats(const cmcd_desc& d, int HP, int64_t n) {
  // a slot per tile of every quad: the same order of memory as the kept trajectory ((3 K + 2) d floats per particle)
  return ((n + 16 * kUhaNW - 1) / (16 * kUhaNW)) * kUhaNW * uha_det_tile_floats(d, HP);
}

int64_t uha_grad_workspace_floats(const cmcd_desc& d, int HP, int64_t n) {
  int64_t oS, oS2, ob, oe, tot;
  uha_grad_offsets(d, HP, oS, oS2, ob, oe, tot);
  return tot + uha_dds_tail_floats(d) + uha_slab_floats(d, HP) * kUhaSlabs + uha_item_floats(d, HP / 16, n) +
         uha_det_floats(d, HP, n);
}

static float* g_uha_xdump = nullptr;   // tests: cmcd_debug_uha_xdump — the next sweeps write the adjoint state they carry
#ifndef CMCD_NO_DIAG_HOOKS   // include/cmcd_hip_diag.h
extern "C" void cmcd_debug_uha_xdump(float* buf) { g_uha_xdump = buf; }
#endif

// ws_fwd / traj as left by the forward launch on the SAME desc / params; gws: uha_grad_workspace_floats; grad: [n_params],
// fully overwritten (zeros for the leaves the loss does not reach).
int uha_grad_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, int64_t n, const float* params,
                    int64_t n_params, const float* ws_fwd, const float* traj, float* gws, float omega, float* grad,
                    void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  uha_grad_fn fn = uha_grad_pick(d, w.T);
  if (!fn || !traj) return CMCD_ERR_UNSUPPORTED;
  const int D = d.dim, HP = w.HP, K = d.nbridges, DIN = 2 * D, XT = (DIN + 15) / 16;
  UhaGradArgs ga{};
  int64_t tot;
  uha_grad_offsets(d, HP, ga.o_S, ga.o_S2, ga.o_gbeta, ga.o_geps, tot);
  const int nw = kUhaNW;
  const int64_t nquads = (n + 16 * nw - 1) / (16 * nw);
  const int nslabs = (int)(nquads < 256 ? nquads : 256);
  float* tailbuf = gws + tot;
  float* slabs = tailbuf + uha_dds_tail_floats(d);
  ga.params = params; ga.ws = ws_fwd; ga.traj = traj; ga.gtab = gws; ga.slabs = slabs; ga.lay = lay; ga.w = w; ga.n = n;
  ga.K = K; ga.nquads = (int)nquads; ga.omega = omega; ga.slab_stride = uha_slab_floats(d, HP);
  const bool items = uha_item_mode(d, w.T, n);
  // (behind the item buffers, whether this call uses them or not)
  ga.det = slabs + uha_slab_floats(d, HP) * kUhaSlabs + uha_item_floats(d, w.T, n);
  if (!items) {   // (the work-item path's Jacobian launch zeroes both on its way)
    if (hipMemsetAsync(gws, 0, sizeof(float) * tot, stream) != hipSuccess) return CMCD_ERR_HIP;
    if (hipMemsetAsync(grad, 0, sizeof(float) * n_params, stream) != hipSuccess) return CMCD_ERR_HIP;
  }
  const bool wglobal = w.T > 4;
  // (keep the small-batch sweep's workgroup under 80 KB: two then share a CU — measured 510 against 668 us for the named shape
  // when a 13 KB table pushed it to 87 KB, profiles/r04_uha_sweep_variants.txt)
  const size_t stg = size_t(3 * HP * 16 + 3 * 256 + XT * 256 + (DIN + 1) * HP);
  const size_t lds_bytes = size_t((wglobal ? 0 : 2 * HP * HP) + DIN * HP + D * HP + HP + 16 + w.tgt_floats + nw * stg) * 4;
  if (lds_bytes > 160 * 1024) return CMCD_ERR_UNSUPPORTED;
  if (!ensure_dynamic_lds(reinterpret_cast<const void*>(fn), lds_bytes)) return CMCD_ERR_HIP;
  int nslabs_used = nslabs;
  if (items) {
    // small-batch path: Jacobian launch over (quad, point) -> per-particle scan -> the sweep over (quad, chunk) work items
    uha_grad_fn jfn = uha_jac_pick(d, w.T);
    float* xbuf = slabs + uha_slab_floats(d, HP) * kUhaSlabs;
    float* jac = xbuf + (int64_t)(K + 1) * 3 * D * n;
    // (no staging area: weights + target constants only, so several workgroups share a CU)
    const size_t jlds = size_t((wglobal ? 0 : 2 * HP * HP) + DIN * HP + D * HP + HP + 16 + w.tgt_floats) * 4;
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(jfn), jlds)) return CMCD_ERR_HIP;
    UhaGradArgs ja = ga;
    ja.jac = jac;
    ja.zero_a = gws; ja.n_a = tot; ja.zero_b = grad; ja.n_b = n_params;
    const int64_t jwork = nquads * K;
    hipLaunchKernelGGL(jfn, dim3((unsigned)(jwork < 2048 ? jwork : 2048)), dim3(64 * nw), jlds, stream, ja);
    // chunks: enough work items for two workgroups on every CU, each at least 4 points long
    int nchunks = (int)((kUhaSlabs + nquads - 1) / nquads);
    if (nchunks > (K + 1) / 4) nchunks = (K + 1) / 4;
    if (nchunks < 1) nchunks = 1;
    ga.chunk_len = (K + 1 + nchunks - 1) / nchunks;
    ga.nchunks = (K + 1 + ga.chunk_len - 1) / ga.chunk_len;
    ga.xbuf = xbuf;
    if (D == 2 && ga.nchunks >= 4) {
      // composite maps per chunk, then a short chain over the chunks: X at the chunk starts only (all the work items read)
      float* cbuf = jac + (int64_t)K * (3 * D * D + 3 * D) * n;
      UhaComposeArgs ca{params, ws_fwd, traj, jac, cbuf, xbuf, lay, w, n, K, ga.nchunks, ga.chunk_len, omega};
      const int64_t threads = n * ga.nchunks * (3 * D + 1);
      hipLaunchKernelGGL(uha_compose_kernel<2>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, ca);
      hipLaunchKernelGGL(uha_chain_kernel<2>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, ca);
    } else {
      UhaScanArgs sa{params, ws_fwd, traj, jac, xbuf, lay, w, n, K, omega};
      if (D == 2) hipLaunchKernelGGL(uha_scan_kernel<2>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, sa);
      else hipLaunchKernelGGL(uha_scan_rows_kernel<10>, dim3((unsigned)((n + 5) / 6)), dim3(64), 0, stream, sa);
    }
    const int64_t nwork = nquads * ga.nchunks;
    nslabs_used = (int)(nwork < kUhaSlabs ? nwork : kUhaSlabs);
    ga.xdump = g_uha_xdump;
    // (d = 10: the whole-chain instance — its staging fills the CU either way)
    uha_grad_fn ifn = D == 2 ? uha_item_pick(d, w.T) : fn;
    const size_t ilds = D == 2 ? size_t(DIN * HP + D * HP + HP + 16 + w.tgt_floats + nw * stg) * 4 : lds_bytes;
    if (!ensure_dynamic_lds(reinterpret_cast<const void*>(ifn), ilds)) return CMCD_ERR_HIP;
    hipLaunchKernelGGL(ifn, dim3(nslabs_used), dim3(64 * nw), ilds, stream, ga);
  } else {
    ga.xdump = g_uha_xdump;
    // (r04) whole chains of the 2-d targets run on the work-item path's instance too — W2 streamed from L2, 73 KB of LDS, two
    // workgroups per CU on up to 512 slabs (the narrow nets' LDS-resident W2 made it 105 KB and one)
#ifndef UHA_SMALL_CHAIN
#define UHA_SMALL_CHAIN 1
#endif
    uha_grad_fn cfn = (UHA_SMALL_CHAIN && D == 2) ? uha_item_pick(d, w.T) : nullptr;
    if (cfn) {
      const size_t clds = size_t(DIN * HP + D * HP + HP + 16 + w.tgt_floats + nw * stg) * 4;
      if (!ensure_dynamic_lds(reinterpret_cast<const void*>(cfn), clds)) return CMCD_ERR_HIP;
      nslabs_used = (int)(nquads < kUhaSlabs ? nquads : kUhaSlabs);
      hipLaunchKernelGGL(cfn, dim3(nslabs_used), dim3(64 * nw), clds, stream, ga);
    } else {
      hipLaunchKernelGGL(fn, dim3(nslabs_used), dim3(64 * nw), lds_bytes, stream, ga);
    }
  }

  {
    UhaDetArgs da{};
    da.det = ga.det; da.gtab = gws; da.o_S = ga.o_S; da.o_S2 = ga.o_S2; da.o_gbeta = ga.o_gbeta; da.o_geps = ga.o_geps;
    da.tile_stride = uha_det_tile_floats(d, HP); da.obe = da.tile_stride - (int64_t)(K + 1) * 8; da.ntiles = (n + 15) / 16;
    da.K = K; da.HP = HP; da.gef = d.arch == CMCD_ARCH_GEFFNER ? 1 : 0;
    const int64_t douts = (int64_t)K * HP * (da.gef ? 2 : 1) + 2 * (int64_t)K;
    hipLaunchKernelGGL(uha_det_reduce_kernel, dim3((unsigned)((douts + 63) / 64)), dim3(256), 0, stream, da);
  }
  UhaReduceArgs ra{};
  ra.slabs = slabs; ra.grad = grad; ra.lay = lay; ra.slab_stride = ga.slab_stride; ra.nslabs = nslabs_used; ra.nw = nw;
  ra.HP = HP; ra.D = D; ra.wid = d.arch == CMCD_ARCH_DDS ? 64 : DIN + d.emb_dim; ra.arch = d.arch;
  const int64_t outs = (int64_t)ra.wid * ra.wid + (int64_t)ra.wid * D + (int64_t)DIN * ra.wid + ra.wid + 3 * D + 2;
  hipLaunchKernelGGL(uha_reduce_kernel, dim3((unsigned)((outs + 31) / 32)), dim3(32 * kRedG), 0, stream, ra);
  // the particle-independent tails: schedules (cos^2 always), time coder / embedding table with the network's state
  // inputs = 2 dim wide
  const int rc = launch_net_tails(d, DIN, CMCD_EPS_COS_SQ, lay, w, params, gws, ga.o_S, ga.o_S2, ga.o_gbeta, ga.o_geps, HP,
                                  tailbuf, grad, stream_);
  if (rc != CMCD_OK) return rc;
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

}  // namespace cmcd
