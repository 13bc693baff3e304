// The work-item ("small batch") path of the reparameterised gradient, first two launches: the Jacobian rows of every
// (tile, evaluation) and the per-particle lambda recursion.  Split from cmcd_grad.hip so that this file can be built
// WITHOUT the SLP vectoriser (cmcd_amd/build.py): the packed fp32 forms it creates here cost `v_mov` shuffles and hold
// the VALU ~1.8x as long as plain ones — bptt_jac_kernel 211.7 -> 188.3 us, bptt_scan_kernel 38.2 -> 36.1 us at N = 2000,
// K = 256 (r02, rocprofv3) — while grad_kernel, which stays in cmcd_grad.hip, is 0.5 % faster WITH it.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

// ------------------------------------------------------------------------------------------
// Small-batch path of the reparameterised gradient.  lambda_e = M_e lambda_{e+1} + c_e is LINEAR in lambda and
// M_e, c_e depend only on the stored trajectory, so the K-long dependent chain shrinks to a d x d
// matrix-vector recursion:
//   jac kernel  (parallel over (tile, evaluation)):  J_s(z_e) by d forward-mode passes through the net,
//               the target Hessian, g_{e-1}  ->  M_e [d][d], C1_e [d], G_{e-1} [d]
//     M_e  = I - eps_e J_s^T + eps_e beta_e H_p diag(m) - eps_e (1 - beta_e) diag(1/std_q^2)           (e < K; M_K = 0)
//     C1_e = g_{e-1} + eps_{e-1} [J_s^T + beta_{e-1} H_p diag(m) - (1 - beta_{e-1}) diag(1/std_q^2)] g_{e-1}
//            - omega grad log p(z_K) [e = K] + omega grad log q(z_0) [e = 0]
//   scan kernel (one thread per particle):           lambda_e = M_e lambda_{e+1} + C1_e - G_e
//   grad_kernel<..., BPTT, ITEM> (parallel over (tile, evaluation)): parameter contractions with lambda known.
// ------------------------------------------------------------------------------------------
struct JacArgs {
  const float* params;
  const float* ws;
  const float* traj;      // [K+1][n][D]
  float* jac;             // [K+1][n][D*D + 2*D]
  cmcd_layout lay;
  WsLayout w;
  int64_t n, nitems;
  int32_t K, grad_clipping, ula;
  float omega;
  // the gradient's accumulation tables and output, zeroed by this launch (it precedes every launch that adds into them):
  // two memset launches fewer per gradient — ~5 us each, 7 % of a small configuration's training iteration
  float* zero_a = nullptr;
  int64_t n_a = 0;
  float* zero_b = nullptr;
  int64_t n_b = 0;
};

template <int TARGET, int ARCH, int D, int T>
__global__ __launch_bounds__(256) void bptt_jac_kernel(JacArgs a) {
  constexpr int HP = 16 * T;
  constexpr bool GEF = ARCH == CMCD_ARCH_GEFFNER;
  constexpr int S = D * D + 2 * D;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_w2 = lds;
  float* lds_w1z = lds_w2 + HP * HP;
  float* lds_w3t = lds_w1z + D * HP;
  float* lds_b2 = lds_w3t + D * HP;
  float* lds_b3 = lds_b2 + HP;
  float* lds_tgt = lds_b3 + 16;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.ws + a.w.w2);
    f32x4* dst = reinterpret_cast<f32x4*>(lds_w2);
    for (int i = threadIdx.x; i < HP * HP / 4; i += blockDim.x) dst[i] = src[i];
    src = reinterpret_cast<const f32x4*>(a.ws + a.w.w1z);
    dst = reinterpret_cast<f32x4*>(lds_w1z);
    for (int i = threadIdx.x; i < D * HP / 4; i += blockDim.x) dst[i] = src[i];
    src = reinterpret_cast<const f32x4*>(a.ws + a.w.w3t);
    dst = reinterpret_cast<f32x4*>(lds_w3t);
    for (int i = threadIdx.x; i < D * HP / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i < HP; i += blockDim.x) lds_b2[i] = a.ws[a.w.b2 + i];
    for (int i = threadIdx.x; i < 16; i += blockDim.x) lds_b3[i] = a.ws[a.w.b3 + i];
    for (int i = threadIdx.x; i < a.w.tgt_floats; i += blockDim.x) lds_tgt[i] = a.ws[a.w.tgt + i];
  }
  __syncthreads();
  {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < a.n_a; i += nth) a.zero_a[i] = 0.f;
    for (int64_t i = tid; i < a.n_b; i += nth) a.zero_b[i] = 0.f;
  }
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int K = a.K;
  const float factor = lds_b3[15];
  float qmean[D], qiv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qmean[j] = a.params[a.lay.vd_mean + j];
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (sd * sd);
  }
  const bool clip_p = a.grad_clipping != 0;
  const float clipv = 1e3f;
  const float* bias1 = a.ws + a.w.bias1;
  const float* utab = a.ws + a.w.utab;

  for (int64_t item = (int64_t)blockIdx.x * 4 + wv; item < a.nitems; item += (int64_t)gridDim.x * 4) {
    const int64_t tile = item / (K + 1);
    const int e = (int)(item - tile * (K + 1));
    const int64_t p = tile * 16 + c;
    const bool valid = p < a.n;
    const int64_t pc = valid ? p : a.n - 1;
    const float om = valid ? a.omega : 0.f;
    float z[D], zp[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      z[j] = a.traj[((int64_t)e * a.n + pc) * D + j];
      zp[j] = e > 0 ? a.traj[((int64_t)(e - 1) * a.n + pc) * D + j] : 0.f;
    }
    // ---- forward, keeping the activation derivatives
    const int64_t erow = a.ula ? (e > 0 ? e - 1 : 0) : e;
    const float fsn = a.ula ? 0.f : 1.f;
    const float* brow = bias1 + erow * HP;
    f32x4 u1[T], s1[T], a2[T], s2[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
      f32x4 pre = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
      for (int j = 0; j < D; ++j) pre += z[j] * *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
      if (!GEF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float dv;
          u1[t][r] = gelu_fast_both(pre[r], dv);
          s1[t][r] = dv;
        }
      } else {
        f32x4 u = *reinterpret_cast<const f32x4*>(utab + erow * HP + 16 * t + 4 * g);
        if (16 * t < D) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int nidx = 16 * t + 4 * g + r;
#pragma unroll
            for (int j = 0; j < D; ++j)
              if (j >= 16 * t && j < 16 * t + 16) u[r] = (nidx == j) ? z[j] : u[r];
          }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float dv;
          u1[t][r] = u[r] + softplus_both(pre[r], dv);
          s1[t][r] = dv;
        }
      }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) a2[t] = *reinterpret_cast<const f32x4*>(lds_b2 + 16 * t + 4 * g);
#pragma unroll
    for (int ti = 0; ti < T; ++ti) {
      asm volatile("" ::: "memory");
#pragma unroll
      for (int to = 0; to < T; ++to) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(lds_w2 + ((ti * T + to) * 64 + lane) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) a2[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r], u1[ti][r], a2[to], 0, 0, 0);
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
          float dv;
          u2t[r] = GEF ? u1[t][r] + softplus_both(a2[t][r], dv) : gelu_fast_both(a2[t][r], dv);
          s2[t][r] = dv;
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
    // ---- J_s: forward-mode pass per input coordinate;  Js[j][k] = d s_k / d z_j
    float Js[D][D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      f32x4 t1[T], t2[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const f32x4 wz = *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          t1[t][r] = s1[t][r] * wz[r];
          if (GEF && 16 * t <= j && j < 16 * t + 16) t1[t][r] += (16 * t + 4 * g + r == j) ? 1.0f : 0.f;
        }
        t2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int ti = 0; ti < T; ++ti) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int to = 0; to < T; ++to) {
          const f32x4 af = *reinterpret_cast<const f32x4*>(lds_w2 + ((ti * T + to) * 64 + lane) * 4);
#pragma unroll
          for (int r = 0; r < 4; ++r) t2[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r], t1[ti][r], t2[to], 0, 0, 0);
        }
      }
      float part[D];
#pragma unroll
      for (int k = 0; k < D; ++k) part[k] = 0.f;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f32x4 d2;
#pragma unroll
        for (int r = 0; r < 4; ++r) d2[r] = GEF ? t1[t][r] + s2[t][r] * t2[t][r] : s2[t][r] * t2[t][r];
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const f32x4 wv4 = *reinterpret_cast<const f32x4*>(lds_w3t + k * HP + 16 * t + 4 * g);
          part[k] += d2[0] * wv4[0] + d2[1] * wv4[1] + d2[2] * wv4[2] + d2[3] * wv4[3];
        }
      }
#pragma unroll
      for (int k = 0; k < D; ++k) {
        const float ds = group_sum(part[k]);
        Js[j][k] = GEF ? ds * factor : (fabsf(opre[k]) < 1e4f ? ds : 0.f);
      }
    }
    // ---- target, q
    constexpr int HN = Target<TARGET, D>::HN;
    float gp[D], gq[D], hs[HN], logp, gpraw[D], m[D];
    Target<TARGET, D>::eval_hess(z, g, lds_tgt, logp, gp, hs);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gq[j] = -(z[j] - qmean[j]) * qiv[j];
      gpraw[j] = gp[j];
      m[j] = (!clip_p || fabsf(gp[j]) < clipv) ? 1.0f : 0.f;
      if (clip_p) gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
    }
    float Hm[D][D];   // Hm[j][k] = H[j][k] m_k
#pragma unroll
    for (int k = 0; k < D; ++k) {
      float v[D], hv[D];
#pragma unroll
      for (int j = 0; j < D; ++j) v[j] = (j == k) ? m[k] : 0.f;
      Target<TARGET, D>::hvp(hs, z, v, hv);
#pragma unroll
      for (int j = 0; j < D; ++j) Hm[j][k] = hv[j];
    }
    float gprev[D], c1[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { gprev[j] = 0.f; c1[j] = 0.f; }
    if (e > 0) {
      const float pb = a.ws[a.w.beta + e - 1], pe = a.ws[a.w.eps + e - 1];
      const float inv2e = 0.5f / pe;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float ub = -1.0f * (pb * gp[j] + (1.0f - pb) * gq[j]);
        gprev[j] = -om * ((zp[j] - z[j]) + pe * (ub - sn[j])) * inv2e;
      }
#pragma unroll
      for (int j = 0; j < D; ++j) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < D; ++k) acc += (Js[j][k] + pb * Hm[j][k]) * gprev[k];
        c1[j] = gprev[j] + pe * (acc - (1.0f - pb) * qiv[j] * gprev[j]);
      }
    }
    if (e == K) {
#pragma unroll
      for (int j = 0; j < D; ++j) c1[j] -= om * gpraw[j];
    }
    if (e == 0) {
#pragma unroll
      for (int j = 0; j < D; ++j) c1[j] += om * gq[j];
    }
    if (valid && g == 0) {
      float* row = a.jac + ((int64_t)e * a.n + p) * S;
      const float be = e < K ? a.ws[a.w.beta + e] : 0.f, ee = e < K ? a.ws[a.w.eps + e] : 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
#pragma unroll
        for (int k = 0; k < D; ++k) {
          float v = -fsn * ee * Js[j][k] + ee * be * Hm[j][k];
          if (j == k) v += 1.0f - ee * (1.0f - be) * qiv[k];
          row[j * D + k] = e < K ? v : 0.f;
        }
        row[D * D + j] = c1[j];
      }
      if (e > 0) {
        float* prow = a.jac + ((int64_t)(e - 1) * a.n + p) * S;
#pragma unroll
        for (int j = 0; j < D; ++j) prow[D * D + D + j] = gprev[j];
      }
    }
  }
}

struct ScanArgs {
  const float* jac;   // [K+1][n][S]
  float* lam;         // [K+1][n][D]
  int64_t n;
  int32_t K, D;
};

template <int D, int EB>
__global__ __launch_bounds__(64) void bptt_scan_kernel(ScanArgs a) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n) return;
  constexpr int S = D * D + 2 * D;
  float lam[D];
#pragma unroll
  for (int j = 0; j < D; ++j) lam[j] = 0.f;
  // rows do not depend on lambda: fetch EB of them ahead of the EB dependent matrix-vector products — and (r04) the NEXT EB
  // while these are consumed: one thread per particle is 32 waves on the whole chip, so the launch was 17 exposed load round
  // trips (35 us at K = 256); double-buffered, the recursion runs behind one
  float cur[EB][S], nxt[EB][S];
  auto fetch = [&](float (&rows)[EB][S], int e1) {
#pragma unroll
    for (int q = 0; q < EB; ++q) {
      const int e = e1 - q;
      const float* row = a.jac + ((int64_t)(e > 0 ? e : 0) * a.n + p) * S;
#pragma unroll
      for (int i = 0; i < S; ++i) rows[q][i] = row[i];
    }
  };
  fetch(cur, a.K);
  for (int e1 = a.K; e1 >= 0; e1 -= EB) {
    if (e1 - EB >= 0) fetch(nxt, e1 - EB);
    __builtin_amdgcn_sched_barrier(0);      // (keep the next batch's loads in front of this batch's dependent chain)
#pragma unroll
    for (int q = 0; q < EB; ++q) {
      const int e = e1 - q;
      if (e < 0) break;
      float nl[D];
#pragma unroll
      for (int j = 0; j < D; ++j) {
        float acc = cur[q][D * D + j] - (e < a.K ? cur[q][D * D + D + j] : 0.f);
#pragma unroll
        for (int k = 0; k < D; ++k) acc = fmaf(cur[q][j * D + k], lam[k], acc);
        nl[j] = acc;
      }
#pragma unroll
      for (int j = 0; j < D; ++j) {
        lam[j] = nl[j];
        a.lam[((int64_t)e * a.n + p) * D + j] = nl[j];
      }
    }
#pragma unroll
    for (int q = 0; q < EB; ++q)
#pragma unroll
      for (int i = 0; i < S; ++i) cur[q][i] = nxt[q][i];
  }
}

// d = 10: the same recursion with one LANE per row of M_e — lane (particle, j) reads row j (10 floats) and its two offsets,
// the vector lambda passes through LDS inside the wave (a single-wave workgroup holds 64 / d particles): 12 loads and 10 FMAs
// per lane and step instead of 120 / 100 on one thread per particle (funnel, K = 8: 12.5 us for nine dependent steps).
template <int D>
__global__ __launch_bounds__(64) void bptt_scan_rows_kernel(ScanArgs a) {
  constexpr int S = D * D + 2 * D, PPB = 64 / D;
  __shared__ float sh[64 + D];
  const int lane = threadIdx.x, pl = lane / D, j = lane % D;
  const int64_t p0 = (int64_t)blockIdx.x * PPB + pl;
  const bool act = pl < PPB && p0 < a.n;
  const int64_t p = act ? p0 : a.n - 1;
  const int base = pl * D;
  float lam = 0.f;
  float rowv[D], c1, gg;
  auto fetch = [&](int e) {
    const float* row = a.jac + ((int64_t)(e > 0 ? e : 0) * a.n + p) * S;
#pragma unroll
    for (int k = 0; k < D; ++k) rowv[k] = row[j * D + k];
    c1 = row[D * D + j];
    gg = row[D * D + D + j];
  };
  fetch(a.K);
  for (int e = a.K; e >= 0; --e) {
    float r[D];
#pragma unroll
    for (int k = 0; k < D; ++k) r[k] = rowv[k];
    const float cc = c1 - (e < a.K ? gg : 0.f);
    if (e > 0) fetch(e - 1);                       // the next item's row arrives during this step
    sh[lane] = lam;
    __syncthreads();
    float acc = cc;
#pragma unroll
    for (int k = 0; k < D; ++k) acc = fmaf(r[k], sh[base + k], acc);
    __syncthreads();
    lam = acc;
    if (act) a.lam[((int64_t)e * a.n + p) * D + j] = acc;
  }
}

typedef void (*jac_fn)(JacArgs);
static jac_fn pick_jac(const cmcd_desc& d, int T) {
  if (d.arch == CMCD_ARCH_DDS && T == 4) {
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return bptt_jac_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2) return bptt_jac_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return bptt_jac_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_DDS, 10, 4>;
    return nullptr;
  }
  if (d.arch == CMCD_ARCH_GEFFNER) {
    if (d.target == CMCD_TARGET_GMM && d.dim == 2 && T == 2) return bptt_jac_kernel<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, 2>;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2 && T == 2) return bptt_jac_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, 2>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2 && T == 4) return bptt_jac_kernel<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, 4>;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2 && T == 4) return bptt_jac_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, 4>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2 && T == 9) return bptt_jac_kernel<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, 9>;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2 && T == 9) return bptt_jac_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, 9>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10 && T == 4) return bptt_jac_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10, 4>;
  }
  return nullptr;
}


// r04: the same recursion parallel in time for d = 2 (K >= 64): CH = 16 threads per particle, thread c owns the chunk of
// L = ceil((K + 1) / 16) evaluations e_hi(c) = K - c L, ..., and (1) composes its chunk's affine map lambda_{e_lo} = A lambda_{e_hi+1}
// + b from its rows (all fetched up front: one load round trip), (2) finds the lambda entering its chunk by walking the maps of
// the chunks above it (through LDS: <= 15 steps), (3) walks its chunk again from there and stores lambda_e for every e.  Depth
// ~3 L instead of K + 1 dependent steps, 16x the waves: 27 us (one thread per particle, 32 waves) -> see profiles.
template <int L>
__global__ __launch_bounds__(256) void bptt_scan_par_kernel(ScanArgs a) {
  constexpr int D = 2, S = D * D + 2 * D, CH = 16;
  __shared__ float maps[16][CH][6];     // [particle of the block][chunk]{A00, A01, A10, A11, b0, b1}
  const int c = threadIdx.x & (CH - 1), pl = threadIdx.x >> 4;
  const int64_t p = (int64_t)blockIdx.x * 16 + pl;
  const bool on = p < a.n;
  const int64_t pc = on ? p : a.n - 1;
  const int e_hi = a.K - c * L;
  float rows[L][S];
#pragma unroll
  for (int q = 0; q < L; ++q) {
    const int e = e_hi - q;
    const float* row = a.jac + ((int64_t)(e > 0 ? e : 0) * a.n + pc) * S;
#pragma unroll
    for (int i = 0; i < S; ++i) rows[q][i] = row[i];
  }
  // (1) compose: start from the identity, apply M_e in descending e
  float A00 = 1.f, A01 = 0.f, A10 = 0.f, A11 = 1.f, b0 = 0.f, b1 = 0.f;
#pragma unroll
  for (int q = 0; q < L; ++q) {
    const int e = e_hi - q;
    if (e < 0) break;
    const float m00 = rows[q][0], m01 = rows[q][1], m10 = rows[q][2], m11 = rows[q][3];
    const float c0 = rows[q][4] - (e < a.K ? rows[q][6] : 0.f), c1 = rows[q][5] - (e < a.K ? rows[q][7] : 0.f);
    const float n00 = fmaf(m00, A00, m01 * A10), n01 = fmaf(m00, A01, m01 * A11);
    const float n10 = fmaf(m10, A00, m11 * A10), n11 = fmaf(m10, A01, m11 * A11);
    const float nb0 = fmaf(m00, b0, fmaf(m01, b1, c0)), nb1 = fmaf(m10, b0, fmaf(m11, b1, c1));
    A00 = n00; A01 = n01; A10 = n10; A11 = n11; b0 = nb0; b1 = nb1;
  }
  maps[pl][c][0] = A00; maps[pl][c][1] = A01; maps[pl][c][2] = A10; maps[pl][c][3] = A11; maps[pl][c][4] = b0; maps[pl][c][5] = b1;
  __syncthreads();
  // (2) lambda entering chunk c (lambda_{K+1} = 0 enters chunk 0)
  float l0 = 0.f, l1 = 0.f;
  for (int cc = 0; cc < c; ++cc) {
    const float* m = maps[pl][cc];
    const float t0 = fmaf(m[0], l0, fmaf(m[1], l1, m[4])), t1 = fmaf(m[2], l0, fmaf(m[3], l1, m[5]));
    l0 = t0; l1 = t1;
  }
  // (3) the chunk's own steps, in the order (and with the arithmetic) of the serial kernel
#pragma unroll
  for (int q = 0; q < L; ++q) {
    const int e = e_hi - q;
    if (e < 0) break;
    const float c0 = rows[q][4] - (e < a.K ? rows[q][6] : 0.f), c1 = rows[q][5] - (e < a.K ? rows[q][7] : 0.f);
    const float n0 = fmaf(rows[q][1], l1, fmaf(rows[q][0], l0, c0));
    const float n1 = fmaf(rows[q][3], l1, fmaf(rows[q][2], l0, c1));
    l0 = n0; l1 = n1;
    if (on) {
      a.lam[((int64_t)e * a.n + p) * D + 0] = n0;
      a.lam[((int64_t)e * a.n + p) * D + 1] = n1;
    }
  }
}

int bptt_jac_scan_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, int64_t n, int64_t nitems,
                         const float* params, const float* ws_fwd, const float* traj, float* jac, float* lam,
                         float omega_scalar, float* zero_a, int64_t n_a, float* zero_b, int64_t n_b, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int D = d.dim, K = d.nbridges, HP = 16 * w.T;
  jac_fn jf = pick_jac(d, w.T);
  if (!jf) return CMCD_ERR_UNSUPPORTED;
  JacArgs ja{params, ws_fwd, traj, jac, lay, w, n, nitems, K, d.grad_clipping, d.mode == CMCD_MODE_ULA_SN ? 2 : 0, omega_scalar};
  ja.zero_a = zero_a; ja.n_a = n_a; ja.zero_b = zero_b; ja.n_b = n_b;
  const size_t jl = size_t(HP * HP + 2 * D * HP + HP + 16 + w.tgt_floats) * 4;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(jf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jl) != hipSuccess)
    return CMCD_ERR_HIP;
  const int64_t jb = (nitems + 3) / 4;
  hipLaunchKernelGGL(jf, dim3((unsigned)(jb < 2048 ? jb : 2048)), dim3(256), jl, stream, ja);
  ScanArgs sa{jac, lam, n, K, D};
#ifndef CMCD_SCAN_PAR
#define CMCD_SCAN_PAR 1
#endif
  const int Lc = (K + 1 + 15) / 16;     // evaluations per chunk of the parallel form
  if (D == 2 && CMCD_SCAN_PAR && K >= 64 && Lc <= 17)
    hipLaunchKernelGGL(bptt_scan_par_kernel<17>, dim3((unsigned)((n + 15) / 16)), dim3(256), 0, stream, sa);
  else if (D == 2) hipLaunchKernelGGL((bptt_scan_kernel<2, 16>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, sa);
  else if (D == 10) hipLaunchKernelGGL(bptt_scan_rows_kernel<10>, dim3((unsigned)((n + 5) / 6)), dim3(64), 0, stream, sa);
  else return CMCD_ERR_UNSUPPORTED;
  return CMCD_OK;
}

}  // namespace cmcd
