// lgcp path (d = 1600, geffner net of width 1620): per-bridge launch sequence instead of a
// register-resident tile, because one evaluation touches 41.5 MB of weights
// (K^-1 10.2 MB + W1 10.4 MB + W2 10.5 MB + W3 10.4 MB) that cannot live on-chip per tile.
//
// Per evaluation i = 0..K (one evaluation serves the backward kernel of step i-1 and the forward
// kernel of step i, as in traj_kernel):
//   gemm A: [x - mu0] K^-1 -> kr          and   x W1[:d] + bias1_i -> pre1    (one launch, two segments)
//   gemm B: u1 = u + softplus(pre1) (formed in the prologue);  u1 W2 + b2 -> pre2
//   gemm C: u2 = u1 + softplus(pre2) (prologue);  factor_sn (u2 W3 + b3) -> sn
//   step  : grad log p = -kr + counts - a e^x  (model_handler.py:386-396, cp_utils.py:102-104),
//           close step i-1, draw eps_i = normal(G_i, (1600,)) (800 Threefry blocks), open step i.
// The skinny GEMM ([<=24 particles] x [K] x [N]) is weight-bandwidth bound: a workgroup owns 64
// output columns, its 16 waves split K, every lane keeps one column's partial sums for all particles
// in registers, W rows are read once as coalesced 256-byte rows, the activation slice is staged in
// wave-private LDS and broadcast, and the 16 partial tiles are summed through LDS (fixed order).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

constexpr int kMP = 24;      // particles per pass (rows of the skinny GEMM)
constexpr int kGemmWaves = 16;
constexpr int kChunk = 32;   // k rows staged per round

enum { PRO_X_MINUS_MU = 0, PRO_X = 1, PRO_U1 = 2, PRO_U2 = 3 };

struct GemmSeg {
  const float* W;      // [Kdim][ldw]
  const float* bias;   // [N] or nullptr
  float* out;          // [kMP][ldo]
  int N, ldw, ldo, pro;
  float scale;
};

struct GemmArgs {
  GemmSeg seg[2];
  int nblk0;           // blocks of segment 0
  const float* x;      // [kMP][D]
  const float* pre1;   // [kMP][IN]
  const float* pre2;   // [kMP][IN]
  const float* emb;    // [E] embedding row of this evaluation
  const float* factor; // device scalar factor_sn (gemm C) or nullptr
  float mu0;
  int M, Kdim, D, IN;
};

__device__ __forceinline__ float lgcp_a(const GemmArgs& a, int pro, int m, int k) {
  if (m >= a.M) return 0.f;
  if (pro == PRO_X_MINUS_MU) return a.x[m * a.D + k] - a.mu0;
  if (pro == PRO_X) return a.x[m * a.D + k];
  const float u = k < a.D ? a.x[m * a.D + k] : a.emb[k - a.D];
  float v = u + softplus(a.pre1[m * a.IN + k]);                    // nn.py:45-47
  if (pro == PRO_U2) v += softplus(a.pre2[m * a.IN + k]);          // nn.py:48-50
  return v;
}

__global__ __launch_bounds__(64 * kGemmWaves) void lgcp_gemm_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int s = blockIdx.x < a.nblk0 ? 0 : 1;
  const GemmSeg sg = a.seg[s];
  const int n0 = (blockIdx.x - (s ? a.nblk0 : 0)) * 64;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* As = lds + wv * (kChunk * kMP);                             // wave-private [kChunk][kMP]
  float* red = lds + kGemmWaves * kChunk * kMP;                      // [16][kMP][64]
  const int ks = (a.Kdim + kGemmWaves - 1) / kGemmWaves;
  const int k_lo = wv * ks, k_hi = min(a.Kdim, k_lo + ks);
  const int n = n0 + lane;
  const bool ncol = n < sg.N;

  float acc[kMP];
#pragma unroll
  for (int m = 0; m < kMP; ++m) acc[m] = 0.f;

  for (int kc = k_lo; kc < k_hi; kc += kChunk) {
    const int len = min(kChunk, k_hi - kc);
    // W rows of this chunk: issue all loads first (one coalesced 256-byte row per k)
    float wrow[kChunk];
#pragma unroll
    for (int kk = 0; kk < kChunk; ++kk)
      wrow[kk] = (kk < len && ncol) ? sg.W[(int64_t)(kc + kk) * sg.ldw + n] : 0.f;
    // stage the activation slice (prologue applied) into wave-private LDS, k-major
    for (int e = lane; e < len * kMP; e += 64) {
      const int m = e / len, kk = e - m * len;
      As[kk * kMP + m] = lgcp_a(a, sg.pro, m, kc + kk);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int kk = 0; kk < kChunk; ++kk) {
      if (kk < len) {
#pragma unroll
        for (int m4 = 0; m4 < kMP / 4; ++m4) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(As + kk * kMP + 4 * m4);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * m4 + q] = fmaf(av[q], wrow[kk], acc[4 * m4 + q]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int m = 0; m < kMP; ++m) red[(wv * kMP + m) * 64 + lane] = acc[m];
  __syncthreads();
  const float fac = a.factor ? a.factor[0] : 1.0f;
  for (int o = threadIdx.x; o < kMP * 64; o += blockDim.x) {
    const int m = o >> 6, nl = o & 63;
    if (m < a.M && n0 + nl < sg.N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kGemmWaves; ++w) v += red[(w * kMP + m) * 64 + nl];  // fixed order
      if (sg.bias) v += sg.bias[n0 + nl];
      sg.out[(int64_t)m * sg.ldo + n0 + nl] = v * sg.scale * fac;
    }
  }
}

// ------------------------------------------------------------------------------------------
// per-bridge first-layer bias of the 1620-wide geffner net: b1 + emb[min(i, K-1)] W1[d:, :]
// ------------------------------------------------------------------------------------------
struct LgcpPrepArgs {
  const float* params;
  float* bias1;  // [K+1][IN]
  cmcd_layout lay;
  int D, E, K, IN;
};

__global__ void lgcp_prep_kernel(LgcpPrepArgs a) {
  const int row = blockIdx.y, n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= a.IN) return;
  const int ie = row < a.K ? row : a.K - 1;
  const float* emb = a.params + a.lay.g_emb + (int64_t)ie * a.E;
  float b = a.params[a.lay.g_b1 + n];
  for (int j = 0; j < a.E; ++j) b = fmaf(emb[j], a.params[a.lay.g_w1 + (int64_t)(a.D + j) * a.IN + n], b);
  a.bias1[(int64_t)row * a.IN + n] = b;
}

// ------------------------------------------------------------------------------------------
// per-particle state kernels: one 256-thread workgroup per particle
// ------------------------------------------------------------------------------------------
struct LgcpStateArgs {
  const int32_t* seeds;      // [M] (this pass)
  const float* params;
  const float* tc;           // {Kinv[d,d], counts[d], mu0, a, lognorm}
  const float* sched;        // [K][8]
  float* x;                  // [kMP][D]   current z
  float* xp;                 // [kMP][D]   previous z
  const float* kr;           // [kMP][D]   K^-1 (x - mu0)
  const float* sn;           // [kMP][D]   score net output
  float* w;                  // [kMP]
  float* fklp;               // [kMP]
  uint32_t* keys;            // [kMP][2]   gen key of the chain
  float* out_loss;           // [M]
  float* out_z;              // [M][D]
  double* partials;          // [M][5]
  cmcd_layout lay;
  int M, D, K, i, var_mode, grad_clipping;
};

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// z0 = mean + std * normal(A, (D,)); w = -log q(z0); gen = second(split(first(split(B))))
// (mcdboundingmachine.py:151-162, mcd_cais.py:94, diag_gauss.py:26-62)
__global__ __launch_bounds__(256) void lgcp_init_kernel(LgcpStateArgs a) {
  __shared__ float sh[4];
  const int p = blockIdx.x, D = a.D, H = (D + 1) / 2;
  const uint32_t seed = (uint32_t)a.seeds[p];
  uint32_t s0 = 0, s1 = 2, t0 = 1, t1 = 3;
  threefry2x32(0u, seed, s0, s1);   // block (0,2) -> out0, out2
  threefry2x32(0u, seed, t0, t1);   // block (1,3) -> out1, out3
  const uint32_t a0 = s0, a1 = t0, b0 = s1, b1 = t1;   // A = (out0,out1), B = (out2,out3)
  float acc = 0.f;
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    uint32_t y0 = j, y1 = (H + j < D) ? H + j : 0;
    threefry2x32(a0, a1, y0, y1);
    const int idx[2] = {j, H + j};
    const uint32_t bits[2] = {y0, y1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (idx[q] < D) {
        const float mean = a.params[a.lay.vd_mean + idx[q]];
        const float sd = expf(a.params[a.lay.vd_logdiag + idx[q]]);
        const float z = sd * bits_to_normal(bits[q]) + mean;
        a.x[p * D + idx[q]] = z;
        const float dz = z - mean;
        acc += -(dz * dz) / (2.0f * sd * sd) - logf(sd) - kHalfLog2Pi;
      }
    }
  }
  const float lq = block_sum_256(acc, sh);
  if (threadIdx.x == 0) {
    a.w[p] = -lq;
    a.fklp[p] = 0.f;
    uint32_t c0 = 0, c2 = 2, c1 = 1, c3 = 3;
    threefry2x32(b0, b1, c0, c2);
    threefry2x32(b0, b1, c1, c3);       // C = (c0, c1)
    uint32_t g0 = 0, g2 = 2, g1 = 1, g3 = 3;
    threefry2x32(c0, c1, g0, g2);
    threefry2x32(c0, c1, g1, g3);       // gen = second(split(C)) = (g2, g3)
    a.keys[2 * p] = g2;
    a.keys[2 * p + 1] = g3;
  }
}

// evaluation i at z_i: closes step i-1, opens step i (or, at i = K, writes the outputs)
__global__ __launch_bounds__(256) void lgcp_step_kernel(LgcpStateArgs a) {
  __shared__ float sh[4];
  const int p = blockIdx.x, D = a.D, H = (D + 1) / 2, i = a.i;
  const float* counts = a.tc + (int64_t)D * D;
  const float mu0 = a.tc[(int64_t)D * D + D], pa = a.tc[(int64_t)D * D + D + 1];
  const float lognorm = a.tc[(int64_t)D * D + D + 2];
  const float clipv = a.var_mode ? 1e2f : 1e3f;
  const bool clip_p = a.grad_clipping != 0, clip_q = clip_p && a.var_mode;
  const bool last = i == a.K;
  const float* sp = a.sched + 8 * (i > 0 ? i - 1 : 0);
  const float pbeta = sp[0], peps = sp[1], pcst = sp[3], pinv2s2 = sp[4];
  const float* sc = a.sched + 8 * (last ? a.K - 1 : i);
  const float beta = sc[0], eps = sc[1], sig = sc[2], cst = sc[3], inv2s2 = sc[4];

  // (G, H) = split(gen); eps_i = normal(G, (D,)); gen' = second(split(H))   mcd_cais.py:66-67,87
  const uint32_t k0 = a.keys[2 * p], k1 = a.keys[2 * p + 1];
  uint32_t g0 = 0, h0 = 2, g1 = 1, h1 = 3;
  threefry2x32(k0, k1, g0, h0);
  threefry2x32(k0, k1, g1, h1);

  float bk_acc = 0.f, fk_acc = 0.f, lp_acc = 0.f;
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    uint32_t y0 = j, y1 = (H + j < D) ? H + j : 0;
    if (!last) threefry2x32(g0, g1, y0, y1);
    const int idx[2] = {j, H + j};
    const uint32_t bits[2] = {y0, y1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int e = idx[q];
      if (e < D) {
        const float z = a.x[p * D + e];
        const float kr = a.kr[p * D + e];
        const float ez = expf(z);
        float gp = -kr + counts[e] - pa * ez;                                  // grad log p
        const float mean = a.params[a.lay.vd_mean + e];
        const float sd = expf(a.params[a.lay.vd_logdiag + e]);
        float gq = -(z - mean) / (sd * sd);
        if (clip_p) gp = fminf(fmaxf(gp, -clipv), clipv);
        if (clip_q) gq = fminf(fmaxf(gq, -clipv), clipv);
        const float s = a.sn[p * D + e];
        if (i > 0) {   // backward kernel of step i-1                           mcd_cais.py:71-86
          const float ub = -1.0f * (pbeta * gp + (1.0f - pbeta) * gq);
          const float bk = z - peps * ub + peps * s;
          const float db = a.xp[p * D + e] - bk;
          bk_acc += -(db * db) * pinv2s2 - pcst;
        }
        if (last) {    // log p(z_K)                                            model_handler.py:386-396
          lp_acc += -0.5f * (z - mu0) * kr + z * counts[e] - pa * ez;
        } else {       // forward kernel of step i                              mcd_cais.py:52-67
          const float uf = -1.0f * (beta * gp + (1.0f - beta) * gq);
          const float fk = z - eps * uf - eps * s;
          const float zn = fk + sig * bits_to_normal(bits[q]);
          const float df = zn - fk;
          fk_acc += -(df * df) * inv2s2 - cst;
          a.xp[p * D + e] = z;
          a.x[p * D + e] = zn;
        }
      }
    }
  }
  const float bk_lp = block_sum_256(bk_acc, sh);
  const float fk_lp = block_sum_256(fk_acc, sh);
  const float lp = block_sum_256(lp_acc, sh);
  if (threadIdx.x == 0) {
    float w = a.w[p];
    if (i > 0) w += bk_lp - a.fklp[p];
    if (!last) {
      a.fklp[p] = fk_lp;
      uint32_t n0 = 0, n2 = 2, n1 = 1, n3 = 3;
      threefry2x32(h0, h1, n0, n2);
      threefry2x32(h0, h1, n1, n3);
      a.keys[2 * p] = n2;
      a.keys[2 * p + 1] = n3;
      a.w[p] = w;
    } else {
      w += lp + lognorm;                        // + log p(z_K)   mcdboundingmachine.py:178
      const float loss = -w;
      a.out_loss[p] = loss;
      double* o = a.partials + (int64_t)p * CMCD_NSTATS;
      o[0] = isfinite(loss) ? 1.0 : 0.0;
      o[1] = loss;
      o[2] = (double)loss * (double)loss;
      o[3] = -(double)loss;
      o[4] = isfinite(loss) ? 1.0 : 0.0;
    }
  }
  if (last)
    for (int e = threadIdx.x; e < D; e += blockDim.x) a.out_z[(int64_t)p * D + e] = a.x[p * D + e];
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
struct LgcpWs {
  int64_t bias1, x, xp, kr, pre1, pre2, sn, w, fklp, keys, partials, total;
};

static LgcpWs lgcp_ws(const cmcd_desc& d, int64_t n, int64_t base) {
  const int64_t D = d.dim, IN = D + d.emb_dim, K = d.nbridges;
  LgcpWs w;
  int64_t o = base;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  w.bias1 = take((K + 1) * IN);
  w.x = take(kMP * D); w.xp = take(kMP * D); w.kr = take(kMP * D);
  w.pre1 = take(kMP * IN); w.pre2 = take(kMP * IN); w.sn = take(kMP * D);
  w.w = take(kMP); w.fklp = take(kMP); w.keys = take(2 * kMP);
  o = (o + 1) & ~int64_t(1);
  w.partials = take(n * CMCD_NSTATS * 2);
  w.total = o;
  return w;
}

int64_t lgcp_workspace_floats(const cmcd_desc& d, int64_t n, int64_t base) { return lgcp_ws(d, n, base).total; }

int lgcp_forward(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const int32_t* seeds, int64_t n,
                 const float* params, const float* tc, float* ws, float* out_loss, float* out_z,
                 double** partials_out, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int D = d.dim, E = d.emb_dim, IN = D + E, K = d.nbridges;
  const LgcpWs w = lgcp_ws(d, n, sw.total_floats);
  {
    LgcpPrepArgs pa{params, ws + w.bias1, lay, D, E, K, IN};
    hipLaunchKernelGGL(lgcp_prep_kernel, dim3((IN + 255) / 256, K + 1), dim3(256), 0, stream, pa);
  }
  const size_t gemm_lds = size_t(kGemmWaves * kChunk * kMP + kGemmWaves * kMP * 64) * 4;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lgcp_gemm_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_lds);
  if (e != hipSuccess) return CMCD_ERR_HIP;
  const float* kinv = tc;
  double* partials = reinterpret_cast<double*>(ws + w.partials);
  *partials_out = partials;
  // mu0 is needed on the host side of the launch (prologue constant): it is a model constant,
  // log(126) - 0.5 * 1.91 (model_handler.py:346); the device copy in tc is used by the step kernel.
  const float mu0 = 3.8812819069514780f;

  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    LgcpStateArgs st{};
    st.seeds = seeds + base; st.params = params; st.tc = tc; st.sched = ws + sw.sched;
    st.x = ws + w.x; st.xp = ws + w.xp; st.kr = ws + w.kr; st.sn = ws + w.sn;
    st.w = ws + w.w; st.fklp = ws + w.fklp; st.keys = reinterpret_cast<uint32_t*>(ws + w.keys);
    st.out_loss = out_loss + base; st.out_z = out_z + base * D; st.partials = partials + base * CMCD_NSTATS;
    st.lay = lay; st.M = M; st.D = D; st.K = K;
    st.var_mode = d.mode == CMCD_MODE_CAIS_VAR_SN ? 1 : 0; st.grad_clipping = d.grad_clipping;
    hipLaunchKernelGGL(lgcp_init_kernel, dim3(M), dim3(256), 0, stream, st);

    GemmArgs g{};
    g.x = ws + w.x; g.pre1 = ws + w.pre1; g.pre2 = ws + w.pre2; g.mu0 = mu0; g.M = M; g.D = D; g.IN = IN;
    for (int i = 0; i <= K; ++i) {
      const int ie = i < K ? i : K - 1;
      g.emb = params + lay.g_emb + (int64_t)ie * E;
      // A: [x - mu0] Kinv -> kr  |  x W1[:D] + bias1_i -> pre1
      g.Kdim = D; g.factor = nullptr;
      g.seg[0] = GemmSeg{kinv, nullptr, ws + w.kr, D, D, D, PRO_X_MINUS_MU, 1.0f};
      g.seg[1] = GemmSeg{params + lay.g_w1, ws + w.bias1 + (int64_t)i * IN, ws + w.pre1, IN, IN, IN, PRO_X, 1.0f};
      g.nblk0 = (D + 63) / 64;
      hipLaunchKernelGGL(lgcp_gemm_kernel, dim3(g.nblk0 + (IN + 63) / 64), dim3(64 * kGemmWaves), gemm_lds, stream, g);
      // B: u1 W2 + b2 -> pre2
      g.Kdim = IN;
      g.seg[0] = GemmSeg{params + lay.g_w2, params + lay.g_b2, ws + w.pre2, IN, IN, IN, PRO_U1, 1.0f};
      g.nblk0 = (IN + 63) / 64;
      hipLaunchKernelGGL(lgcp_gemm_kernel, dim3(g.nblk0), dim3(64 * kGemmWaves), gemm_lds, stream, g);
      // C: factor_sn (u2 W3 + b3) -> sn
      g.seg[0] = GemmSeg{params + lay.g_w3, params + lay.g_b3, ws + w.sn, D, D, D, PRO_U2, 1.0f};
      g.nblk0 = (D + 63) / 64;
      g.factor = params + lay.g_factor;
      hipLaunchKernelGGL(lgcp_gemm_kernel, dim3(g.nblk0), dim3(64 * kGemmWaves), gemm_lds, stream, g);
      st.i = i;
      hipLaunchKernelGGL(lgcp_step_kernel, dim3(M), dim3(256), 0, stream, st);
    }
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

}  // namespace cmcd
