// lgcp path (d = 1600, geffner net of width 1620): per-bridge launch sequence instead of a
// register-resident tile, because one evaluation touches 41.5 MB of weights
// (K^-1 10.2 MB + W1 10.4 MB + W2 10.5 MB + W3 10.4 MB) that cannot live on-chip per tile.
//
// Per evaluation i = 0..K (one evaluation serves the backward kernel of step i-1 and the forward
// kernel of step i, as in traj_kernel), 7 launches:
//   act 0 : x - mu0
//   gemm A: [x - mu0] K^-1 -> kr slabs      and   x W1[:d] -> pre1 slabs      (one launch, two segments)
//   act 1 : pre1 = sum(slabs) + bias1_i ;  u1 = [x; emb_i] + softplus(pre1)           nn.py:45-47,68
//   gemm B: u1 W2 -> pre2 slabs
//   act 2 : pre2 = sum(slabs) + b2 ;       u2 = u1 + softplus(pre2)                   nn.py:48-50
//   gemm C: u2 W3 -> sn slabs
//   step  : sn = factor_sn (sum(slabs) + b3); grad log p = -kr + counts - a e^x
//           (model_handler.py:386-396, cp_utils.py:102-104); close step i-1, draw
//           eps_i = normal(G_i, (1600,)) (800 Threefry blocks), open step i.
// The skinny GEMM ([<=24 particles] x [K] x [N]) is weight-bandwidth bound: a workgroup owns 64
// output columns of one of kSplit K-slices, its 16 waves split the slice, every lane keeps one
// column's partial sums for all particles in registers, W rows are read once as coalesced 256-byte
// rows, the activation slice is staged in wave-private LDS and broadcast, the 16 partial tiles are
// summed through LDS and written as a partial slab; the consumer sums the kSplit slabs in a fixed
// order (bitwise deterministic, no atomics).
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

constexpr int kSplit = 4;    // K split across workgroups (partial slabs, summed in fixed order downstream)

struct GemmSeg {
  const float* A;      // [kMP][lda]  activations (already formed)
  const float* W;      // [Kdim][ldw]
  float* out;          // [kSplit][kMP][ldo] partial slabs
  int N, lda, ldw, ldo;
};

struct GemmArgs {
  GemmSeg seg[2];
  int nblk0;           // column blocks of segment 0
  int M, Kdim;
};

// out[ks][m][n] = sum_{k in slice ks, wave w} A[m][k] W[k][n]   (no bias: added when the slabs are summed)
__global__ __launch_bounds__(64 * kGemmWaves) void lgcp_gemm_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int s = blockIdx.x < a.nblk0 ? 0 : 1;
  const GemmSeg sg = a.seg[s];
  const int n0 = (blockIdx.x - (s ? a.nblk0 : 0)) * 64;
  const int ksplit = blockIdx.y;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* As = lds + wv * (kChunk * kMP);                             // wave-private [kChunk][kMP]
  float* red = lds + kGemmWaves * kChunk * kMP;                      // [16][kMP][64]
  const int kslice = (a.Kdim + kSplit - 1) / kSplit;
  const int s_lo = ksplit * kslice, s_hi = min(a.Kdim, s_lo + kslice);
  const int ks = (s_hi - s_lo + kGemmWaves - 1) / kGemmWaves;
  const int k_lo = s_lo + wv * ks, k_hi = min(s_hi, k_lo + ks);
  const int n = n0 + lane;
  const bool ncol = n < sg.N;

  float acc[kMP];
#pragma unroll
  for (int m = 0; m < kMP; ++m) acc[m] = 0.f;

  for (int kc = k_lo; kc < k_hi; kc += kChunk) {
    const int len = min(kChunk, k_hi - kc);
    float wrow[kChunk];
#pragma unroll
    for (int kk = 0; kk < kChunk; ++kk)
      wrow[kk] = (kk < len && ncol) ? sg.W[(int64_t)(kc + kk) * sg.ldw + n] : 0.f;
    // stage A[:, kc:kc+len] into wave-private LDS, k-major: lane -> (m = e / 32, kk = e % 32), coalesced in k
#pragma unroll
    for (int e0 = 0; e0 < kChunk * kMP; e0 += 64) {
      const int e = e0 + lane, m = e >> 5, kk = e & 31;
      As[kk * kMP + m] = (m < a.M && kk < len) ? sg.A[(int64_t)m * sg.lda + kc + kk] : 0.f;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
    for (int kk = 0; kk < kChunk; ++kk) {
#pragma unroll
      for (int m4 = 0; m4 < kMP / 4; ++m4) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(As + kk * kMP + 4 * m4);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 * m4 + q] = fmaf(av[q], wrow[kk], acc[4 * m4 + q]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
#pragma unroll
  for (int m = 0; m < kMP; ++m) red[(wv * kMP + m) * 64 + lane] = acc[m];
  __syncthreads();
  float* slab = sg.out + (int64_t)ksplit * kMP * sg.ldo;
  for (int o = threadIdx.x; o < kMP * 64; o += blockDim.x) {
    const int m = o >> 6, nl = o & 63;
    if (m < a.M && n0 + nl < sg.N) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < kGemmWaves; ++w) v += red[(w * kMP + m) * 64 + nl];  // fixed order
      slab[(int64_t)m * sg.ldo + n0 + nl] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// activation kernels: sum the split-K slabs in fixed order, add the bias, form the next GEMM's input
// ------------------------------------------------------------------------------------------
struct ActArgs {
  const float* x;        // [kMP][D]
  const float* emb;      // [E]
  const float* slab_a;   // [kSplit][kMP][lda]  partials of the previous GEMM (segment a)
  const float* bias_a;   // [IN] or nullptr
  float* sum_a;          // [kMP][IN] summed pre-activation (kept: the next layer's residual needs it)
  const float* u_prev;   // [kMP][IN] previous layer's u (act2 only)
  float* u_out;          // [kMP][IN]
  float* xm;             // [kMP][D]   x - mu0 (act0 only)
  float mu0;
  int M, D, IN, mode;    // mode 0: xm = x - mu0 ; 1: u1 = u + softplus(pre1) ; 2: u2 = u1 + softplus(pre2)
};

__global__ void lgcp_act_kernel(ActArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (a.mode == 0) {
    if (idx < a.M * a.D) a.xm[idx] = a.x[idx] - a.mu0;
    return;
  }
  if (idx >= a.M * a.IN) return;
  const int m = idx / a.IN, k = idx - m * a.IN;
  float pre = a.bias_a[k];
#pragma unroll
  for (int ks = 0; ks < kSplit; ++ks) pre += a.slab_a[((int64_t)ks * kMP + m) * a.IN + k];
  a.sum_a[idx] = pre;
  float u;
  if (a.mode == 1) u = k < a.D ? a.x[m * a.D + k] : a.emb[k - a.D];   // u = [x; emb_i]      nn.py:68-69
  else u = a.u_prev[idx];
  a.u_out[idx] = u + softplus(pre);                                   // nn.py:45-50
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
  const float* kr;           // [kSplit][kMP][D]   partial slabs of K^-1 (x - mu0)
  const float* sn;           // [kSplit][kMP][D]   partial slabs of u2 W3
  const float* b3;           // [D]
  const float* factor;       // factor_sn (device scalar)
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
        float kr = 0.f, s = a.b3[e];
#pragma unroll
        for (int ks = 0; ks < kSplit; ++ks) {   // fixed-order sum of the split-K slabs
          kr += a.kr[((int64_t)ks * kMP + p) * D + e];
          s += a.sn[((int64_t)ks * kMP + p) * D + e];
        }
        s *= a.factor[0];                       // factor_sn (u2 W3 + b3)           nn.py:70
        const float ez = expf(z);
        float gp = -kr + counts[e] - pa * ez;                                  // grad log p
        const float mean = a.params[a.lay.vd_mean + e];
        const float sd = expf(a.params[a.lay.vd_logdiag + e]);
        float gq = -(z - mean) / (sd * sd);
        if (clip_p) gp = fminf(fmaxf(gp, -clipv), clipv);
        if (clip_q) gq = fminf(fmaxf(gq, -clipv), clipv);
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
  int64_t bias1, x, xp, xm, u1, u2, pre1, pre2, kr, slab1, slab2, sn, w, fklp, keys, partials, total;
};

static LgcpWs lgcp_ws(const cmcd_desc& d, int64_t n, int64_t base) {
  const int64_t D = d.dim, IN = D + d.emb_dim, K = d.nbridges;
  LgcpWs w;
  int64_t o = base;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  w.bias1 = take((K + 1) * IN);
  w.x = take(kMP * D); w.xp = take(kMP * D); w.xm = take(kMP * D);
  w.u1 = take(kMP * IN); w.u2 = take(kMP * IN); w.pre1 = take(kMP * IN); w.pre2 = take(kMP * IN);
  w.kr = take(kSplit * kMP * D); w.slab1 = take(kSplit * kMP * IN); w.slab2 = take(kSplit * kMP * IN);
  w.sn = take(kSplit * kMP * D);
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
  // mu0 is a model constant, log(126) - 0.5 * 1.91 (model_handler.py:346); the step kernel reads the
  // device copy in tc, the activation kernel takes it as a launch argument.
  const float mu0 = 3.8812819069514780f;
  const dim3 gblock(64 * kGemmWaves);
  const int cbD = (D + 63) / 64, cbIN = (IN + 63) / 64;

  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    LgcpStateArgs st{};
    st.seeds = seeds + base; st.params = params; st.tc = tc; st.sched = ws + sw.sched;
    st.x = ws + w.x; st.xp = ws + w.xp; st.kr = ws + w.kr; st.sn = ws + w.sn;
    st.b3 = params + lay.g_b3; st.factor = params + lay.g_factor;
    st.w = ws + w.w; st.fklp = ws + w.fklp; st.keys = reinterpret_cast<uint32_t*>(ws + w.keys);
    st.out_loss = out_loss + base; st.out_z = out_z + base * D; st.partials = partials + base * CMCD_NSTATS;
    st.lay = lay; st.M = M; st.D = D; st.K = K;
    st.var_mode = d.mode == CMCD_MODE_CAIS_VAR_SN ? 1 : 0; st.grad_clipping = d.grad_clipping;
    hipLaunchKernelGGL(lgcp_init_kernel, dim3(M), dim3(256), 0, stream, st);

    ActArgs act{};
    act.x = ws + w.x; act.mu0 = mu0; act.M = M; act.D = D; act.IN = IN; act.xm = ws + w.xm;
    GemmArgs g{};
    g.M = M;
    for (int i = 0; i <= K; ++i) {
      const int ie = i < K ? i : K - 1;
      act.emb = params + lay.g_emb + (int64_t)ie * E;
      // x - mu0
      act.mode = 0;
      hipLaunchKernelGGL(lgcp_act_kernel, dim3((M * D + 255) / 256), dim3(256), 0, stream, act);
      // A: [x - mu0] Kinv -> kr slabs  |  x W1[:D] -> pre1 slabs
      g.Kdim = D;
      g.seg[0] = GemmSeg{ws + w.xm, kinv, ws + w.kr, D, D, D, D};
      g.seg[1] = GemmSeg{ws + w.x, params + lay.g_w1, ws + w.slab1, IN, D, IN, IN};
      g.nblk0 = cbD;
      hipLaunchKernelGGL(lgcp_gemm_kernel, dim3(cbD + cbIN, kSplit), gblock, gemm_lds, stream, g);
      // u1 = u + softplus(pre1 + bias1_i)
      act.mode = 1; act.slab_a = ws + w.slab1; act.bias_a = ws + w.bias1 + (int64_t)i * IN;
      act.sum_a = ws + w.pre1; act.u_prev = nullptr; act.u_out = ws + w.u1;
      hipLaunchKernelGGL(lgcp_act_kernel, dim3((M * IN + 255) / 256), dim3(256), 0, stream, act);
      // B: u1 W2 -> pre2 slabs
      g.Kdim = IN;
      g.seg[0] = GemmSeg{ws + w.u1, params + lay.g_w2, ws + w.slab2, IN, IN, IN, IN};
      g.nblk0 = cbIN;
      hipLaunchKernelGGL(lgcp_gemm_kernel, dim3(cbIN, kSplit), gblock, gemm_lds, stream, g);
      // u2 = u1 + softplus(pre2 + b2)
      act.mode = 2; act.slab_a = ws + w.slab2; act.bias_a = params + lay.g_b2;
      act.sum_a = ws + w.pre2; act.u_prev = ws + w.u1; act.u_out = ws + w.u2;
      hipLaunchKernelGGL(lgcp_act_kernel, dim3((M * IN + 255) / 256), dim3(256), 0, stream, act);
      // C: u2 W3 -> sn slabs (bias and factor_sn applied in the step kernel)
      g.seg[0] = GemmSeg{ws + w.u2, params + lay.g_w3, ws + w.sn, D, IN, D, D};
      g.nblk0 = cbD;
      hipLaunchKernelGGL(lgcp_gemm_kernel, dim3(cbD, kSplit), gblock, gemm_lds, stream, g);
      st.i = i;
      hipLaunchKernelGGL(lgcp_step_kernel, dim3(M), dim3(256), 0, stream, st);
    }
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}


// ------------------------------------------------------------------------------------------
// Mean-field VI on the lgcp target (nbridges = 0; /root/reference/src/boundingmachine.py:73-111 with
// /root/reference/src/main.py:82-109): z = mean + std e, loss = log q(z) - log p(z), and under the
// reparameterisation d loss / d mean = -grad log p(z), d loss / d logdiag = -1 - grad log p(z) std e.
// Per pass of <= kMP particles: init (z, -log q) -> x - mu0 -> one skinny GEMM with K^-1 -> finish.
// ------------------------------------------------------------------------------------------
struct LgcpMfviArgs {
  const float* params;
  const float* tc;
  const float* x;        // [kMP][D]
  const float* kr;       // [kSplit][kMP][D]
  const float* w;        // [kMP]  -log q(z)
  float* out_loss;       // [M]
  float* out_z;          // [M][D]
  double* partials;      // [M][5]
  float* gbuf;           // [M][2][D] per-particle gradient rows (nullable)
  int64_t o_mean;
  int D;
};

__global__ __launch_bounds__(256) void lgcp_mfvi_finish_kernel(LgcpMfviArgs a) {
  __shared__ float sh[4];
  const int p = blockIdx.x, D = a.D;
  const float* counts = a.tc + (int64_t)D * D;
  const float mu0 = a.tc[(int64_t)D * D + D], pa = a.tc[(int64_t)D * D + D + 1];
  const float lognorm = a.tc[(int64_t)D * D + D + 2];
  float lp_acc = 0.f;
  for (int e = threadIdx.x; e < D; e += blockDim.x) {
    const float z = a.x[p * D + e];
    float kr = 0.f;
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) kr += a.kr[((int64_t)ks * kMP + p) * D + e];
    const float ez = expf(z);
    const float gp = -kr + counts[e] - pa * ez;
    lp_acc += -0.5f * (z - mu0) * kr + z * counts[e] - pa * ez;
    a.out_z[(int64_t)p * D + e] = z;
    if (a.gbuf) {
      a.gbuf[((int64_t)p * 2) * D + e] = -gp;
      a.gbuf[((int64_t)p * 2 + 1) * D + e] = -1.0f - gp * (z - a.params[a.o_mean + e]);
    }
  }
  const float lp = block_sum_256(lp_acc, sh);
  if (threadIdx.x == 0) {
    const float loss = -(a.w[p] + lp + lognorm);
    a.out_loss[p] = loss;
    double* o = a.partials + (int64_t)p * CMCD_NSTATS;
    o[0] = isfinite(loss) ? 1.0 : 0.0;
    o[1] = loss;
    o[2] = (double)loss * (double)loss;
    o[3] = -(double)loss;
    o[4] = isfinite(loss) ? 1.0 : 0.0;
  }
}

int64_t lgcp_mfvi_workspace_floats(int D, int64_t n, bool with_grad) {
  int64_t o = 0;
  auto take = [&](int64_t cnt) { o += (cnt + 3) & ~int64_t(3); };
  take(kMP * (int64_t)D); take(kMP * (int64_t)D); take(kSplit * kMP * (int64_t)D);
  take(kMP); take(kMP); take(2 * kMP);
  take(n * CMCD_NSTATS * 2);
  if (with_grad) take(n * 2 * (int64_t)D);
  return o;
}

int lgcp_mfvi(int D, int64_t o_mean, int64_t o_logdiag, const int32_t* seeds, int64_t n, const float* params,
              const float* tc, float* ws, float* out_loss, float* out_z, double** partials_out, float** gbuf_out,
              bool with_grad, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  int64_t o = 0;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  const int64_t ox = take(kMP * (int64_t)D), oxm = take(kMP * (int64_t)D), okr = take(kSplit * kMP * (int64_t)D);
  const int64_t ow = take(kMP), ofk = take(kMP), okeys = take(2 * kMP);
  const int64_t opart = take(n * CMCD_NSTATS * 2);
  const int64_t ogb = with_grad ? take(n * 2 * (int64_t)D) : 0;
  double* partials = reinterpret_cast<double*>(ws + opart);
  float* gbuf = with_grad ? ws + ogb : nullptr;
  *partials_out = partials;
  *gbuf_out = gbuf;
  const size_t gemm_lds = size_t(kGemmWaves * kChunk * kMP + kGemmWaves * kMP * 64) * 4;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(lgcp_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)gemm_lds) != hipSuccess)
    return CMCD_ERR_HIP;
  const float mu0 = 3.8812819069514780f;
  const int cbD = (D + 63) / 64;
  cmcd_layout lay{};
  lay.vd_mean = o_mean; lay.vd_logdiag = o_logdiag;
  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    LgcpStateArgs st{};
    st.seeds = seeds + base; st.params = params; st.tc = tc; st.x = ws + ox;
    st.w = ws + ow; st.fklp = ws + ofk; st.keys = reinterpret_cast<uint32_t*>(ws + okeys);
    st.lay = lay; st.M = M; st.D = D; st.K = 0;
    hipLaunchKernelGGL(lgcp_init_kernel, dim3(M), dim3(256), 0, stream, st);
    ActArgs act{};
    act.x = ws + ox; act.mu0 = mu0; act.M = M; act.D = D; act.IN = D; act.xm = ws + oxm; act.mode = 0;
    hipLaunchKernelGGL(lgcp_act_kernel, dim3((M * D + 255) / 256), dim3(256), 0, stream, act);
    GemmArgs g{};
    g.M = M; g.Kdim = D;
    g.seg[0] = GemmSeg{ws + oxm, tc, ws + okr, D, D, D, D};
    g.nblk0 = cbD;
    hipLaunchKernelGGL(lgcp_gemm_kernel, dim3(cbD, kSplit), dim3(64 * kGemmWaves), gemm_lds, stream, g);
    LgcpMfviArgs fa{params, tc, ws + ox, ws + okr, ws + ow, out_loss + base, out_z + base * D,
                    partials + base * CMCD_NSTATS, gbuf ? gbuf + base * 2 * D : nullptr, o_mean, D};
    hipLaunchKernelGGL(lgcp_mfvi_finish_kernel, dim3(M), dim3(256), 0, stream, fa);
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

}  // namespace cmcd
