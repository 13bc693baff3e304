// lgcp path (d = 1600, geffner net of width 1620): per-bridge launch sequence instead of a
// register-resident tile, because one evaluation touches 41.5 MB of weights
// (K^-1 10.2 MB + W1 10.4 MB + W2 10.5 MB + W3 10.4 MB) that cannot live on-chip per tile.
//
// Per evaluation i = 0..K (one evaluation serves the backward kernel of step i-1 and the forward
// kernel of step i, as in traj_kernel), 3 launches of one skinny-GEMM kernel with fused consumers:
//   A: [x - mu0] K^-1 -> kr slabs   and   x W1[:d] -> pre1 slabs -> u1 = [x; emb_i] + softplus(pre1 + bias1_i)
//   B: u1 W2 -> pre2 slabs -> u2 = u1 + softplus(pre2 + b2)                              nn.py:45-50,68
//   C: u2 W3 -> sn slabs -> state update: sn = factor_sn (sum(slabs) + b3); grad log p = -kr + counts - a e^x
//      (model_handler.py:386-396, cp_utils.py:102-104); close step i-1, draw eps_i = normal(G_i, (1600,)),
//      open step i; per-column-block partial log-weights (summed once at the end, fixed order).
// The skinny GEMM ([<=32 particles] x [K] x [N]) is weight-bandwidth / latency bound: a workgroup owns 64
// output columns of one of kSplit K-slices; it stages its <= 32 x 208 operand slice once in LDS (k-major), each of its 8 waves takes a 32-column half and a quarter of the slice on
// v_mfma_f32_32x32x2_f32 (exact fp32; W rows read once, 2 x 128 B per load, all loads in flight before the first
// wait), the 4 quarter tiles are summed through LDS and written as a partial slab.  The kSplit workgroups of a
// column block count arrivals on a device counter; the last one sums the slabs in fixed order (bitwise
// deterministic, no float atomics) and applies the consumer.  Slabs of a fused launch are written / read with
// agent-scope (sc1) accesses, so the protocol needs no L2 write-back fence.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMP = 32;      // particles per pass = the rows of the 32-row MFMA tile (same time per pass as 24 at N = 20)
constexpr int kGemmWaves = 8;
constexpr int kQuarters = kGemmWaves / 2;   // k quarters of a round (x 2 column halves = the 8 waves)
constexpr int kQLen = 52;                   // k rows per quarter per round (even: 32x32x2 takes two per MFMA)
constexpr int kStage = kQuarters * kQLen;   // k rows staged per round (208 >= 1620 / 8)
constexpr int kAsLd = 33;                   // LDS row pitch of the staged slice (odd: conflict-free writes)

constexpr int kSplit = 8;    // K split across workgroups (partial slabs, summed in fixed order downstream)

struct GemmSeg {
  const float* A;      // [kMP][lda]  activations (already formed)
  const float* W;      // [Kdim][ldw]
  float* out;          // [kSplit][kMP][ldo] partial slabs
  int N, lda, ldw, ldo;
  float a_shift;       // the operand is A - a_shift (K^-1 (x - mu0) without a separate subtraction pass)
};

// Fused consumers ("epilogues"): the kSplit workgroups of one 64-column block publish their partial slabs, fence, and
// bump the block's counter; the LAST one to arrive sums the slabs in fixed order (bitwise deterministic: which
// workgroup does the summing does not change the arithmetic) and applies the consumer for its 64 columns.  That
// removes the separate activation / state launches (5.7 us and 13 us of a 61 us evaluation) from the dependent chain.
enum { EPI_NONE = 0, EPI_ACT = 1, EPI_STEP = 2, EPI_STEP_NONET = 3, EPI_ACTB = 4 };   // NONET: MCD_ULA, the K^-1 slabs are this launch's own

struct ActEpi {          // u_out = u + softplus(bias + sum(slabs))      nn.py:45-50,68-69
  const float* bias;     // [IN]
  float* sum_out;        // [kMP][IN] summed pre-activation (kept for the backward pass)
  const float* x;        // [kMP][D]   mode 1: u = [x; emb]
  const float* emb;      // [E]
  const float* u_prev;   // [kMP][IN]  mode 2: u = u1
  float* u_out;          // [kMP][IN]
  int D, IN, mode;
};

struct StepEpi {         // evaluation i at z_i: closes step i-1, opens step i (or, at i = K, collects log p(z_K))
  const float* params;
  const float* tc;           // {Kinv[d,d], counts[d], mu0, a, lognorm}
  const float* sched;        // [K][8]
  float* x;                  // [kMP][D]   current z (updated in place, own columns only, unless xn is set)
  float* xp;                 // [kMP][D]   previous z
  float* xn;                 // MCD_ULA (the state is this launch's own GEMM operand): z_{i+1} goes HERE and the host rotates
                             // cur / prev / next; nullptr: in place (the operand of launch C is u2, never the state)
  const float* kr;           // [kSplit][kMP][D]   partial slabs of K^-1 (x - mu0)   (previous launch)
  const float* b3;           // [D]
  const float* factor;       // factor_sn (device scalar)
  uint32_t* gen;             // [kMP][2]      gen key of the chain (advanced by column block 0's extra wave)
  uint32_t* gkey;            // [2][kMP][2]   G_i, parity-buffered by i: read [i & 1], written [(i + 1) & 1]
  uint32_t* gktab;           // [K][n_total][2] every G_i of every particle (the VarGrad sweep redraws the noise from it)
  float* wslot;              // [ncb][kMP]    running sum over closed steps of (bk - fk) on this block's columns
  float* fkslot;             // [ncb][kMP]    forward-kernel log-density of the open step on this block's columns
  float* lpslot;             // [ncb][kMP]    log p(z_K) on this block's columns (i = K)
  float* out_z;              // [M][D]
  float* traj;               // optional [K+1][n_total][D]
  int64_t n_total, base;
  cmcd_layout lay;
  int D, K, i, var_mode, grad_clipping;
  int ula;                   // 0: CAIS; 1: MCD_ULA (no network at all: the K^-1 product is this launch's own GEMM);
                             // 2: MCD_ULA_sn (network in the backward kernel only)     mcd_over_orig.py:6-65
};

// backward activations (consumer of the two IN-wide backward GEMMs): d u = [d u_next +] sum(slabs), d a = d u sigmoid(pre)
// (constants and the packed-operand index of the no-split-K GEMM, section "r04" below: the adjoint step writes packed operands too)
constexpr int kNskChunks = 104;                  // 16-deep chunks of a packed operand (K <= 1664), 13 per wave
constexpr int kNskCpw = kNskChunks / kGemmWaves;
constexpr int64_t kNskOperand = (int64_t)2 * kNskChunks * 256;   // floats of one packed [32 x 1664] operand
constexpr int kNskOps = 6;                      // per lane: z (three rotating buffers), u1, u2, K^-1 product

// nch: 16-deep chunks of the packed array — kNskChunks (K <= 1664), or 2 kNskChunks for the 3220-wide layers of the 2nd-order
// mode (two rounds of 13 chunks per wave)
__host__ __device__ __forceinline__ int64_t nsk_pack(int r, int k, int nch = kNskChunks) {
  return (((int64_t)(r >> 4) * nch + (k >> 4)) * 64 + (r & 15) + 16 * ((k & 15) >> 2)) * 4 + (k & 3);
}

struct LgcpActbArgs {
  const float* pre;        // [kMP][IN] pre-activation of this layer
  const float* du_prev;    // [kMP][IN] (mode 1: d u2)
  const float* u_src;      // [kMP][IN] this layer's input activation to keep (u2 for mode 2, u1 for mode 1)
  float* du_out;           // [kMP][IN]
  float* da_out;           // [kMP][IN]
  float* da_big;           // [(K+1) n][IN]
  float* u_big;            // [(K+1) n][IN]
  float* S;                // mode 1: S[e][k]  = sum_m d a1   (row of the table)
  float* S2;               // mode 1: S2[e][k] = sum_m d u1
  float* gb;               // mode 2: d b2[k] += sum_m d a2
  int64_t row0;            // e * n + base
  int IN, mode;
};

struct LgcpAdjArgs {
  const float* params;
  const float* tc;
  const float* sched;        // [K][8] {beta, eps, ...}
  const float* traj;         // [K+1][n][D]
  const float* kr;           // [kSplit][kMP][D]
  const float* sn;           // [kSplit][kMP][D]
  const float* b3;
  const float* factor;
  const float* lamn;         // [kMP][D] lambda_{e+1}
  const float* gE;           // [kMP][D] g_e
  float* gprev;              // [kMP][D] g_{e-1}
  float* dO;                 // [kMP][D] cotangent of o = u2 W3 + b3
  float* v;                  // [kMP][D] clipmask . a_gp
  float* lam_part;           // [kMP][D]
  float* gmu_acc;            // [kMP][D] running d / d vd.mean per particle
  float* glam_acc;           // [kMP][D]
  float* part;               // [K+1][n * nbx][8] per-workgroup partial sums {sb, se, r2, sb2, se2, gf} (no atomics)
  float* DObig;              // [(K+1) n][D]
  cmcd_layout lay;
  int64_t n, base;
  int M, D, K, e, grad_clipping;
  float omega;               // weight of every particle's loss (reparameterised gradient of the mean)
  const float* omega_vec;    // [n] VarGrad: weight of particle p's LOG-WEIGHT (cmcd_vargrad_weights), or nullptr
  const uint32_t* gktab;     // [K][n][2] noise keys of the forward pass (VarGrad: z_{e+1} - mean = sigma eps_e, exactly)
  int ula;                   // 0: CAIS; 1: MCD_ULA (no network); 2: MCD_ULA_sn (network in the backward kernel only)
  int bptt;                  // 1: gradient through the trajectory (lambda recursion); 0: z detached (mcd_cais_var.py:59,79)
  int nslab = kSplit;        // slabs of kr / sn to sum: kSplit (recomputed by the split-K GEMM) or 1 (kept by the forward)
  float* dOp = nullptr;      // packed copies of dO / v: operands of the reverse sweep's no-split-K launches (nullptr: none)
  float* vp = nullptr;
  int var_mode;              // MCD_CAIS_var_sn: clip at 1e2 and clip grad log q too (mcd_cais_var.py:29-36)
};

struct LgcpLamArgs {
  const float* params;
  const float* tc;
  const float* traj;
  const float* dxf;        // [kSplit][kMP][D]   d a1 W1[:D]^T partials
  const float* hv;         // [kSplit][kMP][D]   v K^-1 partials
  const float* du1;        // [kMP][IN]
  const float* v;          // [kMP][D]
  const float* lam_part;   // [kMP][D]
  const float* gprev;      // [kMP][D]
  float* lamn;             // [kMP][D]
  float* gE;               // [kMP][D]
  float* gmu_acc;
  float* glam_acc;
  cmcd_layout lay;
  int64_t n, base;
  int M, D, IN, e;
  float omega;
  int no_net;              // MCD_ULA: d x = 0 (no network to go back through)
  int nslab = kSplit;      // slabs of dxf / hv to sum: kSplit, or 1 (row-major outputs of the no-split-K launch)
};

struct GemmArgs {
  GemmSeg seg[2];
  int nblk0;           // column blocks of segment 0
  int M, Kdim;
  int Kdim1;           // K extent of segment 1 when it differs from segment 0's (0: same)
  int* counters;       // [gridDim.x] arrival counters, zero between launches (reset by the last arriver)
  int epi_seg;         // segment the epilogue applies to (-1: all)
  ActEpi act;
  StepEpi step;
  LgcpActbArgs actb;
};

__device__ __forceinline__ float wave_sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// One element (particle p of the pass, column j < D) of the adjoint step at evaluation a.e; ln_in / ge_in are
// lambda_{e+1}[p][j] and g_e[p][j] (reparameterised gradient only); t accumulates {sb, se, r2, sb2, se2, gf}.
__device__ __forceinline__ void lgcp_adj_elem(const LgcpAdjArgs& a, int p, int j, float ln_in, float ge_in, float (&t)[6]) {
  const int D = a.D, e = a.e, K = a.K;
  const float* counts = a.tc + (int64_t)D * D;
  const float pa = a.tc[(int64_t)D * D + D + 1];
  const float clipv = a.var_mode ? 1e2f : 1e3f;
  const bool clip_p = a.grad_clipping != 0, clip_q = clip_p && a.var_mode;
  const bool bptt = a.bptt != 0;
  const float om = a.omega_vec ? -a.omega_vec[a.base + p] : a.omega;      // weight of this particle's loss = -w
  const float pb = e > 0 ? a.sched[8 * (e - 1)] : 0.f, pe = e > 0 ? a.sched[8 * (e - 1) + 1] : 1.f;
  const float be = e < K ? a.sched[8 * e] : 0.f, ee = e < K ? a.sched[8 * e + 1] : 1.f;
  const bool no_net = a.ula == 1;
  const float fsn = a.ula ? 0.f : 1.f;
  const float fac = no_net ? 0.f : a.factor[0];
  const float* ze = a.traj + ((int64_t)e * a.n + a.base + p) * D;
  const float* zpv = a.traj + ((int64_t)(e > 0 ? e - 1 : 0) * a.n + a.base + p) * D;
  const float* znv = a.traj + ((int64_t)(e < K ? e + 1 : K) * a.n + a.base + p) * D;
  {
    const float z = ze[j];
    float kr = 0.f, o = no_net ? 0.f : a.b3[j];
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) {
      if (ks >= a.nslab) break;            // (1: the forward's kept sums; kSplit: the recompute's slabs)
      kr += a.kr[((int64_t)ks * kMP + p) * D + j];
      o += no_net ? 0.f : a.sn[((int64_t)ks * kMP + p) * D + j];
    }
    const float s = o * fac;
    const float graw = -kr + counts[j] - pa * expf(z);
    const float m = (!clip_p || fabsf(graw) < clipv) ? 1.0f : 0.f;
    const float gp = clip_p ? fminf(fmaxf(graw, -clipv), clipv) : graw;
    const float mean = a.params[a.lay.vd_mean + j];
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    const float qiv = 1.0f / (sd * sd);
    const float gqraw = -(z - mean) * qiv;
    const float mq = (!clip_q || fabsf(gqraw) < clipv) ? 1.0f : 0.f;
    const float gq = clip_q ? fminf(fmaxf(gqraw, -clipv), clipv) : gqraw;
    float a_s = 0.f, a_gp = 0.f, a_gq = 0.f, lam = 0.f, gpv = 0.f;
    if (e > 0) {   // backward kernel of step e-1: d loss / d (its mean) = -r / sigma^2
      const float ub = -1.0f * (pb * gp + (1.0f - pb) * gq);
      // r = z_{e-1} - (z - pe ub + pe s), with the O(1) states subtracted first (exact in float32: they differ by a step)
      const float r = (zpv[j] - z) + pe * (ub - s);
      gpv = -om * r * (0.5f / pe);
      a_s += pe * gpv; a_gp += pe * pb * gpv; a_gq += pe * (1.0f - pb) * gpv; lam += gpv;
      t[0] += (gp - gq) * gpv; t[1] += (s - ub) * gpv; t[2] += om * r * r;
    }
    if (e < K) {   // forward kernel of step e
      const float uf = -1.0f * (be * gp + (1.0f - be) * gq);
      float df = (znv[j] - z) + ee * (uf + fsn * s);      // z_{e+1} - (z - ee uf - ee s), same ordering
      if (!bptt) {
        // the weights omega_p sum to zero and |df|^2 / (4 eps^2) is ~1e5 per particle: the float32 difference above loses
        // the digits that survive the cancellation.  z_{e+1} = mean + sigma eps_e, so redraw eps_e (mcd_cais.py:66-67)
        const int H = (D + 1) / 2, jj = j < H ? j : j - H;
        const uint32_t* gk = a.gktab + ((int64_t)e * a.n + a.base + p) * 2;
        uint32_t y0 = jj, y1 = (H + jj < D) ? H + jj : 0;
        threefry2x32(gk[0], gk[1], y0, y1);
        df = a.sched[8 * e + 2] * bits_to_normal(j < H ? y0 : y1);
      }
      const float nsig = df * (0.5f / ee);
      // cotangent of the kernel's mean: lambda_{e+1} when z_{e+1} = mean + noise carries the gradient on; with z detached
      // the density log N(z_{e+1}; mean, sigma) itself: d loss / d mean = +df / sigma^2
      const float ln = bptt ? ln_in : om * nsig;
      a_s -= fsn * ee * ln; a_gp += ee * be * ln; a_gq += ee * (1.0f - be) * ln;
      if (bptt) lam += ln - ge_in;
      t[3] += (gp - gq) * ln;
      t[4] += bptt ? (nsig - uf - fsn * s) * ln : (-uf - fsn * s) * ln + om * df * df * (0.25f / (ee * ee));
    }
    if (e == K) lam -= om * graw;
    if (e == 0) lam += om * gq;
    a.gmu_acc[p * D + j] += a_gq * mq * qiv;
    a.glam_acc[p * D + j] += a_gq * mq * (-2.0f * gqraw) + ((!bptt && e == 0) ? -om : 0.f);   // z detached: d log q(z_0(theta)) / d logdiag = -1
    lam -= a_gq * qiv;
    t[5] += a_s * o;
    if (!no_net) {
      a.dO[p * D + j] = a_s * fac;
      if (a.dOp) a.dOp[nsk_pack(p, j)] = a_s * fac;
      a.DObig[((int64_t)e * a.n + a.base + p) * D + j] = a_s * fac;
    }
    if (bptt) {
      a.gprev[p * D + j] = gpv;
      a.v[p * D + j] = m * a_gp;
      if (a.vp) a.vp[nsk_pack(p, j)] = m * a_gp;
      a.lam_part[p * D + j] = lam;
    }
  }
}

// standalone launch: one wave per (64-column block, particle) — the slot layout of the fused consumer
__global__ __launch_bounds__(64) void lgcp_adj_step_kernel(LgcpAdjArgs a) {
  const int p = blockIdx.y, D = a.D, lane = threadIdx.x, j = blockIdx.x * 64 + lane;
  float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < D) {
    const bool live = a.bptt != 0 && a.e < a.K;
    lgcp_adj_elem(a, p, j, live ? a.lamn[p * D + j] : 0.f, live ? a.gE[p * D + j] : 0.f, t);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) t[q] = wave_sum64(t[q]);
  if (lane == 0) {
    float* o = a.part + (((int64_t)a.e * a.n + a.base + p) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = t[q];
  }
}

// lambda_e of element (m, j) from the summed products dx = d u1_j + (d a1 W1[:D]^T)_j and hv = (v K^-1)_j; also hands g_e on
__device__ __forceinline__ void lgcp_lam_elem(const LgcpLamArgs& a, int m, int j, float dx, float hv, float& lam_out, float& ge_out) {
  const int D = a.D, idx = m * D + j;
  const float pa = a.tc[(int64_t)D * D + D + 1];
  const float z = a.traj[((int64_t)a.e * a.n + a.base + m) * D + j];
  const float lam = a.lam_part[idx] + dx - hv - pa * expf(z) * a.v[idx];   // H_p v = -K^-1 v - a e^z v
  a.lamn[idx] = lam;
  const float ge = a.gprev[idx];
  a.gE[idx] = ge;
  if (a.e == 0) {
    const float mean = a.params[a.lay.vd_mean + j];
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    const float qiv = 1.0f / (sd * sd), dz = z - mean;
    const float gq = -dz * qiv;
    a.gmu_acc[idx] += lam - a.omega * gq;
    a.glam_acc[idx] += lam * dz + a.omega * (dz * dz * qiv - 1.0f);
  }
  lam_out = lam;
  ge_out = ge;
}

__global__ void lgcp_lam_finish_kernel(LgcpLamArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= a.M * a.D) return;
  const int m = idx / a.D, j = idx - m * a.D, D = a.D;
  float dx = a.no_net ? 0.f : a.du1[m * a.IN + j], hv = 0.f;     // residual path: d x_j += d u1_j
#pragma unroll
  for (int ks = 0; ks < kSplit; ++ks) {
    if (ks >= a.nslab) break;
    dx += a.no_net ? 0.f : a.dxf[((int64_t)ks * kMP + m) * D + j];
    hv += a.hv[((int64_t)ks * kMP + m) * D + j];
  }
  float lam, ge;
  lgcp_lam_elem(a, m, j, dx, hv, lam, ge);
}

// r04 (activations kept by the forward: evaluation e - 1 has nothing to wait for): lambda_e from the products of evaluation e,
// then the adjoint step of evaluation e - 1 on the same element — one launch instead of two per evaluation of the sweep
__global__ __launch_bounds__(64) void lgcp_lam_adj_kernel(LgcpLamArgs la, LgcpAdjArgs a) {
  const int p = blockIdx.y, D = a.D, lane = threadIdx.x, j = blockIdx.x * 64 + lane;
  float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < D) {
    float dx = la.no_net ? 0.f : la.du1[p * la.IN + j], hv = 0.f;     // residual path: d x_j += d u1_j
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) {
      if (ks >= la.nslab) break;
      dx += la.no_net ? 0.f : la.dxf[((int64_t)ks * kMP + p) * D + j];
      hv += la.hv[((int64_t)ks * kMP + p) * D + j];
    }
    float lam, ge;
    lgcp_lam_elem(la, p, j, dx, hv, lam, ge);
    lgcp_adj_elem(a, p, j, lam, ge, t);
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) t[q] = wave_sum64(t[q]);
  if (lane == 0) {
    float* o = a.part + (((int64_t)a.e * a.n + a.base + p) * gridDim.x + blockIdx.x) * 8;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = t[q];
  }
}

// (G, H) = split(gen); gen' = second(split(H))     mcd_cais.py:66-67,87
__device__ __forceinline__ void lgcp_key_advance(uint32_t& k0, uint32_t& k1, uint32_t& G0, uint32_t& G1) {
  uint32_t g0 = 0, h0 = 2, g1 = 1, h1 = 3;
  threefry2x32(k0, k1, g0, h0);
  threefry2x32(k0, k1, g1, h1);
  uint32_t n0 = 0, n2 = 2, n1 = 1, n3 = 3;
  threefry2x32(h0, h1, n0, n2);
  threefry2x32(h0, h1, n1, n3);
  G0 = g0; G1 = g1; k0 = n2; k1 = n3;
}

// the state update of lgcp_forward's evaluation i on columns [n0, n0 + 64) of all particles; sn slabs are this
// launch's output (visible after the arrival protocol), everything else is from earlier launches
template <bool NO_NET>   // NO_NET (MCD_ULA): `sn_slabs` are the K^-1 slabs of THIS launch, there is no network output
__device__ __forceinline__ void lgcp_step_tile(const StepEpi& a, const float* sn_slabs, int M, int n0, int cb, int wv,
                                               int lane) {
  constexpr bool no_net = NO_NET;               // compile-time: a run-time branch here split the 16 slab loads apart
  const float fsn = a.ula ? 0.f : 1.f;          // the overdamped baselines have no network in the forward kernel
  const int D = a.D, H = (D + 1) / 2, i = a.i;
  const float* counts = a.tc + (int64_t)D * D;
  const float mu0 = a.tc[(int64_t)D * D + D], pa = a.tc[(int64_t)D * D + D + 1];
  const float clipv = a.var_mode ? 1e2f : 1e3f;
  const bool clip_p = a.grad_clipping != 0, clip_q = clip_p && a.var_mode;
  const bool last = i == a.K;
  const float* sp = a.sched + 8 * (i > 0 ? i - 1 : 0);
  const float pbeta = sp[0], peps = sp[1], pcst = sp[3], pinv2s2 = sp[4];
  const float* sc = a.sched + 8 * (last ? a.K - 1 : i);
  const float beta = sc[0], eps = sc[1], sig = sc[2], cst = sc[3], inv2s2 = sc[4];
  const float fac = no_net ? 0.f : a.factor[0];
  const int e = n0 + lane;
  const bool ecol = e < D;
  const int ecl = min(e, D - 1);
  const float cnt = counts[ecl], b3 = no_net ? 0.f : a.b3[ecl];
  const float mean = a.params[a.lay.vd_mean + ecl];
  const float sd = expf(a.params[a.lay.vd_logdiag + ecl]);
  const uint32_t* gk = a.gkey + (i & 1) * 2 * kMP;
  // A wave = one particle x 64 columns, kMP / 8 particles per wave.  Every load of all of them first, from clamped
  // (always valid) addresses: taken one particle at a time the stores of one would fence the loads of the next (three
  // memory round trips in the tail), and a predicated load gets a branch and an s_waitcnt of its own.
  constexpr int R = kMP / kGemmWaves;
  const int ec = min(e, D - 1);
  float zv[R], krv[R], sv[R], xpv[R];
  uint32_t g0v[R], g1v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int mc = min(wv + kGemmWaves * r, M - 1);
    zv[r] = a.x[mc * D + ec];
    xpv[r] = a.xp[mc * D + ec];
    g0v[r] = gk[2 * mc];
    g1v[r] = gk[2 * mc + 1];
    float kr = 0.f, sacc = b3;
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) {              // fixed-order sums of the split-K slabs
      const float own = __hip_atomic_load(sn_slabs + ((int64_t)ks * kMP + mc) * D + ec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      kr += no_net ? own : a.kr[((int64_t)ks * kMP + mc) * D + ec];
      sacc += no_net ? 0.f : own;
    }
    krv[r] = kr;
    sv[r] = no_net ? 0.f : sacc * fac;                 // factor_sn (u2 W3 + b3)           nn.py:70
  }
  __builtin_amdgcn_sched_barrier(0);
  float znv[R], bkv[R], fkv[R], lpv[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const float z = zv[r], kr = krv[r], sn = sv[r];
    float bk_acc = 0.f, fk_acc = 0.f, lp_acc = 0.f, zn = 0.f;
    znv[r] = 0.f; bkv[r] = 0.f; fkv[r] = 0.f; lpv[r] = 0.f;
    if (wv + kGemmWaves * r >= M) continue;            // wave-uniform: no arithmetic for the clamped duplicates
    const float ez = expf(z);
    float gp = -kr + cnt - pa * ez;                    // grad log p      model_handler.py:386-396
    float gq = -(z - mean) / (sd * sd);
    if (clip_p) gp = fminf(fmaxf(gp, -clipv), clipv);
    if (clip_q) gq = fminf(fmaxf(gq, -clipv), clipv);
    if (i > 0) {   // backward kernel of step i-1                             mcd_cais.py:71-86
      const float ub = -1.0f * (pbeta * gp + (1.0f - pbeta) * gq);
      const float bk = z - peps * ub + peps * sn;
      const float db = xpv[r] - bk;
      bk_acc = -(db * db) * pinv2s2 - pcst;
    }
    if (last) {    // log p(z_K)
      lp_acc = -0.5f * (z - mu0) * kr + z * cnt - pa * ez;
    } else {       // forward kernel of step i                                mcd_cais.py:52-67
      // eps_i = normal(G_i, (D,)): element e is word (e >= H) of the block with counters (j, H + j), j = e mod H
      const int j = ec < H ? ec : ec - H;
      uint32_t y0 = j, y1 = (H + j < D) ? H + j : 0;
      threefry2x32(g0v[r], g1v[r], y0, y1);
      const float uf = -1.0f * (beta * gp + (1.0f - beta) * gq);
      const float fk = z - eps * uf - fsn * eps * sn;
      zn = fk + sig * bits_to_normal(ec < H ? y0 : y1);
      const float df = zn - fk;
      fk_acc = -(df * df) * inv2s2 - cst;
    }
    znv[r] = zn; bkv[r] = ecol ? bk_acc : 0.f; fkv[r] = ecol ? fk_acc : 0.f; lpv[r] = ecol ? lp_acc : 0.f;   // clamped lanes: dropped
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const int m = wv + kGemmWaves * r;
    if (m >= M) continue;                              // wave-uniform
    if (ecol) {
      if (last) {
        a.out_z[(int64_t)m * D + e] = zv[r];
      } else {
        if (a.xn) {
          a.xn[m * D + e] = znv[r];
        } else {
          a.xp[m * D + e] = zv[r];
          a.x[m * D + e] = znv[r];
        }
        if (a.traj) a.traj[((int64_t)(i + 1) * a.n_total + a.base + m) * D + e] = znv[r];
      }
    }
    const float bk_lp = wave_sum64(bkv[r]), fk_lp = wave_sum64(fkv[r]), lp = wave_sum64(lpv[r]);
    if (lane == 0) {
      const int sl = cb * kMP + m;
      if (i > 0) a.wslot[sl] += bk_lp - a.fkslot[sl];
      if (!last) a.fkslot[sl] = fk_lp;
      else a.lpslot[sl] = lp;
    }
  }
}

// out[ks][m][n] = sum_{k in slice ks, wave w} A[m][k] W[k][n]   (no bias: added when the slabs are summed)
template <int EPI>
__global__ __launch_bounds__(64 * (kGemmWaves + ((EPI == EPI_STEP || EPI == EPI_STEP_NONET) ? 1 : 0)), (EPI == EPI_STEP || EPI == EPI_STEP_NONET) ? 1 : 4) void lgcp_gemm_kernel(GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ int s_last;
  const int s = blockIdx.x < a.nblk0 ? 0 : 1;
  const GemmSeg sg = a.seg[s];
  const int Kdim = (s && a.Kdim1) ? a.Kdim1 : a.Kdim;
  const int n0 = (blockIdx.x - (s ? a.nblk0 : 0)) * 64;
  const int ksplit = blockIdx.y;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const bool worker = (EPI != EPI_STEP && EPI != EPI_STEP_NONET) || wv < kGemmWaves;   // EPI_STEP* carry one extra wave (key chain)
  float* As = lds;                                                   // [kStage][kAsLd]  A slice, k-major, particle-minor
  float* red = lds + kStage * kAsLd;                                 // [kQuarters][kMP][64]
  const int kslice = (Kdim + kSplit - 1) / kSplit;
  const int s_lo = ksplit * kslice, s_hi = min(Kdim, s_lo + kslice);
  const int half = wv & 1, q = wv >> 1;                              // wave = (32-column half, k quarter of the round)
  const int l31 = lane & 31, l5 = lane >> 5;
  const int n = n0 + 32 * half + l31;
  const bool ncol = n < sg.N;

  if (kMP < 32) {
    constexpr int kPad = kMP < 32 ? 32 - kMP : 1;
    for (int e = threadIdx.x; e < kStage * kPad; e += blockDim.x) As[(e / kPad) * kAsLd + kMP + e % kPad] = 0.f;
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int r0 = s_lo; r0 < s_hi; r0 += kStage) {                     // one round for Kdim <= kSplit * kStage = 1664
    // All loads of the round first — the operand slice (L2 / Infinity Cache resident: the previous launch wrote it), then
    // this wave's 26 W row pairs (2 x 128 B per load, HBM) — so that the slice is staged while the weights are still in
    // flight.  Clamped, unpredicated addresses (a predicated load gets its own branch and s_waitcnt): weight rows past the
    // slice meet zero operand rows, columns past N are never stored.
    constexpr int kIt = (kMP * kStage + 64 * kGemmWaves - 1) / (64 * kGemmWaves);
    float wreg[kQLen / 2], av[kIt];
    if (r0 > s_lo) __syncthreads();
    if (worker) {
#pragma unroll
      for (int it = 0; it < kIt; ++it) {
        const int e = it * 64 * kGemmWaves + threadIdx.x, m = e / kStage, kk = e - m * kStage;
        av[it] = sg.A[(int64_t)min(m, a.M - 1) * sg.lda + min(r0 + kk, s_hi - 1)];
      }
      const int nc = ncol ? n : sg.N - 1;
#pragma unroll
      for (int j = 0; j < kQLen / 2; ++j)
        wreg[j] = sg.W[(int64_t)min(r0 + q * kQLen + 2 * j + l5, s_hi - 1) * sg.ldw + nc];
      __builtin_amdgcn_sched_barrier(0);        // keep the selects below from being interleaved with (and waiting on) the loads
#pragma unroll
      for (int it = 0; it < kIt; ++it) {        // k-major, odd row pitch: conflict-free LDS writes
        const int e = it * 64 * kGemmWaves + threadIdx.x, m = e / kStage, kk = e - m * kStage;
        if (e < kMP * kStage) As[kk * kAsLd + m] = (m < a.M && r0 + kk < s_hi) ? av[it] - sg.a_shift : 0.f;
      }
    }
    // workgroup barrier that publishes the LDS writes without draining the weight loads still in flight
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (worker) {
#pragma unroll
      for (int j = 0; j < kQLen / 2; ++j) {
        const float aop = As[(q * kQLen + 2 * j + l5) * kAsLd + l31];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aop, wreg[j], acc, 0, 0, 0);
      }
    }
  }
  if (worker) {
    // D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int m = (i & 3) + 8 * (i >> 2) + 4 * l5;
      if (m < kMP) red[(q * kMP + m) * 64 + 32 * half + l31] = acc[i];
    }
  }
  __syncthreads();
  float* slab = sg.out + (int64_t)ksplit * kMP * sg.ldo;
  if (worker) {
    for (int o = threadIdx.x; o < kMP * 64; o += 64 * kGemmWaves) {
      const int m = o >> 6, nl = o & 63;
      if (m < a.M && n0 + nl < sg.N) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kQuarters; ++w) v += red[(w * kMP + m) * 64 + nl];  // fixed order
        // a consumed slab goes out as an agent-scope (write-through, sc1) store: the arrival protocol below then
        // needs no L2 write-back fence (buffer_wbl2 per wave made the launch 4x longer)
        if (EPI != EPI_NONE && (a.epi_seg < 0 || s == a.epi_seg))
          __hip_atomic_store(slab + (int64_t)m * sg.ldo + n0 + nl, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else
          slab[(int64_t)m * sg.ldo + n0 + nl] = v;
      }
    }
  }
  if (EPI == EPI_NONE) return;
  if (a.epi_seg >= 0 && s != a.epi_seg) return;                       // uniform per workgroup

  // ---- arrival protocol: the slab stores are complete at agent scope once vmcnt drains (the workgroup-scope
  // release of __syncthreads waits for that); count; the last workgroup of the block reads the slabs back with
  // agent-scope loads (never served from a stale L1 / remote-XCD L2 line) and consumes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = __hip_atomic_fetch_add(a.counters + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = old == kSplit - 1;
    if (s_last) a.counters[blockIdx.x] = 0;                           // ready for the next launch
  }
  __syncthreads();
  if (!s_last) return;

  if (EPI == EPI_ACT) {
    const ActEpi& ac = a.act;
    const int k = n0 + lane, kc = min(k, ac.IN - 1);
    constexpr int R = kMP / kGemmWaves;
    // all loads of the wave's particles first, clamped and unpredicated (see lgcp_step_tile)
    const float bias = ac.bias[kc];
    float pre[R], u[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int mc = min(wv + kGemmWaves * r, a.M - 1);
      float p = bias;
#pragma unroll
      for (int q = 0; q < kSplit; ++q)
        p += __hip_atomic_load(sg.out + ((int64_t)q * kMP + mc) * sg.ldo + kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pre[r] = p;
      if (ac.mode == 1) {                              // u = [x; emb_i]      nn.py:68-69
        const float xv = ac.x[mc * ac.D + min(kc, ac.D - 1)], ev = ac.emb[max(kc - ac.D, 0)];
        u[r] = kc < ac.D ? xv : ev;
      } else {
        u[r] = ac.u_prev[mc * ac.IN + kc];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = wv + kGemmWaves * r;
      if (m < a.M && k < ac.IN) {
        ac.sum_out[m * ac.IN + k] = pre[r];
        ac.u_out[m * ac.IN + k] = u[r] + softplus(pre[r]);                        // nn.py:45-50
      }
    }
  } else if (EPI == EPI_ACTB) {
    const LgcpActbArgs& ab = a.actb;
    const int k = n0 + lane, kc = min(k, ab.IN - 1);
    constexpr int R = kMP / kGemmWaves;
    float du[R], pr[R], us[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {                      // all loads first, clamped and unpredicated (see lgcp_step_tile)
      const int mc = min(wv + kGemmWaves * r, a.M - 1);
      float d = ab.mode == 1 ? ab.du_prev[mc * ab.IN + kc] : 0.f;      // uniform mode
#pragma unroll
      for (int q = 0; q < kSplit; ++q)
        d += __hip_atomic_load(sg.out + ((int64_t)q * kMP + mc) * sg.ldo + kc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      du[r] = d;
      pr[r] = ab.pre[mc * ab.IN + kc];
      us[r] = ab.u_src[mc * ab.IN + kc];
    }
    __builtin_amdgcn_sched_barrier(0);
    float sda = 0.f, sdu = 0.f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = wv + kGemmWaves * r;
      if (m < a.M && k < ab.IN) {
        const float da = du[r] * sigmoid_fast(pr[r]);
        ab.du_out[m * ab.IN + k] = du[r];
        ab.da_out[m * ab.IN + k] = da;
        ab.da_big[(ab.row0 + m) * ab.IN + k] = da;
        if (ab.u_big) ab.u_big[(ab.row0 + m) * ab.IN + k] = us[r];   // (nullptr: the forward kept u in that very table)
        sda += da; sdu += du[r];
      }
    }
    // sums over the particles of the pass: through LDS across the 8 waves, then ONE writer per column (no atomics; the
    // launches that touch a column's entry are ordered on the stream)
    __syncthreads();                                   // the slab-summing pass above is done with `red`
    red[(wv * 64 + lane) * 2] = sda;
    red[(wv * 64 + lane) * 2 + 1] = sdu;
    __syncthreads();
    if (wv == 0 && k < ab.IN) {
      float ta = 0.f, tu = 0.f;
#pragma unroll
      for (int w = 0; w < kGemmWaves; ++w) { ta += red[(w * 64 + lane) * 2]; tu += red[(w * 64 + lane) * 2 + 1]; }
      if (ab.mode == 1) { ab.S[k] += ta; ab.S2[k] += tu; }
      else ab.gb[k] += ta;
    }
  } else if (EPI == EPI_STEP || EPI == EPI_STEP_NONET) {
    const StepEpi& st = a.step;
    const int cb = blockIdx.x;
    if (wv < kGemmWaves) {
      lgcp_step_tile<EPI == EPI_STEP_NONET>(st, sg.out, a.M, n0, cb, wv, lane);
    } else if (cb == 0 && st.i + 1 < st.K && lane < a.M) {
      // the chain's key for evaluation i + 1 (a dependent integer chain of ~450 instructions: on its own wave it
      // hides behind the state update of the eight others)
      uint32_t k0 = st.gen[2 * lane], k1 = st.gen[2 * lane + 1], G0, G1;
      lgcp_key_advance(k0, k1, G0, G1);
      st.gen[2 * lane] = k0; st.gen[2 * lane + 1] = k1;
      uint32_t* gk = st.gkey + ((st.i + 1) & 1) * 2 * kMP;
      gk[2 * lane] = G0; gk[2 * lane + 1] = G1;
      uint32_t* gt = st.gktab + ((int64_t)(st.i + 1) * st.n_total + st.base + lane) * 2;
      gt[0] = G0; gt[1] = G1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// r04: the FORWARD launch sequence on a skinny GEMM WITHOUT split-K across workgroups ("nsk").
//
// The split-K kernel above pays, per dependent launch, a seam inside the launch: partial slabs published write-through, an
// arrival ticket, the last arriver reading eight slabs back (10 - 13 us per launch, 3 x 129 launches per call).  Here a
// workgroup owns a 16-row x 16-column output tile over the WHOLE contraction and the contraction is split over its 8 WAVES
// only (interleaved 16-deep chunks, 13 per wave for K <= 1664): the cross-wave sum goes through 8 KB of LDS and one barrier,
// no workgroup ever waits for another, no counter, no slab.  What made the earlier no-split-K attempts slow (section 4 of
// 10.5 / 4.9 ms) — half-used 64-byte row segments, duplicated operand rows, the load latency paid several
// times over — is removed by PACKING both operands once so that every wave-load is 1 KB contiguous and a wave holds its whole
// share of the contraction in registers (26 x 16-byte loads per lane, all in flight before the first wait):
//   weights    Wp[tile = n / 16][chunk = k / 16][lane = n % 16 + 16 ((k % 16) / 4)][k % 4]   (re-packed once per call)
//   operands   Ap[half = r / 16][chunk = k / 16][lane = r % 16 + 16 ((k % 16) / 4)][k % 4]   (written by the consumers)
// = the operand layout of v_mfma_f32_16x16x4_f32 (A: lane -> row lane % 16, k lane / 16; B: lane -> column lane % 16, k lane /
// 16): MFMA step s of a chunk contracts k = 16 chunk + 4 (lane / 16) + s, i.e. component s of both 16-byte loads.  A
// workgroup's 16 x 16 outputs are ONE contiguous 1 KB block of the next launch's packed operand.
// Measured (tools/probes/nsk_probe.hip, the three-launch chain x 129 with a stand-in state update): 3.0 ms against the
// split-K sequence's 4.55 ms; per launch 5.1 (204 workgroups) / 9.7 (404) / 5.2 us.
// ------------------------------------------------------------------------------------------
// NSK_OUT: the plain product, stored row-major [kMP][N] (the element-wise kernels of the 2nd-order sequence read it)
enum { NSK_ACT1 = 1, NSK_ACT2 = 2, NSK_KR = 3, NSK_STEP = 4, NSK_STEP_NONET = 5, NSK_OUT = 6, NSK_UHA_MID = 7 };

struct NskSeg {
  const float* A;      // packed operand [2][nch][64][4]
  const float* W;      // packed weights [ceil(N / 16)][nch][64][4]
  int N, epi;
  float a_shift;       // the operand is A - a_shift
  int nch = kNskChunks;   // chunks of THIS segment's contraction (a multiple of kNskChunks)
};

struct NskArgs {
  NskSeg seg[2];
  int nt0;             // column tiles of segment 0
  int M;
  // r05: launches with two 16-row halves per column tile (17 .. 32 particles, not merged) are 1-D grids of 16 ceil(gx / 8)
  // workgroups in XCD-PAIRED order: workgroup L runs on XCD L % 8 (round-robin dispatch), slot s = L / 8 there; it takes tile
  // (L % 8) + 8 (s / 2), half s % 2 — both halves of a tile, which read the SAME weight tile, share an XCD and its L2 (a
  // (tile, half) = (blockIdx.x, blockIdx.y) grid of gx = 102 columns put them on XCDs t % 8 and (t + 6) % 8: every weight
  // byte crossed the fabric twice, profiles/r04_pmc/lgcp_summary.json: 23.7 / 24.0 MB per launch for 10.5 MB of weights).
  // gx = the logical grid width (column tiles + the key-chain workgroup of STEP launches); 0 = plain (blockIdx.x, blockIdx.y).
  int pair_gx = 0;
  // activations
  const float* bias;   // ACT1: bias1_i [IN];  ACT2: b2 [IN] (params)
  const float* emb;    // ACT1: emb_i [E]
  const float* xA;     // packed z of this evaluation (operand of A / of B's second product; ACT1 and STEP read it)
  const float* uA;     // ACT2: packed u1
  float* outA;         // ACT1: packed u1; ACT2: packed u2; KR: packed K^-1 product
  int D, IN;
  int nch_out = kNskChunks;   // chunk count of the packed OUTPUT (outA)
  float* outN = nullptr;      // NSK_OUT / NSK_KR with plain output: row-major [kMP][N]
  // state update (the fields of StepEpi; x / xp / kr in the packed layout, one value per element instead of slabs)
  StepEpi step;
  const float* xpA;    // packed z of the previous evaluation
  float* xnA;          // packed z of the next one.  cur / prev / next ROTATE on the host: in MCD_ULA the state is the operand of
                       // the very launch that updates it, so no launch writes a buffer another workgroup may still be loading
  const float* krA;    // STEP: the K^-1 product of this evaluation (launch B)
  float* krOut;        // KR (packed, the overdamped sequence)
  float* krOutN = nullptr;   // KR, row-major [kMP][N] (the 2nd-order sequence's element-wise kernels read it)
  // r04: what the reverse sweep of the gradient used to RECOMPUTE per evaluation, kept by the forward's consumers instead —
  // row-major [M][N] blocks of this evaluation and pass inside [(K+1) n][N] tables of the gradient workspace (nullptr: no keep):
  // keepPre / keepU the ACT segment's pre-activation and output, keepKr the K^-1 product, keepSn the raw u2 W3 product
  float* keepPre = nullptr;
  float* keepU = nullptr;
  float* keepKr = nullptr;
  float* keepSn = nullptr;
  // KIND 3 (r04): the backward activation consumer of the reverse sweep (the arithmetic of EPI_ACTB): d u = product (+ d u of
  // the layer above), d a = d u sigmoid(pre); d u row-major, d a packed (the next launch's operand) and into the big matrix of
  // the deferred contraction, column sums over the pass's particles (one workgroup per column tile: a single writer)
  const float* bPre = nullptr;      // [kMP][IN] pre-activation of this layer (the forward's kept table)
  const float* bDuPrev = nullptr;   // [kMP][IN] d u of the layer above (nullptr: none)
  float* bDu = nullptr;             // [kMP][IN]
  float* bDaBig = nullptr;          // [(K+1) n][IN] rows of this evaluation and pass
  float* bSumA = nullptr;           // [IN] += sum_m d a
  float* bSumU = nullptr;           // [IN] += sum_m d u (nullptr: not needed)
  // NSK_UHA_MID: F1 of the 2nd-order sequence (momentum refresh + half kick + drift, mcd_under_lp_a_cais.py:52-63) as the
  // consumer of its launch L3 (u2 W3)
  struct UhaMid {
    const float* params; const float* tc; const float* sched;
    const float* zr;           // packed [z | rho]
    float* zrp;                // packed [z | rho']
    float* zn;                 // packed z'
    float* rpp;                // [kMP][D] rho''
    const float* kr;           // [kMP][D] K^-1 (z - mu0)
    const uint32_t* gkey;      // [kMP][2] G_i
    float* fkslot;             // [D / 16][kMP] forward-kernel log-density of the bridge on the tile's columns
    float* traj;
    int64_t n_total, base;
    cmcd_layout lay;
    int D, K, i;
  } um;
};

// One element (row m of the pass, column e) of evaluation i's state update — the arithmetic of lgcp_step_tile — in TWO
// phases around the GEMM: everything that does not depend on this launch's product (the element's state, the target's
// gradient from the previous launch's K^-1 product, grad log q, the Threefry block and its deviate: ~250 of the ~300
// instructions and the one load round trip) is issued BEFORE the consumer wave waits for its GEMM operands, so that the
// launch's tail after the last matrix instruction is the LDS sum, a dozen FMAs and the stores.
struct NskStepPre {
  float z, xp, kr, cnt, b3, gq, gp, ez, noise, fac;
  float mean, ld;
  uint32_t g0, g1;
  int64_t ix;
  // schedule scalars and the row's slot values (writer lanes): loaded up front too — after the last matrix instruction
  // nothing waits for a load any more
  float pbeta, peps, pcst, pinv2s2, beta, eps, sig, cst, inv2s2, mu0, pa, wsl, fksl;
};

// phase 0: the element's loads only (issued before the GEMM's own 26, so that they return first)
template <bool NO_NET>
__device__ __forceinline__ void nsk_step_loads(const NskArgs& a, int m, int e, NskStepPre& t) {
  const StepEpi& s = a.step;
  const int D = s.D;
  const float* counts = s.tc + (int64_t)D * D;
  const int ec = min(e, D - 1), mc = min(m, a.M - 1);        // clamped (always valid) addresses: padding lanes only join the butterfly
  t.ix = nsk_pack(mc, ec);
  t.z = a.xA[t.ix]; t.xp = a.xpA[t.ix];
  t.kr = NO_NET ? 0.f : a.krA[t.ix];
  t.cnt = counts[ec]; t.b3 = NO_NET ? 0.f : s.b3[ec];
  t.mean = s.params[s.lay.vd_mean + ec];
  t.ld = s.params[s.lay.vd_logdiag + ec];
  const uint32_t* gk = s.gkey + (s.i & 1) * 2 * kMP;
  t.g0 = gk[2 * mc]; t.g1 = gk[2 * mc + 1];
  t.fac = NO_NET ? 0.f : s.factor[0];
  const int i = s.i;
  const bool last = i == s.K;
  const float* sp = s.sched + 8 * (i > 0 ? i - 1 : 0);
  t.pbeta = sp[0]; t.peps = sp[1]; t.pcst = sp[3]; t.pinv2s2 = sp[4];
  const float* sc = s.sched + 8 * (last ? s.K - 1 : i);
  t.beta = sc[0]; t.eps = sc[1]; t.sig = sc[2]; t.cst = sc[3]; t.inv2s2 = sc[4];
  t.mu0 = s.tc[(int64_t)D * D + D]; t.pa = s.tc[(int64_t)D * D + D + 1];
  const int sl = (e >> 4) * kMP + mc;                        // the element's (tile, row) slot (e / 16 = the column tile)
  t.wsl = s.wslot[sl]; t.fksl = s.fkslot[sl];
}

// phase 1: everything that does not depend on this launch's product, in the shadow of the GEMM operands' flight
template <bool NO_NET>
__device__ __forceinline__ void nsk_step_pre(const NskArgs& a, int e, NskStepPre& t) {
  const StepEpi& s = a.step;
  const int D = s.D, H = (D + 1) / 2, i = s.i;
  const float pa = t.pa;
  const float clipv = s.var_mode ? 1e2f : 1e3f;
  const bool clip_p = s.grad_clipping != 0, clip_q = clip_p && s.var_mode;
  const int ec = min(e, D - 1);
  const float sd = expf(t.ld);
  t.ez = expf(t.z);
  t.gq = -(t.z - t.mean) / (sd * sd);
  if (clip_q) t.gq = fminf(fmaxf(t.gq, -clipv), clipv);
  t.gp = 0.f;
  if (!NO_NET) {
    t.gp = -t.kr + t.cnt - pa * t.ez;                          // grad log p      model_handler.py:386-396
    if (clip_p) t.gp = fminf(fmaxf(t.gp, -clipv), clipv);
  }
  t.noise = 0.f;
  if (i < s.K) {
    // eps_i = normal(G_i, (D,)): element e is word (e >= H) of the block with counters (j, H + j), j = e mod H
    const int j = ec < H ? ec : ec - H;
    uint32_t y0 = j, y1 = (H + j < D) ? H + j : 0;
    threefry2x32(t.g0, t.g1, y0, y1);
    t.noise = bits_to_normal(ec < H ? y0 : y1);
  }
}

template <bool NO_NET>
__device__ __forceinline__ void nsk_step_post(const NskArgs& a, const NskStepPre& t, float o, int m, int e, int tile, bool live) {
  const StepEpi& s = a.step;
  const int D = s.D, i = s.i;
  const float fsn = s.ula ? 0.f : 1.f;
  const float mu0 = t.mu0, pa = t.pa;
  const float clipv = s.var_mode ? 1e2f : 1e3f;
  const bool clip_p = s.grad_clipping != 0;
  const bool last = i == s.K;
  const float pbeta = t.pbeta, peps = t.peps, pcst = t.pcst, pinv2s2 = t.pinv2s2;
  const float beta = t.beta, eps = t.eps, sig = t.sig, cst = t.cst, inv2s2 = t.inv2s2;
  const float z = t.z, kr = NO_NET ? o : t.kr, gq = t.gq;
  float gp = t.gp;
  if (NO_NET) {                                               // MCD_ULA: the K^-1 product is this launch's own
    gp = -kr + t.cnt - pa * t.ez;
    if (clip_p) gp = fminf(fmaxf(gp, -clipv), clipv);
  }
  const float sn = NO_NET ? 0.f : (o + t.b3) * t.fac;         // factor_sn (u2 W3 + b3)           nn.py:70
  float bk_acc = 0.f, fk_acc = 0.f, lp_acc = 0.f, zn = 0.f;
  if (i > 0) {   // backward kernel of step i-1                             mcd_cais.py:71-86
    const float ub = -1.0f * (pbeta * gp + (1.0f - pbeta) * gq);
    const float bk = z - peps * ub + peps * sn;
    const float db = t.xp - bk;
    bk_acc = -(db * db) * pinv2s2 - pcst;
  }
  if (last) {    // log p(z_K)
    lp_acc = -0.5f * (z - mu0) * kr + z * t.cnt - pa * t.ez;
  } else {       // forward kernel of step i                                mcd_cais.py:52-67
    const float uf = -1.0f * (beta * gp + (1.0f - beta) * gq);
    const float fk = z - eps * uf - fsn * eps * sn;
    zn = fk + sig * t.noise;
    const float df = zn - fk;
    fk_acc = -(df * df) * inv2s2 - cst;
  }
  if (live) {
    if (last) {
      s.out_z[(int64_t)m * D + e] = z;
    } else {
      a.xnA[t.ix] = zn;
      if (s.traj) s.traj[((int64_t)(i + 1) * s.n_total + s.base + m) * D + e] = zn;
    }
    if (NO_NET) { if (a.keepKr) a.keepKr[(int64_t)m * D + e] = o; }
    else if (a.keepSn) a.keepSn[(int64_t)m * D + e] = o;        // the raw product: the sweep adds b3 and the factor itself
  }
  // per-row partial log-weights over the tile's 16 columns: the 16 lanes that share the row, fixed butterfly
  float bk_lp = live ? bk_acc : 0.f, fk_lp = live ? fk_acc : 0.f, lp = live ? lp_acc : 0.f;
#pragma unroll
  for (int ofs = 8; ofs > 0; ofs >>= 1) {
    bk_lp += __shfl_xor(bk_lp, ofs);
    fk_lp += __shfl_xor(fk_lp, ofs);
    lp += __shfl_xor(lp, ofs);
  }
  if ((threadIdx.x & 15) == 0 && m < a.M) {
    const int sl = tile * kMP + m;
    if (i > 0) s.wslot[sl] = t.wsl + (bk_lp - t.fksl);
    if (!last) s.fkslot[sl] = fk_lp;
    else s.lpslot[sl] = lp;
  }
}

// F1 of the 2nd-order sequence as a consumer (the arithmetic of lgcp_uha_mid_kernel), loads and the deviate ahead of the GEMM
struct NskMidPre {
  float z, rho, kr, b3, mean, ld, cnt, noise, fac, beta, eps, gamma, pa;
  uint32_t g0, g1;
  int64_t iz, ir, in;
};
__device__ __forceinline__ void nsk_mid_loads(const NskArgs& a, int m, int e, NskMidPre& t) {
  const NskArgs::UhaMid& u = a.um;
  const int D = u.D, ec = min(e, D - 1), mc = min(m, a.M - 1);
  t.iz = nsk_pack(mc, ec, 2 * kNskChunks); t.ir = nsk_pack(mc, D + ec, 2 * kNskChunks); t.in = t.iz;   // zn shares z's index
  t.z = u.zr[t.iz]; t.rho = u.zr[t.ir];
  t.kr = u.kr[(int64_t)mc * D + ec];
  t.b3 = u.params[u.lay.g_b3 + ec];
  t.mean = u.params[u.lay.vd_mean + ec];
  t.ld = u.params[u.lay.vd_logdiag + ec];
  t.cnt = u.tc[(int64_t)D * D + ec];
  t.pa = u.tc[(int64_t)D * D + D + 1];
  t.g0 = u.gkey[2 * mc]; t.g1 = u.gkey[2 * mc + 1];
  t.fac = u.params[u.lay.g_factor];
  t.beta = u.sched[8 * u.i]; t.eps = u.sched[8 * u.i + 1];
  t.gamma = u.params[u.lay.gamma];
}
__device__ __forceinline__ void nsk_mid_pre(const NskArgs& a, int e, NskMidPre& t) {
  const int D = a.um.D, H = (D + 1) / 2, ec = min(e, D - 1);
  const int j = ec < H ? ec : ec - H;
  uint32_t y0 = j, y1 = (H + j < D) ? H + j : 0;
  threefry2x32(t.g0, t.g1, y0, y1);
  t.noise = bits_to_normal(ec < H ? y0 : y1);
}
__device__ __forceinline__ void nsk_mid_post(const NskArgs& a, const NskMidPre& t, float o, int m, int e, int tile, bool live) {
  const NskArgs::UhaMid& u = a.um;
  const int D = u.D;
  const float eps = t.eps, beta = t.beta;
  const float eta = t.gamma * eps, ome = 1.0f - eta, sig = sqrtf(2.0f * eta);
  const float inv2s2 = 1.0f / (2.0f * sig * sig), cst = logf(sig) + kHalfLog2Pi;
  const float s1 = (o + t.b3) * t.fac;
  const float sd = expf(t.ld);
  float gp = -t.kr + t.cnt - t.pa * expf(t.z);
  gp = fminf(fmaxf(gp, -1e2f), 1e2f);                          // gradU(z, beta, clip=1e2)          :23-30
  const float gq = -(t.z - t.mean) / (sd * sd);
  const float uf = -1.0f * (beta * gp + (1.0f - beta) * gq);
  const float mf = t.rho * ome - 2.0f * eta * s1;               // :52-54
  const float rhop = mf + sig * t.noise;                        // :58-59
  const float df = rhop - mf;
  float fk = -(df * df) * inv2s2 - cst;
  const float rpp = rhop - eps * uf / 2.0f;                     // :62
  if (live) {
    u.zrp[t.iz] = t.z;
    u.zrp[t.ir] = rhop;
    u.zn[t.in] = t.z + eps * rpp;                               // :63
    u.rpp[(int64_t)m * D + e] = rpp;
    if (u.traj) u.traj[((int64_t)(2 * u.K + 2 + u.i) * u.n_total + u.base + m) * D + e] = rhop;
    if (a.keepSn) a.keepSn[(int64_t)m * D + e] = o;              // the raw u2 W3 product of the bridge's first evaluation
  }
  fk = live ? fk : 0.f;
#pragma unroll
  for (int ofs = 8; ofs > 0; ofs >>= 1) fk += __shfl_xor(fk, ofs);   // the 16 lanes that share the row, fixed butterfly
  if ((threadIdx.x & 15) == 0 && m < a.M) u.fkslot[tile * kMP + m] = fk;
}

// MERGED (17 .. 20 particles, the named batch): ONE workgroup per column tile serves rows 0 .. 15 on 16x16x4 and rows
// 16 .. 19 on v_mfma_f32_4x4x1 (16 blocks of 4 rows x 4 columns, block = (column group, k quarter)) against the SAME weight
// registers — the weights are fetched once instead of once per 16-row half (launch B: 202 workgroups, one per CU, instead of
// 404).  Lane l's 4x4x1 operands: A = packed operand of row 16 + l % 4 at the lane's k quarter, B = its 16x16x4 weight
// register; D register r of lane l = row 16 + r, column l % 16, partial over the lane's k quarter (summed over the four
// quarters in the LDS pass).  STEP launches carry one EXTRA workgroup (the last) that advances the chain's key.
// ROUNDS: rounds of 13 chunks per wave (compile time: the one-round instances of the overdamped sequence keep their registers —
// 126, two workgroups per CU for the 404-workgroup launch B of a 21 .. 32-particle pass; a run-time round loop cost them 8
// registers and 4 % of the call); 2 = the 3220-wide layers of the 2nd-order sequence (every segment of such a launch has its
// own round count <= ROUNDS)
// The XCD-paired grid order of the two-half launches (NskArgs::pair_gx).  Measured at N = 20, K = 128 in one lease
// (profiles/r05_d_lgcp_xcd_pairs_ab.txt): launch A 6.52 -> 5.96 us, C 7.76 -> 7.24 us, the call 2.999 -> 2.823 ms; launch B stays
// merged (one workgroup per tile: 8.00 us; two paired halves: 9.12 us).  -DCMCD_LGCP_NO_XCD_PAIRS builds the (tile, half) grid.
static constexpr bool lgcp_xcd_pairs() {
#ifdef CMCD_LGCP_NO_XCD_PAIRS
  return false;
#else
  return true;
#endif
}

template <int KIND, bool MERGED, int ROUNDS = 1>      // KIND 0: activations / plain products; 1: the state update; 2: the 2nd-order F1
__global__ __launch_bounds__(64 * kGemmWaves, (KIND || MERGED) ? 2 : 4) void lgcp_nsk_kernel(NskArgs a) {
  constexpr bool STEP = KIND == 1, MID = KIND == 2;
  __shared__ float red[kGemmWaves][MERGED ? 512 : 256];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int bx = blockIdx.x, by = blockIdx.y, gx = gridDim.x;
  if (!MERGED && a.pair_gx) {
    const int L = blockIdx.x, s = L >> 3;
    bx = (L & 7) + 8 * (s >> 1); by = s & 1; gx = a.pair_gx;
    if (bx >= gx) return;          // the tail of the last group of eight tiles (uniform per workgroup)
  }
  if (STEP && bx == gx - 1) {
    // the chain's key for evaluation i + 1 (a dependent integer chain of ~450 instructions): its own workgroup
    const StepEpi& st = a.step;
    if (by == 0 && wv == 0 && st.i + 1 < st.K && lane < a.M) {
      uint32_t k0 = st.gen[2 * lane], k1 = st.gen[2 * lane + 1], G0, G1;
      lgcp_key_advance(k0, k1, G0, G1);
      st.gen[2 * lane] = k0; st.gen[2 * lane + 1] = k1;
      uint32_t* gk = st.gkey + ((st.i + 1) & 1) * 2 * kMP;
      gk[2 * lane] = G0; gk[2 * lane + 1] = G1;
      uint32_t* gt = st.gktab + ((int64_t)(st.i + 1) * st.n_total + st.base + lane) * 2;
      gt[0] = G0; gt[1] = G1;
    }
    return;
  }
  const int sI = bx >= a.nt0 ? 1 : 0;
  const NskSeg sg = a.seg[sI];
  const int tile = bx - (sI ? a.nt0 : 0), half = MERGED ? 0 : by;
  // the element this thread consumes: waves 0 .. 3 the 16 x 16 block (D layout of 16x16x4: column = lane % 16, row =
  // 4 (lane / 16) + register, register = wave), wave 4 the 4 x 16 block of the merged form
  const bool cons = wv < 4 || (MERGED && wv == 4);
  const int n = tile * 16 + (lane & 15);
  const int row = wv < 4 ? half * 16 + 4 * (lane >> 4) + wv : 16 + (lane >> 4);
  const bool live = cons && n < sg.N && row < a.M;
  // ---- consumer operands first (they come from earlier launches): in flight beside the GEMM's own loads
  NskStepPre sp;
  NskMidPre mp;
  float cu = 0.f, cb = 0.f;
  int64_t cix = 0;
  float bpre = 0.f, bdup = 0.f;
  if (cons) {
    if (KIND == 3) {   // the backward consumer's own operands (earlier launches): issued ahead of the GEMM's loads
      const int64_t ixc = (int64_t)min(row, a.M - 1) * sg.N + min(n, sg.N - 1);
      bpre = a.bPre[ixc];
      bdup = a.bDuPrev ? a.bDuPrev[ixc] : 0.f;
    } else if (MID) {
      nsk_mid_loads(a, row, n, mp);
    } else if (STEP) {
      if (sg.epi == NSK_STEP) nsk_step_loads<false>(a, row, n, sp);
      else nsk_step_loads<true>(a, row, n, sp);
    } else if (sg.epi == NSK_ACT1 || sg.epi == NSK_ACT2) {
      const int nc = min(n, sg.N - 1), rc = min(row, a.M - 1);
      cix = nsk_pack(rc, nc, a.nch_out);
      cb = a.bias[nc];
      // u = [x; emb_i] (nn.py:68-69); x is this launch's own operand (its chunk count), u1 the previous launch's output
      cu = sg.epi == NSK_ACT2 ? a.uA[cix] : (nc < a.D ? a.xA[nsk_pack(rc, nc, sg.nch)] : a.emb[max(nc - a.D, 0)]);
    }
  }
  constexpr int nch = ROUNDS * kNskChunks;                      // every segment of a launch has ROUNDS rounds of 13 chunks per wave
  const f32x4* Ap = reinterpret_cast<const f32x4*>(sg.A) + ((int64_t)half * nch + wv) * 64 + lane;
  const f32x4* Wp = reinterpret_cast<const f32x4*>(sg.W) + ((int64_t)tile * nch + wv) * 64 + lane;
  const f32x4* A2p = reinterpret_cast<const f32x4*>(sg.A) + ((int64_t)nch + wv) * 64 + (lane & 3) + 16 * (lane >> 4);
  f32x4 acc[4], ac2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { acc[q] = f32x4{0.f, 0.f, 0.f, 0.f}; ac2[q] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float shift = sg.a_shift;
  // chunk c = wave + 8 j, j = 0 .. 13 ROUNDS - 1: at every j the workgroup reads 8 KB contiguous of either operand.  The first 13
  // chunks' loads are all in flight before the first wait, chunk by chunk (the first matrix instruction waits for two of them);
  // ROUNDS = 2: as soon as chunk j's matrix instructions have issued, its registers take chunk j + 13 — thirteen chunks stay in
  // flight through the whole contraction instead of two serial rounds (13.2 -> see profiles/r04_s_*)
  f32x4 av[kNskCpw], bv[kNskCpw], a2[MERGED ? kNskCpw : 1];
#pragma unroll
  for (int j = 0; j < kNskCpw; ++j) {
    bv[j] = Wp[j * 512]; av[j] = Ap[j * 512];
    if (MERGED) a2[j] = A2p[j * 512];
  }
  __builtin_amdgcn_sched_barrier(0);        // (the machine scheduler otherwise sinks the loads to their uses: 43 registers, 26 round trips)
  if (STEP && cons) {                        // waits for the consumer's own loads only (issued first): the GEMM's stay in flight
    if (sg.epi == NSK_STEP) nsk_step_pre<false>(a, n, sp);
    else nsk_step_pre<true>(a, n, sp);
  }
  if (MID && cons) nsk_mid_pre(a, n, mp);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < kNskCpw * ROUNDS; ++j) {
    const int sl = j % kNskCpw;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[sl][q] - shift, bv[sl][q], acc[q], 0, 0, 0);
    if (MERGED) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        ac2[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a2[sl][q] - shift, bv[sl][q], ac2[q], 0, 0, 0);
    }
    if (ROUNDS > 1 && j + kNskCpw < kNskCpw * ROUNDS) {
      __builtin_amdgcn_sched_barrier(0);
      bv[sl] = Wp[(j + kNskCpw) * 512]; av[sl] = Ap[(j + kNskCpw) * 512];
      if (MERGED) a2[sl] = A2p[(j + kNskCpw) * 512];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const f32x4 t = (acc[0] + acc[1]) + (acc[2] + acc[3]);      // fixed order
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][r * 64 + lane] = t[r];
  if (MERGED) {
    const f32x4 t2 = (ac2[0] + ac2[1]) + (ac2[2] + ac2[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][256 + r * 64 + lane] = t2[r];
  }
  __syncthreads();
  if (KIND == 3) {
    // every wave stays to the second barrier (the column sums cross the waves); only consumer waves hold an element
    __shared__ float csum[5][2][16];
    float v = 0.f;
    if (wv < 4) {
#pragma unroll
      for (int w = 0; w < kGemmWaves; ++w) v += red[w][wv * 64 + lane];        // fixed order
    } else if (MERGED && wv == 4) {
      const int r = lane >> 4, c = lane & 15;
#pragma unroll
      for (int w = 0; w < kGemmWaves; ++w)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) v += red[w][256 + r * 64 + c + 16 * kq];
    }
    float da = 0.f, du = 0.f;
    if (live) {
      const int64_t ix = (int64_t)row * sg.N + n;
      du = v + bdup;
      da = du * sigmoid_fast(bpre);
      a.bDu[ix] = du;
      a.outA[nsk_pack(row, n, a.nch_out)] = da;
      a.bDaBig[ix] = da;
    }
    if (cons) {
      float sa = da, su = du;     // rows of this wave: the four lane groups that share a column
      sa += __shfl_xor(sa, 16); sa += __shfl_xor(sa, 32);
      su += __shfl_xor(su, 16); su += __shfl_xor(su, 32);
      if (lane < 16) { csum[wv][0][lane] = sa; csum[wv][1][lane] = su; }
    }
    __syncthreads();
    if (wv == 0 && lane < 16 && n < sg.N) {
      float ta = 0.f, tu = 0.f;
      const int nw = MERGED ? 5 : 4;
      for (int w = 0; w < nw; ++w) { ta += csum[w][0][lane]; tu += csum[w][1][lane]; }   // fixed order
      a.bSumA[n] += ta;
      if (a.bSumU) a.bSumU[n] += tu;
    }
    return;
  }
  if (!cons) return;
  float v = 0.f;
  if (wv < 4) {
#pragma unroll
    for (int w = 0; w < kGemmWaves; ++w) v += red[w][wv * 64 + lane];        // fixed order
  } else {
    // row 16 + r, column c: the four k quarters (lanes c, c + 16, c + 32, c + 48) of every wave, fixed order
    const int r = lane >> 4, c = lane & 15;
#pragma unroll
    for (int w = 0; w < kGemmWaves; ++w)
#pragma unroll
      for (int kq = 0; kq < 4; ++kq) v += red[w][256 + r * 64 + c + 16 * kq];
  }
  if (STEP) {
    if (sg.epi == NSK_STEP) nsk_step_post<false>(a, sp, v, row, n, tile, live);
    else nsk_step_post<true>(a, sp, v, row, n, tile, live);
    return;
  }
  if (MID) {
    nsk_mid_post(a, mp, v, row, n, tile, live);
    return;
  }
  if (!live) return;
  if (sg.epi == NSK_KR || sg.epi == NSK_OUT) {                 // plain products: the K^-1 product (second segment of launch B) / u2 W3
    if (a.outN && sg.epi == NSK_OUT) a.outN[(int64_t)row * sg.N + n] = v;
    else if (a.krOutN) a.krOutN[(int64_t)row * sg.N + n] = v;
    else a.krOut[nsk_pack(row, n)] = v;
    if (sg.epi == NSK_KR && a.keepKr) a.keepKr[(int64_t)row * sg.N + n] = v;
    if (sg.epi == NSK_OUT && a.keepSn) a.keepSn[(int64_t)row * sg.N + n] = v;
  } else {
    const float pre = v + cb, u = cu + softplus(pre);          // nn.py:45-50
    a.outA[cix] = u;
    if (a.keepPre) {
      a.keepPre[(int64_t)row * sg.N + n] = pre;
      a.keepU[(int64_t)row * sg.N + n] = u;
    }
  }
}

// packed copies of up to four [K][N] weight matrices (row stride = N) in ONE launch (blockIdx.y = matrix): one thread per
// 16-byte group {k, k+1, k+2, k+3} x column
struct NskPackArgs {
  const float* src[4];
  float* dst[4];
  int K[4], N[4], ntile[4];
  int nch[4];          // chunks per tile of the packed copy (0: kNskChunks)
  int ld[4];           // row stride of the source (0: N)
};
__global__ void lgcp_nsk_pack_kernel(NskPackArgs a) {
  const int q = blockIdx.y;
  const float* __restrict__ src = a.src[q];
  if (!src) return;
  const int K = a.K[q], N = a.N[q], nch = a.nch[q] ? a.nch[q] : kNskChunks, ld = a.ld[q] ? a.ld[q] : N;
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // (tile, chunk, lane)
  if (g >= (int64_t)a.ntile[q] * nch * 64) return;
  const int lane = (int)(g & 63), chunk = (int)((g >> 6) % nch), tile = (int)((g >> 6) / nch);
  const int n = tile * 16 + (lane & 15), k0 = chunk * 16 + 4 * (lane >> 4);
  f32x4 v;
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (n < N && k0 + j < K) ? src[(int64_t)(k0 + j) * ld + n] : 0.f;
  reinterpret_cast<f32x4*>(a.dst[q])[g] = v;
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
  float* x;                  // [kMP][D]   z_0
  float* w;                  // [kMP]      -log q(z_0)
  uint32_t* keys;            // [kMP][2]   gen key of the chain
  uint32_t* gkey;            // [2][kMP][2] noise key of evaluation 0 (forward path only; nullptr: not needed)
  uint32_t* gktab;           // [K][n_total][2] (forward path only)
  float* traj;               // optional [K+1][n_total][D]: z_0..z_K of every particle (reverse sweep of the gradient)
  int64_t n_total, base;     // trajectory row of particle p of this pass: base + p
  cmcd_layout lay;
  int M, D;
  int packed;                // x in the packed operand layout of the no-split-K GEMM (nsk_pack)
};

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

template <int NW>   // NW waves per block, fixed order
__device__ __forceinline__ float block_sum_n(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < NW; ++w) t += sh[w];
  return t;
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
        a.x[a.packed ? nsk_pack(p, idx[q]) : (int64_t)p * D + idx[q]] = z;
        if (a.traj) a.traj[(a.base + p) * D + idx[q]] = z;
        const float dz = z - mean;
        acc += -(dz * dz) / (2.0f * sd * sd) - logf(sd) - kHalfLog2Pi;
      }
    }
  }
  const float lq = block_sum_256(acc, sh);
  if (threadIdx.x == 0) {
    a.w[p] = -lq;
    uint32_t c0 = 0, c2 = 2, c1 = 1, c3 = 3;
    threefry2x32(b0, b1, c0, c2);
    threefry2x32(b0, b1, c1, c3);       // C = (c0, c1)
    uint32_t g0 = 0, g2 = 2, g1 = 1, g3 = 3;
    threefry2x32(c0, c1, g0, g2);
    threefry2x32(c0, c1, g1, g3);       // gen = second(split(C)) = (g2, g3)
    uint32_t k0 = g2, k1 = g3;
    if (a.gkey) {                        // G_0 for evaluation 0; the GEMM epilogue carries the chain from here
      uint32_t G0, G1;
      lgcp_key_advance(k0, k1, G0, G1);
      a.gkey[2 * p] = G0;
      a.gkey[2 * p + 1] = G1;
      a.gktab[(a.base + p) * 2] = G0;
      a.gktab[(a.base + p) * 2 + 1] = G1;
    }
    a.keys[2 * p] = k0;
    a.keys[2 * p + 1] = k1;
  }
}

// loss = -(w_0 + sum over column blocks of the closed steps' (bk - fk) + log p(z_K)); fixed summation order
struct LgcpFinalArgs {
  const float* w0;       // [kMP]  -log q(z_0)
  const float* wslot;    // [ncb][kMP]
  const float* lpslot;   // [ncb][kMP]
  const float* tc;
  float* out_loss;       // [M]
  double* partials;      // [M][5]
  int M, D, ncb;
};

__global__ __launch_bounds__(64) void lgcp_final_kernel(LgcpFinalArgs a) {
  // one wave per particle: lane c takes slot rows c, c + 64, ... (two dependent loads instead of a hundred), then a fixed
  // butterfly — the same order on every call
  const int p = blockIdx.x, lane = threadIdx.x;
  float w = 0.f, lp = 0.f;
  for (int cb = lane; cb < a.ncb; cb += 64) {
    w += a.wslot[cb * kMP + p];
    lp += a.lpslot[cb * kMP + p];
  }
  w = wave_sum64(w);
  lp = wave_sum64(lp);
  if (lane != 0) return;
  w += a.w0[p];
  w += lp + a.tc[(int64_t)a.D * a.D + a.D + 2];     // + log p(z_K)   mcdboundingmachine.py:178
  const float loss = -w;
  a.out_loss[p] = loss;
  double* o = a.partials + (int64_t)p * CMCD_NSTATS;
  o[0] = isfinite(loss) ? 1.0 : 0.0;
  o[1] = loss;
  o[2] = (double)loss * (double)loss;
  o[3] = -(double)loss;
  o[4] = isfinite(loss) ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
// Per-pass buffers (one "lane"): up to kLanes passes of <= kMP particles run concurrently, each on its own HIP stream
// and buffer set (every launch of the sequence is latency-bound, so independent passes overlap almost freely).  Lane 0
// is laid out exactly as the single-lane workspace always was (lgcp_grad reads bias1 / gktab at those offsets); the
// other lanes' copies follow the statistics records.
constexpr int kLanes = 4;
struct LgcpLane {
  int64_t x, xp, u1, u2, pre1, pre2, kr, slab1, slab2, sn, w, keys, gkey, slots, counters;
  int64_t nsk;          // no-split-K form: packed x | xp | xn | u1 | u2 | kr (kNskOperand floats each), then its slots [3][D / 16][kMP]
};
struct LgcpWs {
  int64_t bias1, gktab, partials, total;
  int64_t w1p, w2p, w3p, kip;    // no-split-K form: packed weights (once per call, shared by the lanes)
  LgcpLane lane[kLanes];
};

// the no-split-K forward serves operands of up to 1664 inputs (kNskChunks) whose state part is whole 16-column tiles;
// desc.reserved == 3 pins the split-K sequence (A / B measurements, tests)
static bool lgcp_nsk_ok(const cmcd_desc& d) {
  const int D = d.dim, IN = D + d.emb_dim;
  return d.reserved != 3 && D % 16 == 0 && IN <= 16 * kNskChunks && d.mode != CMCD_MODE_CAIS_UHA_SN;
}

static int lgcp_lanes(int64_t n) {
  const int64_t passes = (n + kMP - 1) / kMP;
  return (int)(passes < kLanes ? passes : kLanes);
}

static LgcpWs lgcp_ws(const cmcd_desc& d, int64_t n, int64_t base) {
  const int64_t D = d.dim, IN = D + d.emb_dim, K = d.nbridges;
  LgcpWs w;
  int64_t o = base;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  auto take_lane_head = [&](LgcpLane& l) {
    l.x = take(kMP * D); l.xp = take(kMP * D);
    l.u1 = take(kMP * IN); l.u2 = take(kMP * IN); l.pre1 = take(kMP * IN); l.pre2 = take(kMP * IN);
    l.kr = take(kSplit * kMP * D); l.slab1 = take(kSplit * kMP * IN); l.slab2 = take(kSplit * kMP * IN);
    l.sn = take(kSplit * kMP * D);
    l.w = take(kMP); l.keys = take(2 * kMP); l.gkey = take(4 * kMP);
  };
  auto take_lane_tail = [&](LgcpLane& l) {
    l.slots = take(3 * ((D + 63) / 64) * kMP);             // wslot | fkslot | lpslot
    l.counters = take(((D + 63) / 64) + ((IN + 63) / 64)); // int arrival counters of the widest launch
  };
  w.bias1 = take((K + 1) * IN);
  take_lane_head(w.lane[0]);
  w.gktab = take(2 * K * n);
  take_lane_tail(w.lane[0]);
  o = (o + 1) & ~int64_t(1);
  w.partials = take(n * CMCD_NSTATS * 2);
  for (int l = 1; l < kLanes; ++l) {
    if (l < lgcp_lanes(n)) {
      take_lane_head(w.lane[l]);
      take_lane_tail(w.lane[l]);
    } else {
      w.lane[l] = w.lane[0];
    }
  }
  w.w1p = w.w2p = w.w3p = w.kip = 0;
  if (lgcp_nsk_ok(d)) {
    const int64_t tIN = (IN + 15) / 16, tD = D / 16, per_tile = (int64_t)kNskChunks * 256;
    w.w1p = take(tIN * per_tile); w.w2p = take(tIN * per_tile); w.w3p = take(tD * per_tile); w.kip = take(tD * per_tile);
    for (int l = 0; l < kLanes; ++l)
      w.lane[l].nsk = (l < lgcp_lanes(n) || l == 0) ? take(kNskOps * kNskOperand + 3 * tD * kMP) : w.lane[0].nsk;
  }
  w.total = o;
  return w;
}

struct LgcpUhaWs;
static LgcpUhaWs lgcp_uha_ws(const cmcd_desc& d, int64_t n, int64_t base);
static int64_t lgcp_uha_ws_total(const cmcd_desc& d, int64_t n, int64_t base);
// r04: tables of the gradient workspace that the forward's consumers fill when it runs for a gradient call (lgcp_keep below
// lgcp_grad_ws): [(K+1) n][IN] pre1, u1, pre2, u2 and [(K+1) n][D] kr, sn.  `on` is false when the forward does not run on the
// no-split-K GEMM or the tables would pass 1 GB: the reverse sweep then recomputes every evaluation as it did in r01 - r03.
struct LgcpKeep { float *pre1, *u1, *pre2, *u2, *kr, *sn; bool on; };
static LgcpKeep lgcp_keep(const cmcd_desc& d, int64_t n, float* gws);

static int lgcp_uha_forward(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const int32_t* seeds, int64_t n,
                            const float* params, const float* tc, float* ws, float* out_loss, float* out_z,
                            double** partials_out, float* traj, hipStream_t stream, float* keep_gws);
// the 2nd-order sequence's kept tables: [2 K n][IN] pre1, u1, pre2, u2 and [2 K n][D] sn (rows 2 i: s([z_i; rho_i], i), 2 i + 1:
// s([z_i; rho'_i], i)), [(K+1) n][D] kr (K^-1 (z_e - mu0))
static LgcpKeep lgcp_uha_keep(const cmcd_desc& d, int64_t n, float* gws);

// desc.reserved (the kernel-variant hook of tests / probes): 1 pins the 32-row passes, 2 the wide-batch path
bool lgcp_use_wide(const cmcd_desc& d, int64_t n, bool keeps_trajectory) {
  if (keeps_trajectory || !lgcp_wide_supported(d) || d.reserved == 1) return false;
  return d.reserved == 2 || n >= kLgcpWideMin;
}

int64_t lgcp_workspace_floats(const cmcd_desc& d, int64_t n, int64_t base) {
  if (d.mode == CMCD_MODE_CAIS_UHA_SN) return lgcp_uha_ws_total(d, n, base);
  // a gradient call keeps the trajectory and stays on the 32-row passes: the query covers whichever form the call takes
  const int64_t narrow = lgcp_ws(d, n, base).total;
  return lgcp_use_wide(d, n, false) ? std::max(narrow, lgcp_wide_workspace_floats(d, n, base)) : narrow;
}

int lgcp_launch_prep(const cmcd_desc& d, const cmcd_layout& lay, const float* params, float* bias1, void* stream) {
  const int D = d.dim, E = d.emb_dim, IN = D + E, K = d.nbridges;
  LgcpPrepArgs pa{params, bias1, lay, D, E, K, IN};
  hipLaunchKernelGGL(lgcp_prep_kernel, dim3((IN + 255) / 256, K + 1), dim3(256), 0, static_cast<hipStream_t>(stream), pa);
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

static int lgcp_gemm_attrs() {
  const int gemm_lds = int(size_t(kStage * kAsLd + kQuarters * kMP * 64) * 4);
  const void* fns[5] = {reinterpret_cast<const void*>(lgcp_gemm_kernel<EPI_ACTB>),
                        reinterpret_cast<const void*>(lgcp_gemm_kernel<EPI_NONE>),
                        reinterpret_cast<const void*>(lgcp_gemm_kernel<EPI_ACT>),
                        reinterpret_cast<const void*>(lgcp_gemm_kernel<EPI_STEP>),
                        reinterpret_cast<const void*>(lgcp_gemm_kernel<EPI_STEP_NONET>)};
  for (const void* fn : fns)
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, gemm_lds) != hipSuccess) return -1;
  return gemm_lds;
}

int lgcp_forward(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const int32_t* seeds, int64_t n,
                 const float* params, const float* tc, float* ws, float* out_loss, float* out_z,
                 double** partials_out, float* traj, void* stream_, bool tables_ready, float* keep_gws) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (d.mode == CMCD_MODE_CAIS_UHA_SN)
    return lgcp_uha_forward(d, lay, sw, seeds, n, params, tc, ws, out_loss, out_z, partials_out, traj, stream, keep_gws);
  if (lgcp_use_wide(d, n, traj != nullptr))
    return lgcp_wide_forward(d, lay, sw, seeds, n, params, tc, ws, out_loss, out_z, partials_out, stream_, tables_ready);
  const int D = d.dim, E = d.emb_dim, IN = D + E, K = d.nbridges;
  const LgcpWs w = lgcp_ws(d, n, sw.total_floats);
  if (d.mode != CMCD_MODE_ULA && !tables_ready) {   // MCD_ULA has no network leaves at all
    LgcpPrepArgs pa{params, ws + w.bias1, lay, D, E, K, IN};
    hipLaunchKernelGGL(lgcp_prep_kernel, dim3((IN + 255) / 256, K + 1), dim3(256), 0, stream, pa);
  }
  const int gemm_lds = lgcp_gemm_attrs();
  if (gemm_lds < 0) return CMCD_ERR_HIP;
  // r04: the forward runs on the no-split-K GEMM (header above lgcp_nsk_kernel); weights re-packed once per call
  const bool nsk = lgcp_nsk_ok(d);
  const int tIN = (IN + 15) / 16, tD = D / 16;
  if (nsk && !tables_ready) {
    NskPackArgs pk{};
    int np = 0;
    auto pack = [&](const float* src, int Kr, int Nr, int64_t dst, int nt) {
      pk.src[np] = src; pk.dst[np] = ws + dst; pk.K[np] = Kr; pk.N[np] = Nr; pk.ntile[np] = nt; ++np;
    };
    pack(tc, D, D, w.kip, tD);
    if (d.mode != CMCD_MODE_ULA) {
      pack(params + lay.g_w1, D, IN, w.w1p, tIN);     // the state rows W1[:d] only: the embedding rows are in bias1
      pack(params + lay.g_w2, IN, IN, w.w2p, tIN);
      pack(params + lay.g_w3, IN, D, w.w3p, tD);
    }
    const int64_t groups = (int64_t)tIN * kNskChunks * 64;
    hipLaunchKernelGGL(lgcp_nsk_pack_kernel, dim3((unsigned)((groups + 255) / 256), np), dim3(256), 0, stream, pk);
  }
  const float* kinv = tc;
  double* partials = reinterpret_cast<double*>(ws + w.partials);
  *partials_out = partials;
  // mu0 is a model constant, log(126) - 0.5 * 1.91 (model_handler.py:346); the state update reads the device copy in
  // tc, the GEMM takes it as the operand shift
  const float mu0 = 3.8812819069514780f;
  const dim3 gblock(64 * kGemmWaves), gblock_step(64 * (kGemmWaves + 1));
  const int cbD = (D + 63) / 64, cbIN = (IN + 63) / 64;

  // Passes of <= kMP particles are independent chains of 3 (K + 1) latency-bound launches: up to kLanes of them run side
  // by side, lane 0 on the caller's stream, the others on per-thread side streams forked from / joined to it by events
  // (capturable in a graph, like the reverse sweep's two streams).  One pass (the named N = 20 batch): no side stream.
  const int lanes = lgcp_lanes(n);
  // side streams and events belong to ONE device: keyed by the current device (a host thread may drive several GPUs)
  struct SideSet { hipStream_t side[kLanes]; hipEvent_t ev_join[kLanes]; hipEvent_t ev_fork; };
  constexpr int kMaxDev = 16;
  static thread_local SideSet sets[kMaxDev] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return CMCD_ERR_HIP;
  SideSet& ss = sets[dev];
  hipStream_t* side = ss.side;
  hipEvent_t* ev_join = ss.ev_join;
  hipEvent_t& ev_fork = ss.ev_fork;
  if (lanes > 1 && !ev_fork) {
    for (int l = 1; l < kLanes; ++l) {
      if (hipStreamCreateWithFlags(&side[l], hipStreamNonBlocking) != hipSuccess) return CMCD_ERR_HIP;
      if (hipEventCreateWithFlags(&ev_join[l], hipEventDisableTiming) != hipSuccess) return CMCD_ERR_HIP;
    }
    if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return CMCD_ERR_HIP;
  }
  if (lanes > 1 && hipEventRecord(ev_fork, stream) != hipSuccess) return CMCD_ERR_HIP;   // behind the prep launch
  // from here on the side streams are forked from the caller's stream: every exit joins them again (an un-joined fork
  // would invalidate a graph capture and leave work of this call running behind the caller's back)
  bool forked[kLanes] = {false, false, false, false};
  auto join_all = [&]() {
    bool ok = true;
    for (int l = 1; l < lanes; ++l) {
      if (!forked[l]) continue;
      ok = hipEventRecord(ev_join[l], side[l]) == hipSuccess && ok;
      ok = hipStreamWaitEvent(stream, ev_join[l], 0) == hipSuccess && ok;
    }
    return ok;
  };
  auto bail = [&]() { join_all(); return (int)CMCD_ERR_HIP; };

  int* counters[kLanes];
  for (int l = 0; l < lanes; ++l) {
    hipStream_t st_l = l == 0 ? stream : side[l];
    if (l > 0) {
      if (hipStreamWaitEvent(st_l, ev_fork, 0) != hipSuccess) return bail();
      forked[l] = true;
    }
    counters[l] = reinterpret_cast<int*>(ws + w.lane[l].counters);
    if (hipMemsetAsync(counters[l], 0, sizeof(int) * (cbD + cbIN), st_l) != hipSuccess) return bail();
  }
  const int ula = d.mode == CMCD_MODE_ULA ? 1 : (d.mode == CMCD_MODE_ULA_SN ? 2 : 0);
  const LgcpKeep keep = keep_gws ? lgcp_keep(d, n, keep_gws) : LgcpKeep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false};
  // groups of `lanes` passes; inside a group the launches are enqueued evaluation by evaluation, round-robin over the
  // lanes, so that the chains advance together (the same weights are in flight for all of them)
  for (int64_t gbase = 0; gbase < n; gbase += (int64_t)lanes * kMP) {
    GemmArgs g[kLanes];
    int M[kLanes], live = 0;
    for (int l = 0; l < lanes; ++l) {
      const int64_t base = gbase + (int64_t)l * kMP;
      if (base >= n) break;
      ++live;
      hipStream_t st_l = l == 0 ? stream : side[l];
      const LgcpLane& wl = w.lane[l];
      M[l] = (int)((n - base) < kMP ? (n - base) : kMP);
      if (hipMemsetAsync(ws + wl.slots, 0, sizeof(float) * 3 * cbD * kMP, st_l) != hipSuccess) return bail();
      // packed operands: the padding (rows >= M, inputs >= IN) must read as zeros; slots start at zero
      if (nsk && hipMemsetAsync(ws + wl.nsk, 0, sizeof(float) * (kNskOps * kNskOperand + 3 * (int64_t)tD * kMP), st_l) != hipSuccess)
        return bail();
      LgcpStateArgs st{};
      st.seeds = seeds + base; st.params = params; st.tc = tc; st.x = nsk ? ws + wl.nsk : ws + wl.x; st.packed = nsk ? 1 : 0;
      st.w = ws + wl.w; st.keys = reinterpret_cast<uint32_t*>(ws + wl.keys);
      st.gkey = reinterpret_cast<uint32_t*>(ws + wl.gkey); st.gktab = reinterpret_cast<uint32_t*>(ws + w.gktab);
      st.lay = lay; st.M = M[l]; st.D = D;
      st.traj = traj; st.n_total = n; st.base = base;
      hipLaunchKernelGGL(lgcp_init_kernel, dim3(M[l]), dim3(256), 0, st_l, st);

      g[l] = GemmArgs{};
      g[l].M = M[l]; g[l].counters = counters[l];
      g[l].act.x = ws + wl.x; g[l].act.D = D; g[l].act.IN = IN;
      StepEpi& se = g[l].step;
      se.params = params; se.tc = tc; se.sched = ws + sw.sched; se.x = ws + wl.x; se.xp = ws + wl.xp;
      se.kr = ws + wl.kr; se.b3 = params + lay.g_b3; se.factor = params + lay.g_factor;
      se.gen = reinterpret_cast<uint32_t*>(ws + wl.keys); se.gkey = reinterpret_cast<uint32_t*>(ws + wl.gkey);
      se.gktab = reinterpret_cast<uint32_t*>(ws + w.gktab);
      se.wslot = ws + wl.slots; se.fkslot = se.wslot + cbD * kMP; se.lpslot = se.fkslot + cbD * kMP;
      if (nsk) {   // one slot row per 16-column tile
        se.wslot = ws + wl.nsk + kNskOps * kNskOperand; se.fkslot = se.wslot + tD * kMP; se.lpslot = se.fkslot + tD * kMP;
      }
      se.out_z = out_z + base * D; se.traj = traj; se.n_total = n; se.base = base; se.lay = lay;
      se.D = D; se.K = K; se.var_mode = d.mode == CMCD_MODE_CAIS_VAR_SN ? 1 : 0; se.grad_clipping = d.grad_clipping;
      se.ula = ula;
    }
    for (int i = 0; i <= K; ++i) {
      // CAIS: s(z_i, i) serves both kernels; MCD_ULA_sn: s(z_i, i - 1) serves the backward kernel only
      const int it = ula == 2 ? (i > 0 ? i - 1 : 0) : i;
      const int ie = it < K ? it : K - 1;
      for (int l = 0; l < live; ++l) {
        hipStream_t st_l = l == 0 ? stream : side[l];
        const LgcpLane& wl = w.lane[l];
        GemmArgs& gl = g[l];
        gl.step.i = i;
        if (nsk) {
          // evaluation i reads z_i from buffer i % 3, z_{i-1} from (i + 2) % 3 and writes z_{i+1} into (i + 1) % 3
          float* const sb = ws + wl.nsk;
          float* const xA = sb + (i % 3) * kNskOperand;
          float* const u1A = sb + 3 * kNskOperand, *u2A = u1A + kNskOperand, *krA = u2A + kNskOperand;
          NskArgs na{};
          na.M = M[l]; na.D = D; na.IN = IN; na.xA = xA; na.xpA = sb + ((i + 2) % 3) * kNskOperand;
          na.xnA = sb + ((i + 1) % 3) * kNskOperand; na.krA = krA; na.krOut = krA; na.step = gl.step;
          // 17 .. 20 particles (the named batch): one workgroup per column tile serves both row blocks (16x16x4 + 4x4x1);
          // otherwise one workgroup per (tile, 16-row half).  State-update launches carry one extra workgroup (key chain).
          // Measured at N = 20 (profiles/r04_f_lgcp_nsk_per_launch_n20.txt): merged helps the launch that is otherwise two
          // workgroups per CU (B: 9.96 -> 7.44 us) and costs the two that are one per CU either way (A 6.48 -> 7.08, C 7.64 ->
          // 8.40: half as many CUs pull the bytes), so only launch B takes it.
          const bool can_merge = M[l] > 16 && M[l] <= 20;
          auto launch = [&](bool step, int tiles, bool want_merged = false) {
            const bool merged = can_merge && want_merged;
            const unsigned gy = (!merged && M[l] > 16) ? 2 : 1;
            dim3 grid((unsigned)tiles + (step ? 1 : 0), gy);
            na.pair_gx = 0;
            if (gy == 2 && lgcp_xcd_pairs()) {   // both row halves of a column tile on one XCD (NskArgs::pair_gx)
              na.pair_gx = (int)grid.x;
              grid = dim3(16u * ((grid.x + 7) / 8), 1);
            }
            if (step) {
              if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<1, true>), grid, gblock, 0, st_l, na);
              else hipLaunchKernelGGL((lgcp_nsk_kernel<1, false>), grid, gblock, 0, st_l, na);
            } else {
              if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<0, true>), grid, gblock, 0, st_l, na);
              else hipLaunchKernelGGL((lgcp_nsk_kernel<0, false>), grid, gblock, 0, st_l, na);
            }
          };
          // (gradient calls: this evaluation's rows of the kept tables)
          const int64_t krow = (int64_t)i * n + gbase + (int64_t)l * kMP;
          if (ula == 1) {   // MCD_ULA: one launch per evaluation, (x - mu0) K^-1 with the state update as its consumer
            na.seg[0] = NskSeg{xA, ws + w.kip, D, NSK_STEP_NONET, mu0}; na.nt0 = tD;
            if (keep.on) na.keepKr = keep.kr + krow * D;
            launch(true, tD);
            continue;
          }
          // A: x W1[:D] -> u1 = [x; emb_i] + softplus(. + bias1_i)
          na.seg[0] = NskSeg{xA, ws + w.w1p, IN, NSK_ACT1, 0.f}; na.nt0 = tIN;
          na.bias = ws + w.bias1 + (int64_t)it * IN; na.emb = params + lay.g_emb + (int64_t)ie * E; na.outA = u1A;
          if (keep.on) { na.keepPre = keep.pre1 + krow * IN; na.keepU = keep.u1 + krow * IN; }
          launch(false, tIN);
          // B: u1 W2 -> u2 = u1 + softplus(. + b2)   |   (x - mu0) K^-1 -> kr
          na.seg[0] = NskSeg{u1A, ws + w.w2p, IN, NSK_ACT2, 0.f};
          na.seg[1] = NskSeg{xA, ws + w.kip, D, NSK_KR, mu0};
          na.bias = params + lay.g_b2; na.uA = u1A; na.outA = u2A;
          if (keep.on) { na.keepPre = keep.pre2 + krow * IN; na.keepU = keep.u2 + krow * IN; na.keepKr = keep.kr + krow * D; }
          launch(false, tIN + tD, true);
          // C: u2 W3 -> the state update of evaluation i on the tile's elements
          na.seg[0] = NskSeg{u2A, ws + w.w3p, D, NSK_STEP, 0.f}; na.nt0 = tD;
          na.keepPre = nullptr; na.keepU = nullptr; na.keepKr = nullptr;
          if (keep.on) na.keepSn = keep.sn + krow * D;
          launch(true, tD);
          continue;
        }
        if (ula == 1) {   // MCD_ULA: one launch per evaluation, [x - mu0] Kinv with the state update as its consumer
          // the state is this launch's operand: z_i / z_{i-1} / z_{i+1} rotate through x, xp and the (unused: no network)
          // u1 buffer instead of an in-place update that a slower workgroup of another column block could still be loading
          float* const sb3[3] = {ws + wl.x, ws + wl.xp, ws + wl.u1};
          gl.step.x = sb3[i % 3]; gl.step.xp = sb3[(i + 2) % 3]; gl.step.xn = sb3[(i + 1) % 3];
          gl.Kdim = D; gl.Kdim1 = 0;
          gl.seg[0] = GemmSeg{gl.step.x, kinv, ws + wl.kr, D, D, D, D, mu0};
          gl.nblk0 = cbD; gl.epi_seg = -1;
          hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_STEP_NONET>, dim3(cbD, kSplit), gblock_step, gemm_lds, st_l, gl);
          continue;
        }
        // A: x W1[:D] -> pre1 slabs -> u1 = [x; emb_i] + softplus(pre1 + bias1_i)            (fused consumer)
        gl.Kdim = D; gl.Kdim1 = 0;
        gl.seg[0] = GemmSeg{ws + wl.x, params + lay.g_w1, ws + wl.slab1, IN, D, IN, IN};
        gl.nblk0 = cbIN; gl.epi_seg = -1;
        gl.act.mode = 1; gl.act.bias = ws + w.bias1 + (int64_t)it * IN; gl.act.emb = params + lay.g_emb + (int64_t)ie * E;
        gl.act.sum_out = ws + wl.pre1; gl.act.u_prev = nullptr; gl.act.u_out = ws + wl.u1;
        hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACT>, dim3(cbIN, kSplit), gblock, gemm_lds, st_l, gl);
        // B: u1 W2 -> pre2 slabs -> u2 = u1 + softplus(pre2 + b2)                             (fused consumer)
        //    [x - mu0] Kinv -> kr slabs (summed by the state update in launch C): this product needs only the state, so
        //    it rides in the LIGHTEST launch of the three (r02: it used to double launch A's grid to 408 workgroups; the
        //    GEMM kernel now fits two workgroups per CU, so B's 408 run side by side: A 15 -> ~10 us, B 9 -> ~11 us)
        gl.Kdim = IN; gl.Kdim1 = D;
        gl.seg[0] = GemmSeg{ws + wl.u1, params + lay.g_w2, ws + wl.slab2, IN, IN, IN, IN};
        gl.seg[1] = GemmSeg{ws + wl.x, kinv, ws + wl.kr, D, D, D, D, mu0};
        gl.nblk0 = cbIN; gl.epi_seg = 0;
        gl.act.mode = 2; gl.act.bias = params + lay.g_b2; gl.act.sum_out = ws + wl.pre2; gl.act.u_prev = ws + wl.u1;
        gl.act.u_out = ws + wl.u2;
        hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACT>, dim3(cbIN + cbD, kSplit), gblock, gemm_lds, st_l, gl);
        gl.Kdim1 = 0;
        // C: u2 W3 -> sn slabs -> state update of evaluation i on the block's columns
        gl.seg[0] = GemmSeg{ws + wl.u2, params + lay.g_w3, ws + wl.sn, D, IN, D, D};
        gl.nblk0 = cbD;
        hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_STEP>, dim3(cbD, kSplit), gblock_step, gemm_lds, st_l, gl);
      }
    }
    for (int l = 0; l < live; ++l) {
      hipStream_t st_l = l == 0 ? stream : side[l];
      const int64_t base = gbase + (int64_t)l * kMP;
      const StepEpi& se = g[l].step;
      LgcpFinalArgs fa{ws + w.lane[l].w, se.wslot, se.lpslot, tc, out_loss + base, partials + base * CMCD_NSTATS, M[l], D,
                       nsk ? tD : cbD};
      hipLaunchKernelGGL(lgcp_final_kernel, dim3(M[l]), dim3(64), 0, st_l, fa);
    }
  }
  if (!join_all()) return CMCD_ERR_HIP;
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}


// ------------------------------------------------------------------------------------------
// Mean-field VI on the lgcp target (nbridges = 0; /root/reference/src/boundingmachine.py:73-111 with
// /root/reference/src/main.py:82-109): z = mean + std e, loss = log q(z) - log p(z), and under the
// reparameterisation d loss / d mean = -grad log p(z), d loss / d logdiag = -1 - grad log p(z) std e.
// Per pass of <= kMP particles: init (z, -log q) -> one skinny GEMM (z - mu0) K^-1 -> finish.
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
  take(kMP * (int64_t)D); take(kSplit * kMP * (int64_t)D);
  take(kMP); take(2 * kMP);
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
  const int64_t ox = take(kMP * (int64_t)D), okr = take(kSplit * kMP * (int64_t)D);
  const int64_t ow = take(kMP), okeys = take(2 * kMP);
  const int64_t opart = take(n * CMCD_NSTATS * 2);
  const int64_t ogb = with_grad ? take(n * 2 * (int64_t)D) : 0;
  double* partials = reinterpret_cast<double*>(ws + opart);
  float* gbuf = with_grad ? ws + ogb : nullptr;
  *partials_out = partials;
  *gbuf_out = gbuf;
  const int gemm_lds = lgcp_gemm_attrs();
  if (gemm_lds < 0) return CMCD_ERR_HIP;
  const float mu0 = 3.8812819069514780f;
  const int cbD = (D + 63) / 64;
  cmcd_layout lay{};
  lay.vd_mean = o_mean; lay.vd_logdiag = o_logdiag;
  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    LgcpStateArgs st{};
    st.seeds = seeds + base; st.params = params; st.tc = tc; st.x = ws + ox;
    st.w = ws + ow; st.keys = reinterpret_cast<uint32_t*>(ws + okeys);
    st.lay = lay; st.M = M; st.D = D;
    hipLaunchKernelGGL(lgcp_init_kernel, dim3(M), dim3(256), 0, stream, st);
    GemmArgs g{};
    g.M = M; g.Kdim = D;
    g.seg[0] = GemmSeg{ws + ox, tc, ws + okr, D, D, D, D, mu0};
    g.nblk0 = cbD;
    hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), dim3(64 * kGemmWaves), gemm_lds, stream, g);
    LgcpMfviArgs fa{params, tc, ws + ox, ws + okr, ws + ow, out_loss + base, out_z + base * D,
                    partials + base * CMCD_NSTATS, gbuf ? gbuf + base * 2 * D : nullptr, o_mean, D};
    hipLaunchKernelGGL(lgcp_mfvi_finish_kernel, dim3(M), dim3(256), 0, stream, fa);
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}


// ------------------------------------------------------------------------------------------
// Reparameterised gradient on the lgcp path (jax.grad(compute_bound, 1) for MCD_CAIS_sn, d = 1600): the same
// reverse recursion as cmcd_grad.hip (header there), as a launch sequence.  Per evaluation e = K..0:
//   recompute the forward at the stored z_e (6 launches of the forward path) ->
//   adjoint step (g_{e-1}, cotangents a_s / a_gp / a_gq, beta / eps / q gradients) ->
//   net backward through three skinny GEMMs against TRANSPOSED weight copies (made once per call), the
//   Hessian product H_p v = -K^-1 v - a e^z v through one more GEMM -> lambda_e.
// The O(width^2) parameter gradients are deferred: every evaluation's (u1, u2, d a1, d a2, d o) rows are kept
// and contracted at the end by three A^T B products on the matrix cores (inner dimension (K+1) n).
// ------------------------------------------------------------------------------------------
__global__ void lgcp_transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C, int lds_, int ldd) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < R && c < C) ? src[(int64_t)r * lds_ + c] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += blockDim.y) {
    const int c = c0 + i, r = r0 + threadIdx.x;
    if (r < R && c < C) dst[(int64_t)c * ldd + r] = tile[threadIdx.x][i];
  }
}

// gbeta / geps / d factor_sn from the adjoint step's per-workgroup partials: one wave per evaluation e, fixed order
struct LgcpAdjRedArgs {
  const float* part;     // [K+1][slots][8]
  const float* sched;
  float* gbeta_lo;       // [K+1] contribution of evaluation e to entry e-1
  float* geps_lo;
  float* gbeta_hi;       // [K+1] contribution of evaluation e to entry e
  float* geps_hi;
  float* gfac_e;         // [K+1]
  int K, slots;
};

__global__ __launch_bounds__(64) void lgcp_adj_reduce_kernel(LgcpAdjRedArgs a) {
  const int e = blockIdx.x, lane = threadIdx.x;
  float t[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int sl = lane; sl < a.slots; sl += 64) {
    const float* o = a.part + ((int64_t)e * a.slots + sl) * 8;
#pragma unroll
    for (int q = 0; q < 6; ++q) t[q] += o[q];
  }
#pragma unroll
  for (int q = 0; q < 6; ++q) t[q] = wave_sum64(t[q]);
  if (lane == 0) {
    const float pe = e > 0 ? a.sched[8 * (e - 1) + 1] : 1.f, ee = e < a.K ? a.sched[8 * e + 1] : 1.f;
    const float inv2e = 0.5f / pe;
    a.gbeta_lo[e] = e > 0 ? pe * t[0] : 0.f;
    a.geps_lo[e] = e > 0 ? t[1] - t[2] * inv2e * inv2e : 0.f;            // t[2] = sum omega r^2
    a.gbeta_hi[e] = e < a.K ? ee * t[3] : 0.f;
    a.geps_hi[e] = e < a.K ? t[4] : 0.f;
    a.gfac_e[e] = t[5];
  }
}

// gbeta[k] = hi[k] + lo[k + 1]  (two addends: order-free), same for geps
__global__ void lgcp_adj_combine_kernel(const float* lo_b, const float* hi_b, const float* lo_e, const float* hi_e, float* gbeta,
                                        float* geps, int K) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= K) return;
  gbeta[k] = hi_b[k] + lo_b[k + 1];
  geps[k] = hi_e[k] + lo_e[k + 1];
}

// C[Ma][Nb] (ldc) += sum_r A[r][Ma]^T B[r][Nb]: contraction over the (K+1) n stored rows on the matrix cores.
// A wave owns a 64 x 64 tile of C as 2 x 2 fp32 32x32x2 MFMA tiles (4 waves: 128 x 128 per workgroup); both operands
// come straight from global memory in MFMA order (lane (h, c): A[r + h][i0 + c], B[r + h][j0 + c]: two 128-byte row
// pieces per load), register double-buffered kTnU row pairs ahead.  blockIdx.z splits the rows; every split adds its
// tile into the zero-initialised gradient (two addends per address at kTnSplit = 2: order-free).
constexpr int kTnU = 8;
constexpr int kTnSplit = 2;

__global__ __launch_bounds__(256) void lgcp_tn_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B,
                                                           float* __restrict__ C, int64_t R, int Ma, int Nb, int lda,
                                                           int ldb, int ldc) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, l5 = lane >> 5;
  const int i0 = blockIdx.y * 128 + (wv >> 1) * 64, j0 = blockIdx.x * 128 + (wv & 1) * 64;
  if (i0 >= Ma || j0 >= Nb) return;
  const int64_t rchunk = (((R + kTnSplit - 1) / kTnSplit) + 1) & ~int64_t(1);
  const int64_t r_lo = blockIdx.z * rchunk, r_hi = min(R, r_lo + rchunk);
  f32x16 acc[2][2];
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[ti][tj][q] = 0.f;
  const bool ia[2] = {i0 + l31 < Ma, i0 + 32 + l31 < Ma}, jb[2] = {j0 + l31 < Nb, j0 + 32 + l31 < Nb};
  const float* Ap = A + i0 + l31;
  const float* Bp = B + j0 + l31;
  float ca[kTnU][2], cb[kTnU][2], na[kTnU][2], nb[kTnU][2];
  auto load = [&](int64_t r0, float (&va)[kTnU][2], float (&vb)[kTnU][2]) {
#pragma unroll
    for (int u = 0; u < kTnU; ++u) {
      const int64_t r = r0 + 2 * u + l5;
      const bool ok = r < r_hi;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        va[u][t] = (ok && ia[t]) ? Ap[r * lda + 32 * t] : 0.f;
        vb[u][t] = (ok && jb[t]) ? Bp[r * ldb + 32 * t] : 0.f;
      }
    }
  };
  load(r_lo, ca, cb);
  for (int64_t r0 = r_lo; r0 < r_hi; r0 += 2 * kTnU) {
    load(r0 + 2 * kTnU, na, nb);                       // rows past r_hi load as zeros
#pragma unroll
    for (int u = 0; u < kTnU; ++u)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
          acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(ca[u][ti], cb[u][tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < kTnU; ++u)
#pragma unroll
      for (int t = 0; t < 2; ++t) { ca[u][t] = na[u][t]; cb[u][t] = nb[u][t]; }
  }
  // D layout: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = i0 + 32 * ti + (q & 3) + 8 * (q >> 2) + 4 * l5, col = j0 + 32 * tj + l31;
        if (row < Ma && col < Nb) atomicAdd(C + (int64_t)row * ldc + col, acc[ti][tj][q]);
      }
}

// column sums of a [R][C] matrix (d b3), and the final reduction of the per-particle q gradients
__global__ void lgcp_colsum_kernel(const float* __restrict__ A, int64_t R, int C, int lda, float* out, float scale, int accumulate) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= C) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  int64_t r = 0;
  for (; r + 4 <= R; r += 4) {
    v0 += A[r * lda + j]; v1 += A[(r + 1) * lda + j]; v2 += A[(r + 2) * lda + j]; v3 += A[(r + 3) * lda + j];
  }
  for (; r < R; ++r) v0 += A[r * lda + j];
  const float v = ((v0 + v1) + (v2 + v3)) * scale;
  out[j] = accumulate ? out[j] + v : v;
}

// buffers of one forward recompute (two sets: evaluation e-1 is recomputed on a side stream while the backward
// pass of evaluation e reads the other set)
struct LgcpFwdSet {
  int64_t kr, slab1, pre1, u1, slab2, pre2, u2, sn;
};

struct LgcpGradWs {
  int64_t wt1, wt2, wt3;                       // transposed weights
  int64_t lamn, lam_part, gE, gprev, dO, v;    // [kMP][D]
  int64_t hv, dxf;                             // [kSplit][kMP][D]
  int64_t du2s, ts;                            // [kSplit][kMP][IN] slabs of the two IN-wide backward GEMMs
  int64_t du2, du1, da2, da1;                  // [kMP][IN]
  int64_t gmu_acc, glam_acc;                   // [kMP][D]
  int64_t U1, U2, DA1, DA2;                    // [(K+1) n][IN]
  int64_t kpre1, kpre2, kkr, ksn;              // [(K+1) n][IN] x 2, [(K+1) n][D] x 2: kept by the forward (lgcp_keep); 0 floats when off
  bool keep;
  int64_t bops, wt3p, wt2p, wt1p;              // keep: packed dO | v | da2 | da1 operands and packed W3^T, W2^T, W1[:D]^T
  int64_t DO;                                  // [(K+1) n][D]
  int64_t S, S2, gbeta, geps, gfac, gb2;       // tables (gfac: [K+1] per-evaluation terms)
  int64_t adjpart, gb_lo, ge_lo, gb_hi, ge_hi; // adjoint step partial sums and their per-evaluation reductions
  int64_t counters;                            // arrival counters of the side stream's fused GEMMs (ints)
  int64_t zero_lo, zero_hi;                    // range to clear per call
  LgcpFwdSet fs[2];
  int64_t total;
};

static LgcpGradWs lgcp_grad_ws(const cmcd_desc& d, int64_t n) {
  const int64_t D = d.dim, IN = D + d.emb_dim, K = d.nbridges, R = (K + 1) * n;
  LgcpGradWs w;
  int64_t o = 0;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  w.wt1 = take(IN * IN); w.wt2 = take(IN * IN); w.wt3 = take(D * IN);
  w.dO = take(kMP * D); w.v = take(kMP * D); w.lam_part = take(kMP * D); w.gprev = take(kMP * D);
  w.hv = take(kSplit * kMP * D); w.dxf = take(kSplit * kMP * D);
  w.du2s = take(kSplit * kMP * IN); w.ts = take(kSplit * kMP * IN);
  w.du2 = take(kMP * IN); w.du1 = take(kMP * IN); w.da2 = take(kMP * IN); w.da1 = take(kMP * IN);
  w.U1 = take(R * IN); w.U2 = take(R * IN); w.DA1 = take(R * IN); w.DA2 = take(R * IN); w.DO = take(R * D);
  for (int b = 0; b < 2; ++b) {
    LgcpFwdSet& f = w.fs[b];
    f.kr = take(kSplit * kMP * D); f.slab1 = take(kSplit * kMP * IN); f.pre1 = take(kMP * IN);
    f.u1 = take(kMP * IN); f.slab2 = take(kSplit * kMP * IN); f.pre2 = take(kMP * IN); f.u2 = take(kMP * IN);
    f.sn = take(kSplit * kMP * D);
  }
  w.zero_lo = o;
  w.lamn = take(kMP * D); w.gE = take(kMP * D);
  w.gmu_acc = take(kMP * D); w.glam_acc = take(kMP * D);
  w.S = take((K + 1) * IN); w.S2 = take((K + 1) * IN);
  w.gbeta = take(K); w.geps = take(K); w.gfac = take(K + 1); w.gb2 = take(IN);
  w.counters = take(2 * ((D + 63) / 64 + (IN + 63) / 64));     // side stream's set | caller's stream's set
  w.zero_hi = o;
  w.adjpart = take((K + 1) * n * ((D + 63) / 64) * 8);
  w.gb_lo = take(K + 1); w.ge_lo = take(K + 1); w.gb_hi = take(K + 1); w.ge_hi = take(K + 1);
  // the forward's kept activations (behind everything else: the layout above does not move when they are off)
  w.keep = lgcp_nsk_ok(d) && d.mode != CMCD_MODE_CAIS_UHA_SN && R * (2 * IN + 2 * D) <= (int64_t(1) << 28);
  w.kpre1 = take(w.keep ? R * IN : 0); w.kpre2 = take(w.keep ? R * IN : 0);
  w.kkr = take(w.keep ? R * D : 0); w.ksn = take(w.keep ? R * D : 0);
  {
    const int64_t tIN = (IN + 15) / 16, tD = D / 16;
    w.bops = take(w.keep ? 4 * kNskOperand : 0);
    w.wt3p = take(w.keep ? tIN * kNskChunks * 256 : 0); w.wt2p = take(w.keep ? tIN * kNskChunks * 256 : 0);
    w.wt1p = take(w.keep ? tD * kNskChunks * 256 : 0);
  }
  w.total = o;
  return w;
}

static LgcpKeep lgcp_keep(const cmcd_desc& d, int64_t n, float* gws) {
  const LgcpGradWs g = lgcp_grad_ws(d, n);
  if (!g.keep) return LgcpKeep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false};
  return LgcpKeep{gws + g.kpre1, gws + g.U1, gws + g.kpre2, gws + g.U2, gws + g.kkr, gws + g.ksn, true};
}

static int64_t lgcp_uha_grad_ws_total(const cmcd_desc& d, int64_t n);
static int lgcp_uha_grad(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, int64_t n, const float* params,
                         int64_t n_params, const float* tc, float* ws, const float* traj, float* gws, float omega,
                         float* grad, hipStream_t stream);

int64_t lgcp_grad_workspace_floats(const cmcd_desc& d, int64_t n) {
  if (d.mode == CMCD_MODE_CAIS_UHA_SN) return lgcp_uha_grad_ws_total(d, n);
  return lgcp_grad_ws(d, n).total;
}

int lgcp_grad(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, int64_t n, const float* params,
              int64_t n_params, const float* tc, float* ws, const float* traj, float* gws, float omega,
              const float* omega_vec, bool bptt, float* grad, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (d.mode == CMCD_MODE_CAIS_UHA_SN) {
    if (!bptt || omega_vec) return CMCD_ERR_UNSUPPORTED;   // the mode has no stop_gradient variant
    return lgcp_uha_grad(d, lay, sw, n, params, n_params, tc, ws, traj, gws, omega, grad, stream);
  }
  const int D = d.dim, E = d.emb_dim, IN = D + E, K = d.nbridges;
  const LgcpWs w = lgcp_ws(d, n, sw.total_floats);
  const LgcpGradWs g = lgcp_grad_ws(d, n);
  if (hipMemsetAsync(grad, 0, sizeof(float) * n_params, stream) != hipSuccess) return CMCD_ERR_HIP;
  if (hipMemsetAsync(gws + g.zero_lo, 0, sizeof(float) * (g.zero_hi - g.zero_lo), stream) != hipSuccess) return CMCD_ERR_HIP;
  const int ula = d.mode == CMCD_MODE_ULA ? 1 : (d.mode == CMCD_MODE_ULA_SN ? 2 : 0);
  const bool net = ula != 1;
  // transposed weight copies: the backward GEMMs are then the forward kernel on W^T
  if (net) {
    const dim3 tb(32, 8);
    hipLaunchKernelGGL(lgcp_transpose_kernel, dim3((IN + 31) / 32, (IN + 31) / 32), tb, 0, stream, params + lay.g_w1,
                       gws + g.wt1, IN, IN, IN, IN);
    hipLaunchKernelGGL(lgcp_transpose_kernel, dim3((IN + 31) / 32, (IN + 31) / 32), tb, 0, stream, params + lay.g_w2,
                       gws + g.wt2, IN, IN, IN, IN);
    hipLaunchKernelGGL(lgcp_transpose_kernel, dim3((D + 31) / 32, (IN + 31) / 32), tb, 0, stream, params + lay.g_w3,
                       gws + g.wt3, IN, D, D, IN);
  }
  const int gemm_lds = lgcp_gemm_attrs();
  if (gemm_lds < 0) return CMCD_ERR_HIP;
  const float* kinv = tc;
  const float mu0 = 3.8812819069514780f;
  const dim3 gblock(64 * kGemmWaves);
  const int cbD = (D + 63) / 64, cbIN = (IN + 63) / 64;
  // r04: with the activations kept, passes of <= 20 particles run the three backward products on the no-split-K GEMM too
  // (lgcp_nsk_kernel<3, ..>: the backward activation as the consumer, column sums by the tile's one workgroup); the packed
  // W3^T / W2^T / W1[:D]^T copies are made once per call, K^-1 is the forward's packed copy
  const int tIN = (IN + 15) / 16, tD = D / 16;
  const bool nskb_ok = g.keep && net;
  if (nskb_ok) {
    NskPackArgs pk{};
    pk.src[0] = gws + g.wt3; pk.dst[0] = gws + g.wt3p; pk.K[0] = D; pk.N[0] = IN; pk.ntile[0] = tIN;
    pk.src[1] = gws + g.wt2; pk.dst[1] = gws + g.wt2p; pk.K[1] = IN; pk.N[1] = IN; pk.ntile[1] = tIN;
    pk.src[2] = gws + g.wt1; pk.dst[2] = gws + g.wt1p; pk.K[2] = IN; pk.N[2] = D; pk.ntile[2] = tD; pk.ld[2] = IN;   // first D columns of W1^T
    const int64_t groups = (int64_t)tIN * kNskChunks * 64;
    hipLaunchKernelGGL(lgcp_nsk_pack_kernel, dim3((unsigned)((groups + 255) / 256), 3), dim3(256), 0, stream, pk);
  }
  // Two streams: the forward recompute of evaluation e-1 (side stream, its own buffer set) overlaps the adjoint /
  // backward launches of evaluation e (caller's stream).  Every kernel here is launch-latency-bound, so the two
  // chains run side by side; events order the hand-overs (fork / join, capturable in a graph).
  struct GradSide { hipStream_t side; hipEvent_t ev_fwd[2], ev_bwd[2], ev_fork; };
  constexpr int kMaxDev = 16;
  static thread_local GradSide gsets[kMaxDev] = {};   // keyed by device, like the forward's side streams
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return CMCD_ERR_HIP;
  hipStream_t& side = gsets[dev].side;
  hipEvent_t* ev_fwd = gsets[dev].ev_fwd;
  hipEvent_t* ev_bwd = gsets[dev].ev_bwd;
  hipEvent_t& ev_fork = gsets[dev].ev_fork;
  if (!side) {
    if (hipStreamCreateWithFlags(&side, hipStreamNonBlocking) != hipSuccess) return CMCD_ERR_HIP;
    for (int b = 0; b < 2; ++b) {
      if (hipEventCreateWithFlags(&ev_fwd[b], hipEventDisableTiming) != hipSuccess) return CMCD_ERR_HIP;
      if (hipEventCreateWithFlags(&ev_bwd[b], hipEventDisableTiming) != hipSuccess) return CMCD_ERR_HIP;
    }
    if (hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming) != hipSuccess) return CMCD_ERR_HIP;
  }
  // r04: the forward kept pre1 / u1 / pre2 / u2 / kr / sn of every evaluation (lgcp_keep): nothing to recompute, one stream
  const LgcpKeep keep = lgcp_keep(d, n, gws);
  const bool kept = keep.on;
  // bias1 rows are still in the forward workspace (lgcp_forward's prep)
  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    if (base > 0) {  // lambda / g start from zero for every pass
      if (hipMemsetAsync(gws + g.lamn, 0, sizeof(float) * 2 * ((kMP * (int64_t)D + 3) & ~3), stream) != hipSuccess) return CMCD_ERR_HIP;
      if (hipMemsetAsync(gws + g.gmu_acc, 0, sizeof(float) * 2 * ((kMP * (int64_t)D + 3) & ~3), stream) != hipSuccess) return CMCD_ERR_HIP;
    }
    if (!kept && (hipEventRecord(ev_fork, stream) != hipSuccess || hipStreamWaitEvent(side, ev_fork, 0) != hipSuccess)) return CMCD_ERR_HIP;
    const bool nskb = nskb_ok && kept && M <= 20;      // (21 .. 32 particles: two workgroups per column tile — the split-K form)
    float* const dOp = gws + g.bops, *vp = dOp + kNskOperand, *da2p = vp + kNskOperand, *da1p = da2p + kNskOperand;
    // the padding of the packed operands (rows >= M, inputs past the contraction) must read as zeros
    if (nskb && hipMemsetAsync(dOp, 0, sizeof(float) * 4 * kNskOperand, stream) != hipSuccess) return CMCD_ERR_HIP;

    // forward recompute at z_e into buffer set e & 1, on stream st: the forward path's three launches (activations
    // fused into the GEMMs; the third has no consumer here: the adjoint step sums its slabs)
    int* side_counters = reinterpret_cast<int*>(gws + g.counters);
    int* main_counters = side_counters + cbD + cbIN;
    auto forward_at = [&](int e, hipStream_t st) {
      const LgcpFwdSet& f = g.fs[e & 1];
      const int er = ula == 2 ? (e > 0 ? e - 1 : 0) : e;       // MCD_ULA_sn: s(z_e, e - 1)
      const int ie = er < K ? er : K - 1;
      const float* xe = traj + ((int64_t)e * n + base) * D;
      GemmArgs gm{};
      gm.M = M; gm.counters = side_counters;
      gm.act.x = xe; gm.act.D = D; gm.act.IN = IN;
      gm.Kdim = D;
      gm.seg[0] = GemmSeg{xe, kinv, gws + f.kr, D, D, D, D, mu0};
      if (!net) {   // MCD_ULA: the target's K^-1 product is all there is to recompute
        gm.nblk0 = cbD;
        hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), gblock, gemm_lds, st, gm);
        return;
      }
      gm.seg[1] = GemmSeg{xe, params + lay.g_w1, gws + f.slab1, IN, D, IN, IN};
      gm.nblk0 = cbD; gm.epi_seg = 1;
      gm.act.mode = 1; gm.act.bias = ws + w.bias1 + (int64_t)er * IN; gm.act.emb = params + lay.g_emb + (int64_t)ie * E;
      gm.act.sum_out = gws + f.pre1; gm.act.u_prev = nullptr; gm.act.u_out = gws + f.u1;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACT>, dim3(cbD + cbIN, kSplit), gblock, gemm_lds, st, gm);
      gm.Kdim = IN;
      gm.seg[0] = GemmSeg{gws + f.u1, params + lay.g_w2, gws + f.slab2, IN, IN, IN, IN};
      gm.nblk0 = cbIN; gm.epi_seg = -1;
      gm.act.mode = 2; gm.act.bias = params + lay.g_b2; gm.act.sum_out = gws + f.pre2; gm.act.u_prev = gws + f.u1;
      gm.act.u_out = gws + f.u2;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACT>, dim3(cbIN, kSplit), gblock, gemm_lds, st, gm);
      gm.seg[0] = GemmSeg{gws + f.u2, params + lay.g_w3, gws + f.sn, D, IN, D, D};
      gm.nblk0 = cbD;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), gblock, gemm_lds, st, gm);
    };

    if (!kept) {
      forward_at(K, side);
      if (hipEventRecord(ev_fwd[K & 1], side) != hipSuccess) return CMCD_ERR_HIP;
    }
    for (int e = K; e >= 0; --e) {
      if (!kept) {
        if (e > 0) {  // next evaluation's recompute: its buffer set was last read by the backward pass of e+1
          if (e + 1 <= K && hipStreamWaitEvent(side, ev_bwd[(e - 1) & 1], 0) != hipSuccess) return CMCD_ERR_HIP;
          forward_at(e - 1, side);
          if (hipEventRecord(ev_fwd[(e - 1) & 1], side) != hipSuccess) return CMCD_ERR_HIP;
        }
        if (hipStreamWaitEvent(stream, ev_fwd[e & 1], 0) != hipSuccess) return CMCD_ERR_HIP;
      }
      const LgcpFwdSet& f = g.fs[e & 1];
      const int64_t row0 = (int64_t)e * n + base;
      // this evaluation's activations: the recompute's buffer set, or its rows of the tables the forward kept
      const float* e_kr = kept ? keep.kr + row0 * D : gws + f.kr;
      const float* e_sn = kept ? keep.sn + row0 * D : gws + f.sn;
      const float* e_pre1 = kept ? keep.pre1 + row0 * IN : gws + f.pre1;
      const float* e_pre2 = kept ? keep.pre2 + row0 * IN : gws + f.pre2;
      const float* e_u1 = kept ? keep.u1 + row0 * IN : gws + f.u1;
      const float* e_u2 = kept ? keep.u2 + row0 * IN : gws + f.u2;
      GemmArgs gm{};
      gm.M = M;
      // ---- adjoint step (of evaluation ev: its kr / sn are the recompute's buffers, or rows of the kept tables)
      auto adj_args = [&](int ev) {
        const int64_t rv = (int64_t)ev * n + base;
        LgcpAdjArgs aa{};
        aa.params = params; aa.tc = tc; aa.sched = ws + sw.sched; aa.traj = traj;
        aa.kr = kept ? keep.kr + rv * D : e_kr; aa.sn = kept ? keep.sn + rv * D : e_sn;
        aa.nslab = kept ? 1 : kSplit;
        aa.b3 = params + lay.g_b3; aa.factor = params + lay.g_factor; aa.lamn = gws + g.lamn; aa.gE = gws + g.gE;
        aa.gprev = gws + g.gprev; aa.dO = gws + g.dO; aa.v = gws + g.v; aa.lam_part = gws + g.lam_part;
        aa.gmu_acc = gws + g.gmu_acc; aa.glam_acc = gws + g.glam_acc; aa.part = gws + g.adjpart;
        aa.DObig = gws + g.DO; aa.lay = lay; aa.n = n; aa.base = base;
        aa.M = M; aa.D = D; aa.K = K; aa.e = ev; aa.grad_clipping = d.grad_clipping; aa.omega = omega;
        aa.ula = ula; aa.omega_vec = omega_vec; aa.gktab = reinterpret_cast<const uint32_t*>(ws + w.gktab); aa.bptt = bptt ? 1 : 0; aa.var_mode = d.mode == CMCD_MODE_CAIS_VAR_SN ? 1 : 0;
        if (nskb) { aa.dOp = dOp; aa.vp = vp; }
        return aa;
      };
      // kept + reparameterised: the step of evaluation e was fused behind lambda_{e+1} by the previous iteration
      const bool fuse = kept && bptt;
      if (!(fuse && e < K)) hipLaunchKernelGGL(lgcp_adj_step_kernel, dim3(cbD, M), dim3(64), 0, stream, adj_args(e));
      if (!net) {   // MCD_ULA: lambda_e = lam_part - H_p v, one GEMM
        gm.Kdim = D; gm.Kdim1 = 0;
        gm.seg[0] = GemmSeg{gws + g.v, kinv, gws + g.hv, D, D, D, D};
        gm.nblk0 = cbD;
        hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), gblock, gemm_lds, stream, gm);
        LgcpLamArgs la{};
        la.params = params; la.tc = tc; la.traj = traj; la.dxf = nullptr; la.hv = gws + g.hv; la.du1 = nullptr;
        la.v = gws + g.v; la.lam_part = gws + g.lam_part; la.gprev = gws + g.gprev; la.lamn = gws + g.lamn;
        la.gE = gws + g.gE; la.gmu_acc = gws + g.gmu_acc; la.glam_acc = gws + g.glam_acc; la.lay = lay; la.n = n;
        la.base = base; la.M = M; la.D = D; la.IN = IN; la.e = e; la.omega = omega; la.no_net = 1;
        if (fuse && e > 0) hipLaunchKernelGGL(lgcp_lam_adj_kernel, dim3(cbD, M), dim3(64), 0, stream, la, adj_args(e - 1));
        else hipLaunchKernelGGL(lgcp_lam_finish_kernel, dim3((M * D + 255) / 256), dim3(256), 0, stream, la);
        if (!kept && hipEventRecord(ev_bwd[e & 1], stream) != hipSuccess) return CMCD_ERR_HIP;
        continue;
      }
      const int er_b = ula == 2 ? (e > 0 ? e - 1 : 0) : e;       // the time index the network saw at this evaluation
      if (nskb) {
        const bool merged = M > 16;
        auto launch3 = [&](NskArgs& na, int tiles) {
          if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<3, true>), dim3((unsigned)tiles, 1), gblock, 0, stream, na);
          else hipLaunchKernelGGL((lgcp_nsk_kernel<3, false>), dim3((unsigned)tiles, 1), gblock, 0, stream, na);
        };
        NskArgs na{};
        na.M = M; na.D = D; na.IN = IN;
        // d u2 = d o W3^T, d a2 = d u2 sigmoid(pre2)
        na.seg[0] = NskSeg{dOp, gws + g.wt3p, IN, 0, 0.f}; na.nt0 = tIN;
        na.bPre = e_pre2; na.bDuPrev = nullptr; na.bDu = gws + g.du2; na.outA = da2p; na.bDaBig = gws + g.DA2 + row0 * IN;
        na.bSumA = gws + g.gb2; na.bSumU = nullptr;
        launch3(na, tIN);
        // d u1 = d u2 + d a2 W2^T, d a1 = d u1 sigmoid(pre1)
        na.seg[0] = NskSeg{da2p, gws + g.wt2p, IN, 0, 0.f};
        na.bPre = e_pre1; na.bDuPrev = gws + g.du2; na.bDu = gws + g.du1; na.outA = da1p; na.bDaBig = gws + g.DA1 + row0 * IN;
        na.bSumA = gws + g.S + (int64_t)er_b * IN; na.bSumU = gws + g.S2 + (int64_t)er_b * IN;
        launch3(na, tIN);
        if (!bptt) continue;
        // d a1 W1[:D]^T | v K^-1: plain products, row-major
        NskArgs nb{};
        nb.M = M; nb.D = D; nb.IN = IN;
        nb.seg[0] = NskSeg{da1p, gws + g.wt1p, D, NSK_OUT, 0.f}; nb.nt0 = tD; nb.outN = gws + g.dxf;
        nb.seg[1] = NskSeg{vp, ws + w.kip, D, NSK_KR, 0.f}; nb.krOutN = gws + g.hv;
        if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<0, true>), dim3((unsigned)(2 * tD), 1), gblock, 0, stream, nb);
        else hipLaunchKernelGGL((lgcp_nsk_kernel<0, false>), dim3((unsigned)(2 * tD), 1), gblock, 0, stream, nb);
        LgcpLamArgs la{};
        la.params = params; la.tc = tc; la.traj = traj; la.dxf = gws + g.dxf; la.hv = gws + g.hv; la.du1 = gws + g.du1;
        la.v = gws + g.v; la.lam_part = gws + g.lam_part; la.gprev = gws + g.gprev; la.lamn = gws + g.lamn;
        la.gE = gws + g.gE; la.gmu_acc = gws + g.gmu_acc; la.glam_acc = gws + g.glam_acc; la.lay = lay; la.n = n;
        la.base = base; la.M = M; la.D = D; la.IN = IN; la.e = e; la.omega = omega; la.nslab = 1;
        if (fuse && e > 0) hipLaunchKernelGGL(lgcp_lam_adj_kernel, dim3(cbD, M), dim3(64), 0, stream, la, adj_args(e - 1));
        else hipLaunchKernelGGL(lgcp_lam_finish_kernel, dim3((M * D + 255) / 256), dim3(256), 0, stream, la);
        continue;
      }
      // ---- net backward: d u2 = d o W3^T, then d a2 = d u2 sigmoid(pre2) as the GEMM's consumer
      gm.counters = main_counters; gm.epi_seg = -1;
      gm.Kdim = D;
      gm.seg[0] = GemmSeg{gws + g.dO, gws + g.wt3, gws + g.du2s, IN, D, IN, IN};
      gm.nblk0 = cbIN;
      LgcpActbArgs& ab = gm.actb;
      ab.pre = e_pre2; ab.du_prev = nullptr; ab.u_src = e_u2;
      ab.du_out = gws + g.du2; ab.da_out = gws + g.da2; ab.da_big = gws + g.DA2; ab.u_big = kept ? nullptr : gws + g.U2;
      ab.gb = gws + g.gb2; ab.row0 = row0; ab.IN = IN; ab.mode = 2;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACTB>, dim3(cbIN, kSplit), gblock, gemm_lds, stream, gm);
      // d u1 = d u2 + d a2 W2^T, d a1 = d u1 sigmoid(pre1)
      gm.Kdim = IN;
      gm.seg[0] = GemmSeg{gws + g.da2, gws + g.wt2, gws + g.ts, IN, IN, IN, IN};
      gm.nblk0 = cbIN;
      ab.pre = e_pre1; ab.du_prev = gws + g.du2; ab.u_src = e_u1;
      ab.du_out = gws + g.du1; ab.da_out = gws + g.da1; ab.da_big = gws + g.DA1; ab.u_big = kept ? nullptr : gws + g.U1;
      {
        const int er = ula == 2 ? (e > 0 ? e - 1 : 0) : e;     // the time index the network saw at this evaluation
        ab.S = gws + g.S + (int64_t)er * IN; ab.S2 = gws + g.S2 + (int64_t)er * IN; ab.mode = 1;
      }
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACTB>, dim3(cbIN, kSplit), gblock, gemm_lds, stream, gm);
      if (!bptt) {   // z detached: no lambda, no Hessian product; this evaluation's buffers are free after actb
        if (!kept && hipEventRecord(ev_bwd[e & 1], stream) != hipSuccess) return CMCD_ERR_HIP;
        continue;
      }
      gm.Kdim = IN; gm.Kdim1 = D;                       // two independent products, one launch
      gm.seg[0] = GemmSeg{gws + g.da1, gws + g.wt1, gws + g.dxf, D, IN, IN, D};
      gm.seg[1] = GemmSeg{gws + g.v, kinv, gws + g.hv, D, D, D, D};
      gm.nblk0 = cbD;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(2 * cbD, kSplit), gblock, gemm_lds, stream, gm);
      gm.Kdim1 = 0;
      LgcpLamArgs la{};
      la.params = params; la.tc = tc; la.traj = traj; la.dxf = gws + g.dxf; la.hv = gws + g.hv; la.du1 = gws + g.du1;
      la.v = gws + g.v; la.lam_part = gws + g.lam_part; la.gprev = gws + g.gprev; la.lamn = gws + g.lamn;
      la.gE = gws + g.gE; la.gmu_acc = gws + g.gmu_acc; la.glam_acc = gws + g.glam_acc; la.lay = lay; la.n = n;
      la.base = base; la.M = M; la.D = D; la.IN = IN; la.e = e; la.omega = omega;
      if (fuse && e > 0) hipLaunchKernelGGL(lgcp_lam_adj_kernel, dim3(cbD, M), dim3(64), 0, stream, la, adj_args(e - 1));
      else hipLaunchKernelGGL(lgcp_lam_finish_kernel, dim3((M * D + 255) / 256), dim3(256), 0, stream, la);
      if (!kept && hipEventRecord(ev_bwd[e & 1], stream) != hipSuccess) return CMCD_ERR_HIP;
    }
    // q gradients of this pass: sum over its particles, accumulated into grad
    hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, gws + g.gmu_acc, (int64_t)M, D, D,
                       grad + lay.vd_mean, 1.0f, 1);
    hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, gws + g.glam_acc, (int64_t)M, D, D,
                       grad + lay.vd_logdiag, 1.0f, 1);
  }
  // ---- deferred parameter contractions over all (K+1) n rows
  const int64_t R = (int64_t)(K + 1) * n;
  if (net) {
  hipLaunchKernelGGL(lgcp_tn_gemm_kernel, dim3((IN + 127) / 128, (IN + 127) / 128, kTnSplit), dim3(256), 0, stream, gws + g.U1, gws + g.DA2,
                     grad + lay.g_w2, R, IN, IN, IN, IN, IN);                                   // dW2 = U1^T dA2
  hipLaunchKernelGGL(lgcp_tn_gemm_kernel, dim3((D + 127) / 128, (IN + 127) / 128, kTnSplit), dim3(256), 0, stream, gws + g.U2, gws + g.DO,
                     grad + lay.g_w3, R, IN, D, IN, D, D);                                      // dW3 = U2^T dO
  hipLaunchKernelGGL(lgcp_tn_gemm_kernel, dim3((IN + 127) / 128, (D + 127) / 128, kTnSplit), dim3(256), 0, stream, traj, gws + g.DA1,
                     grad + lay.g_w1, R, D, IN, D, IN, IN);                                     // dW1[:D] = X^T dA1
  hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, gws + g.DO, R, D, D, grad + lay.g_b3, 1.0f, 0);
  hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((IN + 255) / 256), dim3(256), 0, stream, gws + g.gb2, (int64_t)1, IN, IN, grad + lay.g_b2, 1.0f, 0);
  }
  {
    const int slots = (int)(n * cbD);
    LgcpAdjRedArgs ra{gws + g.adjpart, ws + sw.sched, gws + g.gb_lo, gws + g.ge_lo, gws + g.gb_hi, gws + g.ge_hi, gws + g.gfac,
                      K, slots};
    hipLaunchKernelGGL(lgcp_adj_reduce_kernel, dim3(K + 1), dim3(64), 0, stream, ra);
    hipLaunchKernelGGL(lgcp_adj_combine_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, gws + g.gb_lo, gws + g.gb_hi,
                       gws + g.ge_lo, gws + g.ge_hi, gws + g.gbeta, gws + g.geps, K);
  }
  if (net)
    hipLaunchKernelGGL(lgcp_colsum_kernel, dim3(1), dim3(256), 0, stream, gws + g.gfac, (int64_t)(K + 1), 1, 1, grad + lay.g_factor, 1.0f, 0);
  int rc = launch_geffner_tails(d, lay, sw, params, gws, g.S, g.S2, g.gbeta, g.geps, IN, grad, stream_, net);
  if (rc != CMCD_OK) return rc;
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

// =============================================================================================================
// MCD_CAIS_UHA_sn on the lgcp path (d = 1600, geffner net of width 2 d + emb_dim = 3220 on concat(z, rho)):
// /root/reference/src/mcd_under_lp_a_cais.py:42-112 as a launch sequence.  One evaluation of the network touches
// W1[:2d] 41.2 MB + W2 41.5 MB + W3 20.6 MB, a bridge needs two of them (momentum rho and rho', same z, same index)
// plus K^-1 (10.2 MB) once: 8 launches per bridge on the skinny-GEMM kernel above —
//   L1  [z; rho] W1[:2d]  -> u1 = [z; rho; emb_i] + softplus(. + bias1_i)          (fused consumer)
//   L2  u1 W2             -> u2 = u1 + softplus(. + b2)                             (fused consumer)
//   L3  u2 W3             -> s1 slabs
//   F1  m_f, rho' = m_f + sqrt(2 eta) n_i, rho'' = rho' - eps uf / 2, z' = z + eps rho''      (element-wise, one workgroup per particle)
//   L4  [z; rho'] W1[:2d] -> u1  |  (z' - mu0) K^-1 -> kr slabs of z'  (second segment: z' does not depend on s2)
//   L5, L6 as L2, L3      -> s2 slabs
//   F2  m_b, log-weight increment, ub(z'), rho_new, key chain; at the last bridge log N(rho_K; 0, 1) + log p(z_K)
// Passes of <= 32 particles run one after the other on the caller's stream.
// =============================================================================================================
struct LgcpUhaWs {
  int64_t bias1;                                   // [K+1][IN]
  int64_t zr, zrp, zn, rpp;                        // [kMP][2D], [kMP][2D], [kMP][D], [kMP][D]
  int64_t u1, u2, pre1, pre2;                      // [kMP][IN]
  int64_t slab1, slab2, sn, kr;                    // [kSplit][kMP][IN] x2, [kSplit][kMP][D] x2
  int64_t w, fk, keys, gkey, counters, partials, total;
  // r04, no-split-K form: packed weights (W1[:2d] | W2 | W3 | K^-1 with 2 kNskChunks chunks per tile: K^-1 zero beyond d) and
  // packed operands zr | zrp | u1 | u2 | zn (2 kNskOperand floats each)
  int64_t w1p, w2p, w3p, kip, ops;
};

// the 2nd-order sequence on the no-split-K kernel: widths up to 2 * 1664 inputs; desc.reserved == 3 pins the split-K sequence
static bool lgcp_uha_nsk_ok(const cmcd_desc& d) {
  const int D = d.dim, IN = 2 * D + d.emb_dim;
  return d.reserved != 3 && D % 16 == 0 && D <= 16 * kNskChunks && IN <= 32 * kNskChunks;
}

static LgcpUhaWs lgcp_uha_ws(const cmcd_desc& d, int64_t n, int64_t base) {
  const int64_t D = d.dim, IN = 2 * D + d.emb_dim, K = d.nbridges;
  LgcpUhaWs w;
  int64_t o = base;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  w.bias1 = take((K + 1) * IN);
  w.zr = take(kMP * 2 * D); w.zrp = take(kMP * 2 * D); w.zn = take(kMP * D); w.rpp = take(kMP * D);
  w.u1 = take(kMP * IN); w.u2 = take(kMP * IN); w.pre1 = take(kMP * IN); w.pre2 = take(kMP * IN);
  w.slab1 = take(kSplit * kMP * IN); w.slab2 = take(kSplit * kMP * IN);
  w.sn = take(kSplit * kMP * D); w.kr = take(kSplit * kMP * D);
  w.w = take(kMP); w.fk = take(kMP); w.keys = take(2 * kMP); w.gkey = take(2 * kMP);
  w.counters = take(((D + 63) / 64) + ((IN + 63) / 64));
  o = (o + 1) & ~int64_t(1);
  w.partials = take(n * CMCD_NSTATS * 2);
  w.w1p = w.w2p = w.w3p = w.kip = w.ops = 0;
  if (lgcp_uha_nsk_ok(d)) {
    const int64_t tIN = (IN + 15) / 16, tD = D / 16, big = (int64_t)2 * kNskChunks * 256;
    w.w1p = take(tIN * big); w.w2p = take(tIN * big); w.w3p = take(tD * big); w.kip = take(tD * big);
    w.ops = take(5 * 2 * kNskOperand);
  }
  w.total = o;
  return w;
}

struct LgcpUhaStepArgs {
  const int32_t* seeds;      // [M] (this pass; init only)
  const float* params;
  const float* tc;           // {Kinv[d,d], counts[d], mu0, a, lognorm}
  const float* sched;        // [K][8] {beta, eps, ...}: cos^2 schedule (the prep launch is given CMCD_EPS_COS_SQ)
  float* zr;                 // [kMP][2D]  [z | rho]
  float* zrp;                // [kMP][2D]  [z | rho']
  float* zn;                 // [kMP][D]   z'
  float* rpp;                // [kMP][D]   rho''
  const float* kr;           // [kSplit][kMP][D]  K^-1 (z - mu0) slabs: of z in F1, of z' in F2
  const float* sn;           // [kSplit][kMP][D]  u2 W3 slabs
  float* w;                  // [kMP] running log-weight
  float* fk;                 // [kMP] forward-kernel log-density of the open bridge
  uint32_t* gen;             // [kMP][2] chain key
  uint32_t* gkey;            // [kMP][2] G_i: key of this bridge's momentum-refresh noise
  float* out_loss;           // [M]
  float* out_z;              // [M][D]
  double* partials;          // [M][5]
  float* traj;               // optional [3K+2][n_total][D]: z_0..z_K | rho_0..rho_K | rho'_0..rho'_{K-1}
  int64_t n_total, base;
  cmcd_layout lay;
  int M, D, K, i;
  // r04 (the GEMMs on the no-split-K kernel): zr / zrp / zn are packed operands of 2 kNskChunks chunks (nsk_pack); kr / sn are
  // ONE row-major [kMP][D] array each instead of kSplit slabs
  int packed, nslab;
  const float* fkslot;       // r04: F1 fused into its GEMM launch leaves the forward density per 16-column tile, [D / 16][kMP]
};

__device__ __forceinline__ int64_t uha_zr_ix(const LgcpUhaStepArgs& a, int p, int col) {     // element (p, col) of [z | rho]
  return a.packed ? nsk_pack(p, col, 2 * kNskChunks) : (int64_t)p * 2 * a.D + col;
}
__device__ __forceinline__ int64_t uha_zn_ix(const LgcpUhaStepArgs& a, int p, int e) {
  // (the K^-1 product's operand shares the two-round layout of its launch: its weights are zero-padded to 2 kNskChunks chunks)
  return a.packed ? nsk_pack(p, e, 2 * kNskChunks) : (int64_t)p * a.D + e;
}

// z0 = mean + std normal(A); rho0 = normal(R); w = -log q(z0) - log N(rho0; 0, 1); gen_0, G_0
// (mcdboundingmachine.py:151-162, mcd_under_lp_a_cais.py:92-100)
__global__ __launch_bounds__(256) void lgcp_uha_init_kernel(LgcpUhaStepArgs a) {
  __shared__ float sh[4];
  __shared__ uint32_t rk[2];
  const int p = blockIdx.x, D = a.D, H = (D + 1) / 2, K = a.K;
  const uint32_t seed = (uint32_t)a.seeds[p];
  uint32_t s0 = 0, s1 = 2, t0 = 1, t1 = 3;
  threefry2x32(0u, seed, s0, s1);
  threefry2x32(0u, seed, t0, t1);
  const uint32_t a0 = s0, a1 = t0, b0 = s1, b1 = t1;   // A = (out0, out1), B = (out2, out3)
  if (threadIdx.x == 0) {
    uint32_t c0 = 0, c2 = 2, c1 = 1, c3 = 3;
    threefry2x32(b0, b1, c0, c2);
    threefry2x32(b0, b1, c1, c3);       // C = first(split(B)) = (c0, c1): the key handed to evolve
    uint32_t r0 = 0, g0 = 2, r1 = 1, g1 = 3;
    threefry2x32(c0, c1, r0, g0);
    threefry2x32(c0, c1, r1, g1);       // R = (r0, r1), G' = (g0, g1)                   :92
    rk[0] = r0; rk[1] = r1;
    uint32_t n0 = 0, n2 = 2, n1 = 1, n3 = 3;
    threefry2x32(g0, g1, n0, n2);
    threefry2x32(g0, g1, n1, n3);       // gen_0 = second(split(G')) = (n2, n3)          :100
    uint32_t k0 = n2, k1 = n3, G0, G1;
    lgcp_key_advance(k0, k1, G0, G1);   // G_0 and gen_1                                 :55,84
    a.gkey[2 * p] = G0; a.gkey[2 * p + 1] = G1;
    a.gen[2 * p] = k0; a.gen[2 * p + 1] = k1;
  }
  __syncthreads();
  const uint32_t r0 = rk[0], r1 = rk[1];
  float acc = 0.f;
  for (int j = threadIdx.x; j < H; j += blockDim.x) {
    uint32_t y0 = j, y1 = (H + j < D) ? H + j : 0;
    uint32_t q0 = y0, q1 = y1;
    threefry2x32(a0, a1, y0, y1);
    threefry2x32(r0, r1, q0, q1);
    const int idx[2] = {j, H + j};
    const uint32_t bz[2] = {y0, y1}, br[2] = {q0, q1};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (idx[q] < D) {
        const float mean = a.params[a.lay.vd_mean + idx[q]];
        const float sd = expf(a.params[a.lay.vd_logdiag + idx[q]]);
        const float z = sd * bits_to_normal(bz[q]) + mean;
        const float rho = bits_to_normal(br[q]);
        a.zr[uha_zr_ix(a, p, idx[q])] = z;
        a.zr[uha_zr_ix(a, p, D + idx[q])] = rho;
        if (a.packed) a.zn[uha_zn_ix(a, p, idx[q])] = z;       // operand of the K^-1 (z_0 - mu0) launch
        if (a.traj) {
          a.traj[(a.base + p) * D + idx[q]] = z;
          a.traj[((int64_t)(K + 1) * a.n_total + a.base + p) * D + idx[q]] = rho;
        }
        const float dz = z - mean;
        acc += -(dz * dz) / (2.0f * sd * sd) - logf(sd) - kHalfLog2Pi;   // log q(z0)
        acc += -(rho * rho) * 0.5f - kHalfLog2Pi;                         // log N(rho0; 0, 1)
      }
    }
  }
  const float lq = block_sum_256(acc, sh);
  if (threadIdx.x == 0) a.w[p] = -lq;
}

// F1 (see the header above).  NW waves per particle: 4 in the split-K sequence; 16 in the no-split-K one (r04: at 256 threads a
// thread walks 6 - 7 elements one dependent load chain after the other, 9.6 us per launch; 1024 threads: one or two)
template <int NW>
__global__ __launch_bounds__(64 * NW) void lgcp_uha_mid_kernel(LgcpUhaStepArgs a) {
  __shared__ float sh[NW];
  const int p = blockIdx.x, D = a.D, H = (D + 1) / 2, K = a.K, i = a.i;
  const float* counts = a.tc + (int64_t)D * D;
  const float pa = a.tc[(int64_t)D * D + D + 1];
  const float beta = a.sched[8 * i], eps = a.sched[8 * i + 1];
  const float gamma = a.params[a.lay.gamma];
  const float eta = gamma * eps, ome = 1.0f - eta, sig = sqrtf(2.0f * eta);
  const float inv2s2 = 1.0f / (2.0f * sig * sig), cst = logf(sig) + kHalfLog2Pi;
  const float fac = a.params[a.lay.g_factor];
  const uint32_t g0 = a.gkey[2 * p], g1 = a.gkey[2 * p + 1];
  float fk_acc = 0.f;
  for (int e = threadIdx.x; e < D; e += blockDim.x) {
    const float z = a.zr[uha_zr_ix(a, p, e)], rho = a.zr[uha_zr_ix(a, p, D + e)];
    float kr = 0.f, o = a.params[a.lay.g_b3 + e];
    for (int ks = 0; ks < a.nslab; ++ks) {
      kr += a.kr[((int64_t)ks * kMP + p) * D + e];
      o += a.sn[((int64_t)ks * kMP + p) * D + e];
    }
    const float s1 = o * fac;
    const float mean = a.params[a.lay.vd_mean + e];
    const float sd = expf(a.params[a.lay.vd_logdiag + e]);
    float gp = -kr + counts[e] - pa * expf(z);
    gp = fminf(fmaxf(gp, -1e2f), 1e2f);                      // gradU(z, beta, clip=1e2)          :23-30
    const float gq = -(z - mean) / (sd * sd);
    const float uf = -1.0f * (beta * gp + (1.0f - beta) * gq);
    const float mf = rho * ome - 2.0f * eta * s1;             // :52-54
    const int j = e < H ? e : e - H;
    uint32_t y0 = j, y1 = (H + j < D) ? H + j : 0;
    threefry2x32(g0, g1, y0, y1);
    const float rhop = mf + sig * bits_to_normal(e < H ? y0 : y1);   // :58-59
    const float df = rhop - mf;
    fk_acc += -(df * df) * inv2s2 - cst;
    const float rpp = rhop - eps * uf / 2.0f;                 // :62
    a.zrp[uha_zr_ix(a, p, e)] = z;
    a.zrp[uha_zr_ix(a, p, D + e)] = rhop;
    a.zn[uha_zn_ix(a, p, e)] = z + eps * rpp;                 // :63
    a.rpp[p * D + e] = rpp;
    if (a.traj) a.traj[((int64_t)(2 * K + 2 + i) * a.n_total + a.base + p) * D + e] = rhop;
  }
  const float fk = block_sum_n<NW>(fk_acc, sh);
  if (threadIdx.x == 0) a.fk[p] = fk;
}

// F2 (see the header above)
template <int NW>
__global__ __launch_bounds__(64 * NW) void lgcp_uha_close_kernel(LgcpUhaStepArgs a) {
  __shared__ float sh[NW];
  const int p = blockIdx.x, D = a.D, K = a.K, i = a.i;
  const bool last = i == K - 1;
  const float* counts = a.tc + (int64_t)D * D;
  const float mu0 = a.tc[(int64_t)D * D + D], pa = a.tc[(int64_t)D * D + D + 1], lognorm = a.tc[(int64_t)D * D + D + 2];
  const float beta = a.sched[8 * i], eps = a.sched[8 * i + 1];
  const float gamma = a.params[a.lay.gamma];
  const float eta = gamma * eps, ome = 1.0f - eta, sig = sqrtf(2.0f * eta);
  const float inv2s2 = 1.0f / (2.0f * sig * sig), cst = logf(sig) + kHalfLog2Pi;
  const float fac = a.params[a.lay.g_factor];
  // the key chain for bridge i + 1 (:55,84): ~450 dependent integer instructions, on the first lane of the LAST wave (the one
  // with the fewest elements to walk)
  if (threadIdx.x == 64 * (NW - 1) && !last) {
    uint32_t k0 = a.gen[2 * p], k1 = a.gen[2 * p + 1], G0, G1;
    lgcp_key_advance(k0, k1, G0, G1);
    a.gen[2 * p] = k0; a.gen[2 * p + 1] = k1;
    a.gkey[2 * p] = G0; a.gkey[2 * p + 1] = G1;
  }
  float bk_acc = 0.f, lp_acc = 0.f, lr_acc = 0.f;
  for (int e = threadIdx.x; e < D; e += blockDim.x) {
    const float rho = a.zr[uha_zr_ix(a, p, D + e)], rhop = a.zrp[uha_zr_ix(a, p, D + e)];
    const float zn = a.zn[uha_zn_ix(a, p, e)], rpp = a.rpp[p * D + e];
    float kr = 0.f, o = a.params[a.lay.g_b3 + e];
    for (int ks = 0; ks < a.nslab; ++ks) {
      kr += a.kr[((int64_t)ks * kMP + p) * D + e];
      o += a.sn[((int64_t)ks * kMP + p) * D + e];
    }
    const float s2 = o * fac;
    const float mb = rhop * ome + 2.0f * eta * s2;            // :77-80
    const float db = rho - mb;
    bk_acc += -(db * db) * inv2s2 - cst;                      // :84
    const float mean = a.params[a.lay.vd_mean + e];
    const float sd = expf(a.params[a.lay.vd_logdiag + e]);
    const float ez = expf(zn);
    const float graw = -kr + counts[e] - pa * ez;
    const float gp = fminf(fmaxf(graw, -1e2f), 1e2f);
    const float gq = -(zn - mean) / (sd * sd);
    const float ub = -1.0f * (beta * gp + (1.0f - beta) * gq);   // :65
    const float rnew = rpp - eps * ub / 2.0f;                 // :67
    a.zr[uha_zr_ix(a, p, e)] = zn;
    a.zr[uha_zr_ix(a, p, D + e)] = rnew;
    if (a.traj) {
      a.traj[((int64_t)(i + 1) * a.n_total + a.base + p) * D + e] = zn;
      a.traj[((int64_t)(K + 2 + i) * a.n_total + a.base + p) * D + e] = rnew;
    }
    if (last) {
      lp_acc += -0.5f * (zn - mu0) * kr + zn * counts[e] - pa * ez;
      lr_acc += -(rnew * rnew) * 0.5f - kHalfLog2Pi;
      a.out_z[(int64_t)p * D + e] = zn;
    }
  }
  const float bk = block_sum_n<NW>(bk_acc, sh);
  const float lp = block_sum_n<NW>(lp_acc, sh);
  const float lr = block_sum_n<NW>(lr_acc, sh);
  float fk_open = 0.f;
  if (a.fkslot) {                                             // per-tile partials of the fused F1, fixed order
    float t = 0.f;
    for (int tl = threadIdx.x; tl < D / 16; tl += blockDim.x) t += a.fkslot[tl * kMP + p];
    fk_open = block_sum_n<NW>(t, sh);
  }
  if (threadIdx.x == 0) {
    float w = a.w[p] + (bk - (a.fkslot ? fk_open : a.fk[p]));   // :88
    a.w[p] = w;
    if (last) {
      w += lr;                                                // + log N(rho_K; 0, 1)   :112
      w += lp + lognorm;                                      // + log p(z_K)           mcdboundingmachine.py:178
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
}

// the network on [zin | .] (lda = 2 D) at time index i: L1, L2, L3 of the header; `extra` (nullable): a second segment
// of the first launch, (extra - mu0) K^-1 -> kr slabs
static void lgcp_uha_net(const cmcd_desc& d, const cmcd_layout& lay, const float* params, const float* kinv, int M, int i,
                         const float* zin, const float* bias1, float* slab1, float* pre1, float* u1, float* slab2, float* pre2,
                         float* u2, float* sn, const float* extra, float* kr, int* counters, int gemm_lds, hipStream_t st) {
  const int D = d.dim, E = d.emb_dim, IN = 2 * D + E;
  const int cbD = (D + 63) / 64, cbIN = (IN + 63) / 64;
  const float mu0 = 3.8812819069514780f;
  const dim3 gblock(64 * kGemmWaves);
  GemmArgs g{};
  g.M = M; g.counters = counters;
  g.act.x = zin; g.act.D = 2 * D; g.act.IN = IN;
  g.Kdim = 2 * D; g.Kdim1 = extra ? D : 0;
  g.seg[0] = GemmSeg{zin, params + lay.g_w1, slab1, IN, 2 * D, IN, IN};
  if (extra) g.seg[1] = GemmSeg{extra, kinv, kr, D, D, D, D, mu0};
  g.nblk0 = cbIN; g.epi_seg = extra ? 0 : -1;
  g.act.mode = 1; g.act.bias = bias1 + (int64_t)i * IN; g.act.emb = params + lay.g_emb + (int64_t)i * E;
  g.act.sum_out = pre1; g.act.u_prev = nullptr; g.act.u_out = u1;
  hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACT>, dim3(cbIN + (extra ? cbD : 0), kSplit), gblock, gemm_lds, st, g);
  g.Kdim = IN; g.Kdim1 = 0; g.epi_seg = -1;
  g.seg[0] = GemmSeg{u1, params + lay.g_w2, slab2, IN, IN, IN, IN};
  g.nblk0 = cbIN;
  g.act.mode = 2; g.act.bias = params + lay.g_b2; g.act.sum_out = pre2; g.act.u_prev = u1; g.act.u_out = u2;
  hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACT>, dim3(cbIN, kSplit), gblock, gemm_lds, st, g);
  g.seg[0] = GemmSeg{u2, params + lay.g_w3, sn, D, IN, D, D};
  g.nblk0 = cbD;
  hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), gblock, gemm_lds, st, g);
}

static int64_t lgcp_uha_ws_total(const cmcd_desc& d, int64_t n, int64_t base) { return lgcp_uha_ws(d, n, base).total; }

// The 2nd-order sequence on the no-split-K GEMM (r04; header above lgcp_nsk_kernel): the same eight launches per bridge, the six
// GEMMs on 16 x 16 tiles over the whole contraction — 3200 / 3220 inputs are two rounds of 13 chunks per wave — with packed
// operands; the K^-1 product of z' rides in L6 (100 + 100 tiles) instead of L4 (202 + 100 would not fit the chip at one
// workgroup per CU in the 17 .. 20-particle form); F1 / F2 read ONE product array each instead of summing eight slabs.
static int lgcp_uha_forward_nsk(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const LgcpUhaWs& w,
                                const int32_t* seeds, int64_t n, const float* params, const float* tc, float* ws, float* out_loss,
                                float* out_z, double* partials, float* traj, hipStream_t stream, float* keep_gws) {
  const int D = d.dim, E = d.emb_dim, IN = 2 * D + E, K = d.nbridges;
  const int tIN = (IN + 15) / 16, tD = D / 16, big = 2 * kNskChunks;
  const LgcpKeep keep = keep_gws ? lgcp_uha_keep(d, n, keep_gws) : LgcpKeep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false};
  const float mu0 = 3.8812819069514780f;
  {
    NskPackArgs pk{};
    pk.src[0] = params + lay.g_w1; pk.dst[0] = ws + w.w1p; pk.K[0] = 2 * D; pk.N[0] = IN; pk.ntile[0] = tIN; pk.nch[0] = big;
    pk.src[1] = params + lay.g_w2; pk.dst[1] = ws + w.w2p; pk.K[1] = IN; pk.N[1] = IN; pk.ntile[1] = tIN; pk.nch[1] = big;
    pk.src[2] = params + lay.g_w3; pk.dst[2] = ws + w.w3p; pk.K[2] = IN; pk.N[2] = D; pk.ntile[2] = tD; pk.nch[2] = big;
    pk.src[3] = tc; pk.dst[3] = ws + w.kip; pk.K[3] = D; pk.N[3] = D; pk.ntile[3] = tD; pk.nch[3] = big;   // zero rows beyond D
    const int64_t groups = (int64_t)tIN * big * 64;
    hipLaunchKernelGGL(lgcp_nsk_pack_kernel, dim3((unsigned)((groups + 255) / 256), 4), dim3(256), 0, stream, pk);
  }
  float* const zrA = ws + w.ops, *zrpA = zrA + 2 * kNskOperand, *u1A = zrpA + 2 * kNskOperand, *u2A = u1A + 2 * kNskOperand;
  float* const znA = u2A + 2 * kNskOperand;
  const dim3 gblock(64 * kGemmWaves);
  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    // the padding of the packed operands (rows >= M, inputs >= IN) must read as zeros
    if (hipMemsetAsync(zrA, 0, sizeof(float) * (5 * 2 * kNskOperand), stream) != hipSuccess) return CMCD_ERR_HIP;
    LgcpUhaStepArgs sa{};
    sa.seeds = seeds + base; sa.params = params; sa.tc = tc; sa.sched = ws + sw.sched;
    sa.zr = zrA; sa.zrp = zrpA; sa.zn = znA; sa.rpp = ws + w.rpp; sa.kr = ws + w.kr; sa.sn = ws + w.sn;
    sa.w = ws + w.w; sa.fk = ws + w.fk; sa.gen = reinterpret_cast<uint32_t*>(ws + w.keys);
    sa.gkey = reinterpret_cast<uint32_t*>(ws + w.gkey);
    sa.out_loss = out_loss + base; sa.out_z = out_z + base * D; sa.partials = partials + base * CMCD_NSTATS;
    sa.traj = traj; sa.n_total = n; sa.base = base; sa.lay = lay; sa.M = M; sa.D = D; sa.K = K; sa.i = 0;
    sa.packed = 1; sa.nslab = 1;
    hipLaunchKernelGGL(lgcp_uha_init_kernel, dim3(M), dim3(256), 0, stream, sa);
    const bool can_merge = M > 16 && M <= 20;
    auto launch = [&](NskArgs& na, int tiles) {
      // 17 .. 20 particles: one workgroup per column tile (16x16x4 + 4x4x1 on the same weight registers) — at these widths the
      // weights are the bytes, so every launch takes the form that fetches them once
      if (can_merge) hipLaunchKernelGGL((lgcp_nsk_kernel<0, true, 2>), dim3((unsigned)tiles, 1), gblock, 0, stream, na);
      else hipLaunchKernelGGL((lgcp_nsk_kernel<0, false, 2>), dim3((unsigned)tiles, M > 16 ? 2 : 1), gblock, 0, stream, na);
    };
    auto kinv_seg = [&]() { NskSeg sg{znA, ws + w.kip, D, NSK_KR, mu0}; sg.nch = big; return sg; };
    {   // K^-1 (z_0 - mu0)
      NskArgs na{};
      na.M = M; na.D = 2 * D; na.IN = IN; na.seg[0] = kinv_seg(); na.nt0 = tD; na.krOutN = ws + w.kr;
      if (keep.on) na.keepKr = keep.kr + base * D;
      launch(na, tD);
    }
    float* const fkslot = ws + w.slab1;      // [D / 16][kMP] (the split-K form's slab region is free here)
    sa.fkslot = fkslot;
    // the network on `zin` at time index i: L1, L2, L3 of the header; with_kinv: (z' - mu0) K^-1 beside L3.  The first network of
    // a bridge (!with_kinv) has F1 as the consumer of its L3
    auto net = [&](int i, float* zin, bool with_kinv) {
      NskArgs na{};
      na.M = M; na.D = 2 * D; na.IN = IN; na.nch_out = big;
      na.seg[0] = NskSeg{zin, ws + w.w1p, IN, NSK_ACT1, 0.f}; na.seg[0].nch = big; na.nt0 = tIN;
      na.xA = zin; na.bias = ws + w.bias1 + (int64_t)i * IN; na.emb = params + lay.g_emb + (int64_t)i * E; na.outA = u1A;
      // (gradient calls: this evaluation's rows of the kept tables — 2 i for [z; rho], 2 i + 1 for [z; rho'])
      const int64_t krow = (int64_t)(2 * i + (with_kinv ? 1 : 0)) * n + base;
      if (keep.on) { na.keepPre = keep.pre1 + krow * IN; na.keepU = keep.u1 + krow * IN; }
      launch(na, tIN);
      na.seg[0] = NskSeg{u1A, ws + w.w2p, IN, NSK_ACT2, 0.f}; na.seg[0].nch = big;
      na.bias = params + lay.g_b2; na.uA = u1A; na.outA = u2A;
      if (keep.on) { na.keepPre = keep.pre2 + krow * IN; na.keepU = keep.u2 + krow * IN; }
      launch(na, tIN);
      na.seg[0] = NskSeg{u2A, ws + w.w3p, D, with_kinv ? NSK_OUT : NSK_UHA_MID, 0.f}; na.seg[0].nch = big; na.nt0 = tD;
      na.outN = ws + w.sn;
      na.keepPre = nullptr; na.keepU = nullptr;
      if (keep.on) na.keepSn = keep.sn + krow * D;
      if (with_kinv) {
        na.seg[1] = kinv_seg(); na.krOutN = ws + w.kr;
        if (keep.on) na.keepKr = keep.kr + ((int64_t)(i + 1) * n + base) * D;   // K^-1 (z_{i+1} - mu0)
        launch(na, 2 * tD);
        return;
      }
      NskArgs::UhaMid& um = na.um;
      um.params = params; um.tc = tc; um.sched = ws + sw.sched; um.zr = zrA; um.zrp = zrpA; um.zn = znA; um.rpp = ws + w.rpp;
      um.kr = ws + w.kr; um.gkey = reinterpret_cast<const uint32_t*>(ws + w.gkey); um.fkslot = fkslot; um.traj = traj;
      um.n_total = n; um.base = base; um.lay = lay; um.D = D; um.K = K; um.i = i;
      if (can_merge) hipLaunchKernelGGL((lgcp_nsk_kernel<2, true, 2>), dim3((unsigned)tD, 1), gblock, 0, stream, na);
      else hipLaunchKernelGGL((lgcp_nsk_kernel<2, false, 2>), dim3((unsigned)tD, M > 16 ? 2 : 1), gblock, 0, stream, na);
    };
    for (int i = 0; i < K; ++i) {
      sa.i = i;
      net(i, zrA, false);       // L1, L2, L3 + F1
      net(i, zrpA, true);       // L4, L5, L6 | K^-1 (z' - mu0)
      hipLaunchKernelGGL(lgcp_uha_close_kernel<16>, dim3(M), dim3(1024), 0, stream, sa);
    }
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

static int lgcp_uha_forward(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const int32_t* seeds, int64_t n,
                            const float* params, const float* tc, float* ws, float* out_loss, float* out_z,
                            double** partials_out, float* traj, hipStream_t stream, float* keep_gws) {
  const int D = d.dim, E = d.emb_dim, IN = 2 * D + E, K = d.nbridges;
  const LgcpUhaWs w = lgcp_uha_ws(d, n, sw.total_floats);
  LgcpPrepArgs pa{params, ws + w.bias1, lay, 2 * D, E, K, IN};   // bias1_i = b1 + emb_i W1[2d:, :]
  hipLaunchKernelGGL(lgcp_prep_kernel, dim3((IN + 255) / 256, K + 1), dim3(256), 0, stream, pa);
  if (lgcp_uha_nsk_ok(d)) {
    double* partials_n = reinterpret_cast<double*>(ws + w.partials);
    *partials_out = partials_n;
    return lgcp_uha_forward_nsk(d, lay, sw, w, seeds, n, params, tc, ws, out_loss, out_z, partials_n, traj, stream, keep_gws);
  }
  const int gemm_lds = lgcp_gemm_attrs();
  if (gemm_lds < 0) return CMCD_ERR_HIP;
  double* partials = reinterpret_cast<double*>(ws + w.partials);
  *partials_out = partials;
  const float mu0 = 3.8812819069514780f;
  const int cbD = (D + 63) / 64, cbIN = (IN + 63) / 64;
  int* counters = reinterpret_cast<int*>(ws + w.counters);
  if (hipMemsetAsync(counters, 0, sizeof(int) * (cbD + cbIN), stream) != hipSuccess) return CMCD_ERR_HIP;
  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    LgcpUhaStepArgs sa{};
    sa.seeds = seeds + base; sa.params = params; sa.tc = tc; sa.sched = ws + sw.sched;
    sa.zr = ws + w.zr; sa.zrp = ws + w.zrp; sa.zn = ws + w.zn; sa.rpp = ws + w.rpp; sa.kr = ws + w.kr; sa.sn = ws + w.sn;
    sa.w = ws + w.w; sa.fk = ws + w.fk; sa.gen = reinterpret_cast<uint32_t*>(ws + w.keys);
    sa.gkey = reinterpret_cast<uint32_t*>(ws + w.gkey);
    sa.out_loss = out_loss + base; sa.out_z = out_z + base * D; sa.partials = partials + base * CMCD_NSTATS;
    sa.traj = traj; sa.n_total = n; sa.base = base; sa.lay = lay; sa.M = M; sa.D = D; sa.K = K; sa.i = 0;
    sa.packed = 0; sa.nslab = kSplit;
    hipLaunchKernelGGL(lgcp_uha_init_kernel, dim3(M), dim3(256), 0, stream, sa);
    {   // K^-1 (z_0 - mu0)
      GemmArgs g{};
      g.M = M; g.Kdim = D; g.counters = counters;
      g.seg[0] = GemmSeg{ws + w.zr, tc, ws + w.kr, D, 2 * D, D, D, mu0};
      g.nblk0 = cbD;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), dim3(64 * kGemmWaves), gemm_lds, stream, g);
    }
    for (int i = 0; i < K; ++i) {
      sa.i = i;
      lgcp_uha_net(d, lay, params, tc, M, i, ws + w.zr, ws + w.bias1, ws + w.slab1, ws + w.pre1, ws + w.u1, ws + w.slab2,
                   ws + w.pre2, ws + w.u2, ws + w.sn, nullptr, nullptr, counters, gemm_lds, stream);
      hipLaunchKernelGGL(lgcp_uha_mid_kernel<4>, dim3(M), dim3(256), 0, stream, sa);
      lgcp_uha_net(d, lay, params, tc, M, i, ws + w.zrp, ws + w.bias1, ws + w.slab1, ws + w.pre1, ws + w.u1, ws + w.slab2,
                   ws + w.pre2, ws + w.u2, ws + w.sn, ws + w.zn, ws + w.kr, counters, gemm_lds, stream);
      hipLaunchKernelGGL(lgcp_uha_close_kernel<4>, dim3(M), dim3(256), 0, stream, sa);
    }
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

// -------------------------------------------------------------------------------------------------------------
// Reparameterised gradient of MCD_CAIS_UHA_sn on the lgcp path: the reverse recursion of cmcd_uha.hip (header there)
// as a launch sequence.  Per point e = K..0: element-wise point kernel (cotangents of ub / uf, clip mask, q gradients)
// + one GEMM for H_p v = -K^-1 v - a e^z v; per bridge i = e - 1: recompute both network evaluations at the kept
// (z_i, rho_i, rho'_i), back-propagate s2 then s1 (their cotangents are chained through d rho') through three skinny
// GEMMs each against transposed weight copies.  The O(width^2) parameter gradients are deferred: every evaluation's
// (input, u1, u2, d a1, d a2, d o) rows are kept and contracted at the end by A^T B products (inner dimension 2 K n).
// -------------------------------------------------------------------------------------------------------------
struct LgcpUhaFwdSet { int64_t zin, slab1, pre1, u1, slab2, pre2, u2, sn; };

struct LgcpUhaGradWs {
  int64_t wt1, wt2, wt3;                        // W1[:2D]^T [IN][2D], W2^T [IN][IN], W3^T [D][IN]
  LgcpUhaFwdSet fs[2];                          // evaluation A = s([z; rho]), B = s([z; rho'])
  int64_t kr, hv;                               // [kSplit][kMP][D]
  int64_t v, dO, arp, gb, lrn, arppc;           // [kMP][D]
  int64_t du2s, ts;                             // [kSplit][kMP][IN]
  int64_t du2, du1, da2, da1;                   // [kMP][IN]
  int64_t dxf;                                  // [kSplit][kMP][2D]
  int64_t XIN, U1, U2, DA1, DA2, DO;            // [2 K n][2D | IN | IN | IN | IN | D]
  int64_t zero_lo, zero_hi;
  int64_t lz, lr, gmu_acc, glam_acc;            // [kMP][D]
  int64_t geta, gepsd;                          // [kMP]
  int64_t S, S2, gb2, gbeta, geps, counters;
  int64_t sc;                                   // [(K+1)][n][8]
  int64_t kpre1, kpre2, ksn, kkr;               // kept by the forward (lgcp_uha_keep): [2 K n][IN] x 2, [2 K n][D], [(K+1) n][D]
  bool keep;
  int64_t bops, wt3p, wt2p, wt1p, kip1;         // keep: packed dO | v (one round) | da2 | da1 (two rounds), packed W3^T, W2^T, W1[:2D]^T, K^-1
  int64_t total;
};

static LgcpUhaGradWs lgcp_uha_grad_ws(const cmcd_desc& d, int64_t n) {
  const int64_t D = d.dim, IN = 2 * D + d.emb_dim, K = d.nbridges, R = 2 * K * n;
  LgcpUhaGradWs w;
  int64_t o = 0;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  w.wt1 = take(IN * 2 * D); w.wt2 = take(IN * IN); w.wt3 = take(D * IN);
  for (int b = 0; b < 2; ++b) {
    LgcpUhaFwdSet& f = w.fs[b];
    f.zin = take(kMP * 2 * D); f.slab1 = take(kSplit * kMP * IN); f.pre1 = take(kMP * IN); f.u1 = take(kMP * IN);
    f.slab2 = take(kSplit * kMP * IN); f.pre2 = take(kMP * IN); f.u2 = take(kMP * IN); f.sn = take(kSplit * kMP * D);
  }
  w.kr = take(kSplit * kMP * D); w.hv = take(kSplit * kMP * D);
  w.v = take(kMP * D); w.dO = take(kMP * D); w.arp = take(kMP * D); w.gb = take(kMP * D); w.lrn = take(kMP * D);
  w.arppc = take(kMP * D);
  w.du2s = take(kSplit * kMP * IN); w.ts = take(kSplit * kMP * IN);
  w.du2 = take(kMP * IN); w.du1 = take(kMP * IN); w.da2 = take(kMP * IN); w.da1 = take(kMP * IN);
  w.dxf = take(kSplit * kMP * 2 * D);
  w.XIN = take(R * 2 * D); w.U1 = take(R * IN); w.U2 = take(R * IN); w.DA1 = take(R * IN); w.DA2 = take(R * IN); w.DO = take(R * D);
  w.zero_lo = o;
  w.lz = take(kMP * D); w.lr = take(kMP * D); w.gmu_acc = take(kMP * D); w.glam_acc = take(kMP * D);
  w.geta = take(kMP); w.gepsd = take(kMP);
  w.S = take((K + 1) * IN); w.S2 = take((K + 1) * IN); w.gb2 = take(IN); w.gbeta = take(K); w.geps = take(K);
  w.counters = take(((D + 63) / 64) + ((IN + 63) / 64));
  w.sc = take((K + 1) * n * 8);
  w.zero_hi = o;
  // the forward's kept activations (behind everything else, outside the cleared range)
  w.keep = lgcp_uha_nsk_ok(d) && R * (2 * IN + D) + (K + 1) * n * D <= (int64_t(1) << 28);
  w.kpre1 = take(w.keep ? R * IN : 0); w.kpre2 = take(w.keep ? R * IN : 0);
  w.ksn = take(w.keep ? R * D : 0); w.kkr = take(w.keep ? (K + 1) * n * D : 0);
  {
    const int64_t tIN = (IN + 15) / 16, tD = D / 16, t2D = 2 * D / 16, c1 = kNskChunks * 256, c2 = 2 * c1;
    w.bops = take(w.keep ? 6 * kNskOperand : 0);        // dO, v: kNskOperand each; da2, da1: 2 kNskOperand each
    w.wt3p = take(w.keep ? tIN * c1 : 0); w.wt2p = take(w.keep ? tIN * c2 : 0); w.wt1p = take(w.keep ? t2D * c2 : 0);
    w.kip1 = take(w.keep ? tD * c1 : 0);
  }
  w.total = o;
  return w;
}

static LgcpKeep lgcp_uha_keep(const cmcd_desc& d, int64_t n, float* gws) {
  const LgcpUhaGradWs g = lgcp_uha_grad_ws(d, n);
  if (!g.keep) return LgcpKeep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, false};
  return LgcpKeep{gws + g.kpre1, gws + g.U1, gws + g.kpre2, gws + g.U2, gws + g.kkr, gws + g.ksn, true};
}

struct LgcpUhaAdjArgs {
  const float* params;
  const float* tc;
  const float* sched;
  const float* traj;         // [3K+2][n][D]
  const float* kr;           // [kSplit][kMP][D]  K^-1 (z_e - mu0)
  const float* hv;           // [kSplit][kMP][D]  v K^-1
  const float* snA;          // [kSplit][kMP][D]  u2 W3 slabs of evaluation A / B
  const float* snB;
  int nslab = kSplit;        // slabs of kr / snA / snB to sum: kSplit (recomputed) or 1 (kept by the forward)
  int bslab = kSplit;        // slabs of hv / dxf to sum: kSplit, or 1 (row-major outputs of the no-split-K launches)
  float* dOp = nullptr;      // packed copies of dO / v for the no-split-K launches (nullptr: none)
  float* vp = nullptr;
  const float* du1;          // [kMP][IN]  d u1 of the evaluation just back-propagated (residual path)
  const float* dxf;          // [kSplit][kMP][2D]  d a1 W1[:2D]^T
  float* zinA;               // [kMP][2D]
  float* zinB;
  float* XIN;                // [2 K n][2D]
  float* lz;                 // [kMP][D] dL/dz_e
  float* lr;                 // [kMP][D] dL/drho_e
  float* arppc;              // [kMP][D] dL/drho''_e of the bridge walked last
  float* arp;
  float* gb;
  float* lrn;
  float* v;
  float* dO;
  float* DObig;              // [2 K n][D]
  float* gmu_acc;
  float* glam_acc;
  float* geta;               // [kMP]
  float* gepsd;              // [kMP]
  float* sc;                 // [(K+1)][n][8]
  cmcd_layout lay;
  int64_t n, base;
  int M, D, IN, K, e;        // e: the point (point kernel) or e = i + 1 for the bridge kernels of bridge i
  float omega;
};

// [z_i | rho_i] and [z_i | rho'_i]: the operands of the recomputed network evaluations; also rows 2 i, 2 i + 1 of XIN
__global__ void lgcp_uha_gather_kernel(LgcpUhaAdjArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, D = a.D, K = a.K, i = a.e - 1;
  if (idx >= a.M * D) return;
  const int m = idx / D, j = idx - m * D;
  const int64_t pr = a.base + m;
  const float z = a.traj[((int64_t)i * a.n + pr) * D + j];
  const float rho = a.traj[((int64_t)(K + 1 + i) * a.n + pr) * D + j];
  const float rhop = a.traj[((int64_t)(2 * K + 2 + i) * a.n + pr) * D + j];
  a.zinA[m * 2 * D + j] = z; a.zinA[m * 2 * D + D + j] = rho;
  a.zinB[m * 2 * D + j] = z; a.zinB[m * 2 * D + D + j] = rhop;
  float* xa = a.XIN + ((int64_t)(2 * i) * a.n + pr) * 2 * D;
  float* xb = a.XIN + ((int64_t)(2 * i + 1) * a.n + pr) * 2 * D;
  xa[j] = z; xa[D + j] = rho;
  xb[j] = z; xb[D + j] = rhop;
}

// r04 (activations kept by the forward: no recompute operands needed): all rows of XIN in ONE launch, blockIdx.y = bridge
__global__ void lgcp_uha_xin_kernel(LgcpUhaAdjArgs a) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, D = a.D, K = a.K, i = blockIdx.y;
  if (idx >= a.M * D) return;
  const int m = idx / D, j = idx - m * D;
  const int64_t pr = a.base + m;
  const float z = a.traj[((int64_t)i * a.n + pr) * D + j];
  float* xa = a.XIN + ((int64_t)(2 * i) * a.n + pr) * 2 * D;
  float* xb = a.XIN + ((int64_t)(2 * i + 1) * a.n + pr) * 2 * D;
  xa[j] = z; xa[D + j] = a.traj[((int64_t)(K + 1 + i) * a.n + pr) * D + j];
  xb[j] = z; xb[D + j] = a.traj[((int64_t)(2 * K + 2 + i) * a.n + pr) * D + j];
}

// r04: the per-bridge adjoint kernels run kUhaAdjW waves per particle (20 blocks of one 256-thread workgroup were latency
// chains: what they needed was width, as the forward's closing step got); block sums in wave order (fixed)
constexpr int kUhaAdjW = 16;
// point e: cotangents of ub (bridge e - 1) and uf (bridge e) at z_e, v = clipmask . a_gp for the Hessian product, q gradients
__global__ __launch_bounds__(64 * kUhaAdjW) void lgcp_uha_adj_point_kernel(LgcpUhaAdjArgs a) {
  __shared__ float sh[kUhaAdjW];
  const int p = blockIdx.x, D = a.D, K = a.K, e = a.e;
  const int64_t pr = a.base + p;
  const float* counts = a.tc + (int64_t)D * D;
  const float pa = a.tc[(int64_t)D * D + D + 1];
  const float om = a.omega;
  const float bl = e >= 1 ? a.sched[8 * (e - 1)] : 0.f, el = e >= 1 ? a.sched[8 * (e - 1) + 1] : 0.f;
  const float bh = e < K ? a.sched[8 * e] : 0.f, eh = e < K ? a.sched[8 * e + 1] : 0.f;
  float sbl = 0.f, sel = 0.f, sbh = 0.f, seh = 0.f;
  for (int j = threadIdx.x; j < D; j += blockDim.x) {
    const float z = a.traj[((int64_t)e * a.n + pr) * D + j];
    float kr = 0.f;
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) kr += ks < a.nslab ? a.kr[((int64_t)ks * kMP + p) * D + j] : 0.f;
    const float graw = -kr + counts[j] - pa * expf(z);
    const float gp = fminf(fmaxf(graw, -1e2f), 1e2f);
    const float msk = fabsf(graw) < 1e2f ? 1.0f : 0.f;
    const float mean = a.params[a.lay.vd_mean + j];
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    const float qiv = 1.0f / (sd * sd);
    const float gq = -(z - mean) * qiv;
    float lz = e == K ? 0.f : a.lz[p * D + j];
    float lr = e == K ? om * a.traj[((int64_t)(2 * K + 1) * a.n + pr) * D + j] : a.lr[p * D + j];   // dL/drho_K = omega rho_K
    if (e == K) a.lr[p * D + j] = lr;
    float a_gp = 0.f, a_gq = 0.f;
    if (e >= 1) {
      const float adj = -0.5f * el * lr;
      const float ub = -1.0f * (bl * gp + (1.0f - bl) * gq);
      a_gp -= bl * adj; a_gq -= (1.0f - bl) * adj;
      sbl -= (gp - gq) * adj;
      sel -= 0.5f * ub * lr;
    }
    if (e < K) {
      const float ar = a.arppc[p * D + j];
      const float adj = -0.5f * eh * ar;
      const float uf = -1.0f * (bh * gp + (1.0f - bh) * gq);
      a_gp -= bh * adj; a_gq -= (1.0f - bh) * adj;
      sbh -= (gp - gq) * adj;
      seh -= 0.5f * uf * ar;
    }
    if (e == K) lz -= om * graw;
    if (e == 0) lz += om * gq;
    lz -= a_gq * qiv;
    a.gmu_acc[p * D + j] += a_gq * qiv;
    a.glam_acc[p * D + j] += a_gq * (-2.0f * gq);
    a.v[p * D + j] = msk * a_gp;
    if (a.vp) a.vp[nsk_pack(p, j)] = msk * a_gp;
    a.lz[p * D + j] = lz;
  }
  sbl = block_sum_n<kUhaAdjW>(sbl, sh); sel = block_sum_n<kUhaAdjW>(sel, sh); sbh = block_sum_n<kUhaAdjW>(sbh, sh); seh = block_sum_n<kUhaAdjW>(seh, sh);
  if (threadIdx.x == 0) {
    float* o = a.sc + ((int64_t)e * a.n + pr) * 8;
    o[0] = sbl; o[1] = sel; o[2] = sbh; o[3] = seh;
  }
}

// bridge i = e - 1, first half: finish lz_e with the Hessian product, leap-frog adjoints, g_b and the cotangent of s2
__global__ __launch_bounds__(64 * kUhaAdjW) void lgcp_uha_adj_b_kernel(LgcpUhaAdjArgs a) {
  __shared__ float sh[kUhaAdjW];
  const int p = blockIdx.x, D = a.D, K = a.K, e = a.e, i = e - 1;
  const int64_t pr = a.base + p;
  const float pa = a.tc[(int64_t)D * D + D + 1];
  const float om = a.omega;
  const float eps = a.sched[8 * i + 1];
  const float gamma = a.params[a.lay.gamma];
  const float eta = gamma * eps, ome = 1.0f - eta, inv2eta = 0.5f / eta;
  const float fac = a.params[a.lay.g_factor];
  float geta = 0.f, gepsd = 0.f, gfac = 0.f;
  for (int j = threadIdx.x; j < D; j += blockDim.x) {
    const float ze = a.traj[((int64_t)e * a.n + pr) * D + j];
    const float z = a.traj[((int64_t)i * a.n + pr) * D + j];
    const float rho = a.traj[((int64_t)(K + 1 + i) * a.n + pr) * D + j];
    const float rhop = a.traj[((int64_t)(2 * K + 2 + i) * a.n + pr) * D + j];
    float hv = 0.f, o = a.params[a.lay.g_b3 + j];
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) {
      hv += ks < a.bslab ? a.hv[((int64_t)ks * kMP + p) * D + j] : 0.f;
      o += ks < a.nslab ? a.snB[((int64_t)ks * kMP + p) * D + j] : 0.f;
    }
    const float lz = a.lz[p * D + j] - hv - pa * expf(ze) * a.v[p * D + j];     // H_p v = -K^-1 v - a e^z v
    a.lz[p * D + j] = lz;
    const float arpp = a.lr[p * D + j] + eps * lz;
    a.arppc[p * D + j] = arpp;
    gepsd += ((ze - z) / eps) * lz;
    const float s2 = o * fac;
    const float mb = rhop * ome + 2.0f * eta * s2;
    const float r = rho - mb;
    const float gb = -om * r * inv2eta;
    a.gb[p * D + j] = gb;
    a.arp[p * D + j] = arpp + ome * gb;
    const float cot = 2.0f * eta * gb;
    geta += (2.0f * s2 - rhop) * gb - om * r * r * inv2eta * inv2eta;
    gfac += cot * o;
    a.dO[p * D + j] = cot * fac;
    if (a.dOp) a.dOp[nsk_pack(p, j)] = cot * fac;
    a.DObig[((int64_t)(2 * i + 1) * a.n + pr) * D + j] = cot * fac;
  }
  geta = block_sum_n<kUhaAdjW>(geta, sh); gepsd = block_sum_n<kUhaAdjW>(gepsd, sh); gfac = block_sum_n<kUhaAdjW>(gfac, sh);
  if (threadIdx.x == 0) {
    a.geta[p] = geta; a.gepsd[p] = gepsd;
    a.sc[((int64_t)i * a.n + pr) * 8 + 6] = gfac;
  }
}

// bridge i, second half: d [z; rho'] of s2 arrives; cotangent of s1
__global__ __launch_bounds__(64 * kUhaAdjW) void lgcp_uha_adj_a_kernel(LgcpUhaAdjArgs a) {
  __shared__ float sh[kUhaAdjW];
  const int p = blockIdx.x, D = a.D, K = a.K, i = a.e - 1, IN = a.IN;
  const int64_t pr = a.base + p;
  const float eps = a.sched[8 * i + 1];
  const float gamma = a.params[a.lay.gamma];
  const float eta = gamma * eps, ome = 1.0f - eta, inv2eta = 0.5f / eta;
  const float fac = a.params[a.lay.g_factor];
  float geta = 0.f, gfac = 0.f;
  for (int j = threadIdx.x; j < D; j += blockDim.x) {
    const float rho = a.traj[((int64_t)(K + 1 + i) * a.n + pr) * D + j];
    const float rhop = a.traj[((int64_t)(2 * K + 2 + i) * a.n + pr) * D + j];
    float dz = a.du1[p * IN + j], dr = a.du1[p * IN + D + j], o = a.params[a.lay.g_b3 + j];   // residual path: d x += d u1[:2D]
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) {
      dz += ks < a.bslab ? a.dxf[((int64_t)ks * kMP + p) * 2 * D + j] : 0.f;
      dr += ks < a.bslab ? a.dxf[((int64_t)ks * kMP + p) * 2 * D + D + j] : 0.f;
      o += ks < a.nslab ? a.snA[((int64_t)ks * kMP + p) * D + j] : 0.f;
    }
    a.lz[p * D + j] += dz;
    const float arp = a.arp[p * D + j] + dr;
    const float s1 = o * fac;
    const float mf = rho * ome - 2.0f * eta * s1;
    const float cot = -2.0f * eta * arp;
    geta += ((rhop - mf) * inv2eta - rho - 2.0f * s1) * arp;
    a.lrn[p * D + j] = ome * arp - a.gb[p * D + j];
    gfac += cot * o;
    a.dO[p * D + j] = cot * fac;
    if (a.dOp) a.dOp[nsk_pack(p, j)] = cot * fac;
    a.DObig[((int64_t)(2 * i) * a.n + pr) * D + j] = cot * fac;
  }
  geta = block_sum_n<kUhaAdjW>(geta, sh); gfac = block_sum_n<kUhaAdjW>(gfac, sh);
  if (threadIdx.x == 0) {
    a.geta[p] += geta;
    a.sc[((int64_t)i * a.n + pr) * 8 + 7] = gfac;
  }
}

// bridge i, end: d [z; rho] of s1 arrives -> (lz, lr) of point i (its Hessian part follows in the point kernel)
__global__ __launch_bounds__(64 * kUhaAdjW) void lgcp_uha_adj_fin_kernel(LgcpUhaAdjArgs a) {
  const int p = blockIdx.x, D = a.D, i = a.e - 1, IN = a.IN;
  const int64_t pr = a.base + p;
  for (int j = threadIdx.x; j < D; j += blockDim.x) {
    float dz = a.du1[p * IN + j], dr = a.du1[p * IN + D + j];
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) {
      dz += ks < a.bslab ? a.dxf[((int64_t)ks * kMP + p) * 2 * D + j] : 0.f;
      dr += ks < a.bslab ? a.dxf[((int64_t)ks * kMP + p) * 2 * D + D + j] : 0.f;
    }
    a.lz[p * D + j] += dz;
    a.lr[p * D + j] = a.lrn[p * D + j] + dr;
  }
  if (threadIdx.x == 0) {
    const float eps = a.sched[8 * i + 1], gamma = a.params[a.lay.gamma];
    float* o = a.sc + ((int64_t)i * a.n + pr) * 8;
    o[4] = a.gepsd[p] + gamma * a.geta[p];      // d / d eps_i (leap-frog + eta = gamma eps)
    o[5] = eps * a.geta[p];                     // d / d gamma
  }
}

// point 0, after its Hessian product: z_0 = mean + std e0 and the explicit parameters of log q(z_0)
__global__ __launch_bounds__(256) void lgcp_uha_adj_z0_kernel(LgcpUhaAdjArgs a) {
  const int p = blockIdx.x, D = a.D;
  const int64_t pr = a.base + p;
  const float pa = a.tc[(int64_t)D * D + D + 1];
  for (int j = threadIdx.x; j < D; j += blockDim.x) {
    const float z = a.traj[pr * D + j];
    float hv = 0.f;
#pragma unroll
    for (int ks = 0; ks < kSplit; ++ks) hv += ks < a.bslab ? a.hv[((int64_t)ks * kMP + p) * D + j] : 0.f;
    const float lz = a.lz[p * D + j] - hv - pa * expf(z) * a.v[p * D + j];
    const float mean = a.params[a.lay.vd_mean + j];
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    const float qiv = 1.0f / (sd * sd), dz = z - mean, gq = -dz * qiv;
    a.gmu_acc[p * D + j] += lz - a.omega * gq;
    a.glam_acc[p * D + j] += lz * dz + a.omega * (dz * dz * qiv - 1.0f);
  }
}

// gbeta / geps tables, d gamma, d factor_sn from the per-(point, particle) records; fixed order
__global__ __launch_bounds__(256) void lgcp_uha_scal_reduce_kernel(const float* sc, int64_t n, int K, float* gbeta, float* geps,
                                                                   float* grad_gamma, float* grad_factor) {
  __shared__ float sh[4];
  const int k = blockIdx.x;   // k < K: bridge k;  k == K: the two scalars
  if (k < K) {
    float b = 0.f, e = 0.f;
    for (int64_t p = threadIdx.x; p < n; p += blockDim.x) {
      const float* lo = sc + ((int64_t)(k + 1) * n + p) * 8;   // point k + 1: ub side of bridge k
      const float* hi = sc + ((int64_t)k * n + p) * 8;         // point k: uf side; bridge k's own records
      b += lo[0] + hi[2];
      e += lo[1] + hi[3] + hi[4];
    }
    b = block_sum_256(b, sh); e = block_sum_256(e, sh);
    if (threadIdx.x == 0) { gbeta[k] = b; geps[k] = e; }
  } else {
    float g = 0.f, f = 0.f;
    for (int64_t t = threadIdx.x; t < (int64_t)K * n; t += blockDim.x) {
      const float* r = sc + t * 8;
      g += r[5];
      f += r[6] + r[7];
    }
    g = block_sum_256(g, sh); f = block_sum_256(f, sh);
    if (threadIdx.x == 0) { *grad_gamma = g; *grad_factor = f; }
  }
}

static int64_t lgcp_uha_grad_ws_total(const cmcd_desc& d, int64_t n) { return lgcp_uha_grad_ws(d, n).total; }

static int lgcp_uha_grad(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, int64_t n, const float* params,
                         int64_t n_params, const float* tc, float* ws, const float* traj, float* gws, float omega,
                         float* grad, hipStream_t stream) {
  const int D = d.dim, E = d.emb_dim, IN = 2 * D + E, K = d.nbridges;
  const LgcpUhaWs w = lgcp_uha_ws(d, n, sw.total_floats);
  const LgcpUhaGradWs g = lgcp_uha_grad_ws(d, n);
  if (hipMemsetAsync(grad, 0, sizeof(float) * n_params, stream) != hipSuccess) return CMCD_ERR_HIP;
  if (hipMemsetAsync(gws + g.zero_lo, 0, sizeof(float) * (g.zero_hi - g.zero_lo), stream) != hipSuccess) return CMCD_ERR_HIP;
  {
    const dim3 tb(32, 8);
    hipLaunchKernelGGL(lgcp_transpose_kernel, dim3((IN + 31) / 32, (2 * D + 31) / 32), tb, 0, stream, params + lay.g_w1,
                       gws + g.wt1, 2 * D, IN, IN, 2 * D);
    hipLaunchKernelGGL(lgcp_transpose_kernel, dim3((IN + 31) / 32, (IN + 31) / 32), tb, 0, stream, params + lay.g_w2,
                       gws + g.wt2, IN, IN, IN, IN);
    hipLaunchKernelGGL(lgcp_transpose_kernel, dim3((D + 31) / 32, (IN + 31) / 32), tb, 0, stream, params + lay.g_w3,
                       gws + g.wt3, IN, D, D, IN);
  }
  const int gemm_lds = lgcp_gemm_attrs();
  if (gemm_lds < 0) return CMCD_ERR_HIP;
  const float* kinv = tc;
  const float mu0 = 3.8812819069514780f;
  const dim3 gblock(64 * kGemmWaves);
  const int cbD = (D + 63) / 64, cbIN = (IN + 63) / 64, cb2D = (2 * D + 63) / 64;
  int* counters = reinterpret_cast<int*>(gws + g.counters);
  const float* bias1 = ws + w.bias1;    // still in the forward workspace (lgcp_uha_forward's prep)
  const LgcpKeep keep = lgcp_uha_keep(d, n, gws);
  const bool kept = keep.on;
  // r04: with the activations kept, passes of <= 20 particles run the seven products of a bridge on the no-split-K GEMM too
  const int tIN = (IN + 15) / 16, tD = D / 16, t2D = 2 * D / 16, big = 2 * kNskChunks;
  if (kept) {
    NskPackArgs pk{};
    pk.src[0] = gws + g.wt3; pk.dst[0] = gws + g.wt3p; pk.K[0] = D; pk.N[0] = IN; pk.ntile[0] = tIN;
    pk.src[1] = gws + g.wt2; pk.dst[1] = gws + g.wt2p; pk.K[1] = IN; pk.N[1] = IN; pk.ntile[1] = tIN; pk.nch[1] = big;
    pk.src[2] = gws + g.wt1; pk.dst[2] = gws + g.wt1p; pk.K[2] = IN; pk.N[2] = 2 * D; pk.ntile[2] = t2D; pk.nch[2] = big;
    pk.src[3] = tc; pk.dst[3] = gws + g.kip1; pk.K[3] = D; pk.N[3] = D; pk.ntile[3] = tD;
    const int64_t groups = (int64_t)tIN * big * 64;
    hipLaunchKernelGGL(lgcp_nsk_pack_kernel, dim3((unsigned)((groups + 255) / 256), 4), dim3(256), 0, stream, pk);
  }

  for (int64_t base = 0; base < n; base += kMP) {
    const int M = (int)((n - base) < kMP ? (n - base) : kMP);
    if (base > 0) {   // per-pass accumulators start from zero (lz / lr are initialised at point K)
      if (hipMemsetAsync(gws + g.gmu_acc, 0, sizeof(float) * 2 * ((kMP * (int64_t)D + 3) & ~3), stream) != hipSuccess) return CMCD_ERR_HIP;
    }
    LgcpUhaAdjArgs aa{};
    aa.params = params; aa.tc = tc; aa.sched = ws + sw.sched; aa.traj = traj; aa.kr = gws + g.kr; aa.hv = gws + g.hv;
    aa.snA = gws + g.fs[0].sn; aa.snB = gws + g.fs[1].sn; aa.du1 = gws + g.du1; aa.dxf = gws + g.dxf;
    aa.zinA = gws + g.fs[0].zin; aa.zinB = gws + g.fs[1].zin; aa.XIN = gws + g.XIN;
    aa.lz = gws + g.lz; aa.lr = gws + g.lr; aa.arppc = gws + g.arppc; aa.arp = gws + g.arp; aa.gb = gws + g.gb;
    aa.lrn = gws + g.lrn; aa.v = gws + g.v; aa.dO = gws + g.dO; aa.DObig = gws + g.DO; aa.gmu_acc = gws + g.gmu_acc;
    aa.glam_acc = gws + g.glam_acc; aa.geta = gws + g.geta; aa.gepsd = gws + g.gepsd; aa.sc = gws + g.sc;
    aa.lay = lay; aa.n = n; aa.base = base; aa.M = M; aa.D = D; aa.IN = IN; aa.K = K; aa.omega = omega;

    const bool nskb = kept && M <= 20;
    const bool merged = M > 16;
    float* const dOp = gws + g.bops, *vp = dOp + kNskOperand, *da2p = vp + kNskOperand, *da1p = da2p + 2 * kNskOperand;
    if (nskb) {
      if (hipMemsetAsync(dOp, 0, sizeof(float) * 6 * kNskOperand, stream) != hipSuccess) return CMCD_ERR_HIP;
      aa.dOp = dOp; aa.vp = vp; aa.bslab = 1;
    }
    auto kr_at = [&](int e) {   // K^-1 (z_e - mu0)
      GemmArgs gm{};
      gm.M = M; gm.Kdim = D; gm.counters = counters;
      gm.seg[0] = GemmSeg{traj + ((int64_t)e * n + base) * D, kinv, gws + g.kr, D, D, D, D, mu0};
      gm.nblk0 = cbD;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), gblock, gemm_lds, stream, gm);
    };
    // back-propagation of one network evaluation (buffer set f, table row `row`, time index i) from the cotangent in dO:
    // leaves d u1 (residual path) and the d a1 W1[:2D]^T slabs
    auto net_backward = [&](const LgcpUhaFwdSet& f, int64_t row, int i) {
      if (nskb) {
        NskArgs na{};
        na.M = M; na.D = 2 * D; na.IN = IN; na.nch_out = big;
        // d u2 = d o W3^T (one round of the contraction), d a2 = d u2 sigmoid(pre2)
        na.seg[0] = NskSeg{dOp, gws + g.wt3p, IN, 0, 0.f}; na.nt0 = tIN;
        na.bPre = keep.pre2 + row * IN; na.bDuPrev = nullptr; na.bDu = gws + g.du2; na.outA = da2p;
        na.bDaBig = gws + g.DA2 + row * IN; na.bSumA = gws + g.gb2; na.bSumU = nullptr;
        if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<3, true>), dim3((unsigned)tIN, 1), gblock, 0, stream, na);
        else hipLaunchKernelGGL((lgcp_nsk_kernel<3, false>), dim3((unsigned)tIN, 1), gblock, 0, stream, na);
        // d u1 = d u2 + d a2 W2^T (two rounds), d a1 = d u1 sigmoid(pre1)
        na.seg[0] = NskSeg{da2p, gws + g.wt2p, IN, 0, 0.f}; na.seg[0].nch = big;
        na.bPre = keep.pre1 + row * IN; na.bDuPrev = gws + g.du2; na.bDu = gws + g.du1; na.outA = da1p;
        na.bDaBig = gws + g.DA1 + row * IN; na.bSumA = gws + g.S + (int64_t)i * IN; na.bSumU = gws + g.S2 + (int64_t)i * IN;
        if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<3, true, 2>), dim3((unsigned)tIN, 1), gblock, 0, stream, na);
        else hipLaunchKernelGGL((lgcp_nsk_kernel<3, false, 2>), dim3((unsigned)tIN, 1), gblock, 0, stream, na);
        // d a1 W1[:2D]^T: plain, row-major
        NskArgs nb{};
        nb.M = M; nb.D = 2 * D; nb.IN = IN;
        nb.seg[0] = NskSeg{da1p, gws + g.wt1p, 2 * D, NSK_OUT, 0.f}; nb.seg[0].nch = big; nb.nt0 = t2D; nb.outN = gws + g.dxf;
        if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<0, true, 2>), dim3((unsigned)t2D, 1), gblock, 0, stream, nb);
        else hipLaunchKernelGGL((lgcp_nsk_kernel<0, false, 2>), dim3((unsigned)t2D, 1), gblock, 0, stream, nb);
        return;
      }
      GemmArgs gm{};
      gm.M = M; gm.counters = counters; gm.epi_seg = -1;
      gm.Kdim = D;
      gm.seg[0] = GemmSeg{gws + g.dO, gws + g.wt3, gws + g.du2s, IN, D, IN, IN};
      gm.nblk0 = cbIN;
      LgcpActbArgs& ab = gm.actb;
      ab.pre = kept ? keep.pre2 + row * IN : gws + f.pre2; ab.du_prev = nullptr; ab.u_src = kept ? keep.u2 + row * IN : gws + f.u2;
      ab.du_out = gws + g.du2; ab.da_out = gws + g.da2; ab.da_big = gws + g.DA2; ab.u_big = kept ? nullptr : gws + g.U2;
      ab.gb = gws + g.gb2; ab.row0 = row; ab.IN = IN; ab.mode = 2;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACTB>, dim3(cbIN, kSplit), gblock, gemm_lds, stream, gm);
      gm.Kdim = IN;
      gm.seg[0] = GemmSeg{gws + g.da2, gws + g.wt2, gws + g.ts, IN, IN, IN, IN};
      ab.pre = kept ? keep.pre1 + row * IN : gws + f.pre1; ab.du_prev = gws + g.du2; ab.u_src = kept ? keep.u1 + row * IN : gws + f.u1;
      ab.du_out = gws + g.du1; ab.da_out = gws + g.da1; ab.da_big = gws + g.DA1; ab.u_big = kept ? nullptr : gws + g.U1;
      ab.S = gws + g.S + (int64_t)i * IN; ab.S2 = gws + g.S2 + (int64_t)i * IN; ab.mode = 1;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_ACTB>, dim3(cbIN, kSplit), gblock, gemm_lds, stream, gm);
      gm.seg[0] = GemmSeg{gws + g.da1, gws + g.wt1, gws + g.dxf, 2 * D, IN, 2 * D, 2 * D};
      gm.nblk0 = cb2D;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cb2D, kSplit), gblock, gemm_lds, stream, gm);
    };
    auto hv_product = [&]() {
      if (nskb) {
        NskArgs nb{};
        nb.M = M; nb.D = 2 * D; nb.IN = IN;
        nb.seg[0] = NskSeg{vp, gws + g.kip1, D, NSK_KR, 0.f}; nb.nt0 = tD; nb.krOutN = gws + g.hv;
        if (merged) hipLaunchKernelGGL((lgcp_nsk_kernel<0, true>), dim3((unsigned)tD, 1), gblock, 0, stream, nb);
        else hipLaunchKernelGGL((lgcp_nsk_kernel<0, false>), dim3((unsigned)tD, 1), gblock, 0, stream, nb);
        return;
      }
      GemmArgs gm{};
      gm.M = M; gm.Kdim = D; gm.counters = counters;
      gm.seg[0] = GemmSeg{gws + g.v, kinv, gws + g.hv, D, D, D, D};
      gm.nblk0 = cbD;
      hipLaunchKernelGGL(lgcp_gemm_kernel<EPI_NONE>, dim3(cbD, kSplit), gblock, gemm_lds, stream, gm);
    };

    if (kept) {
      aa.nslab = 1;
      hipLaunchKernelGGL(lgcp_uha_xin_kernel, dim3((M * D + 255) / 256, K), dim3(256), 0, stream, aa);
    } else {
      kr_at(K);
    }
    for (int e = K; e >= 1; --e) {
      const int i = e - 1;
      aa.e = e;
      const int64_t rowA = (int64_t)(2 * i) * n + base, rowB = rowA + n;
      if (kept) {   // the forward's sums of this bridge: nothing to recompute (r04)
        aa.kr = keep.kr + ((int64_t)e * n + base) * D;
        aa.snA = keep.sn + rowA * D; aa.snB = keep.sn + rowB * D;
      }
      hipLaunchKernelGGL(lgcp_uha_adj_point_kernel, dim3(M), dim3(64 * kUhaAdjW), 0, stream, aa);
      hv_product();
      if (!kept) hipLaunchKernelGGL(lgcp_uha_gather_kernel, dim3((M * D + 255) / 256), dim3(256), 0, stream, aa);
      const LgcpUhaFwdSet& fa = g.fs[0];
      const LgcpUhaFwdSet& fb = g.fs[1];
      if (!kept) {
        lgcp_uha_net(d, lay, params, kinv, M, i, gws + fa.zin, bias1, gws + fa.slab1, gws + fa.pre1, gws + fa.u1, gws + fa.slab2,
                     gws + fa.pre2, gws + fa.u2, gws + fa.sn, nullptr, nullptr, counters, gemm_lds, stream);
        lgcp_uha_net(d, lay, params, kinv, M, i, gws + fb.zin, bias1, gws + fb.slab1, gws + fb.pre1, gws + fb.u1, gws + fb.slab2,
                     gws + fb.pre2, gws + fb.u2, gws + fb.sn, nullptr, nullptr, counters, gemm_lds, stream);
      }
      hipLaunchKernelGGL(lgcp_uha_adj_b_kernel, dim3(M), dim3(64 * kUhaAdjW), 0, stream, aa);
      net_backward(fb, rowB, i);
      hipLaunchKernelGGL(lgcp_uha_adj_a_kernel, dim3(M), dim3(64 * kUhaAdjW), 0, stream, aa);
      net_backward(fa, rowA, i);
      hipLaunchKernelGGL(lgcp_uha_adj_fin_kernel, dim3(M), dim3(64 * kUhaAdjW), 0, stream, aa);
      if (!kept) kr_at(i);
    }
    aa.e = 0;
    if (kept) aa.kr = keep.kr + base * D;
    hipLaunchKernelGGL(lgcp_uha_adj_point_kernel, dim3(M), dim3(64 * kUhaAdjW), 0, stream, aa);
    hv_product();
    hipLaunchKernelGGL(lgcp_uha_adj_z0_kernel, dim3(M), dim3(256), 0, stream, aa);
    hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, gws + g.gmu_acc, (int64_t)M, D, D,
                       grad + lay.vd_mean, 1.0f, 1);
    hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, gws + g.glam_acc, (int64_t)M, D, D,
                       grad + lay.vd_logdiag, 1.0f, 1);
  }
  // ---- deferred parameter contractions over all 2 K n rows
  const int64_t R = (int64_t)2 * K * n;
  hipLaunchKernelGGL(lgcp_tn_gemm_kernel, dim3((IN + 127) / 128, (IN + 127) / 128, kTnSplit), dim3(256), 0, stream, gws + g.U1, gws + g.DA2,
                     grad + lay.g_w2, R, IN, IN, IN, IN, IN);                                   // dW2 = U1^T dA2
  hipLaunchKernelGGL(lgcp_tn_gemm_kernel, dim3((D + 127) / 128, (IN + 127) / 128, kTnSplit), dim3(256), 0, stream, gws + g.U2, gws + g.DO,
                     grad + lay.g_w3, R, IN, D, IN, D, D);                                      // dW3 = U2^T dO
  hipLaunchKernelGGL(lgcp_tn_gemm_kernel, dim3((IN + 127) / 128, (2 * D + 127) / 128, kTnSplit), dim3(256), 0, stream, gws + g.XIN, gws + g.DA1,
                     grad + lay.g_w1, R, 2 * D, IN, 2 * D, IN, IN);                             // dW1[:2D] = [z; rho]^T dA1
  hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((D + 255) / 256), dim3(256), 0, stream, gws + g.DO, R, D, D, grad + lay.g_b3, 1.0f, 0);
  hipLaunchKernelGGL(lgcp_colsum_kernel, dim3((IN + 255) / 256), dim3(256), 0, stream, gws + g.gb2, (int64_t)1, IN, IN, grad + lay.g_b2, 1.0f, 0);
  hipLaunchKernelGGL(lgcp_uha_scal_reduce_kernel, dim3(K + 1), dim3(256), 0, stream, gws + g.sc, n, K, gws + g.gbeta, gws + g.geps,
                     grad + lay.gamma, grad + lay.g_factor);
  // schedule (cos^2), embedding table, W1[2d:], b1 from the S / S2 / beta / eps tables
  int rc = launch_net_tails(d, 2 * D, CMCD_EPS_COS_SQ, lay, sw, params, gws, g.S, g.S2, g.gbeta, g.geps, IN, nullptr, grad, stream);
  if (rc != CMCD_OK) return rc;
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

}  // namespace cmcd
