// grad_kernel — VarGrad gradient of the CMCD bound (the build's `compute_log_var_grad`):
//   d/d params_flat of  sum_n omega_n * w_n   with z detached at every step boundary, exactly the
// autodiff graph of `MCD_CAIS_var_sn` (/root/reference/src/mcd_cais_var.py:59,79: z = stop_gradient(z),
// z_new = stop_gradient(z_new)) under jax.grad(compute_bound_var, 1) (/root/reference/src/main.py:161-176).
// The caller supplies omega_n = d value / d w_n (for the variance: -(2/N)(l_n - mean l)).
//
// Because z is detached, the gradient is LOCAL per bridge evaluation e = 0..K:
//   cotangent of s(z_e, e):   c_e = omega/2 [ (z_{e-1} - bk_{e-1}) [e>=1] + (z_{e+1} - fk_e) [e<=K-1] ]
//   d w / d beta_i = 1/2 [ (z_i - bk_i).(gp' - gq') - (z_{i+1} - fk_i).(gp - gq) ]
//   d w / d eps_i  = (|z_i-bk_i|^2 - |z_{i+1}-fk_i|^2)/(4 eps^2) + [(z_i-bk_i).(s'-ub) + (z_{i+1}-fk_i).(uf+s)]/(2 eps)
//   d w / d gq     = -(1-beta)/2 (z_{i+1}-fk_i) at z_i,  +(1-beta)/2 (z_i-bk_i) at z_{i+1};  d w0 / d logdiag = 1
// so the kernel re-runs the trajectory (same seeds => same z) and back-propagates through the MLP at
// every evaluation while the activations are still in registers.
//
// Mapping: 4-wave workgroups, wave q runs forward + backward of its OWN 16-particle tile.  All
// parameter-gradient contractions over particles are MFMA outer products C += X^T Y with X, Y staged
// per tile in LDS as [feature][particle] (XOR-swizzled), e.g. dW2 += u1^T da2.  The dW2 / dW3
// accumulator row-tiles are distributed over the 4 waves (wave w owns row tile w of all four tiles'
// products) so that nothing but 16-32 accumulator registers per wave persists across evaluations.
// W2^T da2 re-uses the forward A-fragment copy in LDS through a gathered read.  Per-evaluation
// quantities (bias-table, beta, eps gradients) go to small global tables by float atomics; the
// particle-independent tails (time coder / embedding / schedules) are small kernels at the end.
//
// BPTT = true is the reparameterised gradient of `MCD_CAIS_sn` (jax.grad(compute_bound, 1), /root/reference/src/
// main.py:174-176 over mcd_cais.py:46-89, no stop_gradient): the forward kernel stores z_0..z_K, this kernel
// walks the evaluations in REVERSE order carrying lambda_e = d L / d z_e (L = sum_n omega loss_n):
//   g_i      = d L / d bk_i = -omega (z_i - bk_i) / (2 eps_i)
//   a_s(e)   = -eps_e lambda_{e+1} [e<K] + eps_{e-1} g_{e-1} [e>0]                 cotangent of s(z_e, e)
//   a_gp(e)  = eps_e beta_e lambda_{e+1} [e<K] + eps_{e-1} beta_{e-1} g_{e-1} [e>0]  (a_gq: 1 - beta)
//   lambda_e = lambda_{e+1} - g_e [e<K] + g_{e-1} [e>0] + J_s(z_e)^T a_s + H_p(z_e) (clipmask a_gp) + H_q a_gq
//              - omega grad log p(z_K) [e=K] + omega grad log q(z_0) [e=0]
//   d/d beta_i = eps_i [(gp_i - gq_i).lambda_{i+1} + (gp_{i+1} - gq_{i+1}).g_i]
//   d/d eps_i  = (-uf_i - s_i + n_i / sigma_i).lambda_{i+1} + (-ub_i + s_{i+1}).g_i - omega |z_i - bk_i|^2 / (4 eps_i^2)
// (the two log-normalisers cancel and log N(z_{i+1}; fk_i, sigma_i) = -|n_i|^2/2 - ... carries no gradient).
// The parameter contractions, slabs and tails are shared with the local gradient.
//
// Widths <= 64 keep W2 / W2^T fragments in LDS with 4 tiles per workgroup; the 132-wide net (T = 9) runs 3
// tiles per workgroup (3 accumulator row tiles per wave) and streams both fragment copies from L2.
#include <hip/hip_runtime.h>
#include <atomic>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

// cmcd_bptt.hip: Jacobian rows of every (tile, evaluation) + the per-particle lambda recursion
int bptt_jac_scan_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, int64_t n, int64_t nitems,
                         const float* params, const float* ws_fwd, const float* traj, float* jac, float* lam,
                         float omega_scalar, float* zero_a, int64_t n_a, float* zero_b, int64_t n_b, void* stream);

struct GradArgs {
  const int32_t* seeds;
  const float* params;
  const float* ws;           // forward workspace (tables + packed weights of cmcd_bound_forward's prep)
  const float* omega;        // [n], or nullptr: omega_scalar for every particle
  const float* traj;         // BPTT / ITEM: [K+1][n][D] trajectory stored by the forward kernel
  const float* lam;          // BPTT + ITEM: [K+1][n][D] adjoints lambda_e from the scan
  float omega_scalar;
  int64_t nitems;            // ITEM: ntiles * (K + 1) independent (tile, evaluation) work items
  float* gtab;               // gradient tables (zeroed): S[(K+1)][HP], S2[(K+1)][HP], gbeta[K], geps[K], gvd[2D], gfac[1]
  float* slabs;              // per-workgroup slabs
  cmcd_layout lay;
  WsLayout w;
  int64_t n;
  int32_t K, var_mode, grad_clipping, nquads;
  int32_t ula;               // 0: CAIS; 2: MCD_ULA_sn — no network in the forward kernel, s(z_{i+1}, i) in the backward one
  int64_t o_S, o_S2, o_gbeta, o_geps, o_gvd, o_gfac;   // offsets inside gtab
  int64_t slab_stride;                                   // floats per workgroup slab
  // fixed-order accumulation (r04): when `det` is set, every (tile, evaluation) work unit STORES its contribution to
  // the bias-table rows and to the schedule gradients in a slot of its own — each unit is visited by exactly one wave
  // in every mode — and grad_det_reduce_kernel sums the slots over tiles in a fixed order.  With float atomics on the
  // shared tables the order of the adds changed from run to run and a training seed did not reproduce.
  // per tile: S [(K+1)][HP] | S2 [(K+1)][HP] (geffner) | [(K+1)][4] = {d beta_e, d eps_e, d beta_{e-1}, d eps_{e-1}}
  float* det;                // nullptr: the atomics (batches whose slot table would pass kDetCapFloats)
  int64_t det_tile_stride, det_tiles, det_obe;
};

// element (feature f, particle p) of a staged [features][16] array
__device__ __forceinline__ int sw(int f, int p) { return f * 16 + (p ^ (f & 15)); }

// NW waves per workgroup (one tile each); WGLOBAL: W2 / W2^T fragments streamed from L2 instead of LDS
// ITEM: with the trajectory stored, every (tile, evaluation) pair is independent (the local gradient is local
// by construction; the reparameterised one once lambda_e is known from the scan), so a wave takes work
// items instead of whole chains: small batches then fill the chip instead of 1/8 of it.
template <int TARGET, int ARCH, int D, int T, int NW, bool WGLOBAL, bool BPTT, bool ITEM>
// (the work-item instances that stream W2 are launched two workgroups per CU: the register budget must allow it as the LDS does)
__global__ __launch_bounds__(64 * NW, (WGLOBAL && T <= 4) ? 2 : 1) void grad_kernel(GradArgs a) {
  constexpr int HP = 16 * T;
  constexpr int Hh = (D + 1) / 2;
  constexpr bool GEF = ARCH == CMCD_ARCH_GEFFNER;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int WLDS = WGLOBAL ? 0 : 2 * HP * HP;
  float* lds_w2 = lds;                    // HP*HP   forward A fragments   (absent when WGLOBAL)
  float* lds_w2t = lds + HP * HP;         // HP*HP   A fragments of W2^T
  float* lds_w1z = lds + WLDS;            // D*HP
  float* lds_w3t = lds_w1z + D * HP;      // D*HP
  float* lds_b2 = lds_w3t + D * HP;       // HP
  float* lds_b3 = lds_b2 + HP;            // 16
  float* lds_tgt = lds_b3 + 16;           // tgt_floats
  // WSHARE (the 132-wide net, r03, MEASURED AND REJECTED — kept compiled out as the record of the experiment): the four
  // waves of a workgroup walk the SAME W2 / W2^T fragment rows (each for its own work item), so a row (T fragments x 1 KB)
  // can be fetched from L2 ONCE per workgroup — wave q fetches fragments q, q + 4, q + 8 — and handed round through a
  // double-buffered LDS row: a quarter of the L2 traffic (r02: 166 KB per work item and wave).  On MI355X the kernel went
  // from 1.099 to 1.587 ms at 2000 particles (profiles/r03_wshare_rejected_kernel_stats.csv): the 20 extra workgroup
  // barriers per work item put the four waves in lockstep, so no wave's matrix chain covers another's load latency any
  // more — the L2 stream was not the bound, the overlap between the waves was the asset.
  constexpr bool WSHARE = false;   // measured and rejected (r03): see the note below
  constexpr int WSH_FLOATS = WSHARE ? 2 * T * 256 : 0;
  float* wsh = lds_tgt + a.w.tgt_floats;             // [2][T][64 lanes][4]
  float* stage = wsh + WSH_FLOATS;
  // per tile: u1T, u2T, da2T [HP][16] (read by the other waves of the workgroup); da1T, du1T (read by this wave only);
  // z1, doT [16][16]; then accZ1 [D][HP], accB2 [HP] (wave-private sums over evaluations).
  // TILE_LOCAL (the 132-wide net, r02): da1 / du1 are consumed by this wave's own products right where they are formed,
  // 16 hidden units at a time through two 1 KB buffers, instead of being staged whole (2 x 9 KB per wave): 50 -> 33.5 KB
  // of staging per wave, so FOUR waves fit a CU's LDS instead of three (the fourth SIMD used to idle).
  // (r04) the work-item instances of the narrow 2-d nets take the same staging AND stream W2 from L2: 69 KB per workgroup
  // instead of 125, so that two share a CU (CMCD_GRAD_SMALL; the 2nd-order sweep gained 1.3x from exactly that)
  constexpr bool TILE_LOCAL = T > 4 || WGLOBAL;
  constexpr int OWNBUF = TILE_LOCAL ? 256 : HP * 16;
  constexpr int OFF_DOT = 3 * HP * 16 + 2 * OWNBUF + 256;
  constexpr int STG = 3 * HP * 16 + 2 * OWNBUF + 512 + (D + 1) * HP;
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.ws + a.w.w1z);
    f32x4* dst = reinterpret_cast<f32x4*>(lds_w1z);
    for (int i = threadIdx.x; i < D * HP / 4; i += blockDim.x) dst[i] = src[i];
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
  float* u2T = u1T + HP * 16;
  float* da2T = u2T + HP * 16;
  float* da1T = da2T + HP * 16;
  float* du1T = da1T + OWNBUF;
  float* z1T = du1T + OWNBUF;             // rows 0..D-1 = z_j, row D = 1, rest 0
  float* doT = z1T + 256;                 // rows 0..D-1 = d o_j, rest 0
  float* accZ1 = doT + 256;               // [D][HP]  dW1[j][n], j < D
  float* accB2 = accZ1 + D * HP;          // [HP]     db2[n]
  for (int i = lane; i < (D + 1) * HP; i += 64) accZ1[i] = 0.f;
  const int K = a.K;
  const float factor = lds_b3[15];
  // staged [feature][particle] tiles: element (f, p) sits at f*16 + (p ^ (f & 15)).  With f = 16 t + 4 g + r
  // (writes) or f = 16 t + c (reads) the swizzle term does not depend on t, so every access is one of
  // these lane-dependent bases plus a compile-time offset of t*256 floats (an LDS immediate).
  int wb[4], rb[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    wb[r] = (4 * g + r) * 16 + (c ^ (4 * g + r));   // write base of register r:  feature 16 t + 4 g + r, particle c
    rb[r] = c * 16 + ((4 * r + g) ^ c);             // read base of k-step r:     feature 16 t + c, particle 4 r + g
  }
  const float* w2f = WGLOBAL ? a.ws + a.w.w2 : lds_w2;
  const float* w2tf = WGLOBAL ? a.ws + a.w.w2t : lds_w2t;

  float qmean[D], qstd[D], qiv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qmean[j] = a.params[a.lay.vd_mean + j];
    qstd[j] = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (qstd[j] * qstd[j]);
  }
  const float clipv = a.var_mode ? 1e2f : 1e3f;
  const bool clip_p = a.grad_clipping != 0;
  const bool clip_q = clip_p && a.var_mode;
  const float* bias1 = a.ws + a.w.bias1;
  const float* utab = a.ws + a.w.utab;
  float* gS = a.gtab + a.o_S;
  float* gS2 = a.gtab + a.o_S2;

  // persistent accumulators (C layout: lane (g,c), reg r <-> row 16*tile + 4g + r, col 16*tile' + c)
  constexpr int OWN = (T + NW - 1) / NW;  // dW2 / dW3 row tiles owned by this wave: ti = wv + NW*k < T
  f32x4 gW2[OWN][T], gW3[OWN];
  f32x4 gB3;                              // row D: db3[j]   (A = z1 against do)
#pragma unroll
  for (int k = 0; k < OWN; ++k) {
    gW3[k] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < T; ++t) gW2[k][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  gB3 = f32x4{0.f, 0.f, 0.f, 0.f};
  float gfac = 0.f, gmu[D], glam[D];
#pragma unroll
  for (int j = 0; j < D; ++j) { gmu[j] = 0.f; glam[j] = 0.f; }

  const int64_t n_outer = ITEM ? (a.nitems + NW - 1) / NW : a.nquads;
  for (int64_t quad = blockIdx.x; quad < n_outer; quad += gridDim.x) {
    int64_t tile = quad * NW + wv;
    int e_item = 0;
    bool live = true;
    if (ITEM) {
      int64_t item = quad * NW + wv;
      live = item < a.nitems;
      if (!live) item = a.nitems - 1;
      tile = item / (K + 1);
      e_item = (int)(item - tile * (K + 1));
    }
    const int64_t p = tile * 16 + c;
    const bool valid = live && p < a.n;
    // padding waves (a clamped work item, a tile past the batch) carry zeros: they must not overwrite a real slot
    float* dslot = (a.det && live && tile < a.det_tiles) ? a.det + tile * a.det_tile_stride : nullptr;
    const int32_t seed = a.seeds[valid ? p : a.n - 1];
    const float om = valid ? (a.omega ? a.omega[p] : a.omega_scalar) : 0.f;
    const int64_t pc = valid ? p : a.n - 1;

    // ---- key chain + z0 (identical to traj_kernel)
    const int gb = g & 1;
    uint32_t x0, x1, k0 = 0u, k1 = (uint32_t)seed;
    float z[D], zp[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { z[j] = 0.f; zp[j] = 0.f; }
    if (!BPTT && !ITEM) {
      x0 = gb; x1 = 2 + gb;
      threefry2x32(k0, k1, x0, x1);
      uint32_t a0, a1, b0, b1;
      rows01(x0, a0, a1);
      rows01(x1, b0, b1);
      float nz[2 * Hh];
#pragma unroll
      for (int j0 = 0; j0 < Hh; j0 += 4) {
        const int j = j0 + g;
        uint32_t y0 = j, y1 = (Hh + j < D) ? Hh + j : 0;
        threefry2x32(a0, a1, y0, y1);
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
#pragma unroll
      for (int j = 0; j < D; ++j) {
        z[j] = qstd[j] * nz[j] + qmean[j];
        zp[j] = 0.f;
        glam[j] += om;   // d(-log q(z0))/d logdiag_j = +1 under the reparameterisation; mean: 0
      }
      x0 = gb; x1 = 2 + gb;
      threefry2x32(b0, b1, x0, x1);
      uint32_t c0, c1;
      rows01(x0, c0, c1);
      x0 = gb; x1 = 2 + gb;
      threefry2x32(c0, c1, x0, x1);
      rows01(x1, k0, k1);
    }
    float pbeta = 0.f, peps = 0.f;
    float fkd[D];  // z_{i+1} - fk_i of the step opened at the previous evaluation
    float puf[D], psn[D], pgd[D];  // uf, s, (gp - gq) of that step's forward side
#pragma unroll
    for (int j = 0; j < D; ++j) { fkd[j] = 0.f; puf[j] = 0.f; psn[j] = 0.f; pgd[j] = 0.f; }
    float pend_beta = 0.f, pend_eps = 0.f;   // forward-side contributions of step i, completed at e = i+1
    float lamn[D], gE[D], znext[D];          // BPTT: lambda_{e+1}, g_e, z_{e+1}
#pragma unroll
    for (int j = 0; j < D; ++j) { lamn[j] = 0.f; gE[j] = 0.f; znext[j] = 0.f; }

    const int n_it = ITEM ? 1 : K + 1;
    for (int it = 0; it < n_it; ++it) {
      const int e = ITEM ? e_item : (BPTT ? K - it : it);
      // LDS contents are loop-invariant, so LICM would hoist every w1z / w3t / b2 read out of the work loop into
      // ~180 registers (132-wide net) and spill them to scratch: a compiler-level fence keeps them in LDS
      asm volatile("" ::: "memory");
      if (BPTT || ITEM) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          z[j] = a.traj[((int64_t)e * a.n + pc) * D + j];
          zp[j] = e > 0 ? a.traj[((int64_t)(e - 1) * a.n + pc) * D + j] : 0.f;
          if (ITEM) {
            znext[j] = e < K ? a.traj[((int64_t)(e + 1) * a.n + pc) * D + j] : 0.f;
            if (BPTT) lamn[j] = (valid && e < K) ? a.lam[((int64_t)(e + 1) * a.n + pc) * D + j] : 0.f;   // padding lanes carry no adjoint
            else if (e == 0) glam[j] += om;   // d(-log q(z0))/d logdiag_j, once per particle
          }
        }
        if (ITEM && e > 0) { pbeta = a.ws[a.w.beta + e - 1]; peps = a.ws[a.w.eps + e - 1]; }
      }
      // ---------------------------------------------------------------- forward (keeps pre-activations)
      // MCD_ULA_sn (mcd_over_orig.py:40-46): evaluation e >= 1 serves only the backward kernel of step e-1, index e-1
      const int64_t erow = a.ula ? (e > 0 ? e - 1 : 0) : e;
      const float fsn = a.ula ? 0.f : 1.f;
      const float* brow = bias1 + erow * HP;
      // wide nets (T > 4): the first pre-activation is not kept, it is rebuilt from the bias row and z where the
      // backward pass needs it (2 D FMAs per value against 4 T registers held across both MFMA phases)
      constexpr bool KEEP_A1 = T <= 4;
      f32x4 a1[KEEP_A1 ? T : 1], u1[T], a2[T];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f32x4 pre = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
        for (int j = 0; j < D; ++j) pre += z[j] * *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
        // a1 / a2 hold the activation DERIVATIVES from here on (one polynomial + exponential for both)
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
            float dv = 0.f;
            u1[t][r] = u[r] + (KEEP_A1 ? softplus_both(pre[r], dv) : softplus(pre[r]));
            dact[r] = dv;
          }
        }
        if (KEEP_A1) a1[KEEP_A1 ? t : 0] = dact;
        // stage u1^T now (every wave finished reading the previous evaluation's tiles at the closing barrier)
#pragma unroll
        for (int r = 0; r < 4; ++r) u1T[wb[r] + 256 * t] = u1[t][r];
      }
#pragma unroll
      for (int t = 0; t < T; ++t) a2[t] = *reinterpret_cast<const f32x4*>(lds_b2 + 16 * t + 4 * g);
      // fragments (LDS, or L2 when WGLOBAL): the address arithmetic stays inside the loop (an opaque lane offset), or
      // the T*T loop-invariant 64-bit addresses are hoisted into registers and spilled (324 dwords for T = 9); the row
      // of fragments for ti+1 is requested before the MFMAs of ti so that the L2 latency hides behind them
      if constexpr (WSHARE) {
        constexpr int SHR = (T + NW - 1) / NW;
        f32x4 nxt[SHR];
        auto fetch = [&](int row) {
          int lofs = lane * 4;
          asm volatile("" : "+v"(lofs));
#pragma unroll
          for (int k = 0; k < SHR; ++k) {
            const int to = wv + NW * k;
            if (to < T) nxt[k] = *reinterpret_cast<const f32x4*>(w2f + (row * T + to) * 256 + lofs);
          }
        };
        auto publish = [&](int bufi) {
#pragma unroll
          for (int k = 0; k < SHR; ++k) {
            const int to = wv + NW * k;
            if (to < T) *reinterpret_cast<f32x4*>(wsh + (bufi * T + to) * 256 + lane * 4) = nxt[k];
          }
        };
        fetch(0);
        publish(0);
        __syncthreads();
#pragma unroll
        for (int ti = 0; ti < T; ++ti) {
          asm volatile("" ::: "memory");
          if (ti + 1 < T) fetch(ti + 1);            // the next row's L2 requests ride behind this row's matrix instructions
          f32x4 afc[T];
#pragma unroll
          for (int to = 0; to < T; ++to) afc[to] = *reinterpret_cast<const f32x4*>(wsh + ((ti & 1) * T + to) * 256 + lane * 4);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int to = 0; to < T; ++to) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a2[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(afc[to][r], u1[ti][r], a2[to], 0, 0, 0);
          }
          if (ti + 1 < T) publish((ti + 1) & 1);
          __syncthreads();                          // row ti consumed by every wave, row ti + 1 visible
        }
      } else {
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
        // pin the requests above the MFMAs: left alone, the scheduler sinks each one down to its first use to shorten the
        // live range, and every pair of fragments then waits out a full L2 round trip (s_waitcnt vmcnt(0) per 8 MFMAs)
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
            u2T[wb[r] + 256 * t] = u2t[r];
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
      float gp[D], gq[D], logp;
      bool gq_live[D];
      constexpr int HN = Target<TARGET, D>::HN;
      float hs[HN], gpraw[D];
      if (BPTT && !ITEM) Target<TARGET, D>::eval_hess(z, g, lds_tgt, logp, gp, hs);
      else Target<TARGET, D>::eval(z, g, lds_tgt, logp, gp);
#pragma unroll
      for (int j = 0; j < D; ++j) {
        gq[j] = -(z[j] - qmean[j]) * qiv[j];
        gq_live[j] = true;
        gpraw[j] = gp[j];
        if (clip_p) gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
        if (clip_q) {
          gq_live[j] = fabsf(gq[j]) < clipv;
          gq[j] = fminf(fmaxf(gq[j], -clipv), clipv);
        }
      }

      // ---------------------------------------------------------------- cotangents and scalar gradients
      float cot[D];   // local gradient: cotangent of s / omega;  BPTT: a_s (omega included)
      float lam[D];
#pragma unroll
      for (int j = 0; j < D; ++j) { cot[j] = 0.f; lam[j] = 0.f; }
      if (BPTT) {
        float a_gp[D], a_gq[D], gprev[D];
#pragma unroll
        for (int j = 0; j < D; ++j) { a_gp[j] = 0.f; a_gq[j] = 0.f; gprev[j] = 0.f; }
        float npb = 0.f, npe = 0.f;
        if (e > 0) {  // backward kernel of step i = e-1, whose mean is built from this evaluation
          const float pb = a.ws[a.w.beta + e - 1], pe = a.ws[a.w.eps + e - 1];
          const float inv2e = 0.5f / pe;
          float sb = 0.f, se = 0.f, r2 = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const float ub = -1.0f * (pb * gp[j] + (1.0f - pb) * gq[j]);
            // z_{e-1} - (z - pe ub + pe s) with the O(1) states subtracted first (exact in float32: one step apart)
            const float r = (zp[j] - z[j]) + pe * (ub - sn[j]);
            gprev[j] = -om * r * inv2e;
            cot[j] += pe * gprev[j];
            a_gp[j] += pe * pb * gprev[j];
            a_gq[j] += pe * (1.0f - pb) * gprev[j];
            lam[j] += gprev[j];
            sb += (gp[j] - gq[j]) * gprev[j];
            se += (sn[j] - ub) * gprev[j];
            r2 += r * r;
          }
          npb = pe * sb;
          npe = se - om * r2 * inv2e * inv2e;
        }
        if (e < K) {  // forward kernel of step e, which produced z_{e+1}
          const float be = a.ws[a.w.beta + e], ee = a.ws[a.w.eps + e];
          const float inv2e = 0.5f / ee;
          float sb = 0.f, se = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const float uf = -1.0f * (be * gp[j] + (1.0f - be) * gq[j]);
            const float nsig = ((znext[j] - z[j]) + ee * (uf + fsn * sn[j])) * inv2e;        // n_e / sigma_e
            cot[j] -= fsn * ee * lamn[j];
            a_gp[j] += ee * be * lamn[j];
            a_gq[j] += ee * (1.0f - be) * lamn[j];
            lam[j] += lamn[j] - gE[j];
            sb += (gp[j] - gq[j]) * lamn[j];
            se += (nsig - uf - fsn * sn[j]) * lamn[j];
          }
          const float tb = row_sum16(pend_beta + ee * sb), te = row_sum16(pend_eps + se);
          if (lane == 0) {
            if (a.det) {
              if (dslot) { dslot[a.det_obe + 4 * e + 0] = tb; dslot[a.det_obe + 4 * e + 1] = te; }
            } else {
              atomicAdd(a.gtab + a.o_gbeta + e, tb);
              atomicAdd(a.gtab + a.o_geps + e, te);
            }
          }
        }
        pend_beta = npb; pend_eps = npe;
        if (ITEM && e > 0) {  // no carry between work items: the backward-kernel part of step e-1 goes out now
          const float tb = row_sum16(npb), te = row_sum16(npe);
          if (lane == 0) {
            if (a.det) {
              if (dslot) { dslot[a.det_obe + 4 * e + 2] = tb; dslot[a.det_obe + 4 * e + 3] = te; }
            } else {
              atomicAdd(a.gtab + a.o_gbeta + (e - 1), tb);
              atomicAdd(a.gtab + a.o_geps + (e - 1), te);
            }
          }
        }
        float v[D], hv[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
          if (e == K) lam[j] -= om * gpraw[j];                 // - log p(z_K)
          if (e == 0) lam[j] += om * gq[j];                    // + log q(z_0)
          gmu[j] += a_gq[j] * qiv[j];
          glam[j] += a_gq[j] * (-2.0f * gq[j]);
          lam[j] -= a_gq[j] * qiv[j];                          // H_q = -diag(1 / std^2)
          v[j] = (!clip_p || fabsf(gpraw[j]) < clipv) ? a_gp[j] : 0.f;
          gE[j] = gprev[j];
        }
        if (!ITEM) {
          Target<TARGET, D>::hvp(hs, z, v, hv);
#pragma unroll
          for (int j = 0; j < D; ++j) lam[j] += hv[j];
        }
      }
      if (!BPTT && e > 0) {  // backward kernel of step i = e-1 at z' = z (mcd_cais_var.py:81-89)
        float sb = 0.f, se = 0.f, dn2 = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          const float ub = -1.0f * (pbeta * gp[j] + (1.0f - pbeta) * gq[j]);
          const float db = (zp[j] - z[j]) + peps * (ub - sn[j]);
          cot[j] += 0.5f * db;
          sb += db * (gp[j] - gq[j]);
          se += db * (sn[j] - ub);
          dn2 += db * db;
          const float cq = om * 0.5f * (1.0f - pbeta) * db;          // d w / d gq_j(z')
          if (gq_live[j]) { gmu[j] += cq * qiv[j]; glam[j] += cq * (-2.0f * gq[j]); }
        }
        const float inv2e = 0.5f / peps;
        const float gbe = pend_beta + om * 0.5f * sb;
        const float gep = pend_eps + om * (dn2 * inv2e * inv2e + se * inv2e);
        // one value per particle; lanes g = 0 hold it: sum over the tile, one atomic per tile and step
        const float tb = row_sum16(gbe), te = row_sum16(gep);
        if (lane == 0) {
          if (a.det) {
            if (dslot) { dslot[a.det_obe + 4 * e + 2] = tb; dslot[a.det_obe + 4 * e + 3] = te; }
          } else {
            atomicAdd(a.gtab + a.o_gbeta + (e - 1), tb);
            atomicAdd(a.gtab + a.o_geps + (e - 1), te);
          }
        }
      }
      float beta = 0.f, eps = 0.f, sig = 0.f;
      float zn[D];
      if (!BPTT && e < K) {
        beta = a.ws[a.w.beta + e]; eps = a.ws[a.w.eps + e]; sig = a.ws[a.w.sig + e];
        float nz[2 * Hh];
#pragma unroll
        for (int j = 0; j < 2 * Hh; ++j) nz[j] = 0.f;
        if (!ITEM) {
        x0 = gb; x1 = 2 + gb;
        threefry2x32(k0, k1, x0, x1);
        uint32_t g0, g1, h0, h1;
        rows01(x0, g0, g1);
        rows01(x1, h0, h1);
        constexpr int NB = 2 + Hh;
#pragma unroll
        for (int b0 = 0; b0 < NB; b0 += 4) {
          const int b = b0 + g;
          const bool is_split = b < 2;
          const int jn = b - 2;
          uint32_t y0 = is_split ? b : jn;
          uint32_t y1 = is_split ? 2 + b : ((Hh + jn < D) ? Hh + jn : 0);
          threefry2x32(is_split ? h0 : g0, is_split ? h1 : g1, y0, y1);
          if (b0 == 0) rows01(y1, k0, k1);
          uint32_t r0[4], r1[4];
          rows0123(__float_as_uint(bits_to_normal(y0)), r0);
          rows0123(__float_as_uint(bits_to_normal(y1)), r1);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int jj = b0 + q - 2;
            if (jj >= 0 && jj < Hh) {
              nz[jj] = __uint_as_float(r0[q]);
              nz[Hh + jj] = __uint_as_float(r1[q]);
            }
          }
        }
        }
        float sb = 0.f, se = 0.f, dn2 = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {  // forward kernel of step e
          const float uf = -1.0f * (beta * gp[j] + (1.0f - beta) * gq[j]);
          const float fk = z[j] - eps * uf - eps * sn[j];
          zn[j] = ITEM ? znext[j] : fk + sig * nz[j];
          const float df = ITEM ? (znext[j] - z[j]) + eps * (uf + sn[j]) : sig * nz[j];
          cot[j] += 0.5f * df;
          sb += df * (gp[j] - gq[j]);
          se += df * (uf + sn[j]);
          dn2 += df * df;
          const float cq = -om * 0.5f * (1.0f - beta) * df;           // d w / d gq_j(z)
          if (gq_live[j]) { gmu[j] += cq * qiv[j]; glam[j] += cq * (-2.0f * gq[j]); }
        }
        const float inv2e = 0.5f / eps;
        pend_beta = -om * 0.5f * sb;
        pend_eps = om * (-dn2 * inv2e * inv2e + se * inv2e);
        if (ITEM) {  // no carry between work items: the forward-kernel part of step e goes out now
          const float tb = row_sum16(pend_beta), te = row_sum16(pend_eps);
          if (lane == 0) {
            if (a.det) {
              if (dslot) { dslot[a.det_obe + 4 * e + 0] = tb; dslot[a.det_obe + 4 * e + 1] = te; }
            } else {
              atomicAdd(a.gtab + a.o_gbeta + e, tb);
              atomicAdd(a.gtab + a.o_geps + e, te);
            }
          }
        }
      }

      // ---------------------------------------------------------------- MLP backward
      asm volatile("" ::: "memory");
      float dob[D];  // d (sum omega w) / d o_j  (pre-clip / pre-factor output)
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float cj = BPTT ? cot[j] : om * cot[j];
        if (GEF) {
          dob[j] = cj * factor;
          if (g == 0) gfac += cj * opre[j];
        } else {
          dob[j] = fabsf(opre[j]) < 1e4f ? cj : 0.f;
        }
      }
      f32x4 d2[T];  // d / d a2
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f32x4 du2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < D; ++j) du2 += dob[j] * *reinterpret_cast<const f32x4*>(lds_w3t + j * HP + 16 * t + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          d2[t][r] = du2[r] * a2[t][r];
          da2T[wb[r] + 256 * t] = d2[t][r];
        }
        if (GEF) a2[t] = du2;  // keep d u2 (residual path) in a2's registers
      }
      // d u1 = [d u2 +] W2 d a2 : rows = input neuron k, contraction over output neuron n (W2^T fragments)
      f32x4 d1[T];
#pragma unroll
      for (int tk = 0; tk < T; ++tk) d1[tk] = GEF ? a2[tk] : f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (WSHARE) {
        constexpr int SHR = (T + NW - 1) / NW;
        f32x4 nxt[SHR];
        auto fetch = [&](int tn) {                  // fragments (tk, tn) for this wave's tk
          int lofs = lane * 4;
          asm volatile("" : "+v"(lofs));
#pragma unroll
          for (int k = 0; k < SHR; ++k) {
            const int tk = wv + NW * k;
            if (tk < T) nxt[k] = *reinterpret_cast<const f32x4*>(w2tf + (tk * T + tn) * 256 + lofs);
          }
        };
        auto publish = [&](int bufi) {
#pragma unroll
          for (int k = 0; k < SHR; ++k) {
            const int tk = wv + NW * k;
            if (tk < T) *reinterpret_cast<f32x4*>(wsh + (bufi * T + tk) * 256 + lane * 4) = nxt[k];
          }
        };
        fetch(0);
        publish(0);
        __syncthreads();
#pragma unroll
        for (int tn = 0; tn < T; ++tn) {
          asm volatile("" ::: "memory");
          if (tn + 1 < T) fetch(tn + 1);
          f32x4 atc[T];
#pragma unroll
          for (int tk = 0; tk < T; ++tk) atc[tk] = *reinterpret_cast<const f32x4*>(wsh + ((tn & 1) * T + tk) * 256 + lane * 4);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int tk = 0; tk < T; ++tk) {
#pragma unroll
            for (int r = 0; r < 4; ++r) d1[tk] = __builtin_amdgcn_mfma_f32_16x16x4f32(atc[tk][r], d2[tn][r], d1[tk], 0, 0, 0);
          }
          if (tn + 1 < T) publish((tn + 1) & 1);
          __syncthreads();
        }
      } else {
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
      float jpart[D];  // BPTT: J_s(z_e)^T a_s, this lane's share of the hidden units
#pragma unroll
      for (int j = 0; j < D; ++j) jpart[j] = 0.f;
      float za_own[4] = {0.f, 0.f, 0.f, 0.f};
      if (TILE_LOCAL) {   // the small tiles first: this wave's own products below need z1
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 4 * g + r;
          float zv = 0.f, dv = 0.f;
#pragma unroll
          for (int j = 0; j < D; ++j) {
            zv = (f == j) ? z[j] : zv;
            dv = (f == j) ? dob[j] : dv;
          }
          if (f == D) zv = 1.0f;
          z1T[wb[r]] = zv;
          doT[wb[r]] = dv;
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) za_own[s] = z1T[rb[s]];
      }
#pragma unroll
      for (int t = 0; t < T; ++t) {
        f32x4 pre1;
        if (KEEP_A1) {
          pre1 = a1[KEEP_A1 ? t : 0];
        } else {
          pre1 = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
          for (int j = 0; j < D; ++j) pre1 += z[j] * *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = wb[r] + (TILE_LOCAL ? 0 : 256 * t);
          if (GEF) du1T[o] = d1[t][r];
          if (BPTT && !ITEM && GEF && 16 * t < D) {  // residual path of the first block: d x_j += d u1_j
#pragma unroll
            for (int j = 0; j < D; ++j)
              if (j >= 16 * t && j < 16 * t + 16) jpart[j] += (16 * t + 4 * g + r == j) ? d1[t][r] : 0.f;
          }
          d1[t][r] *= KEEP_A1 ? pre1[r] : (GEF ? sigmoid_fast(pre1[r]) : gelu_grad_fast(pre1[r]));
          da1T[o] = d1[t][r];
        }
        if (BPTT && !ITEM) {
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const f32x4 wv4 = *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
            jpart[j] += d1[t][0] * wv4[0] + d1[t][1] * wv4[1] + d1[t][2] * wv4[2] + d1[t][3] * wv4[3];
          }
        }
        if (TILE_LOCAL) {
          // this wave's own outer products for tile t, straight from the two 1 KB buffers just written (same wave: the
          // LDS queue is in order, no barrier): dW1 / bias-table row (d a1), residual table (d u1)
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, s2acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(za_own[s4], da1T[rb[s4]], sacc, 0, 0, 0);
            if (GEF) s2acc = __builtin_amdgcn_mfma_f32_16x16x4f32(za_own[s4], du1T[rb[s4]], s2acc, 0, 0, 0);
          }
          asm volatile("" ::: "memory");   // the next tile's stores stay behind these reads
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            if (row < D) accZ1[row * HP + 16 * t + c] += sacc[r];   // (wave-private: a plain update; ds_add_f32 measured 1 % slower)
            if (row == D) {
              if (a.det) {
                if (dslot) {
                  dslot[(int64_t)e * HP + 16 * t + c] = sacc[r];
                  if (GEF) dslot[(int64_t)(K + 1 + e) * HP + 16 * t + c] = s2acc[r];
                }
              } else {
                atomicAdd(gS + erow * HP + 16 * t + c, sacc[r]);
                if (GEF) atomicAdd(gS2 + erow * HP + 16 * t + c, s2acc[r]);
              }
            }
          }
        }
      }
      if (BPTT) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
          if (!ITEM) lam[j] += group_sum(jpart[j]);
          if (e == 0) {  // z_0 = mean + std e0 and the explicit parameters of log q(z_0)
            const float dz = z[j] - qmean[j];
            const float l0 = ITEM ? (valid ? a.lam[pc * D + j] : 0.f) : lam[j];
            gmu[j] += l0 - om * gq[j];
            glam[j] += l0 * dz + om * (dz * dz * qiv[j] - 1.0f);
          }
          lamn[j] = lam[j];
          znext[j] = z[j];
        }
      }
      // ---------------------------------------------------------------- stage the two small tiles
      if (!TILE_LOCAL) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = 4 * g + r;
        float zv = 0.f, dv = 0.f;
#pragma unroll
        for (int j = 0; j < D; ++j) {
          zv = (f == j) ? z[j] : zv;
          dv = (f == j) ? dob[j] : dv;
        }
        if (f == D) zv = 1.0f;
        z1T[wb[r]] = zv;
        doT[wb[r]] = dv;
      }
      }
      __syncthreads();
      // ---------------------------------------------------------------- outer products over particles
      {
        float za[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) za[s] = z1T[rb[s]];
#pragma unroll
        for (int t = 0; t < T; ++t) {
          f32x4 sacc = {0.f, 0.f, 0.f, 0.f}, s2acc = {0.f, 0.f, 0.f, 0.f}, bacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            const float b2v = da2T[rb[s] + 256 * t];
            bacc = __builtin_amdgcn_mfma_f32_16x16x4f32(za[s], b2v, bacc, 0, 0, 0);
            if (!TILE_LOCAL) {
              const float b1v = da1T[rb[s] + 256 * t];
              sacc = __builtin_amdgcn_mfma_f32_16x16x4f32(za[s], b1v, sacc, 0, 0, 0);
              if (GEF) {
                const float b3v = du1T[rb[s] + 256 * t];
                s2acc = __builtin_amdgcn_mfma_f32_16x16x4f32(za[s], b3v, s2acc, 0, 0, 0);
              }
            }
          }
          // C rows are the features of z1 = [z_0 .. z_{D-1}, 1, 0 ...]: lane (g, c), register r <-> row 4 g + r
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 4 * g + r;
            if (!TILE_LOCAL && row < D) accZ1[row * HP + 16 * t + c] += sacc[r];   // dW1[row][n] += z_row . d a1[n]
            if (row == D) {
              accB2[16 * t + c] += bacc[r];                                            // db2[n] += sum_p d a2
              if (!TILE_LOCAL) {                                                       // d / d bias-table row used
                if (a.det) {
                  if (dslot) {
                    dslot[(int64_t)e * HP + 16 * t + c] = sacc[r];
                    if (GEF) dslot[(int64_t)(K + 1 + e) * HP + 16 * t + c] = s2acc[r];
                  }
                } else {
                  atomicAdd(gS + erow * HP + 16 * t + c, sacc[r]);
                  if (GEF) atomicAdd(gS2 + erow * HP + 16 * t + c, s2acc[r]);
                }
              }
            }
          }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) gB3 = __builtin_amdgcn_mfma_f32_16x16x4f32(za[s], doT[rb[s]], gB3, 0, 0, 0);
      }
#pragma unroll
      for (int k = 0; k < OWN; ++k) {
        const int ti = wv + NW * k;
        if (ti < T) {
#pragma unroll
          for (int q = 0; q < NW; ++q) {  // all tiles of the workgroup
            const float* base = stage + q * STG;
            float xa[4], x2[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              xa[s] = base[rb[s] + 256 * ti];                            // u1T
              x2[s] = base[HP * 16 + rb[s] + 256 * ti];                  // u2T
            }
#pragma unroll
            for (int to = 0; to < T; ++to)
#pragma unroll
              for (int s = 0; s < 4; ++s)
                gW2[k][to] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s], base[2 * HP * 16 + rb[s] + 256 * to],
                                                                  gW2[k][to], 0, 0, 0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
              gW3[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(x2[s], base[OFF_DOT + rb[s]], gW3[k], 0, 0, 0);
          }
        }
      }
      __syncthreads();
      // ---------------------------------------------------------------- advance
      if (!BPTT && !ITEM && e < K) {
#pragma unroll
        for (int j = 0; j < D; ++j) { zp[j] = z[j]; z[j] = zn[j]; }
        pbeta = beta; peps = eps;
      }
    }
  }

  // ---------------------------------------------------------------- write the workgroup slab
  float* slab = a.slabs + (int64_t)blockIdx.x * a.slab_stride;
  // layout: dW2 [HP][HP] | dW3 [HP][16] | per wave: gZ1 [16][HP], gB2 [16][HP], gB3 [16][16], scalars [32]
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
  float* pw = slab + HP * HP + HP * 16 + wv * (2 * 16 * HP + 256 + 32);
  for (int i = lane; i < D * HP; i += 64) pw[i] = accZ1[i];                       // rows j < D of the [16][HP] region
  for (int i = lane; i < HP; i += 64) pw[16 * HP + D * HP + i] = accB2[i];      // row D of the second region
#pragma unroll
  for (int r = 0; r < 4; ++r) pw[2 * 16 * HP + (4 * g + r) * 16 + c] = gB3[r];
  {
    float* sc = pw + 2 * 16 * HP + 256;
    const float tf = row_sum16(gfac);
    if (lane == 0) sc[0] = tf;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float tm = row_sum16(gmu[j]), tl = row_sum16(glam[j]);
      if (lane == 0) { sc[1 + 2 * j] = tm; sc[2 + 2 * j] = tl; }
    }
  }
}

// (the Jacobian and scan kernels of the work-item reparameterised gradient live in cmcd_bptt.hip)

// fixed-order sum over tiles of per-tile rows of `len` floats: 16 lanes per output, lane l the tiles l, l + 16, ...
// (entries with o % period >= used are padding nobody writes or reads)
__global__ __launch_bounds__(256) void tile_rows_reduce_kernel(const float* rows, int64_t ntiles, int64_t len, int64_t period,
                                                               int64_t used, float* out) {
  const int sub = threadIdx.x & 15;
  const int64_t o = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const bool on = o < len && o % period < used;
  float v = 0.f;
  if (on) {
    int64_t t = sub;
    for (; t + 16 * 7 < ntiles; t += 16 * 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = rows[(t + 16 * u) * len + o];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; t < ntiles; t += 16) v += rows[t * len + o];
  }
#pragma unroll
  for (int sh = 8; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
  if (on && sub == 0) out[o] = v;
}

// ------------------------------------------------------------------------------------------
// MCD_ULA (no network; /root/reference/src/mcd_over_orig.py with use_sn = False): the reverse recursion
// without the MLP — target Hessian product, q, schedules.  One wave per 16-particle tile, whole chain.
// ------------------------------------------------------------------------------------------
struct UlaGradArgs {
  const float* params;
  const float* ws;
  const float* traj;     // [K+1][n][D]
  float* gtab;           // gbeta at o_gbeta, geps at o_geps
  float* gvd;            // [ntiles][2 D] per-tile q-gradient rows
  float* det;            // [ntiles][2][K4] per-tile schedule-gradient rows (nullptr: float atomics on gtab)
  cmcd_layout lay;
  WsLayout w;
  int64_t n, o_gbeta, o_geps;
  int32_t K;
  float omega;
};

template <int TARGET, int D>
__global__ __launch_bounds__(256) void ula_grad_kernel(UlaGradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_tgt[];
  for (int i = threadIdx.x; i < a.w.tgt_floats; i += blockDim.x) lds_tgt[i] = a.ws[a.w.tgt + i];
  __syncthreads();
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  if (tile * 16 >= a.n) return;
  const int64_t p = tile * 16 + c;
  const bool valid = p < a.n;
  const int64_t pc = valid ? p : a.n - 1;
  const float om = valid ? a.omega : 0.f;
  const int K = a.K;
  float qmean[D], qiv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qmean[j] = a.params[a.lay.vd_mean + j];
    const float sd = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (sd * sd);
  }
  constexpr int HN = Target<TARGET, D>::HN;
  float lamn[D], gE[D], znext[D], gmu[D], glam[D];
#pragma unroll
  for (int j = 0; j < D; ++j) { lamn[j] = 0.f; gE[j] = 0.f; znext[j] = 0.f; gmu[j] = 0.f; glam[j] = 0.f; }
  float pend_beta = 0.f, pend_eps = 0.f;
  for (int e = K; e >= 0; --e) {
    float z[D], zp[D], gp[D], gq[D], hs[HN], logp;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      z[j] = a.traj[((int64_t)e * a.n + pc) * D + j];
      zp[j] = e > 0 ? a.traj[((int64_t)(e - 1) * a.n + pc) * D + j] : 0.f;
    }
    Target<TARGET, D>::eval_hess(z, g, lds_tgt, logp, gp, hs);
#pragma unroll
    for (int j = 0; j < D; ++j) gq[j] = -(z[j] - qmean[j]) * qiv[j];
    float a_gp[D], a_gq[D], gprev[D], lam[D];
#pragma unroll
    for (int j = 0; j < D; ++j) { a_gp[j] = 0.f; a_gq[j] = 0.f; gprev[j] = 0.f; lam[j] = 0.f; }
    float npb = 0.f, npe = 0.f;
    if (e > 0) {
      const float pb = a.ws[a.w.beta + e - 1], pe = a.ws[a.w.eps + e - 1];
      const float inv2e = 0.5f / pe;
      float sb = 0.f, se = 0.f, r2 = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float ub = -1.0f * (pb * gp[j] + (1.0f - pb) * gq[j]);
        const float r = zp[j] - (z[j] - pe * ub);
        gprev[j] = -om * r * inv2e;
        a_gp[j] += pe * pb * gprev[j];
        a_gq[j] += pe * (1.0f - pb) * gprev[j];
        lam[j] += gprev[j];
        sb += (gp[j] - gq[j]) * gprev[j];
        se += -ub * gprev[j];
        r2 += r * r;
      }
      npb = pe * sb;
      npe = se - om * r2 * inv2e * inv2e;
    }
    if (e < K) {
      const float be = a.ws[a.w.beta + e], ee = a.ws[a.w.eps + e];
      const float inv2e = 0.5f / ee;
      float sb = 0.f, se = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float uf = -1.0f * (be * gp[j] + (1.0f - be) * gq[j]);
        const float nsig = ((znext[j] - z[j]) + ee * uf) * inv2e;
        a_gp[j] += ee * be * lamn[j];
        a_gq[j] += ee * (1.0f - be) * lamn[j];
        lam[j] += lamn[j] - gE[j];
        sb += (gp[j] - gq[j]) * lamn[j];
        se += (nsig - uf) * lamn[j];
      }
      const float tb = row_sum16(pend_beta + ee * sb), te = row_sum16(pend_eps + se);
      if (lane == 0) {
        if (a.det) {   // a.o_geps = K4 = the padded row length
          a.det[tile * 2 * a.o_geps + e] = tb;
          a.det[tile * 2 * a.o_geps + a.o_geps + e] = te;
        } else {
          atomicAdd(a.gtab + a.o_gbeta + e, tb);
          atomicAdd(a.gtab + a.o_geps + e, te);
        }
      }
    }
    pend_beta = npb; pend_eps = npe;
    float hv[D];
    Target<TARGET, D>::hvp(hs, z, a_gp, hv);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (e == K) lam[j] -= om * gp[j];
      if (e == 0) lam[j] += om * gq[j];
      gmu[j] += a_gq[j] * qiv[j];
      glam[j] += a_gq[j] * (-2.0f * gq[j]);
      lam[j] += hv[j] - a_gq[j] * qiv[j];
      if (e == 0) {
        const float dz = z[j] - qmean[j];
        gmu[j] += lam[j] - om * gq[j];
        glam[j] += lam[j] * dz + om * (dz * dz * qiv[j] - 1.0f);
      }
      gE[j] = gprev[j];
      lamn[j] = lam[j];
      znext[j] = z[j];
    }
  }
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float tm = row_sum16(gmu[j]), tl = row_sum16(glam[j]);
    if (lane == 0) {
      a.gvd[tile * (2 * D) + j] = tm;
      a.gvd[tile * (2 * D) + D + j] = tl;
    }
  }
}

// d vd.mean / d vd.logdiag: fixed-order sum of the per-tile rows
__global__ void ula_vd_reduce_kernel(const float* rows, int64_t ntiles, int D, float* grad, int64_t o_mean, int64_t o_logdiag) {
  const int j = threadIdx.x;
  if (j >= 2 * D) return;
  float v = 0.f;
  for (int64_t t = 0; t < ntiles; ++t) v += rows[t * 2 * D + j];
  if (j < D) grad[o_mean + j] = v;
  else grad[o_logdiag + (j - D)] = v;
}

// ------------------------------------------------------------------------------------------
// reduction of the workgroup slabs into grad_flat (fixed order) + particle-independent tails
// ------------------------------------------------------------------------------------------
struct TailArgs {
  const float* params;
  const float* ws;
  const float* gtab;
  const float* slabs;
  float* grad;         // [n_params]
  float* tail;         // dds: [(K+1)][kTailRow] per-evaluation vectors of the time-coder tail
  cmcd_layout lay;
  WsLayout w;
  int64_t o_S, o_S2, o_gbeta, o_geps, o_gvd, o_gfac, slab_stride, n_params;
  int32_t K, D, E, IN, HP, arch, nslabs, eps_schedule, ngrid, nw;
};

// Every entry of grad_flat that is a plain sum of slab entries: 16 lanes per output element, each lane a fixed
// strided subset of the slabs (or of the (slab, wave) regions), then a fixed butterfly: deterministic, and
// 256 slabs cost 16 loads per lane instead of a 256-long serial walk.
__device__ __forceinline__ void grad_reduce_body(const TailArgs& a, const unsigned bidx, const unsigned gdim) {
  const int HP = a.HP, D = a.D, IN = a.IN;
  const bool dds = a.arch == CMCD_ARCH_DDS;
  const int64_t o_w1 = dds ? a.lay.d_sw1 : a.lay.g_w1, o_w2 = dds ? a.lay.d_sw2 : a.lay.g_w2;
  const int64_t o_b2 = dds ? a.lay.d_sb2 : a.lay.g_b2, o_w3 = dds ? a.lay.d_sw3 : a.lay.g_w3;
  const int64_t o_b3 = dds ? a.lay.d_sb3 : a.lay.g_b3;
  const int wid = dds ? 64 : IN;  // true width, row length of W1 / W2
  const int sub = threadIdx.x & 15;
  int64_t i = ((int64_t)bidx * blockDim.x + threadIdx.x) >> 4;
  const int64_t per = 2 * 16 * HP + 256 + 32, base = (int64_t)HP * HP + HP * 16;   // per-wave regions of a slab
  int64_t dst = -1, off = 0;
  bool perwave = false;
  if (i < (int64_t)wid * wid) {                                   // dW2[k][n]
    dst = o_w2 + i; off = (i / wid) * HP + (i % wid);
  } else if ((i -= (int64_t)wid * wid) < (int64_t)wid * D) {      // dW3[n][j]
    dst = o_w3 + i; off = (int64_t)HP * HP + (i / D) * 16 + (i % D);
  } else if ((i -= (int64_t)wid * D) < (int64_t)D * wid) {        // dW1[j][n], j < D (z rows)
    dst = o_w1 + i; off = (i / wid) * HP + (i % wid); perwave = true;
  } else if ((i -= (int64_t)D * wid) < wid) {                     // db2
    dst = o_b2 + i; off = 16 * HP + (int64_t)D * HP + i; perwave = true;
  } else if ((i -= wid) < D) {                                    // db3
    dst = o_b3 + i; off = 2 * 16 * HP + D * 16 + i; perwave = true;
  } else if ((i -= D) < D) {                                      // d vd.mean
    dst = a.lay.vd_mean + i; off = 2 * 16 * HP + 256 + 1 + 2 * i; perwave = true;
  } else if ((i -= D) < D) {                                      // d vd.logdiag
    dst = a.lay.vd_logdiag + i; off = 2 * 16 * HP + 256 + 2 + 2 * i; perwave = true;
  } else if ((i -= D) < 1 && !dds) {                              // d factor_sn
    dst = a.lay.g_factor; off = 2 * 16 * HP + 256; perwave = true;
  }
  // lane `sub` of an output sums the terms sub, sub + 16, ... in that order — as before, but with eight loads in flight: the
  // per-wave outputs of a 256-slab work-item gradient are 1024 terms = 64 DEPENDENT loads per lane otherwise (29 us of a
  // 320 us funnel K = 64 training iteration, 42 us of the named shape's)
  float v = 0.f;
  if (dst >= 0) {
    const int tot = perwave ? a.nslabs * a.nw : a.nslabs;
    auto term = [&](int t) -> int64_t {
      if (!perwave) return (int64_t)t * a.slab_stride + off;
      const int sl = t / a.nw, q = t - sl * a.nw;
      return (int64_t)sl * a.slab_stride + base + q * per + off;
    };
    int t = sub;
    for (; t + 16 * 7 < tot; t += 16 * 8) {
      float x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) x[u] = a.slabs[term(t + 16 * u)];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += x[u];
    }
    for (; t < tot; t += 16) v += a.slabs[term(t)];
  }
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  if (dst >= 0 && sub == 0) a.grad[dst] = v;
}

// d / d eps0 and d / d mgridref_y from the per-step tables (one 256-thread block, thread per bridge)
__device__ __forceinline__ void grad_sched_tail_body(const TailArgs& a, const unsigned bidx, const unsigned gdim) {
  __shared__ float gyg[40], gyv[40], red[256], ms[40];
  __shared__ float c_lo[1024], c_hi[1024];
  __shared__ int c_j[1024], seg[41];
  const int K = a.K, G = a.ngrid;
  const float* gbeta = a.gtab + a.o_gbeta;
  const float* geps = a.gtab + a.o_geps;
  // mgridref_y once, in parallel: thread 0's three serial passes below read it from LDS instead of issuing ~3 (G + 1)
  // dependent global loads (no measurable change of the tails launch: 0.1 us level)
  if ((int)threadIdx.x <= G) ms[threadIdx.x] = a.params[a.lay.mgridref_y + threadIdx.x];
  float ge = 0.f, gq = 0.f;   // gq: this thread's grid node q = threadIdx.x (q <= G + 1)
  for (int base = 0; base < K; base += 1024) {
    __syncthreads();
    if (threadIdx.x < 41) seg[threadIdx.x] = -1;
    __syncthreads();
    for (int i = base + threadIdx.x; i < K && i < base + 1024; i += blockDim.x) {
      float dedeps0 = 1.0f;                                                         // constant schedule
      if (a.eps_schedule == CMCD_EPS_COS_SQ) {
        const float cs = cosf(((float)i / (float)K + 0.008f) / 1.008f * 0.5f * 3.14159265358979323846f);
        dedeps0 = cs * cs;
      } else if (a.eps_schedule == CMCD_EPS_LINEAR) {
        dedeps0 = 1.0f - (float)i / (float)(K - 1);
      }
      ge += geps[i] * dedeps0;
      // beta_i = gy[j-1] + frac_i (gy[j] - gy[j-1]),  gy = [0, cumsum(m)/sum(m)]
      auto cell = [&](int ii) {
        const float x = (float)(ii + 1) / (float)(K + 1);
        int j = 1;
        while (j < G + 1 && (float)j / (float)(G + 1) <= x) ++j;
        return j;
      };
      const float x = (float)(i + 1) / (float)(K + 1);
      const int j = cell(i);
      const float x0 = (float)(j - 1) / (float)(G + 1), x1 = (float)j / (float)(G + 1);
      const float fr = (x - x0) / (x1 - x0);
      c_j[i - base] = j;
      c_lo[i - base] = gbeta[i] * (1.0f - fr);
      c_hi[i - base] = gbeta[i] * fr;
      if (i == base || cell(i - 1) != j) seg[j] = i - base;   // the cell index grows with i: the steps of a cell are a run
    }
    __syncthreads();
    // node q collects the run of cell q (upper ends) and then the run of cell q + 1 (lower ends), each in step order
    // (LDS float atomics from 256 threads summed them in whatever order the waves arrived: the last source of
    // run-to-run differences in a training step, r04)
    if ((int)threadIdx.x <= G + 1) {
      const int q = threadIdx.x, lim = K - base < 1024 ? K - base : 1024;
      if (q >= 1)
        for (int i = seg[q]; i >= 0 && i < lim && c_j[i] == q; ++i) gq += c_hi[i];
      if (q <= G)
        for (int i = seg[q + 1]; i >= 0 && i < lim && c_j[i] == q + 1; ++i) gq += c_lo[i];
    }
  }
  if ((int)threadIdx.x <= G + 1) gyg[threadIdx.x] = gq;
  red[threadIdx.x] = ge;
  __syncthreads();
  for (int s2 = 128; s2 > 0; s2 >>= 1) {
    if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    a.grad[a.lay.eps] = red[0];
    // gy[q] = C_q / S, C_q = sum_{r<q} m_r (q >= 1), S = sum m:  d gy[q] / d m_r = ([r < q] - gy[q]) / S
    const float* m = ms;
    float S = 0.f;
    for (int r = 0; r <= G; ++r) S += m[r];
    float run = 0.f, dot = 0.f;
    gyv[0] = 0.f;
    for (int r = 0; r <= G; ++r) { run += m[r]; gyv[r + 1] = run / S; }
    for (int q = 1; q <= G + 1; ++q) dot += gyg[q] * gyv[q];
    float suffix = 0.f;  // sum_{q > r} gyg[q]
    for (int r = G; r >= 0; --r) {
      suffix += gyg[r + 1];
      a.grad[a.lay.mgridref_y + r] = (suffix - dot) / S;
    }
  }
}

// geffner tail: S[e][n] = d/d bias-table, S2[e][n] = sum_p d u1.   emb row of evaluation e is min(e, K-1).
//   db1[n] = sum_e S[e][n];  dW1[D+j][n] = sum_e emb[ie][j] S[e][n];
//   demb[i][j] = sum_{e: ie(e) = i} ( S2[e][D+j] + sum_n W1[D+j][n] S[e][n] )
__device__ __forceinline__ void grad_geffner_tail_body(const TailArgs& a, const unsigned bidx, const unsigned gdim) {
  const int K = a.K, D = a.D, E = a.E, IN = a.IN, HP = a.HP;
  const float* S = a.gtab + a.o_S;
  const float* S2 = a.gtab + a.o_S2;
  const float* P = a.params;
  const int64_t tid = (int64_t)bidx * blockDim.x + threadIdx.x, stride = (int64_t)gdim * blockDim.x;
  // db1 and dW1[D:, :]: 16 lanes per output, lane `part` sums the evaluations e = part, part + 16, ... and a fixed
  // butterfly adds the 16 partial sums (one thread per output walked all K + 1 evaluations serially — 257 dependent
  // L2 round trips: 0.23 ms of config 4's 2 ms training step at a 2000-particle shard)
  const int64_t n_out = (int64_t)IN + (int64_t)E * IN;
  for (int64_t t = tid; t < n_out * 16; t += stride) {
    const int64_t o = t >> 4;
    const int part = int(t & 15);
    // (same per-lane order, eight loads — or pairs of loads — in flight instead of one dependent round trip per evaluation)
    float v = 0.f;
    if (o < IN) {
      int e = part;
      for (; e + 16 * 7 <= K; e += 16 * 8) {
        float x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = S[(int64_t)(e + 16 * u) * HP + o];
#pragma unroll
        for (int u = 0; u < 8; ++u) v += x[u];
      }
      for (; e <= K; e += 16) v += S[(int64_t)e * HP + o];
    } else {
      const int j = int((o - IN) / IN), n = int((o - IN) % IN);
      auto emb = [&](int e) { return P[a.lay.g_emb + (int64_t)(e < K ? e : K - 1) * E + j]; };
      int e = part;
      for (; e + 16 * 7 <= K; e += 16 * 8) {
        float x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = emb(e + 16 * u); y[u] = S[(int64_t)(e + 16 * u) * HP + n]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) v += x[u] * y[u];
      }
      for (; e <= K; e += 16) v += emb(e) * S[(int64_t)e * HP + n];
    }
#pragma unroll
    for (int sh = 8; sh > 0; sh >>= 1) v += __shfl_xor(v, sh);
    if (part == 0) {
      if (o < IN) a.grad[a.lay.g_b1 + o] = v;
      else a.grad[a.lay.g_w1 + (int64_t)D * IN + (o - IN)] = v;   // row D + j, column n: (D + j) IN + n
    }
  }
  // one wave per (row, j): lanes stride the IN-long dot product (coalesced; one thread per output walked W1 rows IN
  // apart: 370 us at IN = 1620)
  const int lane = threadIdx.x & 63;
  for (int64_t i = tid >> 6; i < (int64_t)K * E; i += stride >> 6) {
    const int row = int(i / E), j = int(i % E);
    float v = 0.f;
    for (int e = row; e <= (row == K - 1 ? K : row); ++e) {
      for (int n = lane; n < IN; n += 64) v += P[a.lay.g_w1 + (int64_t)(D + j) * IN + n] * S[(int64_t)e * HP + n];
      if (lane == 0) v += S2[(int64_t)e * HP + D + j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if (lane == 0) a.grad[a.lay.g_emb + i] = v;
  }
}

// enough 256-thread blocks that every 16-lane group / wave of the kernel above has about one output (capped)
static inline unsigned geffner_tail_blocks(const TailArgs& a) {
  const int64_t groups = ((int64_t)a.IN + (int64_t)a.E * a.IN) * 16, waves = (int64_t)a.K * a.E * 64;
  const int64_t threads = groups > waves ? groups : waves;
  const int64_t blocks = (threads + 255) / 256;
  return (unsigned)(blocks < 64 ? 64 : (blocks > 4096 ? 4096 : blocks));
}

// dds tail: S[e][n] = d / d bias1[e][n] with bias1[e] = sb1 + tau(e) sw1[D:, :], tau(e) the time coder
// (nn_dds.py:131-143,155-158).  Two launches: (A) one 64-thread block per e recomputes the time path and
// back-propagates S[e] through it, leaving its per-e vectors in a table; (B) one thread per parameter sums the
// per-e outer products over e in a fixed order (no atomics: 4.3 M float atomics took 50 us, this takes ~15).
constexpr int kTailRow = 448;   // per e: emb[128] | hh[64] | tau[64] | dtau[64] | dact[64] | dphase[64]

// ONE wave runs this body (64 threads): its LDS hand-overs need no workgroup barrier — the LDS queue of a wave is in
// order — only a compiler / counter fence.  (r02 called it from the first wave of a 256-thread block with
// __syncthreads() inside: a barrier not reached by every thread of the block is undefined behaviour.)
__device__ __forceinline__ void tail_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): every LDS write of this wave has landed
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void grad_dds_tail_body(const TailArgs& a, const unsigned bidx, const unsigned gdim) {
  __shared__ float emb[128], ha[64], hh[64], dtau[64], dh[64], dact[64];
  const int j = threadIdx.x, t = bidx, D = a.D;
  const float* P = a.params;
  const float* S = a.gtab + a.o_S + (int64_t)t * 64;
  float* row = a.tail + (int64_t)t * kTailRow;
  {
    // timestep_coeff = jnp.linspace(0.1, 100, 64): float32 arithmetic, start (1 - s) + stop s with s = iota / 63, the
    // end point appended exactly (nn_dds.py:108; `np` there is jax.numpy).  Unfused: XLA folds it as written.
    float cj;
    {
#pragma clang fp contract(off)
      const float sj = (float)j / 63.0f;
      const float lo_part = 0.1f * (1.0f - sj), hi_part = 100.0f * sj;
      cj = (j == 63) ? 100.0f : lo_part + hi_part;
    }
    const float arg = cj * (float)t + P[a.lay.d_phase + j];
    emb[j] = sinf(arg);
    emb[64 + j] = cosf(arg);
  }
  tail_wave_sync();
  float acc = P[a.lay.d_tb1 + j];
  for (int k = 0; k < 128; ++k) acc = fmaf(emb[k], P[a.lay.d_tw1 + k * 64 + j], acc);
  ha[j] = acc;
  hh[j] = gelu_exact(acc);
  tail_wave_sync();
  acc = P[a.lay.d_tb2 + j];
  for (int k = 0; k < 64; ++k) acc = fmaf(hh[k], P[a.lay.d_tw2 + k * 64 + j], acc);
  const float tau = acc;
  // d tau_k = sum_n sw1[D+k][n] S[n]
  acc = 0.f;
  for (int n = 0; n < 64; ++n) acc = fmaf(P[a.lay.d_sw1 + (int64_t)(D + j) * 64 + n], S[n], acc);
  dtau[j] = acc;
  tail_wave_sync();
  acc = 0.f;
  for (int n = 0; n < 64; ++n) acc = fmaf(P[a.lay.d_tw2 + j * 64 + n], dtau[n], acc);
  dh[j] = acc;
  {  // exact gelu': Phi(x) + x phi(x)
    const float x = ha[j];
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
    dact[j] = dh[j] * (cdf + x * pdf);
  }
  tail_wave_sync();
  float de[2];
  for (int q = 0; q < 2; ++q) {
    const int k = j + 64 * q;
    acc = 0.f;
    for (int n = 0; n < 64; ++n) acc = fmaf(P[a.lay.d_tw1 + k * 64 + n], dact[n], acc);
    de[q] = acc;
  }
  row[j] = emb[j];
  row[64 + j] = emb[64 + j];
  row[128 + j] = hh[j];
  row[192 + j] = tau;
  row[256 + j] = dtau[j];
  row[320 + j] = dact[j];
  // emb = [sin(arg), cos(arg)], arg = c t + phase:  d phase_j = demb_j cos(arg_j) - demb_{64+j} sin(arg_j)
  row[384 + j] = de[0] * emb[64 + j] - de[1] * emb[j];
}

// (B) d sb1, d sw1[D:], d tb2, d tw2, d tb1, d tw1, d phase: 16 lanes per entry, lane l sums e = l, l+16, ...,
// then a fixed butterfly (deterministic; a one-thread 257-long walk is latency-bound: 67 us)
// thin launchers of the bodies above (the MCD_ULA and lgcp sweeps launch the schedule / geffner tails on their own)
__global__ __launch_bounds__(256) void grad_sched_tail_kernel(TailArgs a) { grad_sched_tail_body(a, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void grad_geffner_tail_kernel(TailArgs a) { grad_geffner_tail_body(a, blockIdx.x, gridDim.x); }

// Everything behind the gradient kernel that only reads its tables and slabs, as ONE launch (r02: four dependent
// 9 - 28 us launches were 10 % of the VarGrad step at N = 2000): blocks [0, n_reduce) the slab reduction, block n_reduce
// the schedule tail, then either the K + 1 time-coder blocks of the dds tail (64 threads each; its fixed-order sum over
// evaluations stays a second launch) or the geffner tail's blocks.  The groups write disjoint leaves of grad_flat.
__global__ __launch_bounds__(256) void grad_tails_fused_kernel(TailArgs a, unsigned n_reduce, unsigned n_third) {
  const unsigned b = blockIdx.x;
  if (b < n_reduce) {
    grad_reduce_body(a, b, n_reduce);
  } else if (b == n_reduce) {
    grad_sched_tail_body(a, 0, 1);
  } else if (a.arch == CMCD_ARCH_DDS) {
    if (threadIdx.x < 64) grad_dds_tail_body(a, b - n_reduce - 1, n_third);   // wave 0 only; the body synchronises at wave level
  } else {
    grad_geffner_tail_body(a, b - n_reduce - 1, n_third);
  }
}

__global__ __launch_bounds__(256) void grad_dds_tail_sum_kernel(TailArgs a) {
  const int sub = threadIdx.x & 15, D = a.D, E1 = a.K + 1;
  const int i = (blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const float* T0 = a.tail;
  const float* S = a.gtab + a.o_S;
  // segments: [0,64) sb1 | 4096 sw1[D+k][j] | 64 tb2 | 4096 tw2 | 64 tb1 | 8192 tw1 | 64 phase
  int o = i;
  int64_t dst = -1;
  int xa = -1, xb = -1;     // offsets inside the per-e row (xa < 0: factor 1);  xb < 0: second factor is S[e][-xb-1]
  if (o < 64) { dst = a.lay.d_sb1 + o; xb = -(o + 1); }
  else if ((o -= 64) < 4096) { dst = a.lay.d_sw1 + (int64_t)(D + (o >> 6)) * 64 + (o & 63); xa = 192 + (o >> 6); xb = -((o & 63) + 1); }
  else if ((o -= 4096) < 64) { dst = a.lay.d_tb2 + o; xb = 256 + o; }
  else if ((o -= 64) < 4096) { dst = a.lay.d_tw2 + o; xa = 128 + (o >> 6); xb = 256 + (o & 63); }
  else if ((o -= 4096) < 64) { dst = a.lay.d_tb1 + o; xb = 320 + o; }
  else if ((o -= 64) < 8192) { dst = a.lay.d_tw1 + o; xa = o >> 6; xb = 320 + (o & 63); }
  else if ((o -= 8192) < 64) { dst = a.lay.d_phase + o; xb = 384 + o; }
  float v = 0.f;
  if (dst >= 0) {
    // lane `sub` sums the evaluations sub, sub + 16, ... in that order, eight pairs of loads in flight (K = 256: 17 terms per
    // lane were 17 dependent round trips of two loads)
    auto fac_a = [&](int e) { return xa >= 0 ? T0[(int64_t)e * kTailRow + xa] : 1.0f; };
    auto fac_b = [&](int e) { return xb >= 0 ? T0[(int64_t)e * kTailRow + xb] : S[(int64_t)e * 64 + (-xb - 1)]; };
    int e = sub;
    for (; e + 16 * 7 < E1; e += 16 * 8) {
      float fa[8], fb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { fa[u] = fac_a(e + 16 * u); fb[u] = fac_b(e + 16 * u); }
#pragma unroll
      for (int u = 0; u < 8; ++u) v = fmaf(fa[u], fb[u], v);
    }
    for (; e < E1; e += 16) v = fmaf(fac_a(e), fac_b(e), v);
  }
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  if (dst >= 0 && sub == 0) a.grad[dst] = v;
}

typedef void (*grad_fn)(GradArgs);
#ifndef CMCD_GRAD_SMALL
#define CMCD_GRAD_SMALL 1
#endif
constexpr bool kGradSmall = CMCD_GRAD_SMALL != 0;   // work items of the narrow 2-d nets: L2-streamed W2 + tile-local staging
#ifndef CMCD_GRAD_SMALL_CHAIN
#define CMCD_GRAD_SMALL_CHAIN 1
#endif
constexpr bool kGradSmallChain = CMCD_GRAD_SMALL_CHAIN != 0;   // ... and the whole-chain instances too

static int grad_nw(int T) { (void)T; return 4; }   // (r02: the 132-wide net too — its staging shrank to 33.5 KB per wave)

template <bool BPTT, bool ITEM>
static grad_fn pick_grad_t(const cmcd_desc& d, int T) {
  if (d.arch == CMCD_ARCH_DDS && T == 4) {
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return grad_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4, 4, (ITEM ? kGradSmall : kGradSmallChain), BPTT, ITEM>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2) return grad_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4, 4, (ITEM ? kGradSmall : kGradSmallChain), BPTT, ITEM>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return grad_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_DDS, 10, 4, 4, false, BPTT, ITEM>;
    return nullptr;
  }
  if (d.arch == CMCD_ARCH_GEFFNER) {
    if (d.target == CMCD_TARGET_GMM && d.dim == 2 && T == 2) return grad_kernel<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, 2, 4, (ITEM ? kGradSmall : kGradSmallChain), BPTT, ITEM>;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2 && T == 2) return grad_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, 2, 4, (ITEM ? kGradSmall : kGradSmallChain), BPTT, ITEM>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2 && T == 4) return grad_kernel<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, 4, 4, (ITEM ? kGradSmall : kGradSmallChain), BPTT, ITEM>;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2 && T == 4) return grad_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, 4, 4, (ITEM ? kGradSmall : kGradSmallChain), BPTT, ITEM>;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2 && T == 9) return grad_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, 9, 4, true, BPTT, ITEM>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2 && T == 9) return grad_kernel<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, 9, 4, true, BPTT, ITEM>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10 && T == 4) return grad_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10, 4, 4, false, BPTT, ITEM>;
  }
  return nullptr;
}
static grad_fn pick_grad(const cmcd_desc& d, int T, bool bptt, bool item = false) {
  if (item) return bptt ? pick_grad_t<true, true>(d, T) : pick_grad_t<false, true>(d, T);
  return bptt ? pick_grad_t<true, false>(d, T) : pick_grad_t<false, false>(d, T);
}

bool grad_available(const cmcd_desc& d, int T) { return pick_grad(d, T, false) != nullptr; }
bool bptt_available(const cmcd_desc& d, int T) { return pick_grad(d, T, true) != nullptr; }

// Work-item (small-batch) path: worthwhile while whole-chain waves cannot fill the chip.  cmcd_debug_grad_item(0 / 1)
// pins it process-wide (tests and probes run every case through both paths; the Python binding forwards
// CMCD_GRAD_ITEM from the environment), -1 returns to the measured rule.  No getenv on the per-call path.
static std::atomic<int> grad_item_override{-1};   // process-wide (diagnostic); host threads may call concurrently
void set_grad_item_override(int v) { grad_item_override.store(v, std::memory_order_relaxed); }
int get_grad_item_override() { return grad_item_override.load(std::memory_order_relaxed); }
bool grad_item_mode(const cmcd_desc& d, int T, int64_t n) {
  if (pick_grad(d, T, false, true) == nullptr) return false;
  if (const int ov = get_grad_item_override(); ov >= 0) return ov != 0;
  // measured crossover on MI355X (tools/probes/grad_item_sweep.py, dds net, K = 256): ~11k particles for the
  // reparameterised gradient (it pays the Jacobian pass), ~17k for the local one
  // the 132-wide net (3-wave workgroups, fragments from L2) is faster item-wise at every size measured (N = 2000:
  // 14.5 -> 5.2 ms, N = 16000: 39.3 -> 33.6 ms)
  if (T > 4) return n <= 65536;
  return n <= (d.mode == CMCD_MODE_CAIS_VAR_SN ? 16384 : 10240);
}
// extra floats the work-item path of the reparameterised gradient keeps: jac rows + lambda table
int64_t bptt_item_floats(const cmcd_desc& d, int64_t n) {
  const int64_t D = d.dim;
  return (int64_t)(d.nbridges + 1) * n * (D * D + 2 * D + D);
}

static void grad_offsets(const cmcd_desc& d, int HP, int64_t& o_S, int64_t& o_S2, int64_t& o_gbeta, int64_t& o_geps,
                         int64_t& o_gvd, int64_t& o_gfac, int64_t& total) {
  const int64_t K = d.nbridges;
  int64_t o = 0;
  o_S = o; o += (K + 1) * HP;
  o_S2 = o; o += (K + 1) * HP;
  o_gbeta = o; o += (K + 3) & ~int64_t(3);
  o_geps = o; o += (K + 3) & ~int64_t(3);
  o_gvd = o; o += 32;
  o_gfac = o; o += 4;
  total = o;
}
// the dds tail's per-evaluation table sits after the zero-initialised tables
static int64_t grad_tail_floats(const cmcd_desc& d) { return d.arch == CMCD_ARCH_DDS ? (int64_t)(d.nbridges + 1) * 448 : 0; }

// ---- fixed-order sums of the per-(tile, evaluation) slots (GradArgs::det) into the shared tables the tails read
struct DetArgs {
  const float* det;
  float* gtab;
  int64_t o_S, o_S2, o_gbeta, o_geps, tile_stride, obe, ntiles;
  int32_t K, HP, gef, ula, has_fwd, has_bwd;
};
// A block = 64 consecutive outputs x 4 tile classes: thread (q, l) sums the tiles q, q + 4, ... of output l in that order
// (eight loads in flight), the four partial sums are added in the order q = 0..3.  Outputs: S[row][col] (and S2) for the
// K + 1 bias-table rows, then d beta_i / d eps_i.  Consecutive outputs are consecutive floats of a slot: 256 B per wave load.
__device__ __forceinline__ void grad_det_reduce_body(const DetArgs& a, const unsigned bidx, float (*part)[64]) {
  const int l = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int64_t nS = (int64_t)(a.K + 1) * a.HP, nSS = a.gef ? 2 * nS : nS;
  const int64_t o = (int64_t)bidx * 64 + l;
  // up to two source evaluations per output row: MCD_ULA_sn reads bias-table row e - 1 at evaluation e (and row 0 at e = 0)
  int64_t src[2] = {-1, -1};
  float* dst = nullptr;
  if (o < nSS) {
    const int64_t oo = o < nS ? o : o - nS, half = o < nS ? 0 : nS;
    const int row = (int)(oo / a.HP), col = (int)(oo % a.HP);
    if (!a.ula) {
      src[0] = half + (int64_t)row * a.HP + col;
    } else {
      if (row == 0) src[0] = half + col;
      if (row + 1 <= a.K) src[1] = half + (int64_t)(row + 1) * a.HP + col;
    }
    dst = a.gtab + (o < nS ? a.o_S : a.o_S2) + oo;
  } else if (o < nSS + 2 * (int64_t)a.K) {
    const int64_t oo = o - nSS;
    const int i = (int)(oo >> 1), w = (int)(oo & 1);            // w = 0: d beta_i, 1: d eps_i
    if (a.has_fwd) src[0] = a.obe + 4 * (int64_t)i + w;          // forward-kernel part, written at evaluation i
    if (a.has_bwd) src[1] = a.obe + 4 * (int64_t)(i + 1) + 2 + w; // backward-kernel part, written at evaluation i + 1
    dst = a.gtab + (w ? a.o_geps : a.o_gbeta) + i;
  }
  float v = 0.f;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
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
__global__ __launch_bounds__(256) void grad_det_reduce_kernel(DetArgs a) {
  __shared__ float part[4][64];
  grad_det_reduce_body(a, blockIdx.x, part);
}
// geffner nets (r04): the two reductions that depend on the gradient kernel only — slabs -> grad_flat, slots -> tables — as ONE
// launch, then the two consumers of the tables (schedule tail, embedding tail) as ONE launch: 2 launches behind the gradient
// kernel instead of 3 (the fixed-order slot reduction had added one: 49 -> 54 us per gmm training iteration, 9 %)
__global__ __launch_bounds__(256) void grad_reduce_det_kernel(TailArgs a, DetArgs da, unsigned n_reduce) {
  __shared__ float part[4][64];
  if (blockIdx.x < n_reduce) grad_reduce_body(a, blockIdx.x, n_reduce);
  else grad_det_reduce_body(da, blockIdx.x - n_reduce, part);
}
__global__ __launch_bounds__(256) void grad_sched_geffner_kernel(TailArgs a) {
  if (blockIdx.x == 0) grad_sched_tail_body(a, 0, 1);
  else grad_geffner_tail_body(a, blockIdx.x - 1, gridDim.x - 1);
}


// slot floats per tile, and the cap above which the accumulation falls back to float atomics (the table is written and
// read once per gradient: 1 GB is ~0.3 ms of HBM time against the ~30 ms such a batch's sweep takes; a 2000-particle shard of
// K = 256 needs 30 MB, 65 536 particles 290 MB)
constexpr int64_t kDetCapFloats = int64_t(1) << 28;
static int64_t grad_det_tile_floats(const cmcd_desc& d, int HP) {
  const int64_t K1 = d.nbridges + 1;
  return K1 * HP * (d.arch == CMCD_ARCH_GEFFNER ? 2 : 1) + K1 * 4;
}
// CMCD_GRAD_ATOMICS=1: the float-atomic accumulation of rounds 1-3 (A/B measurements only; read once)
static bool grad_atomics_forced() {
  static const bool v = [] { const char* e = getenv("CMCD_GRAD_ATOMICS"); return e && e[0] == '1'; }();
  return v;
}
static int64_t grad_det_floats(const cmcd_desc& d, int HP, int64_t n) {
  const int64_t f = ((n + 15) / 16) * grad_det_tile_floats(d, HP);
  return f <= kDetCapFloats ? f : 0;
}

static int grad_nslabs(int64_t n, int nw, int max_slabs) {
  const int64_t nquads = (n + 16 * nw - 1) / (16 * nw);
  return (int)(nquads < max_slabs ? nquads : max_slabs);
}

int64_t grad_workspace_floats(const cmcd_desc& d, int HP, int64_t n) {
  int64_t oS, oS2, ob, oe, ov, of, tot;
  grad_offsets(d, HP, oS, oS2, ob, oe, ov, of, tot);
  const int64_t slab = (int64_t)HP * HP + HP * 16 + 4 * (2 * 16 * HP + 256 + 32);
  // up to one slab per workgroup of a full-chip launch (either path), then the fixed-order slots
  return tot + grad_tail_floats(d) + slab * 512 + grad_det_floats(d, HP, n);
}

// ws_fwd: the forward workspace as left by cmcd_bound_forward's prep on the SAME desc/params;
// gws: gradient workspace (grad_workspace_floats).  grad: [n_params], fully overwritten.
// bptt: reparameterised gradient (needs traj) vs local gradient; item: work-item path (needs traj; with bptt
// also item_ws = bptt_item_floats floats for the jac rows and the lambda table).
int grad_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, const int32_t* seeds, int64_t n,
                const float* params, int64_t n_params, const float* ws_fwd, const float* omega, float omega_scalar,
                bool bptt, bool item, const float* traj, float* item_ws, float* gws, float* grad, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  grad_fn fn = pick_grad(d, w.T, bptt, item);
  if (!fn || ((bptt || item) && !traj) || (bptt && item && !item_ws)) return CMCD_ERR_UNSUPPORTED;
  const int D = d.dim, HP = w.HP, K = d.nbridges;
  GradArgs ga{};
  int64_t tot;
  grad_offsets(d, HP, ga.o_S, ga.o_S2, ga.o_gbeta, ga.o_geps, ga.o_gvd, ga.o_gfac, tot);
  const int nw = grad_nw(w.T);
  const int64_t ntiles = (n + 15) / 16;
  const int64_t nitems = ntiles * (K + 1);
  const int64_t n_outer = (nitems + nw - 1) / nw;
  const bool small = (item ? kGradSmall : kGradSmallChain) && D == 2 && w.T <= 4;   // two workgroups per CU: twice the slabs
  const int max_slabs = small ? 512 : 256;
  const int nslabs = item ? (int)(n_outer < max_slabs ? n_outer : max_slabs) : grad_nslabs(n, nw, max_slabs);
  ga.seeds = seeds; ga.params = params; ga.ws = ws_fwd; ga.omega = omega; ga.omega_scalar = omega_scalar; ga.traj = traj; ga.gtab = gws; ga.slabs = gws + tot + grad_tail_floats(d);
  ga.lay = lay; ga.w = w; ga.n = n; ga.K = K; ga.var_mode = d.mode == CMCD_MODE_CAIS_VAR_SN ? 1 : 0;
  ga.grad_clipping = d.grad_clipping; ga.nquads = (int)((n + 16 * nw - 1) / (16 * nw));
  ga.ula = d.mode == CMCD_MODE_ULA_SN ? 2 : 0;
  ga.nitems = nitems;
  ga.slab_stride = (int64_t)HP * HP + HP * 16 + 4 * (2 * 16 * HP + 256 + 32);
  const bool det = grad_det_floats(d, HP, n) > 0 && !grad_atomics_forced();
  if (det) {
    ga.det = gws + tot + grad_tail_floats(d) + ga.slab_stride * 512;
    ga.det_tile_stride = grad_det_tile_floats(d, HP);
    ga.det_tiles = ntiles;
    ga.det_obe = ga.det_tile_stride - (int64_t)(K + 1) * 4;
  }
  if (bptt && item) {
    // (the Jacobian launch zeroes the accumulation tables and the output on its way: no memset launches on this path)
    const int64_t S = (int64_t)D * D + 2 * D;
    float* jac = item_ws;
    float* lam = item_ws + (int64_t)(K + 1) * n * S;
    const int rc = bptt_jac_scan_launch(d, lay, w, n, nitems, params, ws_fwd, traj, jac, lam, omega_scalar, gws, tot, grad,
                                        n_params, stream);
    if (rc != CMCD_OK) return rc;
    ga.lam = lam;
  } else {
    if (hipMemsetAsync(gws, 0, sizeof(float) * tot, stream) != hipSuccess) return CMCD_ERR_HIP;
    if (hipMemsetAsync(grad, 0, sizeof(float) * n_params, stream) != hipSuccess) return CMCD_ERR_HIP;
  }

  const bool tile_local = w.T > 4 || small;
  const size_t stg = tile_local ? size_t(3 * HP * 16 + 2 * 256 + 512 + (D + 1) * HP) : size_t((5 * HP + 32) * 16 + (D + 1) * HP);
  const size_t wsh = 0;   // (the shared fragment rows of the rejected WSHARE experiment would need 2 * T * 256 floats)
  const size_t lds_bytes = size_t((tile_local ? 0 : 2 * HP * HP) + 2 * D * HP + HP + 16 + w.tgt_floats + wsh + nw * stg) * 4;
  if (lds_bytes > 160 * 1024) return CMCD_ERR_UNSUPPORTED;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds_bytes) != hipSuccess)
    return CMCD_ERR_HIP;
  hipLaunchKernelGGL(fn, dim3(nslabs), dim3(64 * nw), lds_bytes, stream, ga);
  DetArgs da{};
  unsigned n_det = 0;
  const bool pair_launches = det && d.arch == CMCD_ARCH_GEFFNER;   // (see grad_reduce_det_kernel)
  if (det) {
    da.det = ga.det; da.gtab = gws; da.o_S = ga.o_S; da.o_S2 = ga.o_S2; da.o_gbeta = ga.o_gbeta; da.o_geps = ga.o_geps;
    da.tile_stride = ga.det_tile_stride; da.obe = ga.det_obe; da.ntiles = ntiles;
    da.K = K; da.HP = HP; da.gef = d.arch == CMCD_ARCH_GEFFNER ? 1 : 0; da.ula = ga.ula ? 1 : 0;
    da.has_fwd = (item || bptt) ? 1 : 0; da.has_bwd = (!bptt || item) ? 1 : 0;
    const int64_t outs = (int64_t)(K + 1) * HP * (da.gef ? 2 : 1) + 2 * (int64_t)K;
    n_det = (unsigned)((outs + 63) / 64);
    if (!pair_launches) hipLaunchKernelGGL(grad_det_reduce_kernel, dim3(n_det), dim3(256), 0, stream, da);
  }

  TailArgs ta{};
  ta.params = params; ta.ws = ws_fwd; ta.gtab = gws; ta.slabs = gws + tot + grad_tail_floats(d); ta.tail = gws + tot;
  ta.grad = grad; ta.lay = lay; ta.w = w;
  ta.o_S = ga.o_S; ta.o_S2 = ga.o_S2; ta.o_gbeta = ga.o_gbeta; ta.o_geps = ga.o_geps; ta.o_gvd = ga.o_gvd;
  ta.o_gfac = ga.o_gfac; ta.slab_stride = ga.slab_stride; ta.n_params = n_params;
  ta.K = K; ta.D = D; ta.E = d.emb_dim; ta.IN = D + d.emb_dim; ta.HP = HP; ta.arch = d.arch; ta.nslabs = nslabs;
  ta.eps_schedule = d.eps_schedule; ta.ngrid = d.ngrid; ta.nw = nw;
  {
    const int64_t wid = d.arch == CMCD_ARCH_DDS ? 64 : D + d.emb_dim;
    const int64_t outs = wid * wid + 2 * wid * D + wid + 3 * D + 1;
    const unsigned n_reduce = (unsigned)((outs * 16 + 255) / 256);
    // dds: reduction + schedule tail + the K + 1 time-coder blocks in one launch (65 -> 47 us of kernel time at N = 2000).
    // geffner: reduction + schedule tail; its embedding tail stays a launch of its own (4096 light blocks behind the
    // reduction's heavy ones in one grid measured 50 us against 24.5 + 17.9 apart)
    const unsigned n_third = d.arch == CMCD_ARCH_DDS ? (unsigned)(K + 1) : 0u;
    if (pair_launches) {
      hipLaunchKernelGGL(grad_reduce_det_kernel, dim3(n_reduce + n_det), dim3(256), 0, stream, ta, da, n_reduce);
      hipLaunchKernelGGL(grad_sched_geffner_kernel, dim3(1 + geffner_tail_blocks(ta)), dim3(256), 0, stream, ta);
      return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
    }
    hipLaunchKernelGGL(grad_tails_fused_kernel, dim3(n_reduce + 1 + n_third), dim3(256), 0, stream, ta, n_reduce, n_third);
  }
  if (d.arch == CMCD_ARCH_DDS)
    hipLaunchKernelGGL(grad_dds_tail_sum_kernel, dim3(((64 + 4096 + 64 + 4096 + 64 + 8192 + 64) * 16 + 255) / 256), dim3(256), 0, stream, ta);
  else
    hipLaunchKernelGGL(grad_geffner_tail_kernel, dim3(geffner_tail_blocks(ta)), dim3(256), 0, stream, ta);
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

typedef void (*ula_fn)(UlaGradArgs);
static ula_fn pick_ula(const cmcd_desc& d) {
  if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return ula_grad_kernel<CMCD_TARGET_MANY_GMM, 2>;
  if (d.target == CMCD_TARGET_GMM && d.dim == 2) return ula_grad_kernel<CMCD_TARGET_GMM, 2>;
  if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return ula_grad_kernel<CMCD_TARGET_FUNNEL, 10>;
  return nullptr;
}
bool ula_grad_available(const cmcd_desc& d) { return pick_ula(d) != nullptr; }
static int64_t ula_det_floats(int64_t K4, int64_t ntiles) {
  const int64_t f = ntiles * 2 * K4;
  return f <= kDetCapFloats ? f : 0;
}
int64_t ula_grad_workspace_floats(const cmcd_desc& d, int64_t n) {
  const int64_t K4 = ((int64_t)d.nbridges + 3) & ~int64_t(3), ntiles = (n + 15) / 16;
  return K4 * 2 + ntiles * 2 * d.dim + 8 + ula_det_floats(K4, ntiles);
}

// MCD_ULA: reverse sweep without a network.  gws: ula_grad_workspace_floats; ws_fwd / traj as left by the forward.
int ula_grad_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, int64_t n, const float* params,
                    int64_t n_params, const float* ws_fwd, const float* traj, float* gws, float omega, float* grad,
                    void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  ula_fn fn = pick_ula(d);
  if (!fn) return CMCD_ERR_UNSUPPORTED;
  const int64_t K4 = ((int64_t)d.nbridges + 3) & ~int64_t(3), ntiles = (n + 15) / 16;
  if (hipMemsetAsync(grad, 0, sizeof(float) * n_params, stream) != hipSuccess) return CMCD_ERR_HIP;
  if (hipMemsetAsync(gws, 0, sizeof(float) * 2 * K4, stream) != hipSuccess) return CMCD_ERR_HIP;
  const bool det = ula_det_floats(K4, ntiles) > 0 && !grad_atomics_forced();
  float* det_rows = det ? gws + 2 * K4 + ntiles * 2 * d.dim + 8 : nullptr;
  UlaGradArgs a{params, ws_fwd, traj, gws, gws + 2 * K4, det_rows, lay, w, n, 0, K4, d.nbridges, omega};
  hipLaunchKernelGGL(fn, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), (size_t)(w.tgt_floats + 4) * 4, stream, a);
  if (det)
    hipLaunchKernelGGL(tile_rows_reduce_kernel, dim3((unsigned)((2 * K4 * 16 + 255) / 256)), dim3(256), 0, stream, det_rows,
                       ntiles, 2 * K4, K4, (int64_t)d.nbridges, gws);
  hipLaunchKernelGGL(ula_vd_reduce_kernel, dim3(1), dim3(64), 0, stream, gws + 2 * K4, ntiles, d.dim, grad, lay.vd_mean,
                     lay.vd_logdiag);
  TailArgs ta{};
  ta.params = params; ta.gtab = gws; ta.grad = grad; ta.lay = lay; ta.w = w; ta.o_gbeta = 0; ta.o_geps = K4;
  ta.K = d.nbridges; ta.D = d.dim; ta.eps_schedule = CMCD_EPS_CONST; ta.ngrid = d.ngrid;
  hipLaunchKernelGGL(grad_sched_tail_kernel, dim3(1), dim3(256), 0, stream, ta);
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

__global__ __launch_bounds__(64) void grad_dds_tail_kernel(TailArgs a) { grad_dds_tail_body(a, blockIdx.x, gridDim.x); }

// The particle-independent tails for a network whose state inputs are `din` wide (din = 2 dim for the momentum mode's
// concat(z, rho); cmcd_uha.hip): schedule tail (d eps0, d mgridref_y from the per-bridge tables), then the embedding
// table / W1[din:] / b1 (geffner) or the time coder (dds; `dds_tail` = (K + 1) * 448 floats of scratch).
int launch_net_tails(const cmcd_desc& d, int din, int eps_schedule, const cmcd_layout& lay, const WsLayout& w,
                     const float* params, const float* gtab, int64_t o_S, int64_t o_S2, int64_t o_gbeta, int64_t o_geps,
                     int HP, float* dds_tail, float* grad, void* stream_) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TailArgs ta{};
  ta.params = params; ta.gtab = gtab; ta.grad = grad; ta.lay = lay; ta.w = w; ta.tail = dds_tail;
  ta.o_S = o_S; ta.o_S2 = o_S2; ta.o_gbeta = o_gbeta; ta.o_geps = o_geps;
  ta.K = d.nbridges; ta.D = din; ta.E = d.emb_dim; ta.IN = din + d.emb_dim; ta.HP = HP; ta.arch = d.arch;
  ta.eps_schedule = eps_schedule; ta.ngrid = d.ngrid;
  hipLaunchKernelGGL(grad_sched_tail_kernel, dim3(1), dim3(256), 0, stream, ta);
  if (d.arch == CMCD_ARCH_DDS) {
    if (!dds_tail) return CMCD_ERR_BAD_ARG;
    hipLaunchKernelGGL(grad_dds_tail_kernel, dim3((unsigned)(d.nbridges + 1)), dim3(64), 0, stream, ta);
    hipLaunchKernelGGL(grad_dds_tail_sum_kernel, dim3(((64 + 4096 + 64 + 4096 + 64 + 8192 + 64) * 16 + 255) / 256), dim3(256), 0, stream, ta);
  } else {
    hipLaunchKernelGGL(grad_geffner_tail_kernel, dim3(geffner_tail_blocks(ta)), dim3(256), 0, stream, ta);
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

int launch_geffner_tails(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, const float* params,
                         const float* gtab, int64_t o_S, int64_t o_S2, int64_t o_gbeta, int64_t o_geps, int HP,
                         float* grad, void* stream_, bool with_net) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  TailArgs ta{};
  ta.params = params; ta.gtab = gtab; ta.grad = grad; ta.lay = lay; ta.w = w;
  ta.o_S = o_S; ta.o_S2 = o_S2; ta.o_gbeta = o_gbeta; ta.o_geps = o_geps;
  ta.K = d.nbridges; ta.D = d.dim; ta.E = d.emb_dim; ta.IN = d.dim + d.emb_dim; ta.HP = HP; ta.arch = d.arch;
  ta.eps_schedule = d.eps_schedule; ta.ngrid = d.ngrid;
  hipLaunchKernelGGL(grad_sched_tail_kernel, dim3(1), dim3(256), 0, stream, ta);
  if (with_net) hipLaunchKernelGGL(grad_geffner_tail_kernel, dim3(geffner_tail_blocks(ta)), dim3(256), 0, stream, ta);
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

}  // namespace cmcd
