// coop_wide8_kernel — the CU-cooperative trajectory kernel for WIDE states (d = 10: the funnel; d = 12 fits) on 8-particle
// tiles: one workgroup per tile, T + 4 waves with the roles of coop_kernel (cmcd_coop.hip) — but laid out for a state that
// is too wide to be carried redundantly.  r04's funnel instance was coop_kernel<.., D = 10, .., HALF> with its 2-d habits:
// every lane of the accounting wave formed all ten coordinates of z_{i+1} (eight-fold redundant, ~250 instructions behind
// barrier 2) and published them behind a THIRD barrier, the key chain ran three Threefry passes per bridge on a wave whose
// twin columns repeated each other, and the ten deviates were three dependent erfinv evaluations on the same wave
// (profiles/r05_stamps_funnel_8tile_before.txt: 3 513 stamped cycles per bridge, 97.0 us per call of N = 300, K = 64).
// Here every per-coordinate job is DEALT to the lanes of its particle, and the state never leaves the waves that use it
// (profiles/r05_stamps_funnel_8tile_after.txt: 2 040 cycles, 56.5 us):
//
//   waves 0..T-1  MLP   lane (qi, pg, kh, ng) = particle 4 pg + qi, neurons 16 wv + 4 ng + 2 kh + {0, 1} (v_mfma_f32_4x4x1
//                       order, as coop_kernel on 8-particle tiles).  Behind barrier 2 each lane takes the step of ITS
//                       coordinates {s8, s8 + 8} (s8 = ng + 4 kh: sums the layer-3 partials of the T waves, fk = base - eps s,
//                       z' = fk + sigma noise) and the particle's 8 lanes exchange their coordinates inside the wave (one DPP
//                       rotate + row swaps): every MLP wave holds all of z_{i+1} with no LDS hand-over and no third barrier.
//                       Layers 1 and 3 run on packed fp32 pairs (the lane's two neurons / two outputs per instruction);
//                       layer 3 ends in a reduce-SCATTER over the particle's 8 lanes (1 DPP add + 2 row-swap adds per four
//                       outputs) instead of an all-reduce per output pair, each lane writing the output it ends up owning.
//   waves T, T+1  TGT   16 lanes per particle, lane `sub` owns coordinate `sub`: takes the same step for its coordinate (same
//                       function, same bits), accumulates the log-weight terms of that coordinate (the backward kernel of the
//                       step before, the forward kernel of this one) in a per-lane sum that is added over the particle's lanes
//                       once, at the end of the launch; then its share of grad log p(z_{i+1}) (the funnel's sum of squares is
//                       one 16-lane reduction), the clipped scores and base_j of the next forward mean.
//   wave  T+2     RNG   the jax Threefry key chain one bridge ahead, 8 lanes per particle: pass A = split(gen) (2 blocks),
//                       pass B = split(H) (2 blocks) + the d / 2 blocks of normal(G, (d,)) in ONE pass (7 of the 8 lanes
//                       busy) — two dependent passes per bridge, the minimum the chain allows.  The ten groups of four rounds
//                       of the two passes are cut at kCut between the two intervals of the bridge, so that the wave is not
//                       the last to arrive at either barrier.
//   wave  T+3     ACC   random words -> Gaussian deviates (Giles' erfinv), 8 lanes per particle, lane s8 converts words
//                       {s8, s8 + 8}: one conversion per interval; at the end of the launch, the tile's statistics record.
//
// Per evaluation i, TWO raw s_barriers:  [step -> z_i] interval 1 (layer 1 | grad log p | chain segment | conversion) —
// barrier 1 — interval 2 (layers 2, 3 | base | chain segment | conversion) — barrier 2.  Every exchange row is single-buffered
// except the raw words (written one bridge ahead): a row is written in the interval AFTER the barrier behind which its last
// reader finished.  Same arithmetic as coop_kernel / traj_kernel up to the association of the sums over coordinates and
// neurons, e^{-v} through v_exp_f32 and the funnel's constants as reciprocals; the PRNG path is bit-exact
// (tests/test_gpu_prng.py).  Reference lines: /root/reference/src/mcd_cais.py:46-89, src/mcdboundingmachine.py:151-179,
// src/model_handler.py:124-143, src/nn.py:42-72; cited per statement in cmcd_kernels.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

#ifdef CMCD_STAMPS   // diagnostic build only (tools/probes/stamp_probe.py)
__device__ unsigned long long g_stamps_wide[16][16];
#define WSTAMP(slot)                                                                     \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    st_acc[slot] += t_ - st_last;                                                        \
    st_last = t_;                                                                        \
  } while (0)
#else
#define WSTAMP(slot)
#endif

namespace {

__device__ __forceinline__ void wbar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Threefry-2x32 cut into its five groups of four rounds (cmcd_device.h: threefry2x32 is the same sequence in one piece)
struct TfState {
  uint32_t k0, k1, k2, x0, x1;
};
__device__ __forceinline__ void tf_begin(TfState& t, uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) {
  t.k0 = k0; t.k1 = k1; t.k2 = k0 ^ k1 ^ 0x1BD11BDAu;
  t.x0 = c0 + k0; t.x1 = c1 + k1;
}
template <int G>
__device__ __forceinline__ void tf_group(TfState& t) {
#define CMCD_TFW_ROUND(r) \
  t.x0 += t.x1;           \
  t.x1 = rotl32(t.x1, r); \
  t.x1 ^= t.x0;
  if (G % 2 == 0) { CMCD_TFW_ROUND(13) CMCD_TFW_ROUND(15) CMCD_TFW_ROUND(26) CMCD_TFW_ROUND(6) }
  else            { CMCD_TFW_ROUND(17) CMCD_TFW_ROUND(29) CMCD_TFW_ROUND(16) CMCD_TFW_ROUND(24) }
#undef CMCD_TFW_ROUND
  if (G == 0) { t.x0 += t.k1; t.x1 += t.k2 + 1u; }
  if (G == 1) { t.x0 += t.k2; t.x1 += t.k0 + 2u; }
  if (G == 2) { t.x0 += t.k0; t.x1 += t.k1 + 3u; }
  if (G == 3) { t.x0 += t.k1; t.x1 += t.k2 + 4u; }
  if (G == 4) { t.x0 += t.k2; t.x1 += t.k0 + 5u; }
}
__device__ __forceinline__ void tf_all(TfState& t) {
  tf_group<0>(t); tf_group<1>(t); tf_group<2>(t); tf_group<3>(t); tf_group<4>(t);
}

// lanes 8 .. 15 of every 16-lane row take the value of lanes 0 .. 7 (DPP row_ror:8 into banks 2, 3 only)
__device__ __forceinline__ uint32_t twin_from_low(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x128, 0xf, 0xC, false);
}

// reduce-scatter over the four rows of the wave (values of one column):
//   rs32(a, b): rows 0, 1 <- a(row r) + a(row r + 2);  rows 2, 3 <- b(row r - 2) + b(row r)
//   rs16(p, q): rows 0, 2 <- p(row r) + p(row r + 1);  rows 1, 3 <- q(row r - 1) + q(row r)
__device__ __forceinline__ float rs32(float a, float b) {
  uint32_t r0, r1;
  swap32(__float_as_uint(a), __float_as_uint(b), r0, r1);
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float rs16(float p, float q) {
  uint32_t r0, r1;
  swap16(__float_as_uint(p), __float_as_uint(q), r0, r1);
  return __uint_as_float(r0) + __uint_as_float(r1);
}

}  // namespace

// the RNG wave's ten round groups per bridge (pass A: 0 - 4, pass B: 5 - 9): [0, kCut) in interval 1, the rest in interval 2
#ifndef CMCD_WIDE_CUT
#define CMCD_WIDE_CUT 6
#endif
#ifndef CMCD_WIDE_PK
#define CMCD_WIDE_PK 1
#endif

// s(z_e, e)_j from the layer-3 partials of the T MLP waves and z_{e+1, j} = fk + sigma noise, fk = base - eps s: ONE function
// for every wave that keeps a copy of the state (MLP lanes: two coordinates each, TGT lanes: one), so that all copies are
// the same bits.   /root/reference/src/mcd_cais.py:61-67, src/nn.py:70, src/nn_dds.py:162
template <int ARCH, int T>
__device__ __forceinline__ void wide_step(const float* prow, float b3, float factor, float base, float noise, float seps,
                                          float sig, float& sn, float& zn) {
#pragma clang fp contract(off)
  float o;
  if (T == 4) {
    const f32x4 t4 = *reinterpret_cast<const f32x4*>(prow);
    o = b3 + ((t4[0] + t4[1]) + (t4[2] + t4[3]));
  } else {
    const f32x2 t2 = *reinterpret_cast<const f32x2*>(prow);
    o = b3 + (t2[0] + t2[1]);
  }
  sn = (ARCH == CMCD_ARCH_DDS) ? __builtin_amdgcn_fmed3f(o, -1e4f, 1e4f) : o * factor;
  zn = __builtin_fmaf(sig, noise, __builtin_fmaf(seps, sn, base));
}

template <int ARCH, int D, int T>
__global__ __launch_bounds__(64 * (T + 4)) void coop_wide8_kernel(TrajArgs a) {
  static_assert(D == 10 || D == 12, "7 or 8 Threefry blocks per particle on 8 lanes; coordinates {s8, s8 + 8} per 8-lane group");
  static_assert(T == 2 || T == 4, "partials of one coordinate are one 8- or 16-byte LDS read");
  constexpr int HP = 16 * T, NR = 2, Hh = D / 2, DP = (D + 3) & ~3;
  constexpr int HQP = HP + 4, NQ = HP / 2, RSA = ((HP / 2 + 15) / 16) * 4;
  constexpr int PTW = (D * T + 3) & ~3;           // layer-3 partials of one particle: [j][wave]
  constexpr int kCut = CMCD_WIDE_CUT;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const hbuf = lds;                        // [8][HQP]   layer-1 activations                          (MLP -> MLP)
  float* const part = hbuf + 8 * HQP;             // [8][PTW]   layer-3 partials [j][wave]                   (MLP -> MLP, TGT)
  float* const baseb = part + 8 * PTW;            // [8][DP]    base_j of the forward mean                   (TGT -> MLP)
  float* const nzb = baseb + 8 * DP;              // [8][DP]    deviates of the bridge                       (ACC -> MLP, TGT)
  float* const zpub = nzb + 8 * DP;               // [8][DP]    z_0 (prologue only)                          (ACC -> MLP, TGT)
  float* const lossb = zpub + 8 * DP;             // [8]        per-particle loss at the end                 (TGT -> ACC)
  uint32_t* const raw = reinterpret_cast<uint32_t*>(lossb + 8);   // [2][8][DP] random words, one bridge ahead (RNG -> ACC)

  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const bool is_mlp = wv < T, is_tgt = wv == T || wv == T + 1, is_rng = wv == T + 2, is_acc = wv == T + 3;
  const int64_t tile = blockIdx.x;
  const int K = a.K;
  switch ((a.prio >> (is_mlp ? 0 : is_tgt ? 2 : is_rng ? 4 : 6)) & 3) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }
#ifdef CMCD_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
#endif
  // the per-bridge tables into this XCD's L2 while the prologue runs (cmcd_coop.hip: the prep launch has just rewritten them)
  {
    float warm = 0.f;
    const int64_t t0 = a.w.sched;
    const int64_t t1 = (ARCH == CMCD_ARCH_GEFFNER ? a.w.utab : a.w.bias1) + (int64_t)(K + 1) * HP;
    const int64_t per_xcd = (gridDim.x + 7) >> 3, rank = blockIdx.x >> 3;     // dealt to the workgroups that share an XCD
    for (int64_t i = t0 + 32 * (rank * blockDim.x + threadIdx.x); i < t1; i += 32 * per_xcd * blockDim.x) warm += a.ws[i];
    asm volatile("" ::"v"(warm));
  }
  const float clipv = a.var_mode ? 1e2f : 1e3f;
  const bool clip_p = a.grad_clipping != 0, clip_q = clip_p && a.var_mode;
  const float cp = clip_p ? clipv : INFINITY, cq = clip_q ? clipv : INFINITY;
  const float* const sched_p = a.ws + a.w.sched;
  // the schedule row of evaluation i: {beta, eps, sigma, log sigma + log sqrt(2 pi) | 1 / (2 sigma^2), eps beta, eps (1 - beta), 0},
  // requested as a scalar load and complete behind barrier 1's own wait, through which the values pass (cmcd_coop.hip)
#define CMCD_WIDE_SCHED_LOAD(i_)                                                                              \
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};                                                 \
  {                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);   /* not above the LDS reads of the step: their wait would include this load */ \
    const float* rowp = sched_p + __builtin_amdgcn_readfirstlane(8 * ((i_) < K ? (i_) : K - 1));              \
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x10" : "=&s"(sc), "=&s"(sd) : "s"(rowp)); \
    __builtin_amdgcn_sched_barrier(0);                                                                        \
  }
#define CMCD_WIDE_BAR1_SCHED() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(sc), "+s"(sd)::"memory")

  // =============================================================================================== MLP
  if (is_mlp) {
    const int pc = lane & 7, kh = (lane >> 3) & 1, ng = lane >> 4;
    const int nb = 16 * wv + 4 * ng + 2 * kh;
    float aq[NQ], b2p[NR], w1[D][NR], w3s[D][NR];
#pragma unroll
    for (int q = 0; q < NQ; ++q) aq[q] = a.ws[a.w.w2q + (int64_t)(wv * NQ + q) * 64 + lane];
#pragma unroll
    for (int r = 0; r < NR; ++r) b2p[r] = a.ws[a.w.b2 + nb + r];
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        w1[j][r] = a.ws[a.w.w1z + j * HP + nb + r];
        // outputs in swapped pairs on the kh = 1 lanes: slot j holds output j ^ kh, so that the first stage of the layer-3
        // reduce-scatter is `own slot 2m + partner's slot 2m + 1` on both lanes of a pair (no select)
        w3s[j][r] = a.ws[a.w.w3t + (j ^ kh) * HP + nb + r];
      }
    }
    auto load_row = [&](const float* ptr) -> f32x2 { return *reinterpret_cast<const f32x2*>(ptr); };
    f32x2 brow = load_row(a.ws + a.w.bias1 + nb), urow = {0.f, 0.f};
    if (ARCH == CMCD_ARCH_GEFFNER) urow = load_row(a.ws + a.w.utab + nb);
    float* const my_h = hbuf + pc * HQP + nb;
    const float* const rd_h = hbuf + pc * HQP + (HP / 2) * kh + RSA * ng;
    // the coordinates whose step this lane takes (8 lanes per particle): jA = s8 always, jB = s8 + 8 on lanes s8 < D - 8
    const int s8 = ng + 4 * kh;
    const int jA = s8, jB = s8 + 8 < D ? s8 + 8 : s8;
    const float b3A = a.ws[a.w.b3 + jA], b3B = a.ws[a.w.b3 + jB], factor = a.ws[a.w.b3 + 15];
    const float* const rd_pA = part + pc * PTW + jA * T;
    const float* const rd_pB = part + pc * PTW + jB * T;
    // the first D neurons of the geffner residual stream are z itself: wave 0, neurons nb, nb + 1 < D
    const bool z_in_u = ARCH == CMCD_ARCH_GEFFNER && wv == 0 && nb < D;
    // which output this lane owns after the reduce-scatter of a quad of pair-slots: row g -> slot {0, 2, 1, 3}[g]
    const int own_slot = ((ng & 1) << 1) | (ng >> 1);
    float z[D], uz[2] = {0.f, 0.f};
    auto residual_z = [&]() {   // uz = (z[nb], z[nb + 1]) on the lanes of wave 0 whose neurons are coordinates (nb = 4 ng + 2 kh)
      if (ARCH == CMCD_ARCH_GEFFNER && wv == 0) {
        const float e0 = kh ? z[2] : z[0], e1 = kh ? z[3] : z[1];
        const float f0 = kh ? z[6] : z[4], f1 = kh ? z[7] : z[5];
        const float g0 = (D > 10 && kh) ? z[D > 10 ? 10 : 0] : z[8], g1 = (D > 10 && kh) ? z[D > 10 ? 11 : 0] : z[9];
        const float u0 = ng == 0 ? e0 : (ng == 1 ? f0 : g0), u1 = ng == 0 ? e1 : (ng == 1 ? f1 : g1);
        uz[0] = z_in_u ? u0 : 0.f;
        uz[1] = z_in_u ? u1 : 0.f;
      }
    };
    wbar();   // P1: random words of z_0
    wbar();   // P3: z_0 published
#pragma unroll
    for (int q = 0; q < DP; q += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(zpub + pc * DP + q);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (q + r < D) z[q + r] = v[r];
    }
    residual_z();
    float sepsP = 0.f, sigP = 0.f;   // -eps and sigma of the previous evaluation's row
#ifdef CMCD_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int i = 0; i <= K; ++i) {
      if (i > 0) {
        // ---- z_i from evaluation i - 1 (everything it needs was published before barrier 2): this lane takes the step of its
        // own coordinates, then the particle's 8 lanes exchange theirs inside the wave (DPP + row swaps: no LDS, no barrier)
        const float baseA = baseb[pc * DP + jA], baseB = baseb[pc * DP + jB];
        const float nzA = nzb[pc * DP + jA], nzB = nzb[pc * DP + jB];
        float snA, snB, znA, znB;
        wide_step<ARCH, T>(rd_pA, b3A, factor, baseA, nzA, sepsP, sigP, snA, znA);
        wide_step<ARCH, T>(rd_pB, b3B, factor, baseB, nzB, sepsP, sigP, snB, znB);
        const float ax = xor8(znA);                 // coordinate ng + 4 (1 - kh)
        const float lo = kh ? ax : znA;             // coordinate ng
        const float hi = kh ? znA : ax;             // coordinate ng + 4
        uint32_t r4[4];
        rows0123(__float_as_uint(lo), r4);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = __uint_as_float(r4[r]);
        rows0123(__float_as_uint(hi), r4);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[4 + r] = __uint_as_float(r4[r]);
        const float bx = xor8(znB);
        const float bb = kh ? bx : znB;             // coordinate 8 + ng, from the kh = 0 lane of the pair
        if (D > 10) {
          rows0123(__float_as_uint(bb), r4);
#pragma unroll
          for (int r = 0; r < D - 8; ++r) z[8 + r] = __uint_as_float(r4[r]);
        } else {
          uint32_t r0, r1;
          rows01(__float_as_uint(bb), r0, r1);
          z[8] = __uint_as_float(r0);
          z[9] = __uint_as_float(r1);
        }
        residual_z();
      }
      WSTAMP(6);
      CMCD_WIDE_SCHED_LOAD(i);
      // ---- interval 1: layer 1
      float pre[NR], h[NR];
#if CMCD_WIDE_PK
      {   // the lane's two neurons as one packed pair: v_pk_fma_f32, one issue slot per coordinate instead of two
        f32x2 pv = brow;
#pragma unroll
        for (int j = 0; j < D; ++j) pv = __builtin_elementwise_fma(f32x2{z[j], z[j]}, f32x2{w1[j][0], w1[j][1]}, pv);
        pre[0] = pv[0];
        pre[1] = pv[1];
      }
#else
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        pre[r] = brow[r];
#pragma unroll
        for (int j = 0; j < D; ++j) pre[r] = fmaf(z[j], w1[j][r], pre[r]);
      }
#endif
#pragma unroll
      for (int r = 0; r < NR; ++r)
        h[r] = (ARCH == CMCD_ARCH_DDS) ? gelu_fast(pre[r]) : (urow[r] + uz[r]) + softplus(pre[r]);
      *reinterpret_cast<float2*>(my_h) = float2{h[0], h[1]};
      WSTAMP(0);
      CMCD_WIDE_BAR1_SCHED();   // barrier 1
      WSTAMP(1);
      sepsP = a.ula ? 0.f : -sc[1];
      sigP = sc[2];
      // ---- interval 2: layers 2 and 3
      const int nrow = (i < K ? i + 1 : K) - (a.ula == 2 ? 1 : 0);   // CAIS: s(z_{i+1}, i + 1); MCD_ULA_sn: s(z_{i+1}, i)
      brow = load_row(a.ws + a.w.bias1 + (int64_t)nrow * HP + nb);
      if (ARCH == CMCD_ARCH_GEFFNER) urow = load_row(a.ws + a.w.utab + (int64_t)nrow * HP + nb);
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
      float av[NR], h2[NR];
      av[0] = ((kh ? acc[2] : acc[0]) + xor8(kh ? acc[0] : acc[2])) + b2p[0];
      av[1] = ((kh ? acc[3] : acc[1]) + xor8(kh ? acc[1] : acc[3])) + b2p[1];
#pragma unroll
      for (int r = 0; r < NR; ++r) h2[r] = (ARCH == CMCD_ARCH_DDS) ? gelu_fast(av[r]) : h[r] + softplus(av[r]);
      WSTAMP(7);
      // layer 3: this lane's two neurons against all D outputs, then the sum over the particle's 8 lanes as a
      // reduce-scatter: stage 1 over kh (DPP), stages 2 and 3 over the four rows (row-swap instructions)
      float ps[D];
#if CMCD_WIDE_PK
#pragma unroll
      for (int m = 0; m < Hh; ++m) {   // outputs in packed pairs
        const f32x2 pv = __builtin_elementwise_fma(f32x2{h2[1], h2[1]}, f32x2{w3s[2 * m][1], w3s[2 * m + 1][1]},
                                                   f32x2{h2[0], h2[0]} * f32x2{w3s[2 * m][0], w3s[2 * m + 1][0]});
        ps[2 * m] = pv[0];
        ps[2 * m + 1] = pv[1];
      }
#else
#pragma unroll
      for (int j = 0; j < D; ++j) ps[j] = fmaf(h2[1], w3s[j][1], h2[0] * w3s[j][0]);
#endif
      constexpr int DQ = (Hh + 3) / 4;   // quads of pair-slots
      float vq[4 * DQ];
#pragma unroll
      for (int m = 0; m < 4 * DQ; ++m) vq[m] = m < Hh ? ps[2 * m] + xor8(ps[2 * m + 1]) : 0.f;   // pair-slot m = output 2 m + kh
#pragma unroll
      for (int q = 0; q < DQ; ++q) {
        const float w01 = rs32(vq[4 * q], vq[4 * q + 1]), w23 = rs32(vq[4 * q + 2], vq[4 * q + 3]);
        const float tot = rs16(w01, w23);    // row 0: slot 4q, row 1: slot 4q + 2, row 2: slot 4q + 1, row 3: slot 4q + 3
        const int m = 4 * q + own_slot;
        if (m < Hh) part[pc * PTW + (2 * m + kh) * T + wv] = tot;
      }
      WSTAMP(2);
      wbar();   // barrier 2
      WSTAMP(3);
    }
    wbar();   // F1: per-particle losses handed to the ACC wave
  }
  // =============================================================================================== TGT
  else if (is_tgt) {
    const int c4 = lane & 3, sub = lane >> 2;            // 4 particles per wave, 16 lanes each
    const int pc = 4 * (wv - T) + c4;
    const int64_t p = tile * 8 + pc;
    const bool valid = p < a.n;
    const int j = sub < D ? sub : 0;                      // the lane's coordinate (lanes sub >= D idle along on coordinate 0)
    const bool act = sub < D;
    const float qmean = a.params[a.lay.vd_mean + j];
    const float qstd = expf(a.params[a.lay.vd_logdiag + j]);
    const float qiv = 1.0f / (qstd * qstd);
    const float b3j = a.ws[a.w.b3 + j], factor = a.ws[a.w.b3 + 15];
    const float* const rd_p = part + pc * PTW + j * T;
    wbar();   // P1
    wbar();   // P3
    float zj = zpub[pc * DP + j], v = zpub[pc * DP];
    // this lane's share of the log-weight: coordinate j of w = -log q(z_0) + sum_i [log N(z_i; bk_i, sigma_i) - log N(z_{i+1}; fk_i, sigma_i)]
    float wl = 0.f, zp = 0.f, lp = 0.f;
    {
      const float dz = zj - qmean;                        // -log q(z_0)      diag_gauss.py:49-62, mcdboundingmachine.py:157
      wl = (dz * dz) / (2.0f * qstd * qstd) + logf(qstd) + kHalfLog2Pi;
    }
    float gpc = 0.f, gqc = 0.f, base = 0.f;
    float seps = 0.f, sig = 0.f, cst = 0.f, inv2s2 = 0.f, cA = 0.f, cB = 0.f, epsv = 0.f;   // row of the current evaluation
    float peps = 0.f, pinv2s2 = 0.f, pcst = 0.f, pA = 0.f, pB = 0.f;                         // row of the previous step
    // closes evaluation e (its partials, base and noise are published): s(z_e, e)_j, the backward kernel of step e - 1
    // [bk = z - eps ub + eps s(z_e, e), ub = -(beta gp + (1 - beta) gq) at z_e] and, for e < K, the forward kernel of step e
    // [fk = base - eps s, z_{e+1} = fk + sigma noise]: the log-weight terms of coordinate j      mcd_cais.py:52-86
    auto finish = [&](int e) {
      float sn, zn;
      wide_step<ARCH, T>(rd_p, b3j, factor, base, nzb[pc * DP + j], seps, sig, sn, zn);
      if (e > 0) {
        const float bk = fmaf(pA, gpc, fmaf(pB, gqc, fmaf(peps, sn, zj)));
        const float db = zp - bk;
        wl += -(db * db) * pinv2s2 - pcst;
      }
      if (e < K) {
        const float fk = fmaf(seps, sn, base);
        const float df = zn - fk;
        wl -= -(df * df) * inv2s2 - cst;
        zp = zj;
        zj = zn;
        v = part_sum<16>(sub == 0 ? zn : 0.f);            // coordinate 0 to all 16 lanes of the particle
        if (a.traj && valid && act) a.traj[((int64_t)(e + 1) * a.n + p) * D + j] = zn;
        peps = epsv; pinv2s2 = inv2s2; pcst = cst; pA = cA; pB = cB;
      }
    };
#ifdef CMCD_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int i = 0; i <= K; ++i) {
      if (i > 0) finish(i - 1);
      WSTAMP(6);
      CMCD_WIDE_SCHED_LOAD(i);
      // funnel (/root/reference/src/model_handler.py:124-143; the scale of v is the hard-coded 3.0):
      //   log p = logN(v; 0, 3) + sum_{j >= 1} logN(z_j; 0, e^{v / 2})
      //   d / dv = -v / 9 - (d - 1) / 2 + e^{-v} ss / 2,   d / dz_j = -z_j e^{-v},   ss = sum_{j >= 1} z_j^2
      // e^{-v} through v_exp_f32, the constants as reciprocals (no IEEE division on these waves).
      const float ss = part_sum<16>((act && j >= 1) ? zj * zj : 0.f);
      const float emv = __builtin_amdgcn_exp2f(-1.44269504088896340736f * v);
      constexpr float c0 = -0.5f * kLog2Pi - 1.0986122886681098f, c1 = -0.5f * (D - 1) * kLog2Pi;
      const float hes = 0.5f * emv * ss;
      const float g0 = fmaf(v, -1.0f / 9.0f, hes - 0.5f * (D - 1));
      const float gp = j == 0 ? g0 : -zj * emv;
      gpc = __builtin_amdgcn_fmed3f(gp, -cp, cp);
      gqc = __builtin_amdgcn_fmed3f((qmean - zj) * qiv, -cq, cq);
      lp = fmaf(v * v, -1.0f / 18.0f, c0 + c1) - 0.5f * (D - 1) * v - hes;
      WSTAMP(0);
      CMCD_WIDE_BAR1_SCHED();   // barrier 1: every copy of z_i has been formed, base_{i-1} may be overwritten
      WSTAMP(1);
      epsv = sc[1]; sig = sc[2]; cst = sc[3]; inv2s2 = sd[0]; cA = sd[1]; cB = sd[2];
      seps = a.ula ? 0.f : -epsv;
      base = fmaf(cA, gpc, fmaf(cB, gqc, zj));   // base_j = z_j + eps beta clip(gp_j) + eps (1 - beta) clip(gq_j)
      if (act) baseb[pc * DP + j] = base;
      WSTAMP(2);
      wbar();   // barrier 2
      WSTAMP(3);
    }
    finish(K);
    // w = sum over coordinates + log p(z_K); loss = -w                              mcdboundingmachine.py:178-179
    const float wtot = part_sum<16>(act ? wl : 0.f) + lp;
    if (sub == 0) lossb[pc] = -wtot;
    if (valid && act) a.out_z[p * D + j] = zj;
    wbar();   // F1
  }
  // =============================================================================================== RNG
  else if (is_rng) {
    const int c = lane & 15, g = lane >> 4, pc = c & 7, tw = c >> 3, s8 = g + 4 * tw;
    const int64_t p = tile * 8 + pc;
    const bool valid = p < a.n;
    const uint32_t gb = g & 1;
    const bool is_split = s8 < 2;                 // pass B: lanes 0, 1 of a particle run split(H), lanes 2 .. 2 + Hh - 1 the normal blocks
    const int jn = s8 - 2;
    const bool writes = jn >= 0 && jn < Hh;
    const uint32_t cb0 = is_split ? (uint32_t)s8 : (uint32_t)(writes ? jn : 0);
    const uint32_t cb1 = is_split ? (uint32_t)(2 + s8) : (uint32_t)(writes ? Hh + jn : 0);
    uint32_t k0 = 0, k1 = 0;
    const int32_t seed = a.seeds[valid ? p : a.n - 1];
    TfState t;
    // (A, B) = split(PRNGKey(seed))                                   mcdboundingmachine.py:151-152
    tf_begin(t, 0u, (uint32_t)seed, gb, 2 + gb);
    tf_all(t);
    uint32_t a0, a1, b0, b1;
    rows01(t.x0, a0, a1);
    rows01(t.x1, b0, b1);
    // one pass: z_0 words = bits of normal(A, (D,)) on the normal lanes, C = first(split(B)) on the split lanes      :153-158
    tf_begin(t, is_split ? b0 : a0, is_split ? b1 : a1, cb0, cb1);
    tf_all(t);
    if (writes) {
      raw[(8 + pc) * DP + jn] = t.x0;
      raw[(8 + pc) * DP + Hh + jn] = t.x1;
    }
    uint32_t c0, c1;
    rows01(t.x0, c0, c1);
    c0 = twin_from_low(c0);
    c1 = twin_from_low(c1);
    // gen_0 = second(split(C))                                          mcd_cais.py:94
    tf_begin(t, c0, c1, gb, 2 + gb);
    tf_all(t);
    rows01(t.x1, k0, k1);
    if (a.dbg_keys && valid && s8 == 0) {
      a.dbg_keys[p * 2] = k0;
      a.dbg_keys[p * 2 + 1] = k1;
    }
    // one chain step: gen (k0, k1) -> words of the bridge in raw[buf], gen of the next bridge in (k0, k1); `stage` = the
    // debug-capture index of the key it derives.  Ten round groups, `from` .. `to` of them per call.
    uint32_t g0 = 0, g1 = 0, h0 = 0, h1 = 0;
    auto chain = [&](auto from_tag, auto to_tag, int buf, int stage) {
      constexpr int from = decltype(from_tag)::value, to = decltype(to_tag)::value;
#pragma unroll
      for (int s = from; s < to; ++s) {
        if (s == 0) tf_begin(t, k0, k1, gb, 2 + gb);                  // (G, H) = split(gen)              mcd_cais.py:66
        if (s == 5) tf_begin(t, is_split ? h0 : g0, is_split ? h1 : g1, cb0, cb1);   // split(H) | normal(G, (D,)) blocks   :67,87
        switch (s % 5) {
          case 0: tf_group<0>(t); break;
          case 1: tf_group<1>(t); break;
          case 2: tf_group<2>(t); break;
          case 3: tf_group<3>(t); break;
          default: tf_group<4>(t); break;
        }
        if (s == 4) {
          rows01(t.x0, g0, g1);
          rows01(t.x1, h0, h1);
        }
        if (s == 9) {
          if (writes) {
            raw[(buf * 8 + pc) * DP + jn] = t.x0;
            raw[(buf * 8 + pc) * DP + Hh + jn] = t.x1;
          }
          rows01(t.x1, k0, k1);                                        // gen = second(split(H))             mcd_cais.py:87
          k0 = twin_from_low(k0);
          k1 = twin_from_low(k1);
          if (a.dbg_keys && valid && s8 == 0) {
            a.dbg_keys[((int64_t)stage * a.n + p) * 2] = k0;
            a.dbg_keys[((int64_t)stage * a.n + p) * 2 + 1] = k1;
          }
        }
      }
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, kCut>;
    using I10 = std::integral_constant<int, 10>;
    chain(I0{}, I10{}, 0, 1);   // bridge 0
    wbar();   // P1
    wbar();   // P3
#ifdef CMCD_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int i = 0; i <= K; ++i) {
      const bool more = i + 1 < K;       // the words of bridge i + 1 and gen_{i + 2}
      const int buf = (i + 1) & 1;
      if (more) chain(I0{}, I1{}, buf, i + 2);
      WSTAMP(0);
      wbar();   // barrier 1
      WSTAMP(1);
      if (more) chain(I1{}, I10{}, buf, i + 2);
      WSTAMP(2);
      wbar();   // barrier 2
      WSTAMP(3);
    }
    wbar();   // F1
  }
  // =============================================================================================== ACC
  else {
    const int c = lane & 15, g = lane >> 4, pc = c & 7, tw = c >> 3, s8 = g + 4 * tw;
    const int64_t p = tile * 8 + pc;
    const bool valid = p < a.n;
    // the random words this lane turns into deviates: coordinates jA = s8 (always), jB = s8 + 8 (lanes s8 < D - 8)
    const int jc[2] = {s8, s8 + 8 < D ? s8 + 8 : s8};
    const bool on[2] = {true, s8 + 8 < D};
    // random word -> deviate of coordinate jc[q] (jax.random.normal: Giles' erfinv); `stage` = debug-capture index
    float nzv[2] = {0.f, 0.f};
    auto convert = [&](int q, int buf, int stage) {
      const uint32_t bits = raw[(buf * 8 + pc) * DP + jc[q]];
      nzv[q] = bits_to_normal(bits);
      if (a.dbg_bits && valid && on[q]) {
        a.dbg_bits[((int64_t)stage * a.n + p) * D + jc[q]] = bits;
        a.dbg_noise[((int64_t)stage * a.n + p) * D + jc[q]] = nzv[q];
      }
    };
    wbar();   // P1
    convert(0, 1, 0);
    convert(1, 1, 0);
    // z_0 = mean + std * normal(A, (D,))                               diag_gauss.py:49-62, mcdboundingmachine.py:157
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float z0 = expf(a.params[a.lay.vd_logdiag + jc[q]]) * nzv[q] + a.params[a.lay.vd_mean + jc[q]];
      if (on[q]) {
        zpub[pc * DP + jc[q]] = z0;
        if (a.traj && valid) a.traj[p * D + jc[q]] = z0;
      }
    }
    wbar();   // P3
#ifdef CMCD_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int i = 0; i <= K; ++i) {
      // the deviates of bridge i (words written during iteration i - 1): one conversion per interval; published after
      // barrier 1, when the last reader of bridge i - 1's deviates is done, and read behind barrier 2
      if (i < K) convert(0, i & 1, i + 1);
      WSTAMP(0);
      wbar();   // barrier 1
      WSTAMP(1);
      if (i < K) {
        nzb[pc * DP + jc[0]] = nzv[0];
        convert(1, i & 1, i + 1);
        if (on[1]) nzb[pc * DP + jc[1]] = nzv[1];
      }
      WSTAMP(2);
      wbar();   // barrier 2
      WSTAMP(3);
    }
    wbar();   // F1
    // ---- outputs: the loss the TGT waves summed over the coordinates, the tile's statistics record
    const float loss = lossb[pc];
    const bool use = valid && s8 == 0;
    if (use) a.out_loss[p] = loss;
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
    if (!a.fin_out) {
      if (lane == 0) {
        double* o = a.partials + tile * CMCD_NSTATS;
        o[0] = cnt; o[1] = sm; o[2] = sq; o[3] = mx; o[4] = ex;
      }
    } else {
      // fused merge by the last workgroup to arrive (cmcd_coop.hip: same protocol, same five doubles as finalize_kernel)
      int last = 0;
      if (lane == 0) {
        double* o = a.partials + tile * CMCD_NSTATS;
        __hip_atomic_store(o + 0, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 1, sm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 2, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 3, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 4, ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = __hip_atomic_fetch_add(a.fin_counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
      }
      last = __builtin_amdgcn_readfirstlane(last);
      if (last) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        wave_merge_stats(a.partials, (int)gridDim.x, a.fin_out, lane);
        if (a.stamp_slot && lane < CMCD_NSTATS && *a.stamp_slot != a.stamp_expect) a.fin_out[lane] = __builtin_nan("");
        if (lane == 0) __hip_atomic_store(a.fin_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
#undef CMCD_WIDE_SCHED_LOAD
#undef CMCD_WIDE_BAR1_SCHED
#ifdef CMCD_STAMPS
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 16; ++k) g_stamps_wide[wv][k] = st_acc[k];
#endif
}

#ifdef CMCD_STAMPS
extern "C" int cmcd_debug_read_stamps_wide(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_wide), sizeof(unsigned long long) * 16 * 16);
}
#endif

typedef void (*wide_fn)(TrajArgs);

static wide_fn pick_wide(const cmcd_desc& d, int T) {
  if (d.target != CMCD_TARGET_FUNNEL || d.dim != 10) return nullptr;
  if (d.arch == CMCD_ARCH_DDS) return T == 4 ? coop_wide8_kernel<CMCD_ARCH_DDS, 10, 4> : nullptr;
  if (T == 2) return coop_wide8_kernel<CMCD_ARCH_GEFFNER, 10, 2>;
  if (T == 4) return coop_wide8_kernel<CMCD_ARCH_GEFFNER, 10, 4>;
  return nullptr;
}

bool coop_wide8_available(const cmcd_desc& d, int T) { return pick_wide(d, T) != nullptr; }

int coop_wide8_launch(const cmcd_desc& d, const TrajArgs& ta, size_t lds_claim_min, void* stream) {
  const int T = ta.w.T, D = d.dim, DP = (D + 3) & ~3;
  wide_fn fn = pick_wide(d, T);
  if (!fn) return CMCD_ERR_UNSUPPORTED;
  const int HP = 16 * T, PTW = (D * T + 3) & ~3;
  size_t lds_bytes = size_t(8 * (HP + 4) + 8 * PTW + 3 * 8 * DP + 8 + 2 * 8 * DP) * 4;   // hbuf, part, baseb / nzb / zpub, lossb, raw
  if (lds_claim_min > lds_bytes) {
    // the caller's CU-exclusive claim (cmcd_coop.hip: coop_launch): the opt-in is per function and device, raised once
    static std::atomic<int> raised[64][2][8];
    int dev = 0;
    bool ok = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (ok) {
      std::atomic<int>& r = raised[dev][d.arch == CMCD_ARCH_DDS][T & 7];
      if (!r.load(std::memory_order_relaxed)) {
        ok = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds_claim_min) == hipSuccess;
        if (ok) r.store(1, std::memory_order_relaxed);
      }
    }
    if (ok) lds_bytes = lds_claim_min;
  }
  const unsigned tiles = unsigned((ta.n + 7) / 8);
  hipLaunchKernelGGL(fn, dim3(tiles), dim3(64 * (T + 4)), lds_bytes, static_cast<hipStream_t>(stream), ta);
  return CMCD_OK;
}

}  // namespace cmcd
