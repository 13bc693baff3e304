// coop_wide8_kernel — the CU-cooperative trajectory kernel for WIDE states (8 < d <= 12: the funnel, d = 10) on 8-particle
// tiles: one workgroup per tile, T + 4 waves with the roles of coop_kernel (cmcd_coop.hip) — but laid out for a state that
// is too wide to be carried redundantly.  r04's funnel instance was coop_kernel<.., D = 10, .., HALF> with its 2-d habits:
// every lane of the accounting wave formed all ten coordinates of z_{i+1} (eight-fold redundant, ~250 instructions behind
// barrier 2), the key chain ran three Threefry passes per bridge on a wave whose twin columns repeated each other, and the
// ten deviates were three dependent erfinv evaluations on the same wave (profiles/r05_stamps_funnel_8tile_before.txt:
// 1.45 us per bridge against the 2-d headline's 0.70).  Here every per-coordinate job is DEALT to the lanes of its particle:
//
//   waves 0..T-1  MLP   lane (qi, pg, kh, ng) = particle 4 pg + qi, neurons 16 wv + 4 ng + 2 kh + {0, 1} (v_mfma_f32_4x4x1
//                       order, as coop_kernel on 8-particle tiles).  Layer 3 ends in a reduce-SCATTER over the particle's 8
//                       lanes (1 DPP add + 2 row-swap adds per four outputs) instead of an all-reduce per output pair, and
//                       each lane writes the one or two outputs it ends up owning.
//   waves T, T+1  TGT   16 lanes per particle: lane `sub` owns coordinate `sub` — its share of grad log p (the funnel's
//                       sum of squares is one 16-lane reduction), the clipped scores and base_j of the forward mean; in
//                       interval 1 it converts random word `sub` of the bridge into its Gaussian deviate (one erfinv per lane).
//   wave  T+2     RNG   the jax Threefry key chain one bridge ahead, 8 lanes per particle: pass A = split(gen) (2 blocks),
//                       pass B = split(H) (2 blocks) + the Hh = d / 2 blocks of normal(G, (d,)) in ONE pass (7 of the 8 lanes
//                       busy) — two dependent passes per bridge, the minimum the chain allows.  The ten groups of four rounds
//                       of the two passes are cut into three segments that sit in the three intervals of the bridge
//                       (kCut1 / kCut2), so the wave is never the last to arrive at a barrier.
//   wave  T+3     ACC   8 lanes per particle, lane s8 owns coordinates {s8, s8 + 8}: between barrier 2 and barrier 3 it sums
//                       the layer-3 partials of ITS coordinates, forms z_{i+1} and publishes it (3 LDS reads, ~10 instructions
//                       per coordinate); the log-weight terms of its coordinates are accumulated per lane AFTER barrier 3 (off
//                       the critical path) and summed over the particle's lanes once, at the end of the launch.
//
// Per evaluation i, three raw s_barriers:  interval 1 (layer 1 | deviates | chain segment) — barrier 1 — interval 2 (layer 2,
// 3 | grad log p, base | chain segment) — barrier 2 — interval 3 (ACC: z_{i+1} | chain segment) — barrier 3 — every wave that
// keeps a copy of z reads the published state.  Every exchange row is single-buffered except the raw words (written one
// bridge ahead).  Same arithmetic as coop_kernel / traj_kernel up to the association of the sums over coordinates and
// neurons; the PRNG path is bit-exact (tests/test_gpu_prng.py).  Reference lines: /root/reference/src/mcd_cais.py:46-89,
// src/mcdboundingmachine.py:151-179, src/model_handler.py:124-143, src/nn.py:42-72; cited per statement in cmcd_kernels.hip.
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

// segments of the RNG wave's ten round groups (pass A: 0 - 4, pass B: 5 - 9): [0, kCut1) in interval 1, [kCut1, kCut2) in
// interval 2, the rest in interval 3 (profiles/r05_stamps_funnel_8tile_after.txt: no wave waits for the RNG wave)
#ifndef CMCD_WIDE_CUT1
#define CMCD_WIDE_CUT1 3
#endif
#ifndef CMCD_WIDE_CUT2
#define CMCD_WIDE_CUT2 8
#endif

template <int ARCH, int D, int T>
__global__ __launch_bounds__(64 * (T + 4)) void coop_wide8_kernel(TrajArgs a) {
  static_assert(D > 8 && D <= 12 && D % 2 == 0, "7 Threefry blocks per particle on 8 lanes, coordinates {s8, s8 + 8} per ACC lane");
  constexpr int HP = 16 * T, NR = 2, Hh = D / 2, DP = (D + 3) & ~3;
  constexpr int HQP = HP + 4, NQ = HP / 2, RSA = ((HP / 2 + 15) / 16) * 4;
  constexpr int PTW = (D * T + 3) & ~3;           // layer-3 partials of one particle: [j][wave]
    constexpr int kCut1 = CMCD_WIDE_CUT1, kCut2 = CMCD_WIDE_CUT2;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const hbuf = lds;                        // [8][HQP]   layer-1 activations
  float* const part = hbuf + 8 * HQP;             // [8][PTW]   layer-3 partials [j][wave]
  float* const baseb = part + 8 * PTW;            // [8][DP]    base_j of the forward mean (TGT -> ACC)
  float* const spub = baseb + 8 * DP;             // [8][DP]    s(z_i, i)_j (ACC -> TGT)
  float* const zpub = spub + 8 * DP;              // [8][DP]    the published state z_{i+1} (ACC -> MLP, TGT)
  float* const lossb = zpub + 8 * DP;             // [8]        per-particle loss at the end (TGT -> ACC)
  uint32_t* const raw = reinterpret_cast<uint32_t*>(lossb + 8);   // [2][8][DP] random words, one bridge ahead

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
  const float clipv = a.var_mode ? 1e2f : 1e3f;
  const bool clip_p = a.grad_clipping != 0, clip_q = clip_p && a.var_mode;
  const float cp = clip_p ? clipv : INFINITY, cq = clip_q ? clipv : INFINITY;
  const float* const sched_p = a.ws + a.w.sched;
  // the schedule row of evaluation i: {beta, eps, sigma, log sigma + log sqrt(2 pi) | 1 / (2 sigma^2), eps beta, eps (1 - beta), 0},
  // requested as a scalar load at the top of the iteration and complete behind barrier 1's own wait (cmcd_coop.hip)
#define CMCD_WIDE_SCHED_LOAD(i_)                                                                              \
  f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};                                                 \
  {                                                                                                           \
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
    const float* const rd_z = zpub + pc * DP;
    // the first D neurons of the geffner residual stream are z itself: wave 0, neurons nb, nb + 1 < D
    const bool z_in_u = ARCH == CMCD_ARCH_GEFFNER && wv == 0 && nb < D;
    const float* const rd_u = zpub + pc * DP + (z_in_u ? nb : 0);
    // which output this lane owns after the reduce-scatter of a quad of pair-slots: row g -> slot {0, 2, 1, 3}[g]
    const int own_slot = ((ng & 1) << 1) | (ng >> 1);
    float z[D], uz[2] = {0.f, 0.f};
    wbar();   // P1: random words of z_0
    wbar();   // P3: z_0 published
    auto read_state = [&]() {
#pragma unroll
      for (int q = 0; q < DP; q += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rd_z + q);
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (q + r < D) z[q + r] = v[r];
      }
      if (ARCH == CMCD_ARCH_GEFFNER) {
        const f32x2 u2 = *reinterpret_cast<const f32x2*>(rd_u);
        uz[0] = z_in_u ? u2[0] : 0.f;
        uz[1] = z_in_u ? u2[1] : 0.f;
      }
    };
    read_state();
#ifdef CMCD_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int i = 0; i <= K; ++i) {
      // ---- interval 1: layer 1
      float pre[NR], h[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        pre[r] = brow[r];
#pragma unroll
        for (int j = 0; j < D; ++j) pre[r] = fmaf(z[j], w1[j][r], pre[r]);
      }
#pragma unroll
      for (int r = 0; r < NR; ++r)
        h[r] = (ARCH == CMCD_ARCH_DDS) ? gelu_fast(pre[r]) : (urow[r] + uz[r]) + softplus(pre[r]);
      *reinterpret_cast<float2*>(my_h) = float2{h[0], h[1]};
      WSTAMP(0);
      wbar();   // barrier 1
      WSTAMP(1);
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
#pragma unroll
      for (int j = 0; j < D; ++j) ps[j] = fmaf(h2[1], w3s[j][1], h2[0] * w3s[j][0]);
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
      wbar();   // barrier 3: z_{i+1} published
      WSTAMP(5);
      if (i < K) read_state();
      WSTAMP(6);
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
    wbar();   // P1
    wbar();   // P3
    float zj = zpub[pc * DP + j], v = zpub[pc * DP];
    // this lane's share of the log-weight: coordinate j of w = -log q(z_0) + sum_i [log N(z_i; bk_i, sigma_i) - log N(z_{i+1}; fk_i, sigma_i)]
    float wl = 0.f, zp = 0.f, lp = 0.f;
    {
      const float dz = zj - qmean;                        // -log q(z_0)      diag_gauss.py:49-62, mcdboundingmachine.py:157
      wl = (dz * dz) / (2.0f * qstd * qstd) + logf(qstd) + kHalfLog2Pi;
    }
    float peps = 0.f, pinv2s2 = 0.f, pcst = 0.f, pA = 0.f, pB = 0.f;
#ifdef CMCD_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int i = 0; i <= K; ++i) {
      CMCD_WIDE_SCHED_LOAD(i);
      // funnel (/root/reference/src/model_handler.py:124-143; the scale of v is the hard-coded 3.0):
      //   log p = logN(v; 0, 3) + sum_{j >= 1} logN(z_j; 0, e^{v / 2})
      //   d / dv = -v / 9 - (d - 1) / 2 + e^{-v} ss / 2,   d / dz_j = -z_j e^{-v},   ss = sum_{j >= 1} z_j^2
      // Everything that does not need the schedule row runs in interval 1 (the MLP partners of these waves are short there and
      // long in interval 2); e^{-v} through v_exp_f32, the constants as reciprocals (no IEEE division on these waves).
      const float ss = part_sum<16>((act && j >= 1) ? zj * zj : 0.f);
      const float emv = __builtin_amdgcn_exp2f(-1.44269504088896340736f * v);
      constexpr float c0 = -0.5f * kLog2Pi - 1.0986122886681098f, c1 = -0.5f * (D - 1) * kLog2Pi;
      const float hes = 0.5f * emv * ss;
      const float g0 = fmaf(v, -1.0f / 9.0f, hes - 0.5f * (D - 1));
      const float gp = j == 0 ? g0 : -zj * emv;
      const float gpc = __builtin_amdgcn_fmed3f(gp, -cp, cp);
      const float gqc = __builtin_amdgcn_fmed3f((qmean - zj) * qiv, -cq, cq);
      lp = fmaf(v * v, -1.0f / 18.0f, c0 + c1) - 0.5f * (D - 1) * v - hes;
      WSTAMP(0);
      CMCD_WIDE_BAR1_SCHED();
      WSTAMP(1);
      const float eps = sc[1], cst = sc[3], inv2s2 = sd[0], cA = sd[1], cB = sd[2];
      const float base = fmaf(cA, gpc, fmaf(cB, gqc, zj));   // base_j = z_j + eps beta clip(gp_j) + eps (1 - beta) clip(gq_j)
      if (act) baseb[pc * DP + j] = base;
      WSTAMP(2);
      wbar();   // barrier 2
      WSTAMP(3);
      wbar();   // barrier 3
      WSTAMP(5);
      // ---- the log-weight terms of coordinate j (mcd_cais.py:71-86), off the critical path: s(z_i, i)_j and z_{i+1, j} from the ACC wave
      const float sn = spub[pc * DP + j], zn = zpub[pc * DP + j], vn = zpub[pc * DP];
      if (i > 0) {   // backward kernel of step i - 1: bk = z - eps ub + eps s(z_i, i), ub = -(beta gp + (1 - beta) gq) at z_i
        const float bk = fmaf(pA, gpc, fmaf(pB, gqc, fmaf(peps, sn, zj)));
        const float db = zp - bk;
        wl += -(db * db) * pinv2s2 - pcst;
      }
      if (i < K) {   // forward kernel of step i: fk = base - eps s, z_{i+1} = fk + sigma noise
        const float fk = fmaf(a.ula ? 0.f : -eps, sn, base);
        const float df = zn - fk;
        wl -= -(df * df) * inv2s2 - cst;
        zp = zj;
        zj = zn;
        v = vn;
        peps = eps; pinv2s2 = inv2s2; pcst = cst; pA = cA; pB = cB;
      }
      WSTAMP(6);
    }
    // w = sum over coordinates + log p(z_K); loss = -w                              mcdboundingmachine.py:178-179
    const float wtot = part_sum<16>(act ? wl : 0.f) + lp;
    if (sub == 0) lossb[pc] = -wtot;
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
    using I1 = std::integral_constant<int, kCut1>;
    using I2 = std::integral_constant<int, kCut2>;
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
      if (more) chain(I1{}, I2{}, buf, i + 2);
      WSTAMP(2);
      wbar();   // barrier 2
      WSTAMP(3);
      if (more) chain(I2{}, I10{}, buf, i + 2);
      WSTAMP(4);
      wbar();   // barrier 3
      WSTAMP(5);
    }
    wbar();   // F1
  }
  // =============================================================================================== ACC
  else {
    const int c = lane & 15, g = lane >> 4, pc = c & 7, tw = c >> 3, s8 = g + 4 * tw;
    const int64_t p = tile * 8 + pc;
    const bool valid = p < a.n;
    // coordinates of this lane: jA = s8 (always), jB = s8 + 8 (lanes s8 < D - 8)
    const int jc[2] = {s8, s8 + 8 < D ? s8 + 8 : s8};
    const bool on[2] = {true, s8 + 8 < D};
    float qmean[2], qstd[2], b3[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      qmean[q] = a.params[a.lay.vd_mean + jc[q]];
      qstd[q] = expf(a.params[a.lay.vd_logdiag + jc[q]]);
      b3[q] = a.ws[a.w.b3 + jc[q]];
    }
    const float factor = a.ws[a.w.b3 + 15];
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
    float z[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      z[q] = qstd[q] * nzv[q] + qmean[q];
      if (on[q]) {
        zpub[pc * DP + jc[q]] = z[q];
        if (a.traj && valid) a.traj[p * D + jc[q]] = z[q];
      }
    }
    wbar();   // P3
#ifdef CMCD_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    for (int i = 0; i <= K; ++i) {
      CMCD_WIDE_SCHED_LOAD(i);
      // the deviates of bridge i for this lane's two coordinates: one conversion in interval 1, the other in interval 2
      if (i < K) convert(0, i & 1, i + 1);
      WSTAMP(0);
      CMCD_WIDE_BAR1_SCHED();
      WSTAMP(1);
      if (i < K) convert(1, i & 1, i + 1);
      WSTAMP(2);
      wbar();   // barrier 2
      WSTAMP(3);
      // ---- interval 3: s(z_i, i) of this lane's coordinates from the layer-3 partials, the forward mean, z_{i+1}
      const float eps = sc[1], sig = sc[2];
      float ptv[2][T], basev[2];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (T == 4) {
          const f32x4 t4 = *reinterpret_cast<const f32x4*>(part + pc * PTW + jc[q] * T);
#pragma unroll
          for (int r = 0; r < T; ++r) ptv[q][r] = t4[r & 3];
        } else {
          const f32x2 t2 = *reinterpret_cast<const f32x2*>(part + pc * PTW + jc[q] * T);
#pragma unroll
          for (int r = 0; r < T; ++r) ptv[q][r] = t2[r & 1];
        }
        basev[q] = baseb[pc * DP + jc[q]];
      }
      float zn[2];
      const float seps = a.ula ? 0.f : -eps;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        float o = b3[q];
        if (T == 4) o += (ptv[q][0] + ptv[q][1]) + (ptv[q][2] + ptv[q][3]);
        else o += ptv[q][0] + ptv[q][1];
        const float sn = (ARCH == CMCD_ARCH_DDS) ? __builtin_amdgcn_fmed3f(o, -1e4f, 1e4f) : o * factor;
        const float fk = fmaf(seps, sn, basev[q]);    // fk = z - eps uf - eps s                              mcd_cais.py:61
        zn[q] = fmaf(sig, nzv[q], fk);                // z' = fk + sqrt(2 eps) noise                           mcd_cais.py:63-67
        if (on[q]) {
          spub[pc * DP + jc[q]] = sn;
          if (i < K) zpub[pc * DP + jc[q]] = zn[q];
        }
      }
      WSTAMP(4);
      wbar();   // barrier 3
      WSTAMP(5);
      if (i < K) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          z[q] = zn[q];
          if (on[q] && a.traj && valid) a.traj[((int64_t)(i + 1) * a.n + p) * D + jc[q]] = zn[q];
        }
      }
      WSTAMP(6);
    }
    wbar();   // F1
    // ---- outputs: the loss the TGT waves summed over the coordinates, z_K, the tile's statistics record
    const float loss = lossb[pc];
    if (valid) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
        if (on[q]) a.out_z[p * D + jc[q]] = z[q];
    }
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
  size_t lds_bytes = size_t(8 * (HP + 4) + 8 * PTW + 3 * 8 * DP + 8 + 2 * 8 * DP) * 4;
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
