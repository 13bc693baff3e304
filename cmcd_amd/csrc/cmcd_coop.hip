// coop_kernel — the CU-cooperative trajectory kernel: ONE WORKGROUP per 16-particle tile (8-particle tile for batches
// of <= 2048 particles: template parameter HALF, described at the kernel).
//
// Why: the named workload (N = 2000, K = 256) is a 256-long dependent chain on only 125 tiles.  A
// wave-per-tile kernel leaves 7/8 of the SIMDs idle and its per-bridge latency is one wave's
// instruction issue (~1800 VALU instructions at ~4.5 cycles each; integer chains ~8.5).  Here the
// tile's work is spread over T + 4 waves of one CU (T + 3 where RNG and ACC share a wave: MERGE, the 132-wide net):
//   waves 0..T-1  MLP: wave v owns hidden-neuron tile v (16 neurons): its first-layer neurons, its W2 operands
//                 (resident in VGPRs for the whole launch), its matrix accumulator, its slice of the output dot
//                 product; keeps its own copy of z (needs it for layer 1);
//   waves T, T+1  TGT: grad log p(z) for half of the tile's particles each (8, or 16 with HALF, lanes per particle),
//                 own copy of z; also forms base = z + eps beta clip(gp) + eps (1 - beta) clip(gq), the part of the
//                 forward mean that does not need the network;
//   wave  T+2     RNG: the jax Threefry key chain, one bridge AHEAD, raw bits only (integer chain);
//   wave  T+3     ACC: bits -> Gaussian deviates (Giles erfinv), and the only owner of the log-weight w
//                 and of the outputs.
// Per bridge evaluation i, two raw s_barriers (LDS visibility via s_waitcnt lgkmcnt(0) only):
//   interval 1:  MLP: layer 1 + activation -> hbuf | TGT: distance pass of grad log p(z_i) | RNG: split(gen) of bridge i+1
//   barrier 1
//   interval 2:  MLP: layer 2 on the matrix cores from hbuf, activation, layer-3 partials -> part |
//                TGT: exponential pass -> grad log p, log p, base -> gpb | RNG: split(H), normal bits -> raw |
//                ACC: raw -> deviates -> nzb
//   barrier 2
//   phase C(i):  every wave that keeps a z (MLP, TGT, ACC) reads the same exchange rows and forms
//                s = clip(b3 + sum of partials), fk = base - eps s, z_{i+1} = fk + sigma eps_i — identical bits in
//                every copy; ACC also closes step i-1 into the log-weight (mcd_cais.py:71-86).
// Every role runs its OWN instantiation of the bridge loop (role_loop below) with this same barrier sequence.
// The per-bridge schedule row arrives as a scalar load requested at the top of the iteration.
// Same arithmetic as traj_kernel (cmcd_kernels.hip) up to the association of the forward mean; reference lines are
// cited there.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <atomic>
#include <mutex>
#include <set>
#include <type_traits>
#include <utility>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

// Diagnostic build only (-DCMCD_STAMPS, tools/probes/stamp_probe.py): per-wave cycle totals of each
// phase of workgroup 0, written to a buffer nothing else reads.  Never compiled into the product.
#ifdef CMCD_STAMPS
__device__ unsigned long long g_stamps[16][16];
#define STAMP(slot)                                                                      \
  do {                                                                                   \
    unsigned long long t_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);                                                   \
    st_acc[slot] += t_ - st_last;                                                        \
    st_last = t_;                                                                        \
  } while (0)
#else
#define STAMP(slot)
#endif

// workgroup barrier that publishes LDS writes but does not drain outstanding global loads
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// HALF: 8 particles per tile.  The exchange rows keep 16 columns, columns c and c + 8 carry the SAME particle (same
// seed, so the RNG / ACC / TGT waves compute identical values in both), and the MLP waves give every particle 8 lanes
// instead of 4: lane (qi, pg, kh, ng) = particle 4 pg + qi, neurons 4 ng + 2 kh + {0, 1} of the wave's 16 — half the
// first-layer FMAs, half the activations (2 instead of 4 per layer and lane), half the layer-3 products, and layer 2 on
// the 4x4x1 matrix instruction (16 blocks of 4 neurons x 4 particles: no column wasted) with the activations broadcast
// across the neuron groups by the instruction itself (described at the lane mapping below).  A batch of <= 2048
// particles then runs on twice the CUs (N = 2000: 250 workgroups instead of 125 on 256 CUs) with a shorter per-bridge
// chain on each.
// MERGE: the RNG and ACC roles share one wave (T + 3 waves per workgroup).  Used by the 9-tile (132-wide net) instance
// on 8-particle tiles: twelve waves are three per SIMD = 168 registers each, which the 72 resident 4x4x1 operands of
// an MLP wave need; and with 72 matrix instructions per MLP wave and bridge the key chain is no longer the long pole.
template <int TARGET, int ARCH, int D, int T, bool HALF, bool MERGE = false>
__global__ __launch_bounds__(64 * (T + (MERGE ? 3 : 4))) void coop_kernel(TrajArgs a) {
  constexpr int HP = 16 * T;
  constexpr int PPT = HALF ? 8 : 16;         // particles per tile
  constexpr int NR = HALF ? 2 : 4;           // neurons per lane and 16-neuron tile (element-wise work)
  constexpr int Hh = (D + 1) / 2;
  constexpr int NZ = 2 * Hh;                 // noise words per particle (>= D)
  constexpr int GP = (2 * D + 1 + 3) & ~3;   // base [D] (state + score terms of the forward mean), grad log p [D], log p; padded
  constexpr int PT = (T * D + 3) & ~3;       // layer-3 partials per particle, padded
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* hbuf = lds;                         // [T][4][16][4]  layer-1 activations, MFMA-B order
  float* part = hbuf + HP * 16;              // [2][16][PT]    layer-3 partial sums, [c][v*D + j]
  float* gpb = part + 2 * 16 * PT;           // [2][16][GP]    base, grad log p, log p
  float* nzb = gpb + 2 * 16 * GP;            // [2][16][NZ]    Gaussian noise
  uint32_t* raw = reinterpret_cast<uint32_t*>(nzb + 2 * 16 * NZ);  // [2][16][NZ] raw bits
  float* lds_tgt = reinterpret_cast<float*>(raw + 2 * 16 * NZ);
  // PUBZ (d > 4: the funnel): phase C costs ~2 exchange rows per dimension and wave — at d = 10 eighteen LDS reads per
  // lane on each of seven waves, ~1000 cycles of LDS pipe behind barrier 2.  There only the ACC wave forms z_{i+1}; it
  // publishes the state and every other wave picks it up behind a third barrier (three 16-byte reads instead of 18):
  // funnel, N = 300, K = 64: 0.1021 -> 0.0952 ms, bitwise identical.  (Splitting the publication over the ACC wave's four
  // rows in front of its full update made it 0.1115 ms: the full update then runs into barrier 1, where this wave has
  // no slack.)
  constexpr bool PUBZ = D > 4;
  constexpr int DP = (D + 3) & ~3;
  float* const zpub = lds_tgt + a.w.tgt_floats;   // [16][DP], PUBZ only

  // Role = hardware wave index: waves go to SIMD (index % 4), so every SIMD holds one MLP wave and one auxiliary wave.
  // Measured alternative (profiles/r01_r_simd_map.txt): MLP waves paired on SIMDs 0 / 1 and the auxiliary waves on 2 / 3
  // (no auxiliary wave ever blocked by an MFMA, but two MFMA chains per matrix pipe) is 6.7 % slower.
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane = threadIdx.x & 63, g = lane >> 4;
  const bool is_mlp = wv < T, is_tgt = wv == T || wv == T + 1, is_rng = wv == T + 2;
  const bool is_acc = MERGE ? (wv == T + 2) : (wv == T + 3);
  // particle column of this lane: TGT waves hold 8 particles x 8 lanes (HALF: 4 particles x 16 lanes — the tile's 8
  // particles over the two waves, results written to both twin columns), everyone else 16 x 4
  constexpr int LPT = HALF ? 16 : 8;         // lanes per particle on the target waves
  const int sub8 = HALF ? (lane >> 2) : (lane >> 3);
  // HALF MLP waves work in the lane order of v_mfma_f32_4x4x1 (16 blocks of 4 x 4): lane = qi + 4 pg + 8 kh + 16 ng —
  // particle 4 pg + qi, half kh of the contraction, neuron group ng (4 neurons of the wave's 16) = the 16-lane ROW of
  // the wave.  The B operand (the layer-1 activations) does not depend on ng, and the instruction can take it from one
  // row for all four (BLGP = 4 + row, tools/probes/blgp_probe.hip): each row fetches a quarter of the particle's
  // activations from LDS instead of every lane fetching all of them (r02: 8 -> 2 ds_read_b128 per lane and bridge; the
  // four MLP waves' reads were 256 of the bridge's ~1800 cycles of LDS pipe).
  const int ng = lane >> 4, kh = (lane >> 3) & 1;
  const int c = is_tgt ? (HALF ? 4 * (wv - T) + (lane & 3) : 8 * (wv - T) + (lane & 7))
                       : ((HALF && is_mlp) ? (lane & 7) : (lane & 15));
  const int64_t tile = blockIdx.x;
  const int64_t p = tile * PPT + (HALF ? (c & 7) : c);
  const bool valid = p < a.n;
  const bool own = !HALF || c < 8;           // the column that writes its particle's outputs
  const int nb = HALF ? 16 * wv + 4 * ng + 2 * kh : 16 * wv + 4 * g;   // first of the lane's NR neurons (MLP waves)
  constexpr int HQP = HP + 4;                // HALF: pitch of the [particle][neuron] activation buffer (bank spread)
  constexpr int NQ = HALF ? HP / 2 : 1;      // HALF: 4x4x1 MFMA steps per bridge (two contraction halves side by side)
  constexpr int RSA = ((HP / 2 + 15) / 16) * 4;   // HALF: activations one row of an MLP wave fetches (quarter of a half, 16-B units)
  const int K = a.K;

  for (int i = threadIdx.x; i < a.w.tgt_floats; i += blockDim.x) lds_tgt[i] = a.ws[a.w.tgt + i];
  // r05: pull the per-bridge tables (schedule rows, first-layer bias rows [+ residual rows]: one contiguous block of the
  // workspace) into THIS XCD's L2 while the key-chain prologue runs — one touch per 128-byte line, spread over the workgroup.
  // In the default launch sequence the prep launch has just rewritten them (every XCD's copy is gone), each row is requested
  // only one bridge ahead (~0.4 us) and the first reader of a row on an XCD pays a trip to the fabric for all its neighbours:
  // the trajectory kernel ran 192.6 us behind the prep launch against 179.4 us on warm tables
  // (profiles/r05_i_headline_gaps.txt, rocprofv3 kernel trace).  The value is only kept alive until the prologue's barrier.
  // The lines are dealt to the workgroups that share an XCD (round-robin dispatch: workgroup b runs on XCD b % 8, its rank there
  // is b / 8): a workgroup of the named grid touches 18 lines, not 578 (touching all of them in every workgroup cost the kernel
  // 2 us on warm tables).
  float warm = 0.f;
  {
    const int64_t t0 = a.w.sched;
    const int64_t t1 = (ARCH == CMCD_ARCH_GEFFNER ? a.w.utab : a.w.bias1) + (int64_t)(K + 1) * HP;
    const int64_t per_xcd = (gridDim.x + 7) >> 3, rank = blockIdx.x >> 3;
    for (int64_t i = t0 + 32 * (rank * blockDim.x + threadIdx.x); i < t1; i += 32 * per_xcd * blockDim.x) warm += a.ws[i];
  }
  // issue priority of this wave's role against its SIMD partner (s_setprio takes an immediate)
  switch ((a.prio >> (is_mlp ? 0 : is_tgt ? 2 : is_rng ? 4 : 6)) & 3) {
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    case 3: __builtin_amdgcn_s_setprio(3); break;
    default: break;
  }

  // ---- per-role resident operands
  f32x4 afrag[T], b2v = {0.f, 0.f, 0.f, 0.f};
  float aq[NQ], b2p[NR];
  float w1[D][NR], w3[D][NR];   // the lane's own neurons nb .. nb + NR - 1
#pragma unroll
  for (int r = 0; r < NR; ++r) b2p[r] = 0.f;
  if (is_mlp) {
    if (HALF) {
#pragma unroll
      for (int q = 0; q < NQ; ++q) aq[q] = a.ws[a.w.w2q + (int64_t)(wv * NQ + q) * 64 + lane];
#pragma unroll
      for (int r = 0; r < NR; ++r) b2p[r] = a.ws[a.w.b2 + nb + r];
    } else {
#pragma unroll
      for (int ti = 0; ti < T; ++ti)
        afrag[ti] = *reinterpret_cast<const f32x4*>(a.ws + a.w.w2 + ((ti * T + wv) * 64 + lane) * 4);
      b2v = *reinterpret_cast<const f32x4*>(a.ws + a.w.b2 + 16 * wv + 4 * g);
    }
#pragma unroll
    for (int j = 0; j < D; ++j) {
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        w1[j][r] = a.ws[a.w.w1z + j * HP + nb + r];
        w3[j][r] = a.ws[a.w.w3t + j * HP + nb + r];
      }
    }
  }
  float b3[D];
#pragma unroll
  for (int j = 0; j < D; ++j) b3[j] = a.ws[a.w.b3 + j];
  const float factor = a.ws[a.w.b3 + 15];

  float qmean[D], qstd[D], qiv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qmean[j] = a.params[a.lay.vd_mean + j];
    qstd[j] = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (qstd[j] * qstd[j]);
  }

  // ---- RNG wave: key chain prologue (mcdboundingmachine.py:151-162, mcd_cais.py:94), then the bits
  //      of bridge 0.  Lane row g computes block (g & 1) of a split / block g of a normal draw.
  const int gb = g & 1;
  uint32_t k0 = 0u, k1 = 0u;
  // normal(key, (D,)) bits: block j encrypts counters (j, Hh + j) -> words j and Hh + j; pad counter 0
  // `stage`: debug-capture index of the chain key this call derives (gen_stage), see TrajArgs::dbg_keys
  auto normal_bits = [&](uint32_t nk0, uint32_t nk1, uint32_t hk0, uint32_t hk1, int buf, bool with_split, int stage) {
    constexpr int NB = 2 + Hh;
#pragma unroll
    for (int b0 = with_split ? 0 : 2; b0 < NB; b0 += 4) {
      const int b = b0 + g - (with_split ? 0 : 0);
      const bool is_split = with_split && b < 2;
      const int jn = with_split ? b - 2 : b0 - 2 + g;
      uint32_t y0 = is_split ? b : jn;
      uint32_t y1 = is_split ? 2 + b : ((Hh + jn < D) ? Hh + jn : 0);
      threefry2x32(is_split ? hk0 : nk0, is_split ? hk1 : nk1, y0, y1);
      if (with_split && b0 == 0) {
        rows01(y1, k0, k1);  // gen = second(split(H))   mcd_cais.py:87
        if (a.dbg_keys && valid && own && g == 0) {
          a.dbg_keys[((int64_t)stage * a.n + p) * 2] = k0;
          a.dbg_keys[((int64_t)stage * a.n + p) * 2 + 1] = k1;
        }
      }
      if (jn >= 0 && jn < Hh) {
        raw[(buf * 16 + c) * NZ + jn] = y0;
        raw[(buf * 16 + c) * NZ + Hh + jn] = y1;
      }
    }
  };
  if (is_rng) {
    const int32_t seed = a.seeds[valid ? p : a.n - 1];
    uint32_t x0 = gb, x1 = 2 + gb;
    threefry2x32(0u, (uint32_t)seed, x0, x1);  // (A, B) = split(PRNGKey(seed))
    uint32_t a0, a1, bb0, bb1;
    rows01(x0, a0, a1);
    rows01(x1, bb0, bb1);
    normal_bits(a0, a1, 0u, 0u, 1, false, 0);  // z0 noise = normal(A) -> raw[1]
    x0 = gb; x1 = 2 + gb;
    threefry2x32(bb0, bb1, x0, x1);            // C = first(split(B))
    uint32_t c0, c1;
    rows01(x0, c0, c1);
    x0 = gb; x1 = 2 + gb;
    threefry2x32(c0, c1, x0, x1);              // gen = second(split(C))
    rows01(x1, k0, k1);
    if (a.dbg_keys && valid && own && g == 0) {
      a.dbg_keys[p * 2] = k0;
      a.dbg_keys[p * 2 + 1] = k1;
    }
    x0 = gb; x1 = 2 + gb;
    threefry2x32(k0, k1, x0, x1);              // (G, H) = split(gen)             mcd_cais.py:66
    uint32_t g0, g1, h0, h1;
    rows01(x0, g0, g1);
    rows01(x1, h0, h1);
    normal_bits(g0, g1, h0, h1, 0, true, 1);   // bridge 0 bits -> raw[0]
  }
  lds_barrier();
  // bits -> deviates, words dealt to the 4 rows of the wave
#ifndef CMCD_DBG_FLAG
#define CMCD_DBG_FLAG 1
#endif
  // (r04) the diagnostic capture's three-term condition, once per launch instead of once per bridge on the accounting wave —
  // for the 2-d instances only: the d = 10 instance (funnel) pays for every extra live value (95.6 -> 106.9 us with this flag
  // and the trajectory row pointer below, profiles/r04_grad_two_workgroups_per_cu.txt)
  constexpr bool kLeanAcc = D <= 4;
  const bool dbg_on = a.dbg_bits && valid && own;
  auto convert = [&](int buf, int stage) {
#pragma unroll
    for (int q0 = 0; q0 < D; q0 += 4) {
      const int q = q0 + g;
      if (q < D) {
        const uint32_t bits = raw[(buf * 16 + c) * NZ + q];
        const float dev = bits_to_normal(bits);
        nzb[(buf * 16 + c) * NZ + q] = dev;
        if ((CMCD_DBG_FLAG && kLeanAcc) ? dbg_on : (a.dbg_bits && valid && own)) {
          a.dbg_bits[((int64_t)stage * a.n + p) * D + q] = bits;
          a.dbg_noise[((int64_t)stage * a.n + p) * D + q] = dev;
        }
      }
    }
  };
  if (is_acc) convert(1, 0);
  asm volatile("" ::"v"(warm));   // (the table touches above: complete by now, nothing else reads them)
  lds_barrier();

  // z0 = mean + std * normal(A, (D,)); w = -log q(z0)      diag_gauss.py:49-62, mcdboundingmachine.py:157
  float z[D], zp[D];
  float w = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    z[j] = qstd[j] * nzb[(16 + c) * NZ + j] + qmean[j];
    zp[j] = 0.f;
    const float dz = z[j] - qmean[j];
    w -= -(dz * dz) / (2.0f * qstd[j] * qstd[j]) - logf(qstd[j]) - kHalfLog2Pi;
  }
  if (is_acc && a.traj && valid && own && g == 0) {
#pragma unroll
    for (int j = 0; j < D; ++j) a.traj[p * D + j] = z[j];
  }
#ifndef CMCD_TRAJ_PTR
#define CMCD_TRAJ_PTR 1
#endif
  // (r04) gradient calls keep z_1 .. z_K: a per-lane row pointer that advances by n D floats per bridge, one 8-byte store for
  // d = 2 — the row index was rebuilt from (e, n, p) in 64-bit arithmetic on the accounting wave every bridge
  float* tnext = (CMCD_TRAJ_PTR && kLeanAcc && is_acc && a.traj && valid && own && g == 0) ? a.traj + ((int64_t)a.n + p) * D : nullptr;
  const int64_t tstride = (int64_t)a.n * D;

  const float clipv = a.var_mode ? 1e2f : 1e3f;
  const bool clip_p = a.grad_clipping != 0;
  const bool clip_q = clip_p && a.var_mode;
  const float* brow_ptr = a.ws + a.w.bias1 + nb;
  const float* urow_ptr = a.ws + a.w.utab + nb;
  auto load_row = [&](const float* ptr) -> f32x4 {   // the lane's NR entries of a per-bridge row
    if (HALF) {
      const float2 t = *reinterpret_cast<const float2*>(ptr);
      return f32x4{t.x, t.y, 0.f, 0.f};
    }
    return *reinterpret_cast<const f32x4*>(ptr);
  };
  // HALF: hbuf is [8 particles][HQP]; the lane writes its neuron pair, reads its particle's half kh of the neurons
  float* const my_h = HALF ? hbuf + c * HQP + nb : hbuf + ((wv * 4 + g) * 16 + (lane & 15)) * 4;
  const float* const rd_h = HALF ? hbuf + c * HQP + (HP / 2) * kh + RSA * ng : hbuf + (g * 16 + (lane & 15)) * 4;

  float fk_lp = 0.f, peps = 0.f, pinv2s2 = 0.f, pcst = 0.f, logp = 0.f;
  f32x4 brow = {0.f, 0.f, 0.f, 0.f}, urow = {0.f, 0.f, 0.f, 0.f};
  if (is_mlp) {
    brow = load_row(brow_ptr);
    if (ARCH == CMCD_ARCH_GEFFNER) urow = load_row(urow_ptr);
  }

  // Phase C of evaluation e: s(z_e, e) from the layer-3 partials; [track_w: close step e-1 into the log-weight,
  // mcd_cais.py:71-86]; open step e (forward kernel, mcd_cais.py:52-67) -> z_{e+1}.  Run right after barrier 2 by
  // every wave that keeps a copy of z (MLP, TGT, ACC: ~15 instructions each, cheaper than an LDS hand-over of z and a
  // third barrier); only ACC tracks w.
  // Arithmetic note: uf = -(beta gp + (1-beta) gq), fk = z - eps uf - eps s  (mcd_cais.py:52-61) is evaluated as
  // base = fma(eps beta, gp, fma(eps (1-beta), gq, z)) on the target waves (the two products precombined in the schedule
  // table) and fk = fma(-eps, s, base) here: same value up to the last rounding.  Clips are v_med3_f32 against +-inf
  // when clipping is off (no branch).
#ifdef CMCD_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last = 0;
#endif
  const float cp = clip_p ? clipv : INFINITY, cq = clip_q ? clipv : INFINITY;
  float pA = 0.f, pB = 0.f;
  auto phase_c = [&](int e, bool track_w, const f32x4& sc, const f32x4& sd, auto pb_tag) {
    constexpr int pb = decltype(pb_tag)::value;   // = e & 1
    const float eps = sc[1], sig = sc[2], cst = sc[3], inv2s2 = sd[0], cA = sd[1], cB = sd[2];
    // base_j = z_j + eps beta clip(gp_j) + eps (1 - beta) clip(gq_j) arrives from the target waves (r02: they have the
    // slack, the MLP waves are the critical path and need only s: 14 instead of 24 instructions here); the forward mean
    // is fk = base - eps s, in the reference's own order (mcd_cais.py:61: z - eps uf - eps s).  Every wave reads the
    // same base from LDS and forms the same sum over the same partials, so all copies of z stay bitwise equal.
    float sn[D], gp[D], gq[D], base[D];
    // Every exchange row this wave needs is requested before the first use: one LDS round trip, not two (the compiler
    // left the noise and base reads behind the wait for the partials, ISA reading r02).  The empty asm statement takes
    // every loaded register as an in-out operand: all requests are out, and have landed, before any arithmetic.
    f32x2 nzv[NZ / 2];
    f32x4 ptv[PT / 4];
    constexpr int GQ = GP / 4, BQ4 = (D + 3) / 4;
    f32x4 gvv[GQ];
    if (e < K) {
#pragma unroll
      for (int q = 0; q < NZ / 2; ++q) nzv[q] = *reinterpret_cast<const f32x2*>(nzb + (pb * 16 + c) * NZ + 2 * q);
    }
#pragma unroll
    for (int q = 0; q < (track_w ? GQ : BQ4); ++q) gvv[q] = *reinterpret_cast<const f32x4*>(gpb + (pb * 16 + c) * GP + 4 * q);
#pragma unroll
    for (int q = 0; q < PT / 4; ++q) ptv[q] = *reinterpret_cast<const f32x4*>(part + (pb * 16 + c) * PT + 4 * q);
    if (e < K) {
#pragma unroll
      for (int q = 0; q < NZ / 2; ++q) asm volatile("" : "+v"(nzv[q]));
    }
#pragma unroll
    for (int q = 0; q < (track_w ? GQ : BQ4); ++q) asm volatile("" : "+v"(gvv[q]));
#pragma unroll
    for (int q = 0; q < PT / 4; ++q) asm volatile("" : "+v"(ptv[q]));
    float nz[NZ];
#pragma unroll
    for (int q = 0; q < NZ; ++q) nz[q] = (e < K) ? nzv[q / 2][q % 2] : 0.f;
    {
      float pt[PT], gv[GP];
#pragma unroll
      for (int q = 0; q < PT; ++q) pt[q] = ptv[q / 4][q % 4];
#pragma unroll
      for (int q = 0; q < (track_w ? GP : 4 * BQ4); ++q) gv[q] = gvv[q / 4][q % 4];
      if (track_w) {   // ACC: also the clipped scores (backward kernel of the previous step) and log p
#pragma unroll
        for (int j = 0; j < D; ++j) {
          base[j] = gv[j];
          gp[j] = __builtin_amdgcn_fmed3f(gv[D + j], -cp, cp);
          gq[j] = __builtin_amdgcn_fmed3f((qmean[j] - z[j]) * qiv[j], -cq, cq);
        }
        logp = gv[2 * D];
      } else {
#pragma unroll
        for (int j = 0; j < D; ++j) { base[j] = gv[j]; gp[j] = 0.f; gq[j] = 0.f; }
      }
#pragma unroll
      for (int j = 0; j < D; ++j) {
        float o = b3[j];
#pragma unroll
        for (int v = 0; v < T; ++v) o += pt[v * D + j];
        sn[j] = (ARCH == CMCD_ARCH_DDS) ? __builtin_amdgcn_fmed3f(o, -1e4f, 1e4f) : o * factor;
      }
    }
    STAMP(9);   // exchange rows + schedule row read and combined
    // the new state first (PUBZ: published to the other waves before the log-weight bookkeeping of this wave)
    const float seps = a.ula ? 0.f : -eps;
    float fkv[D], znv[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      fkv[j] = fmaf(seps, sn[j], base[j]);
      znv[j] = fmaf(sig, nz[j], fkv[j]);
    }
    if (PUBZ && track_w) {
      if (e < K && g == 0) {
#pragma unroll
        for (int q = 0; q < DP; q += 4) {
          f32x4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (q + r < D) ? znv[(q + r < D) ? q + r : 0] : 0.f;
          *reinterpret_cast<f32x4*>(zpub + c * DP + q) = v;
        }
      }
      STAMP(6);        // (ACC) new state formed and published
      lds_barrier();   // barrier 3 (every wave, every evaluation)
      STAMP(5);
    }
    if (track_w && e > 0) {  // backward kernel of step e-1: bk = z - eps ub + eps s, ub = -(beta gp + (1-beta) gq)
      float bk_lp = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float bk = fmaf(pA, gp[j], fmaf(pB, gq[j], fmaf(peps, sn[j], z[j])));
        const float db = zp[j] - bk;
        bk_lp += -(db * db) * pinv2s2 - pcst;
      }
      w += bk_lp - fk_lp;
    }
    if (e == K) return;
    fk_lp = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (track_w) {
        const float df = znv[j] - fkv[j];
        fk_lp += -(df * df) * inv2s2 - cst;
        zp[j] = z[j];
      }
      z[j] = znv[j];
    }
    if (CMCD_TRAJ_PTR && kLeanAcc) {
      if (track_w && tnext) {
        if (D == 2) {
          *reinterpret_cast<float2*>(tnext) = float2{z[0], z[1]};
        } else {
#pragma unroll
          for (int j = 0; j < D; ++j) tnext[j] = z[j];
        }
        tnext += tstride;
      }
    } else if (track_w && a.traj && valid && own && g == 0) {
#pragma unroll
      for (int j = 0; j < D; ++j) a.traj[((int64_t)(e + 1) * a.n + p) * D + j] = z[j];
    }
    peps = eps; pinv2s2 = inv2s2; pcst = cst; pA = cA; pB = cB;
  };

#ifdef CMCD_STAMPS
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
  typename Target<TARGET, D>::Means tmeans;
  if (is_tgt) Target<TARGET, D>::template load_means<LPT>(sub8, lds_tgt, tmeans);
  // Per-bridge scalars {beta, eps, sigma, log sigma + log sqrt(2 pi) | 1/(2 sigma^2), eps beta, eps (1 - beta), 0}: row i is
  // read by phase C(i) at the END of iteration i and requested at its top as a SCALAR load through the constant address
  // space (the table is written by the prep launch, never by this kernel): eight SGPRs per wave, no LDS traffic in the
  // burst behind barrier 2, and the request has a whole interval to complete before the first lgkmcnt wait (the LDS
  // publication in front of barrier 1).  History: r01 fetched the row with vector loads whose registers were dead on
  // some role paths, got re-used there, and the hazard put `s_waitcnt vmcnt(0)` — an L2 round trip — at the start of
  // interval 1 of every wave (0.2676 ms); an LDS copy of the table removed that (0.2425 ms) at the price of two more
  // 1 KB broadcast reads per wave in phase C.
  const float* const sched_p = a.ws + a.w.sched;
  // ONE LOOP PER ROLE (r02): the roles are wave-uniform, but as branches inside one shared loop body every role's
  // loop-carried registers met at every merge point — the MLP waves executed ~40 `v_mov` phi copies and ~35 scalar
  // branch instructions per bridge beside ~105 useful ones (ISA reading), and a lone wave issues one instruction per
  // ~5 cycles whatever it is.  Each role now runs its own copy of the loop with the same barrier sequence (raw
  // `s_barrier` counts arrivals, not program counters).
  // kTGTF: target waves, register-resident mixture (Target::is_fast); kMLPT: the ninth MLP wave of the 132-wide net (coop_tail4)
  enum { kMLP = 0, kTGT = 1, kRNG = 2, kACC = 3, kRNGACC = 4, kTGTF = 5, kMLPT = 6 };
  auto role_loop = [&](auto role_tag) {
    constexpr int R = decltype(role_tag)::value;
    constexpr bool r_mlp = R == kMLP || R == kMLPT, r_tail = R == kMLPT, r_tgt = R == kTGT || R == kTGTF, r_tgtf = R == kTGTF;
    constexpr bool r_rng = R == kRNG || R == kRNGACC, r_acc = R == kACC || R == kRNGACC;
    // The body is instantiated for both parities of the evaluation index: every double-buffered exchange row is then at
    // a compile-time offset from a loop-invariant address (immediate offset field of the LDS instruction) instead of
    // two to four address instructions per access group and bridge.
    auto body = [&](const int i, auto buf_tag) {
      constexpr int buf = decltype(buf_tag)::value;
      const int srow = i < K ? i : K - 1;
      // hand-issued s_load: the row is requested first thing in the iteration and is complete behind barrier 1's own
      // `s_waitcnt lgkmcnt(0)`; the values pass THROUGH that barrier statement (in-out operands), so no use can be
      // scheduled in front of it and the compiler, which does not see a pending scalar load, adds no wait of its own
      f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sd = {0.f, 0.f, 0.f, 0.f};
      if constexpr (!r_rng || r_acc) {
        const float* rowp = sched_p + __builtin_amdgcn_readfirstlane(8 * srow);   // scalar address whatever the allocator thinks of i
        asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x10" : "=&s"(sc), "=&s"(sd) : "s"(rowp));
        __builtin_amdgcn_sched_barrier(0);
      }
      float h[NR];
#pragma unroll
      for (int r = 0; r < NR; ++r) h[r] = 0.f;
      uint32_t g0 = 0, g1 = 0, h0 = 0, h1 = 0;
      typename Target<TARGET, D>::State tst;
      // ------------------------------------------------------------------ interval 1
      if constexpr (r_mlp) {
        float pre[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          pre[r] = brow[r];
#pragma unroll
          for (int j = 0; j < D; ++j) pre[r] += z[j] * w1[j][r];
        }
        if (ARCH == CMCD_ARCH_DDS) {
#pragma unroll
          for (int r = 0; r < NR; ++r) h[r] = gelu_fast(pre[r]);
        } else {
          float u[NR];
#pragma unroll
          for (int r = 0; r < NR; ++r) u[r] = urow[r];
          if (wv == 0) {  // the first D neurons of u are z itself (D <= 16)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              const int nidx = nb + r;   // wv == 0
#pragma unroll
              for (int j = 0; j < D; ++j) u[r] = (nidx == j) ? z[j] : u[r];
            }
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) h[r] = u[r] + softplus(pre[r]);
        }
        if (HALF) {
          *reinterpret_cast<float2*>(my_h) = float2{h[0], h[1]};
        } else {
          *reinterpret_cast<f32x4*>(my_h) = f32x4{h[0], h[1], h[NR - 2], h[NR - 1]};
        }
      } else if constexpr (r_tgt) {
        // distances / shift of z_i (own z, means in registers)
        if constexpr (r_tgtf) Target<TARGET, D>::template pass1f<LPT>(z, sub8, tmeans, tst);
        else Target<TARGET, D>::template pass1r<LPT>(z, sub8, lds_tgt, tmeans, tst);
      } else if constexpr (r_rng && !MERGE) {
       if (i + 1 < K) {
        uint32_t x0 = gb, x1 = 2 + gb;
        threefry2x32(k0, k1, x0, x1);            // (G, H) = split(gen) of bridge i+1
        rows01(x0, g0, g1);
        rows01(x1, h0, h1);
       }
      }
      STAMP(0);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(sc), "+s"(sd)::"memory");   // barrier 1 (+ the schedule row)
      STAMP(1);
      // ------------------------------------------------------------------ interval 2
      if constexpr (r_mlp) {
        // prefetch the next evaluation's first-layer bias row (L2-resident); lands during the MFMAs.
        // CAIS evaluates s(z_{i+1}, i+1); MCD_ULA_sn evaluates s(z_{i+1}, i) (mcd_over_orig.py:44).
        const int nrow = (i < K ? i + 1 : K) - (a.ula == 2 ? 1 : 0);
        // (uniform row base + the lane's constant neuron offset: scalar address arithmetic, saddr form of the load)
        brow = load_row(a.ws + a.w.bias1 + (int64_t)nrow * HP + nb);
        if (ARCH == CMCD_ARCH_GEFFNER) urow = load_row(a.ws + a.w.utab + (int64_t)nrow * HP + nb);
        // layer 2: rows = my 16 output neurons, cols = particles, k = all HP inputs from LDS
        float av[NR], h2[NR];
        if constexpr (r_tail) {
          // 132 = 8 x 16 + 4: this wave's tile holds 4 real neurons.  All 16 blocks of the instruction work on those
          // four — block (pg, slice) contracts slice `slice` of the 144 inputs (8 slices of 20 / 16, 16-byte aligned
          // starts) — 20 matrix instructions instead of 72 on the SIMD that carries three of the nine chains; every lane
          // fetches its own slice of its particle's activations (no row broadcast), the slices are summed over lanes
          // l ^ 8, l ^ 16, l ^ 32.  Steps 16 .. 19 of the short slices re-read valid activations against zero operands.
          const int sl = (lane >> 3) & 7;
          const float* rd_t = hbuf + c * HQP + coop_tail_start(sl);
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
          f32x4 hb[5];
#pragma unroll
          for (int q = 0; q < 5; ++q) hb[q] = *reinterpret_cast<const f32x4*>(rd_t + 4 * ((q == 4 && sl >= 4) ? 3 : q));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int sq = 0; sq < 20; ++sq) {
            f32x4& ac = (sq & 1) ? acc1 : acc;
            ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], hb[sq / 4][sq % 4], ac, 0, 0, 0);
          }
          acc += acc1;
          STAMP(7);
          float tot[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float t = acc[r];
            t += xor8(t);
            tot[r] = group_sum(t);
          }
          // lane (kh, ng) keeps neurons 4 ng + 2 kh + {0, 1} of the tile: real ones only in group 0
          av[0] = (ng == 0 ? (kh ? tot[2] : tot[0]) : 0.f) + b2p[0];
          av[1] = (ng == 0 ? (kh ? tot[3] : tot[1]) : 0.f) + b2p[1];
        } else if (HALF) {
          // 16 blocks of 4 neurons x 4 particles per instruction: block (pg, kh, ng) accumulates half kh of the
          // contraction; no column of the product is wasted (the 16x16x4 shape would carry every particle twice).
          // Step s takes its activation from the row that fetched it (BLGP broadcast); all LDS reads are in flight
          // before the first MFMA, and two accumulators (even / odd inputs): a 4x4x1 MFMA that reads its predecessor's
          // result needs two wait states.  Same accumulation order as the every-lane-reads-everything layout of r01.
          f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#ifdef CMCD_COOP_ACC4   // probe (r03, rejected: 0.1930 -> 0.2011 ms): four independent accumulator chains instead of two
          f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
#endif
          f32x4 hb[RSA / 4];
#pragma unroll
          for (int q = 0; q < RSA / 4; ++q) hb[q] = *reinterpret_cast<const f32x4*>(rd_h + 4 * q);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int sq = 0; sq < NQ; ++sq) {
            const int row = sq / RSA, t = sq % RSA;
            const float bv = hb[t / 4][t % 4];
#ifdef CMCD_COOP_ACC4
            f32x4& ac = (sq & 3) == 0 ? acc : ((sq & 3) == 1 ? acc1 : ((sq & 3) == 2 ? acc2 : acc3));
#else
            f32x4& ac = (sq & 1) ? acc1 : acc;
#endif
            if (row == 0) ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 4);
            else if (row == 1) ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 5);
            else if (row == 2) ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 6);
            else ac = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[sq], bv, ac, 0, 0, 7);
          }
#ifdef CMCD_COOP_ACC4
          acc = (acc + acc2) + (acc1 + acc3);
#else
          acc += acc1;
#endif
          STAMP(7);   // activations read, matrix instructions done
          // the two halves of the contraction sit in lanes l and l ^ 8; lane kh keeps neurons 2 kh + {0, 1} of its group:
          // it adds its own half of those to the partner's (the partner's register of the pair, through DPP row_ror:8)
          av[0] = ((kh ? acc[2] : acc[0]) + xor8(kh ? acc[0] : acc[2])) + b2p[0];
          av[1] = ((kh ? acc[3] : acc[1]) + xor8(kh ? acc[1] : acc[3])) + b2p[1];
        } else {
          f32x4 acc = b2v;
#pragma unroll
          for (int ti = 0; ti < T; ++ti) {
            const f32x4 hb = *reinterpret_cast<const f32x4*>(rd_h + ti * 256);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[ti][r], hb[r], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < NR; ++r) av[r] = acc[r & 3];
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) h2[r] = (ARCH == CMCD_ARCH_DDS) ? gelu_fast(av[r]) : h[r] + softplus(av[r]);
        STAMP(8);   // contraction halves folded, activation
        if (HALF) {
          // outputs in pairs (j, j + 1): lane kh = 0 collects both contraction halves of output j, lane kh = 1 both of
          // output j + 1 (own term + the partner's other term through DPP row_ror:8), so the sum over the 4 neuron groups
          // (the rows of the wave) runs once per pair instead of once per output
          static_assert(D % 2 == 0, "8-particle tiles pair the outputs");
#pragma unroll
          for (int j = 0; j < D; j += 2) {
            const float p0 = h2[0] * w3[j][0] + h2[1] * w3[j][1];
            const float p1 = h2[0] * w3[j + 1][0] + h2[1] * w3[j + 1][1];
            float pj = (kh ? p1 : p0) + xor8(kh ? p0 : p1);   // kh = 0: output j, kh = 1: output j + 1
            pj = group_sum(pj);
            if (ng == 0) {   // both twin columns
              part[(buf * 16 + c) * PT + wv * D + j + kh] = pj;
              part[(buf * 16 + c + 8) * PT + wv * D + j + kh] = pj;
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < D; ++j) {
            float pj = h2[0] * w3[j][0] + h2[1] * w3[j][1] + h2[NR - 2] * w3[j][NR - 2] + h2[NR - 1] * w3[j][NR - 1];
            pj = group_sum(pj);
            if (g == 0) part[(buf * 16 + c) * PT + wv * D + j] = pj;
          }
        }
      } else if constexpr (r_tgt) {
        // second pass only: its SIMD partner (an MLP wave) blocks the VALU during the 16 fp32 MFMAs
        float gp[D], lp = 0.f;
        if constexpr (r_tgtf) Target<TARGET, D>::template pass2f<LPT>(z, sub8, tmeans, tst, lp, gp);
        else Target<TARGET, D>::template pass2<LPT>(z, sub8, lds_tgt, tst, lp, gp);
        if (sub8 < (HALF ? 2 : 1)) {   // HALF: lane sub 1 (same totals) fills the twin column
          const int col = c + 8 * sub8;
          const float cA = sd[1], cB = sd[2];   // eps_i beta_i, eps_i (1 - beta_i): the row phase C(i) uses
#pragma unroll
          for (int j = 0; j < D; ++j) {
            const float gpc = __builtin_amdgcn_fmed3f(gp[j], -cp, cp);
            const float gqc = __builtin_amdgcn_fmed3f((qmean[j] - z[j]) * qiv[j], -cq, cq);
            gpb[(buf * 16 + col) * GP + j] = fmaf(cA, gpc, fmaf(cB, gqc, z[j]));
            gpb[(buf * 16 + col) * GP + D + j] = gp[j];
          }
          gpb[(buf * 16 + col) * GP + 2 * D] = lp;
        }
        if (MERGE && i < K) {
          // 12-wave instance: the merged RNG / ACC wave is the longest stream of the workgroup and the target waves wait
          // ~1200 cycles at barrier 2, so the bits -> deviates conversion of bridge i (raw[buf], written one iteration
          // ago) runs here: lane `sub` of a particle converts word `sub` and fills both twin columns
          static_assert(!MERGE || D <= LPT, "the conversion is dealt to the lanes of a particle");
          if (sub8 < D) {
            const uint32_t bits = raw[(buf * 16 + c) * NZ + sub8];
            const float dev = bits_to_normal(bits);
            nzb[(buf * 16 + c) * NZ + sub8] = dev;
            if (HALF) nzb[(buf * 16 + c + 8) * NZ + sub8] = dev;
            if (a.dbg_bits && valid) {
              a.dbg_bits[((int64_t)(i + 1) * a.n + p) * D + sub8] = bits;
              a.dbg_noise[((int64_t)(i + 1) * a.n + p) * D + sub8] = dev;
            }
          }
        }
      } else if constexpr (r_rng) {
        // MERGE: nothing this wave produces is read before barrier 2 (the log-weight is its own, the bits and deviates are
        // consumed in phase C), so its phase C runs straight into barrier 1 and the whole key-chain stage sits here — the
        // MLP waves would otherwise wait at barrier 1 for a split they do not need (r02 stamps: 700 - 1100 cycles)
        if (MERGE && i + 1 < K) {
          uint32_t x0 = gb, x1 = 2 + gb;
          threefry2x32(k0, k1, x0, x1);            // (G, H) = split(gen) of bridge i+1
          rows01(x0, g0, g1);
          rows01(x1, h0, h1);
        }
        if (i + 1 < K) normal_bits(g0, g1, h0, h1, buf ^ 1, true, i + 2);   // (MERGE: the target waves convert raw[buf])
      } else {
        if (i < K) convert(buf, i + 1);                // noise of bridge i, read in phase C(i)
      }
      STAMP(2);
      lds_barrier();
      STAMP(3);
      // ------------------------------------------------------------------ phase C: MLP, TGT (own copies of z) and
      // ACC (with the log-weight); ~45 instructions each, so the redundancy is cheaper than an LDS hand-over
      if constexpr (r_acc) {
        phase_c(i, true, sc, sd, buf_tag);            // i = K: closes step K-1 and picks up log p(z_K); PUBZ: barrier 3 inside
      } else if constexpr (PUBZ) {
        lds_barrier();                                // barrier 3: z_{i+1} published by the ACC wave
        STAMP(5);
        if constexpr (r_mlp || r_tgt) {
          if (i < K) {
#pragma unroll
            for (int q = 0; q < DP; q += 4) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(zpub + c * DP + q);
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (q + r < D) z[q + r] = v[r];
            }
          }
        }
      } else if constexpr (r_mlp || r_tgt) {
        if (i < K) phase_c(i, false, sc, sd, buf_tag);
      }
      STAMP(4);
    };
    for (int i = 0; i <= K; i += 2) {
      body(i, std::integral_constant<int, 0>{});
      if (i + 1 <= K) body(i + 1, std::integral_constant<int, 1>{});
    }
  };
  if (is_mlp) {
    if constexpr (HALF && T == 9) {
      if (a.tail && wv == T - 1) role_loop(std::integral_constant<int, kMLPT>{});
      else role_loop(std::integral_constant<int, kMLP>{});
    } else {
      role_loop(std::integral_constant<int, kMLP>{});
    }
  }
  else if (is_tgt) {
    if (Target<TARGET, D>::kHasFast && __builtin_amdgcn_readfirstlane((int)Target<TARGET, D>::is_fast(tmeans)))
      role_loop(std::integral_constant<int, kTGTF>{});
    else
      role_loop(std::integral_constant<int, kTGT>{});
  }
  else if (MERGE) role_loop(std::integral_constant<int, kRNGACC>{});
  else if (is_rng) role_loop(std::integral_constant<int, kRNG>{});
  else role_loop(std::integral_constant<int, kACC>{});
#ifdef CMCD_STAMPS
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 16; ++k) g_stamps[wv][k] = st_acc[k];
#endif

  if (!is_acc) return;
  w += logp;  // + log p(z_K)   mcdboundingmachine.py:178
  const float loss = -w;
  if (valid && own && g == 0) {
    a.out_loss[p] = loss;
#pragma unroll
    for (int j = 0; j < D; ++j) a.out_z[p * D + j] = z[j];
  }
  const bool use = valid && own && g == 0;
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
    return;
  }
  // Fused merge: the record goes out with agent-scope (write-through) stores, the workgroup takes a ticket, and the LAST one
  // to arrive merges all records in finalize_kernel's order — the finalize launch (4.4 us + a kernel boundary) is gone from
  // every forward call; the protocol is the lgcp launch sequence's (cmcd_lgcp.hip): no L2 write-back fence.
  int last = 0;
  if (lane == 0) {
    double* o = a.partials + tile * CMCD_NSTATS;
    __hip_atomic_store(o + 0, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, sm, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 2, sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 3, mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 4, ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the ticket is a release (this workgroup's record happens-before it) / acquire (the last arriver sees every earlier
    // arriver's record) read-modify-write at agent scope: a well-defined protocol under the HIP memory model, not one that
    // leans on the write-through stores draining in program order.  Only grids of <= 64 workgroups take this path, one lane
    // each, once per launch: the L2 write-back the release implies has nothing to write (the records went out write-through)
    last = __hip_atomic_fetch_add(a.fin_counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other lanes of the merging wave read the records too
  wave_merge_stats(a.partials, (int)gridDim.x, a.fin_out, lane);
  if (a.stamp_slot && lane < CMCD_NSTATS && *a.stamp_slot != a.stamp_expect) a.fin_out[lane] = __builtin_nan("");
  // ready for the next launch on these tables (cmcd_bound_forward_prepared skips the prep launch that used to zero it)
  if (lane == 0) __hip_atomic_store(a.fin_counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

typedef void (*coop_fn)(TrajArgs);
struct CoopInstance {
  coop_fn fn = nullptr;
  int waves = 0;   // per workgroup
};

template <int TARGET, int ARCH, int D>
static CoopInstance pick_T(int T, bool half) {
  switch (T) {
    case 2: return {half ? coop_kernel<TARGET, ARCH, D, 2, true> : coop_kernel<TARGET, ARCH, D, 2, false>, 6};
    case 4: return {half ? coop_kernel<TARGET, ARCH, D, 4, true> : coop_kernel<TARGET, ARCH, D, 4, false>, 8};
    case 9:   // 132-wide net; the 2-d targets also on 8-particle tiles (merged RNG / ACC wave: 12 waves)
      if (half) {
        if constexpr (D == 2) return {coop_kernel<TARGET, ARCH, D, 9, true, true>, 12};
        return {};
      }
      // 16-particle tiles: thirteen waves would be four per SIMD = 128 registers (76 spilled for many_gmm); the 2-d
      // targets take the merged RNG / ACC wave here too
      if constexpr (D == 2) return {coop_kernel<TARGET, ARCH, D, 9, false, true>, 12};
      else return {coop_kernel<TARGET, ARCH, D, 9, false>, 13};
    default: return {};
  }
}

static CoopInstance pick(const cmcd_desc& d, int T, bool half) {
  const int arch = d.arch == CMCD_ARCH_DDS ? CMCD_ARCH_DDS : CMCD_ARCH_GEFFNER;
  if (arch == CMCD_ARCH_DDS) {
    if (T != 4) return {};
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2>(T, half);
    if (d.target == CMCD_TARGET_GMM && d.dim == 2) return pick_T<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2>(T, half);
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return pick_T<CMCD_TARGET_FUNNEL, CMCD_ARCH_DDS, 10>(T, half);
    return {};
  }
  if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2>(T, half);
  if (d.target == CMCD_TARGET_GMM && d.dim == 2) return pick_T<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2>(T, half);
  if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return pick_T<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10>(T, half);
  return {};
}

#ifdef CMCD_STAMPS
extern "C" int cmcd_debug_read_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16 * 16);
}
#endif

bool coop_available(const cmcd_desc& d, int T) { return pick(d, T, false).fn != nullptr; }
bool coop_half_available(const cmcd_desc& d, int T) { return pick(d, T, true).fn != nullptr; }

// half: 8-particle tiles (ceil(n / 8) workgroups, as many statistics records); else ta.w.n_waves 16-particle tiles
// Issue priority per role (s_setprio against the SIMD partner), tools/probes/prio_sweep.py, interleaved rounds
// (profiles/r01_p_prio_sweep*.txt).  On 8-particle tiles with the 4x4x1 MFMA chain: target and ACC waves one level above
// their MLP partners -0.5 % (many_gmm, N = 2000) / -2.4 % (funnel d = 10: ten deviates and ten log-weight terms per
// bridge on ACC); RNG above its MLP partner +4 ... +8 % in every build (the MLP chain is what the barrier waits for on
// that SIMD).  On 16-particle tiles no raise helps.  (Before the 4x4x1 chain the target waves were the long pole and
// their raise alone gave -2.1 %: the table follows the balance, re-run the sweep after changing a role.)
static int g_coop_prio = -1;   // -1: the table below
#ifndef CMCD_NO_DIAG_HOOKS   // include/cmcd_hip_diag.h
extern "C" void cmcd_debug_set_coop_prio(int prio) { g_coop_prio = prio; }   // tools/probes/prio_sweep.py
#endif
// 12-wave instance (132-wide net, RNG + ACC merged): the merged wave's stream is the longest of the workgroup (three
// Threefry passes, two deviates and the log-weight per bridge against two partners with 72 matrix instructions each);
// target waves and the merged wave one level up: 0.4957 -> 0.4602 ms at N = 2000 (profiles/r02_t_prio_t9.txt)
static int default_prio(const cmcd_desc&, bool half, int waves) {
  if (waves == 12) return 1 << 2 | 1 << 4;   // both tilings of the 12-wave instance (16-particle tiles: -1.6 %, r02_v)
  if (!half) return 0;
  return 1 << 2 | 1 << 6;
}

int coop_launch(const cmcd_desc& d, const TrajArgs& ta_in, bool half, void* stream, bool narrow) {
  TrajArgs ta = ta_in;
  const int T = ta.w.T, D = d.dim, Hh = (D + 1) / 2, NZ = 2 * Hh;
  const int GP = (2 * D + 1 + 3) & ~3, PT = (T * D + 3) & ~3;
  const CoopInstance inst = pick(d, T, half);
  coop_fn fn = inst.fn;
  if (!fn) return CMCD_ERR_UNSUPPORTED;
  ta.prio = g_coop_prio >= 0 ? g_coop_prio : default_prio(d, half, inst.waves);
  ta.tail = half && d.arch == CMCD_ARCH_GEFFNER && coop_tail4(T, net_in_dim(d) + d.emb_dim);
  const size_t lds_bytes = size_t(16 * T * 16 + 2 * 16 * PT + 2 * 16 * GP + 2 * 16 * NZ + 2 * 16 * NZ + ta.w.tgt_floats +
                                 (D > 4 ? 16 * ((D + 3) & ~3) : 0)) * 4;   // + the published state (d > 4)
  const unsigned tiles = half ? unsigned((ta.n + 7) / 8) : (unsigned)ta.w.n_waves;
  // While there are no more workgroups than CUs, claim more than half of a CU's 160 KB of LDS: the dispatcher can
  // then never put two workgroups on one CU while another CU sits idle (two on a CU share its SIMDs and the
  // slower pair sets the kernel time: measured 0.316 vs 0.297 ms between 250 and 128 workgroups without this).
  size_t lds_claim = lds_bytes;
  int dev = 0, n_cu = 0;
  static std::atomic<int> cu_count[64];   // per device, queried once (the attribute call costs microseconds per launch)
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
    n_cu = cu_count[dev].load(std::memory_order_relaxed);
    if (n_cu == 0 && hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess)
      cu_count[dev].store(n_cu, std::memory_order_relaxed);
  }
  constexpr size_t kExclusive = 84 * 1024;
  if (half && !narrow && coop_wide8_available(d, T)) {   // wide states (d = 10): their own kernel, same claim rule
    // no role raised there: the MLP waves are the critical path on every SIMD (profiles/r05_c_prio_sweep_funnel_wide8.txt:
    // target + ACC one level up, this file's table for 8-particle tiles, +2.5 %; MLP up: +-0.3 %)
    if (g_coop_prio < 0) ta.prio = 0;
    return coop_wide8_launch(d, ta, (n_cu > 0 && (int)tiles <= n_cu) ? kExclusive : 0, stream);
  }
  if (n_cu > 0 && (int)tiles <= n_cu) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> raised;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(reinterpret_cast<const void*>(fn), dev);
    bool ok = raised.count(key) != 0;
    if (!ok && hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)kExclusive) == hipSuccess) {
      raised.insert(key);
      ok = true;
    }
    if (ok && lds_claim < kExclusive) lds_claim = kExclusive;
  }
  hipLaunchKernelGGL(fn, dim3(tiles), dim3(64 * inst.waves), lds_claim,
                     static_cast<hipStream_t>(stream), ta);
  return CMCD_OK;
}

}  // namespace cmcd
