// lgcp (d = 1600, geffner net of width 1620) on WIDE batches — the reference's evaluation batches (`opt.sample`,
// /root/reference/src/opt.py:167-197: n_input_dist_seeds x n_samples = 30 x 500 particles, README.md:63) and any other
// forward-only call of >= kWideMin particles.
//
// cmcd_lgcp.hip serves <= 32 particles per pass: there one evaluation is three weight-STREAMING skinny GEMMs (41.5 MB of
// weights per pass and evaluation) and a batch of 600 re-streams the weights nineteen times.  Above ~50 particles per weight
// pass the evaluation is matrix-pipe bound (2 N FLOP per 4 weight bytes against a machine balance of
// ~25 FLOP / B), so this file runs the same per-evaluation launch sequence
//   A: x W1[:d]           -> u1 = [x; emb_i] + softplus(. + bias1_i)                                nn.py:45-50,68
//   B: u1 W2              -> u2 = u1 + softplus(. + b2)          and   (x - mu0) K^-1 -> kr          model_handler.py:386-396
//   C: u2 W3              -> the state update of evaluation i (closes step i-1, opens step i)       mcd_cais.py:46-89
// on ALL particles at once with a real fp32 GEMM body:
//
//   * a workgroup (4 waves) owns a 32-row x 128-column output tile over the WHOLE contraction: no split-K across
//     workgroups, hence no slab / ticket / last-arriver seam and no inter-workgroup communication at all;
//   * the contraction is split over the four WAVES in interleaved 8-deep chunks; every operand goes global -> register ->
//     v_mfma_f32_32x32x2_f32 (exact fp32) with no LDS staging and no barrier in the loop: per chunk a wave issues one
//     16-byte load of its activation rows (lane = row, 4 consecutive k) and four 16-byte loads of weight rows (lane = 4
//     consecutive columns of one k row: the tile's 128 columns are FOUR 32-column MFMA blocks interleaved by column mod 4,
//     so one load feeds four matrix instructions) — 5 loads per 16 matrix instructions (1024 matrix-pipe cycles), three
//     chunks in flight;
//   * the four partial tiles are summed in fixed order through LDS (67.6 KB) and the consumer runs on the summed tile:
//     4 consecutive columns of one row per thread, vector loads / stores, per-row partial log-weights by a fixed butterfly
//     over the 32 lanes that share the row => bitwise deterministic and independent of a particle's row or tile;
//   * weights are re-packed once per call into zero-padded [K rounded to 32][N rounded to 128] arrays (16-byte aligned rows:
//     `params_flat` leaves sit at arbitrary offsets) and the activations live in [rows rounded to 32][width rounded to 32]
//     buffers whose padding stays zero, so the loop has no edge predicate;
//   * tile -> workgroup mapping is XCD-aware: consecutive tiles (row tiles of one column tile first) go to ONE XCD, so a
//     column tile's 850 KB of weights is fetched into one L2 instead of eight.
//
// The chain keys G_0 .. G_{K-1} of every particle (mcd_cais.py:66-67,87,94) are produced once per call by the init launch.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x3 __attribute__((ext_vector_type(3)));

constexpr int kWRows = 32;       // rows of a tile (one 32x32x2 MFMA block)
constexpr int kWCols = 128;      // columns of a tile (four interleaved MFMA blocks)
constexpr int kRedLd = 132;      // LDS row pitch of the cross-wave sum (floats; 528 B: 16-byte aligned, off the bank period)
constexpr int kWideLds = 4 * kWRows * kRedLd * 4;

static inline int64_t rup(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

enum { WEPI_ACT1 = 1, WEPI_ACT2 = 2, WEPI_KR = 3, WEPI_STEP = 4, WEPI_STEP_NONET = 5 };

struct WideSeg {
  const float* A;    // [Mp][lda]   activations; padding rows / columns hold zeros
  const float* W;    // [Kp][ldw]   packed weights; padding rows / columns hold zeros
  int lda, ldw, Kp;  // Kp: multiple of 32
  int epi;
  float a_shift;     // the operand is A - a_shift   (K^-1 (x - mu0) without a subtraction pass)
  int ncols;         // packed columns that hold weights (the rest of the last column tile is zero padding)
};

struct WideStep {            // evaluation i at z_i: closes step i-1, opens step i (i = K: collects log p(z_K))
  const float* sched;        // [K][8]
  const float* tc;           // {Kinv[d,d], counts[d], mu0, a, lognorm}
  const float* xp;           // [Mp][ldx]  previous z
  float* xn;                 // [Mp][ldx]  next z.  The state rotates through THREE buffers (cur -> prev, next -> cur on the host):
                             // in MCD_ULA the state is the operand of the very launch that updates it, and a workgroup of
                             // another column tile may still be streaming a row this one has finished (r04: an in-place
                             // update there read as a 2.4-nat ELBO shift on 15 000-particle evaluations, and only there)
  const float* kr;           // [Mp][ldx]  (z - mu0) K^-1 of this evaluation (launch B)
  const float* b3;           // packed, [Np]
  const float* mean;         // packed vd.mean
  const float* sd;           // packed 1 / exp(vd.logdiag)^2
  const float* counts;       // packed
  const float* factor;       // factor_sn (device scalar)
  const uint32_t* gktab;     // [K][n][2]
  float* wslot;              // [CT][Mp]  running sum over closed steps of (bk - fk) on the tile's columns
  float* fkslot;             // [CT][Mp]  forward-kernel log-density of the open step on the tile's columns
  float* lpslot;             // [CT][Mp]  log p(z_K) on the tile's columns
  float* out_z;              // [n][D]
  int64_t n;
  int Mp, K, i, var_mode, grad_clipping, ula;
};

struct WideArgs {
  WideSeg seg[2];
  int nct0, CT, RT, M;       // column tiles of segment 0 / of the launch, row tiles, real rows
  int narrow;                // many-round grid: tiles run only the column blocks that hold weights (wide_xcd_ranges)
  int xbeg[9];               // XCD x runs tiles [xbeg[x], xbeg[x + 1]) of the order "row tiles of column tile 0, of column tile 1, ..."
  const float* bias;         // ACT1: bias1_i [IN] (workspace);  ACT2: packed b2
  const float* emb;          // ACT1: emb_i [E] (params)
  const float* x;            // [Mp][ldx]  z of this evaluation (operand of A / B's second product, read by ACT1 and STEP)
  const float* u_prev;       // ACT2: u1
  float* u_out;              // ACT1: u1, ACT2: u2      [Mp][ldu]
  float* kr_out;             // KR: [Mp][ldx]
  int D, IN, ldx, ldu;
  WideStep st;
};

__device__ __forceinline__ float half_sum32(float v) {   // sum over the 32 lanes of a half wave, fixed butterfly
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// The state update of evaluation i for 4 consecutive columns [col, col + 4) of particle row m (the arithmetic of
// lgcp_step_tile in cmcd_lgcp.hip: /root/reference/src/mcd_cais.py:46-89, model_handler.py:386-396); `o` = (u2 W3)[cols]
// (NONET: `o` = the K^-1 product itself).  The three per-row partial log-weights are returned for the caller's butterfly.
// PAIR (launch C with a network, r05): the tile's packed columns are PAIRED — packed column 128 ct + 4 c4 + {0, 1, 2, 3} is
// element {e, e + 1, e + H, e + H + 1}, e = 64 ct + 2 c4, H = D / 2 (wide_nat_col) — because elements e and e + H are the two
// words of ONE Threefry block of normal(G_i, (D,)) (counters (e, H + e)): the thread runs two blocks for its four deviates
// instead of four (the block was ~80 of the ~150 instructions per element of this consumer; launch C at N = 600 40.2 ->
// see profiles/r05_e_lgcp_per_launch_n600.txt).  W3's columns, b3, vd and the counts are packed in that order once per call; the
// state rows x / xp / xn / kr and out_z keep the natural order (two 8-byte accesses per thread instead of one 16-byte).
__host__ __device__ __forceinline__ int wide_nat_col(int q, int H) {       // packed column of a PAIRED tile -> element
  const int ct = q >> 7, r = q & 127, c4 = r >> 2, j = r & 3;
  const int e = 64 * ct + 2 * c4 + (j & 1);
  if (e >= H) return 2 * H + e;                                  // pairs past H (the last tile's tail): outside [0, D) on both sides
  return (j & 2) ? e + H : e;
}

// The consumer's own operands of one (row, 4 columns) element, requested ahead of the arithmetic that uses them: the state
// rows (z, z of the previous evaluation, the K^-1 product of launch B) BEFORE the contraction loop — they come from earlier
// launches — and the per-column vectors and the chain key behind it.  r04 loaded them inside the element loop, after the
// cross-wave sum: four dependent trips to L2 / HBM per thread behind every tile (launch C at N = 600: 40.2 us for 22.2 us of
// matrix time).
struct WideStepRegs {
  f32x4 zv, xpv, krv, b3v, mnv, sdv, cnv;
  uint32_t g0, g1;
  int e0;
  bool in;
};

template <bool NO_NET, bool PAIR>
__device__ __forceinline__ void wide_step_load_state(const WideArgs& a, int m, int col, bool live, WideStepRegs& r) {
  const WideStep& s = a.st;
  const int D = a.D, H = (D + 1) / 2;
  // natural order: col is a multiple of 4 and D % 4 == 0 (checked on the host): the four columns are inside or outside
  // together.  Paired order: element e0 = 64 ct + 2 c4 (even, H even): the pair (e0, e0 + 1) and its partners are in or out together
  r.e0 = PAIR ? wide_nat_col(col, H) : col;
  r.in = live && (PAIR ? r.e0 < H : col < D);
  r.zv = r.xpv = r.krv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!r.in) return;
  if (PAIR) {
    const int64_t r0 = (int64_t)m * a.ldx + r.e0, r1 = r0 + H;
    const f32x2 za = *reinterpret_cast<const f32x2*>(a.x + r0), zb = *reinterpret_cast<const f32x2*>(a.x + r1);
    const f32x2 pa2 = *reinterpret_cast<const f32x2*>(s.xp + r0), pb2 = *reinterpret_cast<const f32x2*>(s.xp + r1);
    const f32x2 ka = *reinterpret_cast<const f32x2*>(s.kr + r0), kb = *reinterpret_cast<const f32x2*>(s.kr + r1);
    r.zv = f32x4{za[0], za[1], zb[0], zb[1]};
    r.xpv = f32x4{pa2[0], pa2[1], pb2[0], pb2[1]};
    r.krv = f32x4{ka[0], ka[1], kb[0], kb[1]};
  } else {
    const int64_t ro = (int64_t)m * a.ldx + col;
    r.zv = *reinterpret_cast<const f32x4*>(a.x + ro);
    r.xpv = *reinterpret_cast<const f32x4*>(s.xp + ro);
    if (!NO_NET) r.krv = *reinterpret_cast<const f32x4*>(s.kr + ro);
  }
}

__device__ __forceinline__ void wide_step_load_cols(const WideArgs& a, int m, int col, WideStepRegs& r) {
  const WideStep& s = a.st;
  r.g0 = r.g1 = 0u;
  r.b3v = r.mnv = r.sdv = r.cnv = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!r.in) return;
  // (the per-column vectors are packed in the tile's own column order)
  r.b3v = *reinterpret_cast<const f32x4*>(s.b3 + col);
  r.mnv = *reinterpret_cast<const f32x4*>(s.mean + col);
  r.sdv = *reinterpret_cast<const f32x4*>(s.sd + col);
  r.cnv = *reinterpret_cast<const f32x4*>(s.counts + col);
  if (s.i < s.K) {
    const uint32_t* gk = s.gktab + ((int64_t)s.i * s.n + m) * 2;
    r.g0 = gk[0]; r.g1 = gk[1];
  }
}

template <bool NO_NET, bool PAIR>
__device__ __forceinline__ void wide_step4(const WideArgs& a, int m, int col, const float (&o)[4], const WideStepRegs& r,
                                           float& bk_s, float& fk_s, float& lp_s) {
  const WideStep& s = a.st;
  const int D = a.D, H = (D + 1) / 2, i = s.i;
  const bool last = i == s.K;
  const float fsn = s.ula ? 0.f : 1.f;
  const float mu0 = s.tc[(int64_t)D * D + D], pa = s.tc[(int64_t)D * D + D + 1];
  const float clipv = s.var_mode ? 1e2f : 1e3f;
  const bool clip_p = s.grad_clipping != 0, clip_q = clip_p && s.var_mode;
  const float* sp = s.sched + 8 * (i > 0 ? i - 1 : 0);
  const float pbeta = sp[0], peps = sp[1], pcst = sp[3], pinv2s2 = sp[4];
  const float* sc = s.sched + 8 * (last ? s.K - 1 : i);
  const float beta = sc[0], eps = sc[1], sig = sc[2], cst = sc[3], inv2s2 = sc[4];
  const float fac = NO_NET ? 0.f : s.factor[0];
  if (!r.in) return;
  const int e0 = r.e0;
  int el[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) el[j] = PAIR ? e0 + (j & 1) + ((j & 2) ? H : 0) : col + j;
  const f32x4 zv = r.zv, xpv = r.xpv;
  f32x4 krv = r.krv;
  if (NO_NET) { krv[0] = o[0]; krv[1] = o[1]; krv[2] = o[2]; krv[3] = o[3]; }
  const f32x4 b3v = r.b3v, mnv = r.mnv, sdv = r.sdv, cnv = r.cnv;
  const uint32_t g0 = r.g0, g1 = r.g1;
  // eps_i = normal(G_i, (D,)): element e is word (e >= H) of the block with counters (jj, H + jj), jj = e mod H
  uint32_t bits[4] = {0u, 0u, 0u, 0u};
  if (!last) {
    if (PAIR) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t y0 = (uint32_t)(e0 + q), y1 = (uint32_t)(H + e0 + q);
        threefry2x32(g0, g1, y0, y1);
        bits[q] = y0; bits[2 + q] = y1;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = col + j, jj = e < H ? e : e - H;
        uint32_t y0 = jj, y1 = (H + jj < D) ? H + jj : 0;
        threefry2x32(g0, g1, y0, y1);
        bits[j] = e < H ? y0 : y1;
      }
    }
  }
  f32x4 znv;
  float bk_acc = 0.f, fk_acc = 0.f, lp_acc = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float z = zv[j], kr = krv[j], cnt = cnv[j], sd = sdv[j];
    const float sn = NO_NET ? 0.f : (o[j] + b3v[j]) * fac;       // factor_sn (u2 W3 + b3)        nn.py:70
    // e^z on v_exp_f32 (<= 4e-7 relative at the log-intensities of this target) and 1 / sd^2 from the packed vector: the libm
    // exponential and an IEEE division were ~35 of this consumer's ~190 VALU instructions per element, on a SIMD where they
    // add to the matrix time (launch C at N = 600: 37.5 -> see profiles/r05_h_lgcp_wide_valu_trims.txt)
    const float ez = __builtin_amdgcn_exp2f(1.44269504088896340736f * z);
    float gp = -kr + cnt - pa * ez;                               // grad log p                    model_handler.py:386-396
    float gq = -(z - mnv[j]) * sd;                                // sd holds 1 / std^2 (lgcp_wide_vec_kernel)
    if (clip_p) gp = fminf(fmaxf(gp, -clipv), clipv);
    if (clip_q) gq = fminf(fmaxf(gq, -clipv), clipv);
    float zn = 0.f;
    if (i > 0) {     // backward kernel of step i-1                                                mcd_cais.py:71-86
      const float ub = -1.0f * (pbeta * gp + (1.0f - pbeta) * gq);
      const float bk = z - peps * ub + peps * sn;
      const float db = xpv[j] - bk;
      bk_acc += -(db * db) * pinv2s2 - pcst;
    }
    if (last) {      // log p(z_K)
      lp_acc += -0.5f * (z - mu0) * kr + z * cnt - pa * ez;
    } else {         // forward kernel of step i                                                   mcd_cais.py:52-67
      const float uf = -1.0f * (beta * gp + (1.0f - beta) * gq);
      const float fk = z - eps * uf - fsn * eps * sn;
      zn = fk + sig * bits_to_normal(bits[j]);
      const float df = zn - fk;
      fk_acc += -(df * df) * inv2s2 - cst;
    }
    znv[j] = zn;
  }
  if (last) {
#pragma unroll
    for (int j = 0; j < 4; ++j) s.out_z[(int64_t)m * D + el[j]] = zv[j];
  } else if (PAIR) {
    const int64_t r0 = (int64_t)m * a.ldx + e0;
    *reinterpret_cast<f32x2*>(s.xn + r0) = f32x2{znv[0], znv[1]};
    *reinterpret_cast<f32x2*>(s.xn + r0 + H) = f32x2{znv[2], znv[3]};
  } else {
    *reinterpret_cast<f32x4*>(s.xn + (int64_t)m * a.ldx + col) = znv;
  }
  bk_s = bk_acc; fk_s = fk_acc; lp_s = lp_acc;
}

#ifndef CMCD_WIDE_BUF
#define CMCD_WIDE_BUF 1
#endif
#ifndef CMCD_WIDE_NB
#define CMCD_WIDE_NB 1        // 0: every tile runs all four column blocks (A / B)
#endif
// column blocks (of 32) a tile with `real` weight columns needs
__host__ __device__ __forceinline__ int wide_tile_blocks(int real) {
  if (!CMCD_WIDE_NB || !CMCD_WIDE_BUF) return 4;
  const int b = (real + 31) >> 5;
  return b < 1 ? 1 : (b > 4 ? 4 : b);
}

__global__ __launch_bounds__(256, 2) void lgcp_wide_gemm_kernel(WideArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red[];   // [4 waves][32 rows][kRedLd]
  // XCD-aware tile mapping: workgroup b runs on XCD b % 8; XCD x takes a contiguous tile range of the order "row tiles of
  // column tile 0, row tiles of column tile 1, ...", so the workgroups that share a column tile's weights share an L2.  The
  // ranges are cut by the host (wide_xcd_ranges): equal counts, or equal COST on many-round grids (a last column tile with
  // fewer than four column blocks is cheaper, below)
  const int xcd = blockIdx.x & 7, t = a.xbeg[xcd] + (int)(blockIdx.x >> 3);
  if (t >= a.xbeg[xcd + 1]) return;
  const int ct = t / a.RT, rt = t - ct * a.RT;
  const int sI = ct >= a.nct0 ? 1 : 0;
  const WideSeg sg = a.seg[sI];
  const int ctl = ct - (sI ? a.nct0 : 0), n0 = ctl * kWCols;
  // r05: column blocks of this tile that hold weights.  The last column tile of a 1620- / 1600-wide layer has 84 / 64 real
  // columns of 128: with nb = 3 / 2 blocks the lane fetches 3 / 2 consecutive columns per k row (block j = columns nb c + j: the
  // packed arrays keep their order) and issues 3 / 2 matrix instructions per step instead of 4
  const int nb = (CMCD_WIDE_BUF && a.narrow) ? __builtin_amdgcn_readfirstlane(wide_tile_blocks(sg.ncols - n0)) : 4;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5;
  const int nch = sg.Kp >> 3, T = nch >> 2;                      // 8-deep chunks; T per wave (Kp is a multiple of 32)
  const float* Ap = sg.A + (int64_t)(rt * kWRows + c) * sg.lda + 4 * h;
  const float* Wp = sg.W + (int64_t)(4 * h) * sg.ldw + n0 + 4 * c;
  const int64_t ldw = sg.ldw;
  const float shift = sg.a_shift;
  // r05: the operands come through buffer descriptors — per-lane byte offset in a register that never changes, the chunk's
  // position in a SCALAR offset (`buffer_load ... s_off offen`): no vector address arithmetic in the loop (the flat form
  // spent ~10 of its ~14 VALU instructions per chunk on 64-bit addresses; on gfx950 VALU and fp32 matrix time add).
  const int wvu = __builtin_amdgcn_readfirstlane(wv);
  const uint32_t voffA = (uint32_t)(((rt * kWRows + c) * sg.lda + 4 * h) * 4);
  const uint32_t voffW = (uint32_t)((4 * h * sg.ldw + n0 + nb * c) * 4);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(sg.A), 0, __builtin_amdgcn_readfirstlane(a.RT * kWRows * sg.lda * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(sg.W), 0, __builtin_amdgcn_readfirstlane(sg.Kp * sg.ldw * 4), 0x00020000);
  const int ldwB = __builtin_amdgcn_readfirstlane(sg.ldw * 4);

  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;

  const int epi = sg.epi;

  // chunk ch = 4 tt + wave (interleaved over the waves: the workgroup walks the contraction front to back together).
  // MFMA step s of a chunk contracts the k pair (k0 + s, k0 + 4 + s): lane half h supplies k0 + 4 h + s for both operands,
  // so its four A values are ONE 16-byte load and step s's four B values (the four column blocks) another.
#ifndef CMCD_WIDE_DEPTH
#define CMCD_WIDE_DEPTH 3
#endif
  constexpr int P = CMCD_WIDE_DEPTH;     // chunks in flight per wave
  f32x4 av[P], bv[P][4];
  auto issue = [&](auto nb_tag, f32x4& a_, f32x4 (&b_)[4], int tt) {
    constexpr int NB = decltype(nb_tag)::value;
    if (CMCD_WIDE_BUF) {
      const int ch = min(4 * tt + wvu, nch - 1);                 // past the end: a valid, unused reload of the last chunk
      a_ = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, voffA, 32 * ch, 0));
      const int wo = 8 * ch * ldwB;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if constexpr (NB == 4) {
          b_[s] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsW, voffW, wo + s * ldwB, 0));
        } else if constexpr (NB == 3) {
          const f32x3 v = __builtin_bit_cast(f32x3, __builtin_amdgcn_raw_buffer_load_b96(rsW, voffW, wo + s * ldwB, 0));
          b_[s] = f32x4{v[0], v[1], v[2], 0.f};
        } else if constexpr (NB == 2) {
          const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rsW, voffW, wo + s * ldwB, 0));
          b_[s] = f32x4{v[0], v[1], 0.f, 0.f};
        } else {
          const float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsW, voffW, wo + s * ldwB, 0));
          b_[s] = f32x4{v, 0.f, 0.f, 0.f};
        }
      }
    } else {
      const int ch = min(4 * tt + wv, nch - 1);
      a_ = *reinterpret_cast<const f32x4*>(Ap + 8 * ch);
      const float* wp = Wp + (int64_t)(8 * ch) * ldw;
#pragma unroll
      for (int s = 0; s < 4; ++s) b_[s] = *reinterpret_cast<const f32x4*>(wp + s * ldw);
    }
  };
  // the operand shift (K^-1 (x - mu0)) only in the segments that have one: the whole contraction loop exists twice behind a
  // wave-uniform branch (as a select inside one loop the compiler kept the four subtractions AND added four selects per chunk)
  auto contract = [&](auto sh_tag, auto nb_tag, const f32x4& a_, const f32x4 (&b_)[4]) {
    constexpr bool SH = decltype(sh_tag)::value;
    constexpr int NB = decltype(nb_tag)::value;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float as = SH ? a_[s] - shift : a_[s];
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(as, b_[s][j], acc[j], 0, 0, 0);
    }
  };
  auto run = [&](auto sh_tag, auto nb_tag) {
#pragma unroll
    for (int u = 0; u < P - 1; ++u) issue(nb_tag, av[u], bv[u], u);
    // P chunks in flight; the body is UNCONDITIONAL (a branch around an issue makes the compiler's wait-count merge
    // pessimistic: it put s_waitcnt vmcnt(0) at the loop head, draining the younger chunks on every trip).  The last
    // T mod P chunks are already in flight when the loop ends.
    int tt = 0;
    for (; tt + P <= T; tt += P) {
#pragma unroll
      for (int u = 0; u < P; ++u) {
        issue(nb_tag, av[(u + P - 1) % P], bv[(u + P - 1) % P], tt + u + P - 1);
        __builtin_amdgcn_sched_barrier(0);     // the machine scheduler otherwise sinks these loads to just before their use
        contract(sh_tag, nb_tag, av[u], bv[u]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int u = 0; u < P - 1; ++u)
      if (tt + u < T) contract(sh_tag, nb_tag, av[u], bv[u]);       // wave-uniform
  };
  auto run_nb = [&](auto sh_tag) {
    // (the narrow forms only where they occur: the last column tile of a segment)
    if (nb == 4) run(sh_tag, std::integral_constant<int, 4>{});
    else if (nb == 3) run(sh_tag, std::integral_constant<int, 3>{});
    else if (nb == 2) run(sh_tag, std::integral_constant<int, 2>{});
    else run(sh_tag, std::integral_constant<int, 1>{});
  };
  if (!CMCD_WIDE_BUF || __builtin_amdgcn_readfirstlane(shift != 0.f ? 1 : 0)) run_nb(std::true_type{});
  else run_nb(std::false_type{});

  // ---- the consumer's own operands (outputs of EARLIER launches), all four elements of this thread requested at once, in
  // flight across the cross-wave sum and its barrier (requested BEFORE the contraction they cost ~50 registers through the
  // whole loop for no measurable gain: the consumer is VALU-bound): element q = (row (q 256 + tid) / 32, columns
  // 4 ((q 256 + tid) % 32) ..+3)
  WideStepRegs sr[4];
  f32x4 eb[4], eu[4];      // ACT: bias and residual input
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = q * 256 + threadIdx.x, row = e >> 5, c4 = e & 31;
    const int m = rt * kWRows + row, col = n0 + 4 * c4;
    const bool live = m < a.M;
    eb[q] = eu[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    sr[q].in = false;
    if (epi == WEPI_ACT1 || epi == WEPI_ACT2) {
      if (live && col < a.IN) {                                  // IN % 4 == 0 (host check): four columns in or out together
        eb[q] = *reinterpret_cast<const f32x4*>(a.bias + col);
        if (epi == WEPI_ACT2) eu[q] = *reinterpret_cast<const f32x4*>(a.u_prev + (int64_t)m * a.ldu + col);
        else if (col < a.D) eu[q] = *reinterpret_cast<const f32x4*>(a.x + (int64_t)m * a.ldx + col);   // u = [x; emb_i]   nn.py:68-69
        else eu[q] = f32x4{a.emb[col - a.D], a.emb[col - a.D + 1], a.emb[col - a.D + 2], a.emb[col - a.D + 3]};
      }
    } else if (epi == WEPI_STEP) {
      wide_step_load_state<false, true>(a, m, col, live, sr[q]);
    } else if (epi == WEPI_STEP_NONET) {
      wide_step_load_state<true, false>(a, m, col, live, sr[q]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- sum of the four waves' partial tiles, fixed order.  D layout of the MFMA: column = lane & 31 (= c, i.e. tile
  // column 4 c + block), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  if (nb == 4) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      f32x4 v;
      v[0] = acc[0][i]; v[1] = acc[1][i]; v[2] = acc[2][i]; v[3] = acc[3][i];
      *reinterpret_cast<f32x4*>(red + (wv * kWRows + row) * kRedLd + 4 * c) = v;
    }
  } else {                                                       // tile column nb c + block (columns past 32 nb: never consumed)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
      float* rp = red + (wv * kWRows + row) * kRedLd + nb * c;
      rp[0] = acc[0][i];
      if (nb > 1) rp[1] = acc[1][i];
      if (nb > 2) rp[2] = acc[2][i];
    }
  }
  if (epi == WEPI_STEP || epi == WEPI_STEP_NONET) {   // the per-column vectors and the chain keys: in flight across the barrier
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = q * 256 + threadIdx.x, row = e >> 5, c4 = e & 31;
      wide_step_load_cols(a, rt * kWRows + row, n0 + 4 * c4, sr[q]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = q * 256 + threadIdx.x, row = e >> 5, c4 = e & 31;
    f32x4 v = *reinterpret_cast<const f32x4*>(red + row * kRedLd + 4 * c4);
#pragma unroll
    for (int w = 1; w < 4; ++w) v += *reinterpret_cast<const f32x4*>(red + (w * kWRows + row) * kRedLd + 4 * c4);
    const int m = rt * kWRows + row, col = n0 + 4 * c4;
    const bool live = m < a.M;                                   // uniform over the 32 lanes that share the row
    if (epi == WEPI_ACT1 || epi == WEPI_ACT2) {
      if (live && col < a.IN) {
        f32x4 out;
#pragma unroll
        for (int j = 0; j < 4; ++j) out[j] = eu[q][j] + softplus(v[j] + eb[q][j]);        // nn.py:45-50
        *reinterpret_cast<f32x4*>(a.u_out + (int64_t)m * a.ldu + col) = out;
      }
    } else if (epi == WEPI_KR) {
      if (live && col < a.D) *reinterpret_cast<f32x4*>(a.kr_out + (int64_t)m * a.ldx + col) = v;
    } else {
      float bk = 0.f, fk = 0.f, lp = 0.f;
      if (live) {
        const float o[4] = {v[0], v[1], v[2], v[3]};
        if (epi == WEPI_STEP) wide_step4<false, true>(a, m, col, o, sr[q], bk, fk, lp);
        else wide_step4<true, false>(a, m, col, o, sr[q], bk, fk, lp);
      }
      bk = half_sum32(bk); fk = half_sum32(fk); lp = half_sum32(lp);
      if (live && c4 == 0) {
        const WideStep& s = a.st;
        const int64_t sl = (int64_t)ctl * s.Mp + m;
        if (s.i > 0) s.wslot[sl] += bk - s.fkslot[sl];
        if (s.i < s.K) s.fkslot[sl] = fk;
        else s.lpslot[sl] = lp;
      }
    }
  }
}

// dst[Kp][Np] = zero-padded copy of src[K][N] (row stride lds); pair_h > 0: columns in the PAIRED order of wide_nat_col (H = pair_h)
__global__ void lgcp_wide_pack_kernel(const float* __restrict__ src, int K, int N, int lds, float* __restrict__ dst, int Kp,
                                      int Np, int pair_h) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (n >= Np) return;
  const int nn = pair_h ? wide_nat_col(n, pair_h) : n;
  dst[(int64_t)k * Np + n] = (k < K && nn < N) ? src[(int64_t)k * lds + nn] : 0.f;
}

struct WideVecArgs {
  const float* params;
  const float* tc;
  float* b2; float* b3; float* mean; float* sd; float* counts;
  cmcd_layout lay;
  int D, IN, NpD, NpIN, has_net;
};
// the per-column vectors of the state update, in the column order of its launch: paired with a network (launch C), natural
// without (MCD_ULA: the K^-1 product is its own launch's operand)
__global__ void lgcp_wide_vec_kernel(WideVecArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n < a.NpIN) a.b2[n] = (a.has_net && n < a.IN) ? a.params[a.lay.g_b2 + n] : 0.f;
  if (n < a.NpD) {
    const int e = a.has_net ? wide_nat_col(n, a.D / 2) : n;
    const bool in = e < a.D;
    a.b3[n] = (a.has_net && in) ? a.params[a.lay.g_b3 + e] : 0.f;
    a.mean[n] = in ? a.params[a.lay.vd_mean + e] : 0.f;
    a.sd[n] = in ? expf(-2.0f * a.params[a.lay.vd_logdiag + e]) : 1.f;     // 1 / std^2: what the state update multiplies by
    a.counts[n] = in ? a.tc[(int64_t)a.D * a.D + e] : 0.f;
  }
}

// z0 = mean + std * normal(A, (D,)); w = -log q(z0); gen = second(split(first(split(B)))); then the whole key chain
// (G_i, H_i) = split(gen_i), gen_{i+1} = second(split(H_i))      mcdboundingmachine.py:151-162, mcd_cais.py:66-67,87,94
struct WideInitArgs {
  const int32_t* seeds;
  const float* params;
  float* x;           // [Mp][ldx]
  float* w0;          // [Mp]
  uint32_t* gktab;    // [K][n][2]
  cmcd_layout lay;
  int64_t n;
  int D, ldx, K;
};

__global__ __launch_bounds__(256) void lgcp_wide_init_kernel(WideInitArgs a) {
  __shared__ float sh[4];
  const int64_t p = blockIdx.x;
  const int D = a.D, H = (D + 1) / 2;
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
        a.x[p * a.ldx + idx[q]] = z;
        const float dz = z - mean;
        acc += -(dz * dz) / (2.0f * sd * sd) - logf(sd) - kHalfLog2Pi;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    a.w0[p] = -(sh[0] + sh[1] + sh[2] + sh[3]);
    uint32_t c0 = 0, c2 = 2, c1 = 1, c3 = 3;
    threefry2x32(b0, b1, c0, c2);
    threefry2x32(b0, b1, c1, c3);       // C = (c0, c1)
    uint32_t g0 = 0, g2 = 2, g1 = 1, g3 = 3;
    threefry2x32(c0, c1, g0, g2);
    threefry2x32(c0, c1, g1, g3);       // gen = second(split(C)) = (g2, g3)
    uint32_t k0 = g2, k1 = g3;
    for (int i = 0; i < a.K; ++i) {
      uint32_t G0 = 0, h0 = 2, G1 = 1, h1 = 3;
      threefry2x32(k0, k1, G0, h0);
      threefry2x32(k0, k1, G1, h1);     // G = (out0, out1), H = (out2, out3)
      uint32_t n0 = 0, n2 = 2, n1 = 1, n3 = 3;
      threefry2x32(h0, h1, n0, n2);
      threefry2x32(h0, h1, n1, n3);     // gen' = second(split(H))
      uint32_t* gt = a.gktab + ((int64_t)i * a.n + p) * 2;
      gt[0] = G0; gt[1] = G1;
      k0 = n2; k1 = n3;
    }
  }
}

// loss = -(w_0 + sum over column tiles of the closed steps' (bk - fk) + log p(z_K)); fixed summation order
struct WideFinalArgs {
  const float* w0; const float* wslot; const float* lpslot; const float* tc;
  float* out_loss; double* partials;
  int64_t n;
  int Mp, D, CT;
};
__global__ void lgcp_wide_final_kernel(WideFinalArgs a) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= a.n) return;
  float w = a.w0[p], lp = 0.f;
  for (int ct = 0; ct < a.CT; ++ct) {
    w += a.wslot[(int64_t)ct * a.Mp + p];
    lp += a.lpslot[(int64_t)ct * a.Mp + p];
  }
  w += lp + a.tc[(int64_t)a.D * a.D + a.D + 2];     // + log p(z_K)   mcdboundingmachine.py:178
  const float loss = -w;
  a.out_loss[p] = loss;
  double* o = a.partials + p * CMCD_NSTATS;
  o[0] = isfinite(loss) ? 1.0 : 0.0;
  o[1] = loss;
  o[2] = (double)loss * (double)loss;
  o[3] = -(double)loss;
  o[4] = isfinite(loss) ? 1.0 : 0.0;
}

// ------------------------------------------------------------------------------------------ host side
struct WideWs {
  int64_t bias1, w1p, w2p, w3p, kip, b2, b3, mean, sd, counts, x, xp, xn, kr, u1, u2, w0, gktab, slots, partials, total;
  int Mp, ldx, ldu, NpD, NpIN, KpD, KpIN, ctD, ctIN;
};

static WideWs wide_ws(const cmcd_desc& d, int64_t n, int64_t base) {
  const int64_t D = d.dim, IN = D + d.emb_dim, K = d.nbridges;
  WideWs w;
  w.Mp = (int)rup(n, kWRows);
  w.ldx = (int)rup(D, 32); w.ldu = (int)rup(IN, 32);
  w.KpD = w.ldx; w.KpIN = w.ldu;
  w.NpD = (int)rup(D, kWCols); w.NpIN = (int)rup(IN, kWCols);
  w.ctD = w.NpD / kWCols; w.ctIN = w.NpIN / kWCols;
  int64_t o = base;
  auto take = [&](int64_t cnt) { int64_t r = o; o += (cnt + 3) & ~int64_t(3); return r; };
  w.bias1 = take((K + 1) * IN);
  w.w1p = take((int64_t)w.KpD * w.NpIN);
  w.w2p = take((int64_t)w.KpIN * w.NpIN);
  w.w3p = take((int64_t)w.KpIN * w.NpD);
  w.kip = take((int64_t)w.KpD * w.NpD);
  w.b2 = take(w.NpIN); w.b3 = take(w.NpD); w.mean = take(w.NpD); w.sd = take(w.NpD); w.counts = take(w.NpD);
  w.x = take((int64_t)w.Mp * w.ldx); w.xp = take((int64_t)w.Mp * w.ldx); w.xn = take((int64_t)w.Mp * w.ldx);
  w.kr = take((int64_t)w.Mp * w.ldx);
  w.u1 = take((int64_t)w.Mp * w.ldu); w.u2 = take((int64_t)w.Mp * w.ldu);
  w.w0 = take(w.Mp);
  w.gktab = take(2 * K * n);
  w.slots = take((int64_t)3 * w.ctD * w.Mp);
  o = (o + 1) & ~int64_t(1);
  w.partials = take(n * CMCD_NSTATS * 2);
  w.total = o;
  return w;
}

// XCD x runs the contiguous tile range [xbeg[x], xbeg[x + 1]) (tiles in the order: row tiles of column tile 0, of column tile
// 1, ...).  Few-round grids (every CU gets about one tile): equal counts, as before.  Many-round grids: equal COST — a tile's
// cost is its number of column blocks (wide_tile_blocks), so the XCD that holds the cheaper last column tiles takes more of them.
// Returns the longest range (the grid is 8 x that).
static int wide_xcd_ranges(WideArgs& g) {
  const int nt = g.CT * g.RT;
  auto blocks = [&](int ct) {
    const int sI = ct >= g.nct0 ? 1 : 0, ctl = ct - (sI ? g.nct0 : 0);
    return wide_tile_blocks(g.seg[sI].ncols - ctl * kWCols);
  };
  int64_t total = 0;
  for (int ct = 0; ct < g.CT; ++ct) total += (int64_t)blocks(ct) * g.RT;
  const bool by_cost = nt >= 8 * 64 && total != (int64_t)4 * nt;
  g.narrow = by_cost ? 1 : 0;   // one-round grids wait for their slowest (full) tile anyway: plain form there
  g.xbeg[0] = 0;
  if (!by_cost) {
    const int per = (nt + 7) / 8;
    for (int x = 1; x <= 8; ++x) g.xbeg[x] = std::min(nt, x * per);
  } else {
    int ct = 0, rt = 0;
    int64_t acc = 0;
    for (int x = 1; x < 8; ++x) {
      const int64_t want = (total * x + 7) / 8;
      while (ct < g.CT && acc < want) {     // whole column tiles first, then rows of the one the cut falls into
        const int b = blocks(ct);
        const int64_t left = (int64_t)(g.RT - rt) * b;
        if (acc + left <= want) { acc += left; ++ct; rt = 0; continue; }
        const int rows = (int)((want - acc + b - 1) / b);
        rt += rows; acc += (int64_t)rows * b;
      }
      g.xbeg[x] = std::min(nt, ct * g.RT + rt);
    }
    g.xbeg[8] = nt;
  }
  int longest = 0;
  for (int x = 0; x < 8; ++x) longest = std::max(longest, g.xbeg[x + 1] - g.xbeg[x]);
  return longest;
}

bool lgcp_wide_supported(const cmcd_desc& d) {
  const int D = d.dim, IN = D + d.emb_dim;
  return d.mode != CMCD_MODE_CAIS_UHA_SN && D % 4 == 0 && IN % 4 == 0 && D >= 32;
}

int64_t lgcp_wide_workspace_floats(const cmcd_desc& d, int64_t n, int64_t base) { return wide_ws(d, n, base).total; }

int lgcp_wide_forward(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const int32_t* seeds, int64_t n,
                      const float* params, const float* tc, float* ws, float* out_loss, float* out_z,
                      double** partials_out, void* stream_, bool tables_ready) {
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  const int D = d.dim, E = d.emb_dim, IN = D + E, K = d.nbridges;
  const WideWs w = wide_ws(d, n, sw.total_floats);
  const int ula = d.mode == CMCD_MODE_ULA ? 1 : (d.mode == CMCD_MODE_ULA_SN ? 2 : 0);
  const bool has_net = ula != 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(lgcp_wide_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          kWideLds) != hipSuccess)
    return CMCD_ERR_HIP;
  double* partials = reinterpret_cast<double*>(ws + w.partials);
  *partials_out = partials;

  // once per call: first-layer bias table, zero-padded weight copies, padded vectors, zeroed activations / slots
  if (has_net && !tables_ready) {
    int rc = lgcp_launch_prep(d, lay, params, ws + w.bias1, stream);
    if (rc != CMCD_OK) return rc;
    auto pack = [&](const float* src, int Kr, int Nr, int64_t dst, int Kp, int Np, int pair_h) {
      hipLaunchKernelGGL(lgcp_wide_pack_kernel, dim3((Np + 255) / 256, Kp), dim3(256), 0, stream, src, Kr, Nr, Nr, ws + dst, Kp, Np,
                         pair_h);
    };
    pack(params + lay.g_w1, D, IN, w.w1p, w.KpD, w.NpIN, 0);     // only the state rows W1[:d]: the embedding rows are in bias1
    pack(params + lay.g_w2, IN, IN, w.w2p, w.KpIN, w.NpIN, 0);
    pack(params + lay.g_w3, IN, D, w.w3p, w.KpIN, w.NpD, D / 2);  // launch C's columns in paired order (wide_step4<.., PAIR>)
  }
  if (!tables_ready) {
    hipLaunchKernelGGL(lgcp_wide_pack_kernel, dim3((w.NpD + 255) / 256, w.KpD), dim3(256), 0, stream, tc, D, D, D, ws + w.kip,
                       w.KpD, w.NpD, 0);
    WideVecArgs va{params, tc, ws + w.b2, ws + w.b3, ws + w.mean, ws + w.sd, ws + w.counts, lay, D, IN, w.NpD, w.NpIN, has_net ? 1 : 0};
    hipLaunchKernelGGL(lgcp_wide_vec_kernel, dim3((w.NpIN + 255) / 256), dim3(256), 0, stream, va);
  }
  // x | xp | xn | kr | u1 | u2 | w0 are contiguous up to alignment: padding rows and columns must read as zeros
  if (hipMemsetAsync(ws + w.x, 0, sizeof(float) * (size_t)(w.w0 + w.Mp - w.x), stream) != hipSuccess) return CMCD_ERR_HIP;
  if (hipMemsetAsync(ws + w.slots, 0, sizeof(float) * (size_t)3 * w.ctD * w.Mp, stream) != hipSuccess) return CMCD_ERR_HIP;
  WideInitArgs ia{seeds, params, ws + w.x, ws + w.w0, reinterpret_cast<uint32_t*>(ws + w.gktab), lay, n, D, w.ldx, K};
  hipLaunchKernelGGL(lgcp_wide_init_kernel, dim3((unsigned)n), dim3(256), 0, stream, ia);

  const float mu0 = 3.8812819069514780f;      // log(126) - 0.5 * 1.91 (model_handler.py:346); the state update reads tc's copy
  WideArgs g{};
  g.M = (int)n; g.RT = w.Mp / kWRows;
  g.D = D; g.IN = IN; g.ldx = w.ldx; g.ldu = w.ldu;
  WideStep& st = g.st;
  st.sched = ws + sw.sched; st.tc = tc; st.kr = ws + w.kr; st.b3 = ws + w.b3; st.mean = ws + w.mean;
  st.sd = ws + w.sd; st.counts = ws + w.counts; st.factor = has_net ? params + lay.g_factor : nullptr;
  st.gktab = reinterpret_cast<const uint32_t*>(ws + w.gktab);
  st.wslot = ws + w.slots; st.fkslot = st.wslot + (int64_t)w.ctD * w.Mp; st.lpslot = st.fkslot + (int64_t)w.ctD * w.Mp;
  st.out_z = out_z; st.n = n; st.Mp = w.Mp; st.K = K;
  st.var_mode = d.mode == CMCD_MODE_CAIS_VAR_SN ? 1 : 0; st.grad_clipping = d.grad_clipping; st.ula = ula;
  // the XCD ranges depend on the launch type only (segments' widths, tile counts): cut once per type, not per evaluation
  struct Cut { int xbeg[9]; int longest = -1; int narrow = 0; };
  Cut cuts[4];
  auto launch = [&](int ct_total, int type) {
    g.CT = ct_total;
    Cut& cu = cuts[type];
    if (cu.longest < 0) {
      cu.longest = wide_xcd_ranges(g);
      cu.narrow = g.narrow;
      for (int x = 0; x < 9; ++x) cu.xbeg[x] = g.xbeg[x];
    }
    g.narrow = cu.narrow;
    for (int x = 0; x < 9; ++x) g.xbeg[x] = cu.xbeg[x];
    hipLaunchKernelGGL(lgcp_wide_gemm_kernel, dim3(8 * cu.longest), dim3(256), kWideLds, stream, g);
  };
  float* xbuf[3] = {ws + w.x, ws + w.xp, ws + w.xn};      // cur, prev, next
  for (int i = 0; i <= K; ++i) {
    st.i = i;
    float* const xc = xbuf[0];
    g.x = xc; st.xp = xbuf[1]; st.xn = xbuf[2];
    // after this evaluation: prev <- cur, cur <- next (the old prev is the buffer the next evaluation writes)
    xbuf[0] = st.xn; xbuf[2] = xbuf[1]; xbuf[1] = xc;
    if (ula == 1) {    // MCD_ULA: one launch per evaluation, (x - mu0) K^-1 with the state update as its consumer
      g.seg[0] = WideSeg{xc, ws + w.kip, w.ldx, w.NpD, w.KpD, WEPI_STEP_NONET, mu0, D};
      g.nct0 = w.ctD;
      launch(w.ctD, 3);
      continue;
    }
    // CAIS: s(z_i, i) serves both kernels; MCD_ULA_sn: s(z_i, i - 1) serves the backward kernel only
    const int it = ula == 2 ? (i > 0 ? i - 1 : 0) : i;
    const int ie = it < K ? it : K - 1;
    // A
    g.seg[0] = WideSeg{xc, ws + w.w1p, w.ldx, w.NpIN, w.KpD, WEPI_ACT1, 0.f, IN};
    g.nct0 = w.ctIN;
    g.bias = ws + w.bias1 + (int64_t)it * IN; g.emb = params + lay.g_emb + (int64_t)ie * E; g.u_out = ws + w.u1;
    launch(w.ctIN, 0);
    // B: the second layer and, beside it, the K^-1 product (needs only the state; consumed by C)
    g.seg[0] = WideSeg{ws + w.u1, ws + w.w2p, w.ldu, w.NpIN, w.KpIN, WEPI_ACT2, 0.f, IN};
    g.seg[1] = WideSeg{xc, ws + w.kip, w.ldx, w.NpD, w.KpD, WEPI_KR, mu0, D};
    g.nct0 = w.ctIN;
    g.bias = ws + w.b2; g.u_prev = ws + w.u1; g.u_out = ws + w.u2; g.kr_out = ws + w.kr;
    launch(w.ctIN + w.ctD, 1);
    // C
    // (paired column order: tile ct holds elements [64 ct, 64 ct + 64) and their partners — 2 H = D packed columns in all)
    g.seg[0] = WideSeg{ws + w.u2, ws + w.w3p, w.ldu, w.NpD, w.KpIN, WEPI_STEP, 0.f, D};
    g.nct0 = w.ctD;
    launch(w.ctD, 2);
  }
  WideFinalArgs fa{ws + w.w0, st.wslot, st.lpslot, tc, out_loss, partials, n, w.Mp, D, w.ctD};
  hipLaunchKernelGGL(lgcp_wide_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fa);
  return hipGetLastError() == hipSuccess ? CMCD_OK : CMCD_ERR_HIP;
}

}  // namespace cmcd
