// libcmcd_hip.so — hand-written gfx950 (CDNA4) kernels for CMCD's annealed-Langevin bound
// (`MCD_CAIS_sn` / `MCD_CAIS_var_sn`) and the C ABI of include/cmcd_hip.h.
//
// Launch sequence of one cmcd_bound_forward (all on the caller's stream, no host sync):
//   1. prep_sched_kernel    betas / eps / sigma tables             (mcdboundingmachine.py:146-149,
//                                                                   mcd_cais.py:34-44,54-63)
//   2. prep_{dds,geffner}_kernel  per-bridge first-layer bias table: the particle-independent
//                           time path (nn_dds.py:155-158 / nn.py:68) folded through W1[d:,:]
//   3. pack_weights_kernel  W2 into MFMA A-fragment order, W1[:d], W3^T, biases, zero padded
//   4. traj_kernel          the whole K-step trajectory of 16 particles per wave
//   5. finalize_kernel      fixed-order merge of per-wave statistics -> out_stats[5]
//
// Data layout of the trajectory kernel (one wave = 16 particles, all K steps):
//   lane l = (g = l >> 4, c = l & 15): particle c of the tile; the four lanes g = 0..3 of a particle
//   each own neurons {16 t + 4 g + r} (t < T tiles, r < 4) of every hidden vector.  That is exactly
//   the C/D register layout of v_mfma_f32_16x16x4_f32 with rows = output neurons and
//   columns = particles, AND (with W2 pre-permuted into A fragments) the B-operand layout of the
//   next MFMA, so activations never leave registers between layers.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"
#include "cmcd_hip_diag.h"

namespace cmcd {

static thread_local char g_err[512] = "";
static thread_local char g_kernel_name[96] = "";   // cmcd_last_kernel_name

// Optional in-library timing of the trajectory kernel: when enabled, every cmcd_bound_forward
// brackets its traj_kernel launch with a hipEvent pair on the caller's stream (bench.py reads
// the average kernel duration from them for the roofline figure).
struct ProfileState {
  static constexpr int kMax = 4096;
  bool on = false;
  int used = 0;
  hipEvent_t ev[kMax][2];
  int created = 0;
};
static thread_local ProfileState g_prof;
// cmcd_debug_capture_noise: armed per host thread, consumed (and cleared) by the next forward call on that thread
struct NoiseCapture { uint32_t* bits = nullptr; uint32_t* keys = nullptr; float* noise = nullptr; };
static thread_local NoiseCapture g_capture;

// Tiles (16 particles each) up to which the CU-cooperative kernel is preferred; measured crossovers on
// MI355X (tools/probes/variant_sweep.py, t9_variants.py): dds/geffner T<=4 between 512 and 1024 tiles; the 132-wide
// net at ~600 (500 tiles: cooperative 1.85 ms against 2.27 ms one wave per tile; 1000 tiles: 3.69 against 2.28).
// r02, after the cooperative kernel's per-bridge time dropped by a sixth (tools/probes/variant_crossover.py,
// profiles/r02_s2e_variant_crossover.txt; the cooperative time is ceil(tiles / 256 CUs) rounds of one workgroup per CU):
//   dds net, 40-mode mixture:     cooperative wins through 6 rounds (1536 tiles: 1.13 against 1.34 ms; 2048: 1.42 / 1.34)
//   132-wide net, 40-mode mixture: through 3 rounds (768 tiles: 1.91 against 2.31 ms; 813: 2.53 / 2.29)
//   funnel / gmm on the narrow geffner nets: 512 tiles still (768: 0.373 / 0.314 ms and 0.0257 / 0.0246 ms)
static int coop_max_tiles(const cmcd_desc& d, int T) {
  if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) {
    if (d.arch == CMCD_ARCH_DDS) return 1536;
    if (T == 9) return 768;
  }
  return 512;
}
static int fail(int code, const char* fmt, const char* a = "", long long b = 0) {
  snprintf(g_err, sizeof(g_err), fmt, a, b);
  return code;
}

static inline int64_t align4(int64_t x) { return (x + 3) & ~int64_t(3); }

int net_in_dim(const cmcd_desc& d);
static bool hidden_width(const cmcd_desc& d, int& HP) {
  if (d.arch == CMCD_ARCH_DDS) {
    HP = 64;
    return true;
  }
  if (d.arch == CMCD_ARCH_GEFFNER) {
    if (d.emb_dim < 1) return false;
    HP = ((net_in_dim(d) + d.emb_dim + 15) / 16) * 16;
    if (d.mode == CMCD_MODE_CAIS_UHA_SN && d.target != CMCD_TARGET_LGCP) {
      // 2nd-order CMCD has its own kernels (cmcd_uha.hip): instances of 2, 4, 5 and 9 neuron tiles (gmm 2*2+20 = 24,
      // funnel 2*10+48 = 68, the 40-mode mixture 2*2+130 = 134), other widths zero-padded to the next one
      const int T = HP / 16;
      HP = 16 * (T <= 2 ? 2 : (T <= 4 ? 4 : (T <= 5 ? 5 : (T <= 9 ? 9 : T))));
      return true;
    }
    // Kernel instances exist for 2, 4 and 9 neuron tiles (the BASELINE widths 22 / 58 / 132); any other width runs
    // on the next larger instance with zero-padded weights: a padded unit has no outgoing weight, so it cannot
    // reach the output, and its gradient entries are never copied out.  (lgcp has its own path: any width.)
    if (d.target != CMCD_TARGET_LGCP) {
      const int T = HP / 16;
      // funnel (d = 10): its gradient kernels start at 4 tiles, and forward / gradient share one workspace layout
      const int tmin = d.target == CMCD_TARGET_FUNNEL ? 4 : 2;
      HP = 16 * (T <= tmin ? tmin : (T <= 4 ? 4 : (T <= 9 ? 9 : T)));
    }
    return true;
  }
  return false;
}

// width of the state part of the network input: z, or concat(z, rho) for the momentum mode (rho_dim = dim,
// /root/reference/src/mcdboundingmachine.py:82-98)
int net_in_dim(const cmcd_desc& d) { return d.mode == CMCD_MODE_CAIS_UHA_SN ? 2 * d.dim : d.dim; }

// floats of the trajectory a gradient call keeps: z_0..z_K, plus rho_0..rho_K and rho'_0..rho'_{K-1} for the momentum mode
static int64_t kept_traj_floats(const cmcd_desc& d, int64_t n) {
  return (int64_t)(d.mode == CMCD_MODE_CAIS_UHA_SN ? 3 * d.nbridges + 2 : d.nbridges + 1) * n * d.dim;
}

static int64_t target_lds_floats(const cmcd_desc& d, int64_t n_target) {
  if (d.target == CMCD_TARGET_MANY_GMM) return 4 + (n_target - 1);  // header + means
  return 0;
}

// lgcp: only the schedule tables live in the common layout; the rest is carved by cmcd_lgcp.hip
static bool make_ws_lgcp(const cmcd_desc& d, int64_t n, WsLayout& w) {
  const int64_t K = d.nbridges;
  memset(&w, 0, sizeof(w));
  int64_t o = 0;
  w.beta = o; o += align4(K);
  w.eps = o; o += align4(K);
  w.sig = o; o += align4(K);
  w.logsig = o; o += align4(K);
  w.sched = o; o += 8 * K;
  w.n_waves = (int32_t)n;  // one statistics record per particle
  w.total_floats = o;
  return true;
}

static bool make_ws(const cmcd_desc& d, int64_t n, int64_t n_target, WsLayout& w) {
  int HP;
  if (!hidden_width(d, HP)) return false;
  const int64_t K = d.nbridges, D = d.dim;
  w.HP = HP;
  w.T = HP / 16;
  int64_t o = 0;
  w.beta = o; o += align4(K);
  w.eps = o; o += align4(K);
  w.sig = o; o += align4(K);
  w.logsig = o; o += align4(K);
  w.sched = o; o += 8 * K;
  w.bias1 = o; o += (K + 1) * HP;
  if (d.arch == CMCD_ARCH_GEFFNER) { w.utab = o; o += (K + 1) * HP; } else { w.utab = w.bias1; }
  w.w1z = o; o += int64_t(net_in_dim(d)) * HP;
  w.w2 = o; o += int64_t(HP) * HP;
  w.w2t = o; o += int64_t(HP) * HP;
  w.w2q = o; o += 2 * int64_t(HP) * HP;
  w.b2 = o; o += HP;
  w.w3t = o; o += D * HP;
  w.b3 = o; o += 16;
  w.tgt_floats = align4(target_lds_floats(d, n_target));
  w.tgt = o; o += w.tgt_floats;
  o = (o + 1) & ~int64_t(1);
  w.n_waves = int32_t((n + 15) / 16);
  // sized for the cooperative kernel's 8-particle tiles (twice the records of the 16-particle tiling)
  w.partials = o; o += int64_t((n + 7) / 8) * CMCD_NSTATS * 2;
  w.total_floats = o;
  return true;
}

// ------------------------------------------------------------------------------------------
// 1. schedules
// ------------------------------------------------------------------------------------------
struct SchedArgs {
  const float* params;
  float* ws;
  cmcd_layout lay;
  WsLayout w;
  int32_t K, ngrid, eps_schedule;
  int64_t gridref_x, target_x;  // offsets in params, or -1: use linspace
  uint32_t stamp = 0;           // tables_stamp() of the call that forms the tables: kept in sched[0][7] (see finalize_kernel)
};

__device__ __forceinline__ void prep_sched_body(const SchedArgs& a) {
  __shared__ float gy[40], gx[40], gm[40];
  const int G = a.ngrid;  // mgridref_y has G+1 entries
  if (threadIdx.x <= G) gm[threadIdx.x] = a.params[a.lay.mgridref_y + threadIdx.x];
  __syncthreads();
  // gridref_y = concat([0], cumsum(m)/sum(m))       mcdboundingmachine.py:147-148.  The running sum stays sequential
  // (same association as a serial cumsum) but runs in registers: all entries are read first, one thread adds them in
  // order, and the G + 1 divisions and the grid abscissae are done by G + 2 threads in parallel (the serial version,
  // ~100 dependent LDS reads and divisions on one thread, was the longest block of the prep launch).
  if (threadIdx.x == 0) {
    float v[33];
#pragma unroll
    for (int i = 0; i < 33; ++i) v[i] = gm[i < G + 1 ? i : G];
    float run = 0.f;
#pragma unroll
    for (int i = 0; i < 33; ++i) {
      if (i <= G) {
        run += v[i];
        gy[i + 1] = run;   // un-normalised
      }
    }
    gy[0] = 0.f;
    gm[39] = run;          // total (ngrid <= 32: slot 39 is free)
  }
  __syncthreads();
  if (threadIdx.x < G + 2) {
    const float tot = gm[39];
    if (threadIdx.x >= 1) gy[threadIdx.x] = gy[threadIdx.x] / tot;
    gx[threadIdx.x] = (float)threadIdx.x / (float)(G + 1);  // linspace(0,1,G+2)  :113
  }
  __syncthreads();
  const float eps0 = a.params[a.lay.eps];
  for (int i = threadIdx.x; i < a.K; i += blockDim.x) {
    // betas = interp(target_x, gridref_x, gridref_y)     :149, target_x = linspace(0,1,K+2)[1:-1]  :114
    const float x = (float)(i + 1) / (float)(a.K + 1);
    // searchsorted(gx, x, side='right') clipped to [1, G+1]: gx is a uniform grid, so the answer is within one of
    // floor(x (G+1)) + 1 — settle it with the exact comparisons instead of a linear scan through LDS
    int j = (int)(x * (float)(G + 1)) + 1;
    j = j < 1 ? 1 : (j > G + 1 ? G + 1 : j);
    while (j > 1 && gx[j - 1] > x) --j;
    while (j < G + 1 && gx[j] <= x) ++j;
    const float dx = gx[j] - gx[j - 1], df = gy[j] - gy[j - 1];
    a.ws[a.w.beta + i] = gy[j - 1] + ((x - gx[j - 1]) / dx) * df;
    // eps_i                                              mcd_cais.py:34-44,54-59
    float e;
    if (a.eps_schedule == CMCD_EPS_COS_SQ) {
      const float phase = (float)i / (float)a.K;
      const float cs = cosf((phase + 0.008f) / 1.008f * 0.5f * 3.14159265358979323846f);
      e = eps0 * (cs * cs);
    } else if (a.eps_schedule == CMCD_EPS_LINEAR) {
      e = (0.0001f - eps0) / (float)(a.K - 1) * (float)i + eps0;
    } else {
      e = eps0;
    }
    const float s = sqrtf(2.0f * e);  // scale = sqrt(2 eps)   mcd_cais.py:63
    a.ws[a.w.eps + i] = e;
    a.ws[a.w.sig + i] = s;
    a.ws[a.w.logsig + i] = logf(s);
    float* sc = a.ws + a.w.sched + 8 * (int64_t)i;
    sc[0] = a.ws[a.w.beta + i];
    sc[1] = e;
    sc[2] = s;
    sc[3] = logf(s) + kHalfLog2Pi;
    sc[4] = 1.0f / (2.0f * s * s);
    sc[5] = e * sc[0];            // eps * beta
    sc[6] = e * (1.0f - sc[0]);   // eps * (1 - beta)
    sc[7] = i == 0 ? __uint_as_float(a.stamp) : 0.f;   // never read as a number
  }
}

__global__ void prep_sched_kernel(SchedArgs a) { prep_sched_body(a); }

// ------------------------------------------------------------------------------------------
// 2a. dds: time path -> per-bridge first-layer bias.  One block (256 threads) per bridge index t.
//     tau(t) = W_b gelu(W_a [sin(c t + phi), cos(c t + phi)] + b_a) + b_b   nn_dds.py:131-143,155-158
//     bias1[t][n] = sb1[n] + sum_j tau_j * sw1[d + j][n]                    (concat at :159)
// ------------------------------------------------------------------------------------------
struct DdsPrepArgs {
  const float* params;
  float* ws;
  cmcd_layout lay;
  WsLayout w;
  int32_t D;
};

// `t` = bridge index.  256 threads: thread (q = tid >> 6, j = tid & 63) owns quarter q of each contraction for output j.
// All weights a thread will need (32 + 16 + 16 values, independent of the data) are requested before the first
// sin / cos, so the three dependent matrix-vector products pay ONE L2 round trip instead of three, and each runs
// 32 / 16 / 16 dependent FMAs instead of 128 / 64 / 64; quarters are summed in fixed order (q = 0..3) through LDS.
__device__ __forceinline__ void prep_dds_body(const DdsPrepArgs& a, int t) {
  __shared__ float e[128], h[64], tau[64], red[4][64];
  const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
  const float* P = a.params;
  float wa[32], wb[16], wc[16];
#pragma unroll
  for (int k = 0; k < 32; ++k) wa[k] = P[a.lay.d_tw1 + (32 * q + k) * 64 + j];
#pragma unroll
  for (int k = 0; k < 16; ++k) wb[k] = P[a.lay.d_tw2 + (16 * q + k) * 64 + j];
#pragma unroll
  for (int k = 0; k < 16; ++k) wc[k] = P[a.lay.d_sw1 + (a.D + 16 * q + k) * 64 + j];
  const float ba = P[a.lay.d_tb1 + j], bb = P[a.lay.d_tb2 + j], bc = P[a.lay.d_sb1 + j];
  if (q == 0) {
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
    e[j] = sinf(arg);
    e[64 + j] = cosf(arg);
  }
  __syncthreads();
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) acc = fmaf(e[32 * q + k], wa[k], acc);
  red[q][j] = acc;
  __syncthreads();
  if (q == 0) h[j] = gelu_exact(ba + ((red[0][j] + red[1][j]) + (red[2][j] + red[3][j])));
  __syncthreads();
  acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc = fmaf(h[16 * q + k], wb[k], acc);
  __syncthreads();   // red is reused
  red[q][j] = acc;
  __syncthreads();
  if (q == 0) tau[j] = bb + ((red[0][j] + red[1][j]) + (red[2][j] + red[3][j]));
  __syncthreads();
  acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc = fmaf(tau[16 * q + k], wc[k], acc);
  __syncthreads();
  red[q][j] = acc;
  __syncthreads();
  if (q == 0) a.ws[a.w.bias1 + (int64_t)t * 64 + j] = bc + ((red[0][j] + red[1][j]) + (red[2][j] + red[3][j]));
}

// ------------------------------------------------------------------------------------------
// 2b. geffner: emb[i] folded through W1[d:, :]; row K re-uses emb[K-1] (JAX clamps the
//     out-of-range gather of nn.py:68 reached from mcd_cais.py:78 at the last bridge).
// ------------------------------------------------------------------------------------------
struct GefPrepArgs {
  const float* params;
  float* ws;
  cmcd_layout lay;
  WsLayout w;
  int32_t D, E, K;
};

__device__ __forceinline__ void prep_geffner_body(const GefPrepArgs& a, int row) {
  const int ie = row < a.K ? row : a.K - 1;
  const int in = a.D + a.E;
  const float* P = a.params;
  const float* emb = P + a.lay.g_emb + (int64_t)ie * a.E;
  for (int n = threadIdx.x; n < a.w.HP; n += blockDim.x) {
    float b = 0.f, u = 0.f;
    if (n < in) {
      b = P[a.lay.g_b1 + n];
      // same serial sum over the embedding (bit for bit), with eight weight loads in flight: one dependent load per term
      // made this block the longest of the prep launch for the wide embeddings (emb_dim 48: 13.7 us, 130: ~30 us)
      const float* wcol = P + a.lay.g_w1 + (int64_t)a.D * in + n;
      int j = 0;
      for (; j + 8 <= a.E; j += 8) {
        float wv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wv[q] = wcol[(int64_t)(j + q) * in];
#pragma unroll
        for (int q = 0; q < 8; ++q) b = fmaf(emb[j + q], wv[q], b);
      }
      for (; j < a.E; ++j) b = fmaf(emb[j], wcol[(int64_t)j * in], b);
      u = n >= a.D ? emb[n - a.D] : 0.f;
    }
    a.ws[a.w.bias1 + (int64_t)row * a.w.HP + n] = b;
    a.ws[a.w.utab + (int64_t)row * a.w.HP + n] = u;
  }
}

// ------------------------------------------------------------------------------------------
// 3. weight packing.  w2[(t_in*T + t_out)*64 + lane][r] = W2[16 t_in + 4 g + r][16 t_out + i],
//    lane = (g, i): the A operand (rows = output neurons, k = input neurons) of MFMA k-step
//    (t_in, r) for output tile t_out.  Everything beyond the true width is zero.
// ------------------------------------------------------------------------------------------
struct PackArgs {
  const float* params;
  const float* tgt;
  float* ws;
  WsLayout w;
  int64_t o_w1, o_w2, o_b2, o_w3, o_b3, o_factor;  // offsets in params (factor: -1 -> 1.0)
  int32_t D, IN;                                   // state inputs of the net (z, or [z; rho]); true hidden width
  int32_t DO;                                      // outputs of the net (= dim)
  int32_t target, n_mix;
  uint32_t stamp;                                  // tables_stamp() of this call -> b3[13]
};

__device__ __forceinline__ void pack_weights_body(const PackArgs& a, int vblock, int nblocks) {
  const int HP = a.w.HP, T = a.w.T;
  const float* P = a.params;
  const int64_t tid = (int64_t)vblock * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)nblocks * blockDim.x;
  const bool has_net = a.o_w2 >= 0;
  for (int64_t idx = tid; has_net && idx < (int64_t)HP * HP; idx += stride) {
    const int r = idx & 3, lane = (idx >> 2) & 63;
    const int tt = int(idx >> 8), t_out = tt % T, t_in = tt / T;
    const int kin = 16 * t_in + 4 * (lane >> 4) + r, nout = 16 * t_out + (lane & 15);
    a.ws[a.w.w2 + idx] = (kin < a.IN && nout < a.IN) ? P[a.o_w2 + (int64_t)kin * a.IN + nout] : 0.f;
    // transposed product d u1[k] = sum_n W2[k][n] d a2[n]: rows = k (tile t_in here), contraction over n
    const int krow = 16 * t_in + (lane & 15), ncon = 16 * t_out + 4 * (lane >> 4) + r;
    a.ws[a.w.w2t + idx] = (krow < a.IN && ncon < a.IN) ? P[a.o_w2 + (int64_t)krow * a.IN + ncon] : 0.f;
  }
  // w2q[(wv * HP/2 + s) * 64 + lane] = W2[s + (HP/2) kh][16 wv + 4 ng + i], lane = i + 4 pg + 8 kh + 16 ng: the A
  // operand of v_mfma_f32_4x4x1 step s — block (pg, kh, ng) multiplies 4 output neurons by 4 particles for input
  // s of contraction half kh (the particle group pg sees the same weights; ng = the row of the wave, so that the
  // activations can be broadcast from one row to all four by the instruction's B lane-group pattern)
  for (int64_t idx = tid; has_net && idx < 2 * (int64_t)HP * HP; idx += stride) {
    const int lane = idx & 63, rest = int(idx >> 6), sq = rest % (HP / 2), wvq = rest / (HP / 2);
    int kin = sq + (HP / 2) * ((lane >> 3) & 1), nout = 16 * wvq + 4 * (lane >> 4) + (lane & 3);
    if (wvq == 8 && coop_tail4(T, a.IN)) {
      // the ninth wave's 4-neuron form: lane = i + 4 pg + 8 slice, step sq of slice `slice` (cmcd_common.h: coop_tail4)
      const int sl = (lane >> 3) & 7;
      kin = sq < coop_tail_len(sl) ? coop_tail_start(sl) + sq : a.IN;   // a.IN: out of range -> 0
      nout = 128 + (lane & 3);
    }
    a.ws[a.w.w2q + idx] = (kin < a.IN && nout < a.IN) ? P[a.o_w2 + (int64_t)kin * a.IN + nout] : 0.f;
  }
  for (int64_t idx = tid; has_net && idx < (int64_t)a.D * HP; idx += stride) {
    const int j = int(idx / HP), n = int(idx % HP);
    a.ws[a.w.w1z + idx] = n < a.IN ? P[a.o_w1 + (int64_t)j * a.IN + n] : 0.f;
  }
  for (int64_t idx = tid; has_net && idx < (int64_t)a.DO * HP; idx += stride) {
    const int j = int(idx / HP), n = int(idx % HP);
    a.ws[a.w.w3t + idx] = n < a.IN ? P[a.o_w3 + (int64_t)n * a.DO + j] : 0.f;
  }
  for (int64_t idx = tid; has_net && idx < HP; idx += stride) a.ws[a.w.b2 + idx] = idx < a.IN ? P[a.o_b2 + idx] : 0.f;
  for (int64_t idx = tid; idx < 16; idx += stride) {
    float v = 0.f;
    if (has_net && idx < a.DO) v = P[a.o_b3 + idx];
    if (idx == 15) v = a.o_factor >= 0 ? P[a.o_factor] : 1.0f;
    if (idx == 13) v = __uint_as_float(a.stamp);   // DO <= 10: slots 12 - 14 hold no bias (14 = the fused merge's counter)
    a.ws[a.w.b3 + idx] = v;
  }
  if (a.target == CMCD_TARGET_MANY_GMM) {
    // tgt = {scale, means[n_mix][2]} -> {1/scale, c2, n_mix bits, c0, means}
    for (int64_t idx = tid; idx < a.w.tgt_floats; idx += stride) {
      float v = 0.f;
      const float s = a.tgt[0];
      // logit_k / ln2 = c0 + c2 |z - mu_k|^2 : c2 = -log2(e) / (2 s^2),
      // c0 = log2(e) * (-2 (log s + log sqrt(2 pi)) - log n_mix)
      if (idx == 0) v = 1.0f / s;
      else if (idx == 1) v = -0.5f * 1.44269504088896340736f / (s * s);
      else if (idx == 2) v = __int_as_float(a.n_mix);
      else if (idx == 3) v = 1.44269504088896340736f * (-2.0f * (logf(s) + kHalfLog2Pi) - logf((float)a.n_mix));
      else if (idx >= 4 && idx < 4 + 2 * a.n_mix) v = a.tgt[1 + (idx - 4)];
      a.ws[a.w.tgt + idx] = v;
    }
  }
}

// One launch for all per-call preparation: blocks [0, K] build the per-bridge bias rows, block K+1 the
// schedule tables, the remaining blocks pack the weights (three dependent-free jobs, one boundary).
struct PrepArgs {
  SchedArgs sched;
  DdsPrepArgs dds;
  GefPrepArgs gef;
  PackArgs pack;
  int32_t K, arch, npack;
};

__global__ __launch_bounds__(256) void prep_fused_kernel(PrepArgs a) {
  const int b = blockIdx.x;
  if (b <= a.K) {
    if (a.arch < 0) return;  // MCD_ULA: no network, no bias table
    if (a.arch == CMCD_ARCH_DDS) prep_dds_body(a.dds, b);
    else prep_geffner_body(a.gef, b);
  } else if (b == a.K + 1) {
    prep_sched_body(a.sched);
  } else {
    pack_weights_body(a.pack, b - a.K - 2, a.npack);
  }
}

// ------------------------------------------------------------------------------------------
// 4. the trajectory kernel
// ------------------------------------------------------------------------------------------

template <int ARCH>
__device__ __forceinline__ float act(float pre) {
  return ARCH == CMCD_ARCH_DDS ? gelu_exact(pre) : softplus(pre);
}

// One evaluation of the score network s(z, idx) for the 16 particles of this wave.
//   dds     (nn_dds.py:159-162): h1 = gelu(W1^T[z; tau] + b1); h2 = gelu(W2^T h1 + b2); clip(W3^T h2 + b3)
//   geffner (nn.py:45-52,66-70): u = [z; emb]; u += softplus(uW1 + b1); u += softplus(uW2 + b2);
//                                factor_sn * (uW3 + b3)
// Diagnostic builds only (-DCMCD_TRAJ_ABL=mask, tools/probes/traj_ablate.py): drop one ingredient of the
// evaluation to see what a saturating batch spends its time on.  1: layer-2 MFMAs, 2: activations,
// 4: target gradient, 8: noise generation.  Results are wrong by design; never defined in the product build.
#ifndef CMCD_TRAJ_ABL
#define CMCD_TRAJ_ABL 0
#endif
// PF (r05): the A fragments of input tile ti + 1 are requested while tile ti's 4 T matrix instructions run (a second set of T
// register quads), the two groups held apart by sched_barriers.  Without them the machine scheduler sinks every fragment read
// to just in front of the four matrix instructions that use it — read, full lgkmcnt(0) wait, 128 cycles of matrix work, T^2 times
// per evaluation; config 4 on one GPU (16 000 particles = 1000 waves on 1024 SIMDs, nothing else to issue) spent 27 % of its
// wave cycles in s_waitcnt (profiles/r05_f_pmc_traj_kernel.json): 2.315 -> 2.003 ms per launch, 65 536 particles 7.63 -> 6.96.
// -DCMCD_TRAJ_PF=0 builds the r04 form (A / B).
#ifndef CMCD_TRAJ_PF
#define CMCD_TRAJ_PF 1
#endif
template <int ARCH, int D, int T, bool PF = (CMCD_TRAJ_PF != 0)>
__device__ __forceinline__ void eval_net(const float (&z)[D], const float* __restrict__ brow,
                                         const float* __restrict__ urow, const float* lds_w2,
                                         const float* lds_w1z, const float* lds_b2, const float* lds_w3t,
                                         const float* lds_b3, int lane, float (&s)[D]) {
  constexpr int HP = 16 * T;
  const int g = lane >> 4;
  asm volatile("" ::: "memory");  // keep the LDS-resident weights streaming (no LICM into VGPRs)
  f32x4 h[T];
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f32x4 pre = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
      pre += z[j] * wv;
    }
    if (ARCH == CMCD_ARCH_DDS) {
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = (CMCD_TRAJ_ABL & 2) ? pre[r] : gelu_fast(pre[r]);
    } else {
      f32x4 u = *reinterpret_cast<const f32x4*>(urow + 16 * t + 4 * g);
      if (16 * t < D) {  // the first D neurons of u are z itself
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int nidx = 16 * t + 4 * g + r;
#pragma unroll
          for (int j = 0; j < D; ++j)
            if (j >= 16 * t && j < 16 * t + 16) u[r] = (nidx == j) ? z[j] : u[r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) h[t][r] = u[r] + softplus(pre[r]);
    }
  }
  // layer 2 on the matrix cores: acc[t_out] (rows = neurons 16 t_out + 4 g + r, cols = particles)
  float dummy = z[0];
  f32x4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = *reinterpret_cast<const f32x4*>(lds_b2 + 16 * t + 4 * g);
  if (PF) {
    f32x4 a[2][T];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int to = 0; to < T; ++to) a[0][to] = *reinterpret_cast<const f32x4*>(lds_w2 + (to * 64 + lane) * 4);
#pragma unroll
    for (int ti = 0; ti < T; ++ti) {
      asm volatile("" ::: "memory");     // (as below: no hoisting of the fragments out of the bridge loop)
      if (ti + 1 < T) {
#pragma unroll
        for (int to = 0; to < T; ++to)
          a[(ti + 1) & 1][to] = *reinterpret_cast<const f32x4*>(lds_w2 + (((ti + 1) * T + to) * 64 + lane) * 4);
      }
      // (the machine scheduler otherwise sinks every fragment read to just in front of the four matrix instructions that
      // use it — a read, a full lgkmcnt(0) wait, 128 cycles of matrix work, 81 times per evaluation: ISA reading r05)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int to = 0; to < T; ++to)
          acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti & 1][to][r], h[ti][r], acc[to], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
  for (int ti = 0; ti < T; ++ti) {
    // wide nets: keep the A fragments streaming from LDS (a compiler-level fence stops LICM from
    // hoisting T*T*4 loop-invariant registers out of the bridge loop and spilling them)
    asm volatile("" ::: "memory");
    f32x4 a[T];
#pragma unroll
    for (int to = 0; to < T; ++to)
      a[to] = *reinterpret_cast<const f32x4*>(lds_w2 + ((ti * T + to) * 64 + lane) * 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int to = 0; to < T; ++to) {
        if (CMCD_TRAJ_ABL & 1) acc[to][r] += a[to][r] * h[ti][r];
        else acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[to][r], h[ti][r], acc[to], 0, 0, 0);
      }
      if (CMCD_TRAJ_ABL & 16) {   // probe: 24 dependent VALU instructions in the shadow of these T MFMAs
#pragma unroll
        for (int q = 0; q < 24; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dummy) : "v"(z[0]));
      }
    }
  }
  }
  if (CMCD_TRAJ_ABL & 32) {       // probe: the same 384 instructions after the MFMA loop
#pragma unroll
    for (int q = 0; q < 24 * 4 * T; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dummy) : "v"(z[0]));
  }
  // layer 3: every lane sums over its 4T neurons, then the 4 lanes of a particle combine
  float part[D];
#pragma unroll
  for (int j = 0; j < D; ++j) part[j] = 0.f;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    f32x4 h2;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      h2[r] = (ARCH == CMCD_ARCH_DDS) ? ((CMCD_TRAJ_ABL & 2) ? acc[t][r] : gelu_fast(acc[t][r])) : h[t][r] + softplus(acc[t][r]);
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
  if (CMCD_TRAJ_ABL & 48) s[0] += dummy * 1e-30f;
}

// r05: the 132-wide net (BASELINE configuration 4: T = 9 = 8 full neuron tiles + 4 real neurons) without its 12 zero
// neurons.  eval_net carries the ninth tile like any other: 4 k-steps of every output tile for 4 real inputs, a whole output
// tile (36 matrix instructions) and 4 activations per lane for 4 real outputs — 324 `16x16x4` per evaluation.  Here:
//   * layer 1: lane (g, c) takes the ONE tail neuron 128 + g of particle c (the C layout of a 17th row group);
//   * tail INPUTS: that activation is exactly the B operand of one `16x16x4` step with k = g — one step per output tile
//     instead of four, the A operand being element g of the packed fragment of lane (0, c);
//   * tail OUTPUTS: `4x4x1` — block (g, c >> 2) = 4 neurons x 4 particles over contraction slice g = the activations the lane
//     already holds (k = 16 t + 4 g + r), the A operand being the packed fragment of lane (g, c & 3) of output tile 8: 33 steps
//     of 2 passes instead of 36 of 8; a reduce-scatter over the four rows leaves neuron 128 + g on lane (g, c).
// 264 `16x16x4` + 33 `4x4x1` = 84 % of the matrix time, 66 activations instead of 72, no new table (every operand is read from
// the fragment table eval_net reads).  Same sums up to the order of the tail's contraction.
__device__ __forceinline__ float traj_rs32(float x, float y) {
  uint32_t r0, r1;
  swap32(__float_as_uint(x), __float_as_uint(y), r0, r1);
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float traj_rs16(float x, float y) {
  uint32_t r0, r1;
  swap16(__float_as_uint(x), __float_as_uint(y), r0, r1);
  return __uint_as_float(r0) + __uint_as_float(r1);
}
template <int ARCH, int D, int T>
__device__ __forceinline__ void eval_net_tail4(const float (&z)[D], const float* __restrict__ brow,
                                               const float* __restrict__ urow, const float* lds_w2,
                                               const float* lds_w1z, const float* lds_b2, const float* lds_w3t,
                                               const float* lds_b3, int lane, float (&s)[D]) {
  static_assert(ARCH == CMCD_ARCH_GEFFNER && T >= 2 && D <= 16, "the tail form is the geffner net's");
  constexpr int HP = 16 * T, TF = T - 1, NX = 16 * TF;
  const int g = lane >> 4, c = lane & 15;
  asm volatile("" ::: "memory");  // keep the LDS-resident weights streaming (no LICM into VGPRs)
  f32x4 h[TF];
#pragma unroll
  for (int t = 0; t < TF; ++t) {
    f32x4 pre = *reinterpret_cast<const f32x4*>(brow + 16 * t + 4 * g);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(lds_w1z + j * HP + 16 * t + 4 * g);
      pre += z[j] * wv;
    }
    f32x4 u = *reinterpret_cast<const f32x4*>(urow + 16 * t + 4 * g);
    if (16 * t < D) {  // the first D neurons of u are z itself
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int nidx = 16 * t + 4 * g + r;
#pragma unroll
        for (int j = 0; j < D; ++j)
          if (j >= 16 * t && j < 16 * t + 16) u[r] = (nidx == j) ? z[j] : u[r];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) h[t][r] = u[r] + softplus(pre[r]);
  }
  float hx;                        // tail neuron NX + g of particle c
  {
    float px = brow[NX + g];
#pragma unroll
    for (int j = 0; j < D; ++j) px = fmaf(z[j], lds_w1z[j * HP + NX + g], px);
    hx = urow[NX + g] + softplus(px);
  }
  f32x4 acc[TF], accx = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < TF; ++t) acc[t] = *reinterpret_cast<const f32x4*>(lds_b2 + 16 * t + 4 * g);
  const float* const w2x = lds_w2 + (TF * 64 + 16 * g + (c & 3)) * 4;   // + ti T 256: fragments of output tile TF, lane (g, c & 3)
  {
    f32x4 a[2][TF], ax[2];
    asm volatile("" ::: "memory");
#pragma unroll
    for (int to = 0; to < TF; ++to) a[0][to] = *reinterpret_cast<const f32x4*>(lds_w2 + (to * 64 + lane) * 4);
    ax[0] = *reinterpret_cast<const f32x4*>(w2x);
#pragma unroll
    for (int ti = 0; ti < TF; ++ti) {
      asm volatile("" ::: "memory");
      if (ti + 1 < TF) {
#pragma unroll
        for (int to = 0; to < TF; ++to)
          a[(ti + 1) & 1][to] = *reinterpret_cast<const f32x4*>(lds_w2 + (((ti + 1) * T + to) * 64 + lane) * 4);
        ax[(ti + 1) & 1] = *reinterpret_cast<const f32x4*>(w2x + (ti + 1) * T * 256);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int to = 0; to < TF; ++to)
          acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ti & 1][to][r], h[ti][r], acc[to], 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_4x4x1f32(ax[ti & 1][r], h[ti][r], accx, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  {
    // the tail inputs (k = NX + g): element g of the fragments of lane (0, c) / lane (0, c & 3) of input tile TF
    float a8[TF];
#pragma unroll
    for (int to = 0; to < TF; ++to) a8[to] = lds_w2[((TF * T + to) * 64 + c) * 4 + g];
    const float axx = lds_w2[((TF * T + TF) * 64 + (c & 3)) * 4 + g];
#pragma unroll
    for (int to = 0; to < TF; ++to) acc[to] = __builtin_amdgcn_mfma_f32_16x16x4f32(a8[to], hx, acc[to], 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_4x4x1f32(axx, hx, accx, 0, 0, 0);
  }
  // the four rows' slices of the tail outputs -> neuron NX + g on lane (g, c)
  float h2x;
  {
    const float u01 = traj_rs16(accx[0], accx[1]), u23 = traj_rs16(accx[2], accx[3]);
    const float avx = traj_rs32(u01, u23) + lds_b2[NX + g];
    h2x = hx + softplus(avx);
  }
  float part[D];
#pragma unroll
  for (int j = 0; j < D; ++j) part[j] = h2x * lds_w3t[j * HP + NX + g];
#pragma unroll
  for (int t = 0; t < TF; ++t) {
    f32x4 h2;
#pragma unroll
    for (int r = 0; r < 4; ++r) h2[r] = h[t][r] + softplus(acc[t][r]);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(lds_w3t + j * HP + 16 * t + 4 * g);
      part[j] += h2[0] * wv[0] + h2[1] * wv[1] + h2[2] * wv[2] + h2[3] * wv[3];
    }
  }
  const float factor = lds_b3[15];
#pragma unroll
  for (int j = 0; j < D; ++j) s[j] = (group_sum(part[j]) + lds_b3[j]) * factor;
}

// which instances keep the next input tile's fragments in flight (eval_net<.., PF>), measured on saturating batches against the
// r04 form (profiles/r05_f_traj_fragment_prefetch_ab.txt): the 9-tile net -11 % (config 4: 2.360 -> 2.107 ms), dds on the 2-d
// targets -1.6 %, the 2-tile net -1.4 %; the 4-tile geffner net (+1.9 %: 12 bytes of scratch at 128 registers) and the funnel
// (+2.6 %) keep the plain loop
constexpr bool traj_pf(int ARCH, int D, int T) {
  return CMCD_TRAJ_PF != 0 && (T == 9 || (D == 2 && (T == 2 || ARCH == CMCD_ARCH_DDS)));
}
// TAIL4: the 132-wide net on eval_net_tail4 (the launch picks the instance: traj_tail4)
#ifndef CMCD_TRAJ_TAIL4
#define CMCD_TRAJ_TAIL4 1
#endif
template <int TARGET, int ARCH, int D, int T, bool PF = traj_pf(ARCH, D, T), bool TAIL4 = false>
__global__ __launch_bounds__(512, (T > 4 || D > 4) ? 2 : 4) void traj_kernel(TrajArgs a) {
  constexpr int HP = 16 * T;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* lds_w2 = lds;                    // HP*HP
  float* lds_w1z = lds_w2 + HP * HP;      // D*HP
  float* lds_w3t = lds_w1z + D * HP;      // D*HP
  float* lds_b2 = lds_w3t + D * HP;       // HP
  float* lds_b3 = lds_b2 + HP;            // 16
  float* lds_tgt = lds_b3 + 16;           // tgt_floats
  {
    // the packed weights sit contiguously in the workspace in this same order
    const f32x4* src = reinterpret_cast<const f32x4*>(a.ws + a.w.w1z);
    f32x4* dst = reinterpret_cast<f32x4*>(lds_w1z);
    for (int i = threadIdx.x; i < D * HP / 4; i += blockDim.x) dst[i] = src[i];
    src = reinterpret_cast<const f32x4*>(a.ws + a.w.w2);
    dst = reinterpret_cast<f32x4*>(lds_w2);
    for (int i = threadIdx.x; i < HP * HP / 4; i += blockDim.x) dst[i] = src[i];
    src = reinterpret_cast<const f32x4*>(a.ws + a.w.w3t);
    dst = reinterpret_cast<f32x4*>(lds_w3t);
    for (int i = threadIdx.x; i < D * HP / 4; i += blockDim.x) dst[i] = src[i];
    for (int i = threadIdx.x; i < HP; i += blockDim.x) lds_b2[i] = a.ws[a.w.b2 + i];
    for (int i = threadIdx.x; i < 16; i += blockDim.x) lds_b3[i] = a.ws[a.w.b3 + i];
    for (int i = threadIdx.x; i < a.w.tgt_floats; i += blockDim.x) lds_tgt[i] = a.ws[a.w.tgt + i];
    // r05: the per-bridge tables (schedule, bias rows [+ residual rows], one contiguous block) into this XCD's L2, one touch per
    // 128-byte line (cmcd_coop.hip: behind the prep launch every XCD's copy is gone and a row is requested only when it is needed)
    float warm = 0.f;
    const int64_t t0 = a.w.sched;
    const int64_t t1 = (ARCH == CMCD_ARCH_GEFFNER ? a.w.utab : a.w.bias1) + (int64_t)(a.K + 1) * HP;
    const int64_t per_xcd = (gridDim.x + 7) >> 3, rank = blockIdx.x >> 3;     // dealt to the workgroups that share an XCD
    for (int64_t i = t0 + 32 * (rank * blockDim.x + threadIdx.x); i < t1; i += 32 * per_xcd * blockDim.x) warm += a.ws[i];
    asm volatile("" ::"v"(warm));
  }
  __syncthreads();

  const int lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wave * 16 >= a.n) return;  // whole wave out of range (after the only barrier)
  const int64_t p = wave * 16 + c;
  const bool valid = p < a.n;
  const int32_t seed = a.seeds[valid ? p : a.n - 1];
  const int K = a.K;

  // q = N(mean, exp(logdiag)^2)                          vardist/diag_gauss.py:15-33
  float qmean[D], qstd[D], qiv[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    qmean[j] = a.params[a.lay.vd_mean + j];
    qstd[j] = expf(a.params[a.lay.vd_logdiag + j]);
    qiv[j] = 1.0f / (qstd[j] * qstd[j]);
  }

  // ---- key chain (mcdboundingmachine.py:151-162, mcd_cais.py:94).  Lane g computes block (g & 1)
  //      of a split: counters (b, 2 + b) -> (out[b], out[2 + b]).
  const int gb = g & 1;
  uint32_t x0, x1, k0 = 0u, k1 = (uint32_t)seed;  // PRNGKey(seed) = (0, seed)
  float z[D];
  {
    x0 = gb; x1 = 2 + gb;
    threefry2x32(k0, k1, x0, x1);  // split(PRNGKey(seed)) -> A = (out0,out1), B = (out2,out3)
    uint32_t a0, a1, b0, b1;
    rows01(x0, a0, a1);
    rows01(x1, b0, b1);
    // z0 = mean + std * normal(A, (D,))                  diag_gauss.py:49-62
    constexpr int Hh = (D + 1) / 2;
    float nz[2 * Hh];
#pragma unroll
    for (int j0 = 0; j0 < Hh; j0 += 4) {
      const int j = j0 + g;  // block j encrypts (ctr[j], ctr[Hh + j]); pad counters are 0
      uint32_t y0 = j, y1 = (Hh + j < D) ? Hh + j : 0;
      threefry2x32(a0, a1, y0, y1);
      if (a.dbg_bits && valid && j < Hh) {
        a.dbg_bits[p * D + j] = y0;
        a.dbg_noise[p * D + j] = bits_to_normal(y0);
        if (Hh + j < D) {
          a.dbg_bits[p * D + Hh + j] = y1;
          a.dbg_noise[p * D + Hh + j] = bits_to_normal(y1);
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
#pragma unroll
    for (int j = 0; j < D; ++j) z[j] = qstd[j] * nz[j] + qmean[j];
    if (a.traj && valid && g == 0) {
#pragma unroll
      for (int j = 0; j < D; ++j) a.traj[p * D + j] = z[j];
    }
    // C = first(split(B)); gen = second(split(C))
    x0 = gb; x1 = 2 + gb;
    threefry2x32(b0, b1, x0, x1);
    uint32_t c0, c1;
    rows01(x0, c0, c1);
    x0 = gb; x1 = 2 + gb;
    threefry2x32(c0, c1, x0, x1);
    rows01(x1, k0, k1);
    if (a.dbg_keys && valid && g == 0) {
      a.dbg_keys[p * 2] = k0;
      a.dbg_keys[p * 2 + 1] = k1;
    }
  }

  // w = -log q(z0)                                       mcdboundingmachine.py:157
  float w = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float dz = z[j] - qmean[j];
    w -= -(dz * dz) / (2.0f * qstd[j] * qstd[j]) - logf(qstd[j]) - kHalfLog2Pi;
  }

  const float clipv = a.var_mode ? 1e2f : 1e3f;  // mcd_cais.py:24 / mcd_cais_var.py:33
  const bool clip_p = a.grad_clipping != 0;
  const bool clip_q = clip_p && a.var_mode;

  const float* bias1 = a.ws + a.w.bias1;
  const float* utab = a.ws + a.w.utab;

  // Rotated loop: iteration i evaluates grad log p and the score net ONCE at (z_i, i); that one
  // evaluation closes step i-1 (its backward kernel, mcd_cais.py:71-79) and opens step i (its
  // forward kernel, :52-67) — the reference evaluates both twice.
  float zp[D];           // z_{i-1}
  float fk_lp = 0.f;     // log F_{i-1}(z_i | z_{i-1})
  float pbeta = 0.f, peps = 0.f, pinv2s2 = 0.f, pcst = 0.f;
  float logp = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) zp[j] = 0.f;

  for (int i = 0; i <= K; ++i) {
    float gp[D], sn[D];
    if (CMCD_TRAJ_ABL & 4) {
#pragma unroll
      for (int j = 0; j < D; ++j) gp[j] = -z[j];
      logp = z[0];
    } else {
      Target<TARGET, D>::eval(z, g, lds_tgt, logp, gp);
    }
    if (a.ula == 1) {
#pragma unroll
      for (int j = 0; j < D; ++j) sn[j] = 0.f;
    } else {
      // CAIS: s(z_i, i) serves both kernels; MCD_ULA_sn: s(z_i, i - 1) serves the backward kernel only
      const int64_t row = (a.ula == 2) ? (i > 0 ? i - 1 : 0) : i;
      if constexpr (TAIL4) eval_net_tail4<ARCH, D, T>(z, bias1 + row * HP, utab + row * HP, lds_w2, lds_w1z, lds_b2, lds_w3t, lds_b3, lane, sn);
      else eval_net<ARCH, D, T, PF>(z, bias1 + row * HP, utab + row * HP, lds_w2, lds_w1z, lds_b2, lds_w3t, lds_b3, lane, sn);
    }
    const float fsn = a.ula ? 0.f : 1.f;  // the ULA forward kernel has no network term
    float gq[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      gq[j] = -(z[j] - qmean[j]) * qiv[j];
      if (clip_p) gp[j] = fminf(fmaxf(gp[j], -clipv), clipv);
      if (clip_q) gq[j] = fminf(fmaxf(gq[j], -clipv), clipv);
    }

    if (i > 0) {
      // ---- backward kernel of step i-1 at z_new = z with net index i      mcd_cais.py:71-86
      float bk_lp = 0.f;
#pragma unroll
      for (int j = 0; j < D; ++j) {
        const float ub = -1.0f * (pbeta * gp[j] + (1.0f - pbeta) * gq[j]);
        const float bk = z[j] - peps * ub + peps * sn[j];
        const float db = zp[j] - bk;
        bk_lp += -(db * db) * pinv2s2 - pcst;  // log_prob_kernel, mcd_utils.py:19-21 (pcst = log sigma + log sqrt(2 pi))
      }
      w += bk_lp - fk_lp;
    }
    if (i == K) break;

    // the packed schedule row {beta, eps, sigma, log sigma + log sqrt(2 pi), 1 / (2 sigma^2), ...} of the prep launch
    // (one scalar load; the division — ~10 VALU instructions per bridge on every lane — is done there once)
    const float* scr = a.ws + a.w.sched + 8 * (int64_t)i;
    const float beta = scr[0], eps = scr[1], sig = scr[2], cst = scr[3], inv2s2 = scr[4];

    // ---- noise: (G, H) = split(gen); eps_i = normal(G, (D,)); gen = second(split(H))
    //      mcd_cais.py:66-67,87
    x0 = gb; x1 = 2 + gb;
    threefry2x32(k0, k1, x0, x1);
    uint32_t g0, g1, h0, h1;
    rows01(x0, g0, g1);
    rows01(x1, h0, h1);
    constexpr int Hh = (D + 1) / 2;
    constexpr int NB = 2 + Hh;  // blocks of this stage: 2 for split(H), Hh for normal(G)
    float nz[2 * Hh];
#pragma unroll
    for (int b0 = 0; b0 < NB; b0 += 4) {
      const int b = b0 + g;               // block handled by this lane in this pass
      const bool is_split = b < 2;
      const int jn = b - 2;               // normal block index
      uint32_t y0 = is_split ? b : jn;
      uint32_t y1 = is_split ? 2 + b : ((Hh + jn < D) ? Hh + jn : 0);
      if (!(CMCD_TRAJ_ABL & 8)) threefry2x32(is_split ? h0 : g0, is_split ? h1 : g1, y0, y1);
      if (b0 == 0) {
        rows01(y1, k0, k1);
        if (a.dbg_keys && valid && g == 0) {
          a.dbg_keys[((int64_t)(i + 1) * a.n + p) * 2] = k0;
          a.dbg_keys[((int64_t)(i + 1) * a.n + p) * 2 + 1] = k1;
        }
      }
      if (a.dbg_bits && valid && jn >= 0 && jn < Hh) {
        const int64_t o = ((int64_t)(i + 1) * a.n + p) * D;
        a.dbg_bits[o + jn] = y0;
        a.dbg_noise[o + jn] = bits_to_normal(y0);
        if (Hh + jn < D) {
          a.dbg_bits[o + Hh + jn] = y1;
          a.dbg_noise[o + Hh + jn] = bits_to_normal(y1);
        }
      }
      if (D == 2 && !(CMCD_TRAJ_ABL & 8)) {
        // d = 2: the one normal block sits on row 2 with both words.  Word 1 moves to row 3 (one row swap), so that ONE
        // bits -> deviate conversion serves both words (row 2 converts word 0, row 3 word 1; rows 0, 1 convert the split
        // blocks' words, ignored) and one broadcast of the rows delivers them: a conversion (~18 instructions) and
        // three row swaps less per evaluation than converting y0 and y1 on every row (r02).
        uint32_t t0, t1;
        swap16(y1, y1, t0, t1);                        // t0 = rows [y1(0) y1(0) y1(2) y1(2)]
        const float dev = bits_to_normal(g == 3 ? t0 : y0);
        uint32_t rr[4];
        rows0123(__float_as_uint(dev), rr);
        nz[0] = __uint_as_float(rr[2]);
        nz[1] = __uint_as_float(rr[3]);
      } else {
        uint32_t r0[4], r1[4];
        rows0123(__float_as_uint((CMCD_TRAJ_ABL & 8) ? __uint_as_float(y0) * 1e-9f : bits_to_normal(y0)), r0);
        rows0123(__float_as_uint((CMCD_TRAJ_ABL & 8) ? __uint_as_float(y1) * 1e-9f : bits_to_normal(y1)), r1);
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

    // ---- forward kernel of step i                                          mcd_cais.py:52-67
    fk_lp = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float uf = -1.0f * (beta * gp[j] + (1.0f - beta) * gq[j]);
      const float fk = z[j] - eps * uf - eps * (fsn * sn[j]);
      const float zn = fk + sig * nz[j];
      const float df = zn - fk;
      fk_lp += -(df * df) * inv2s2 - cst;
      zp[j] = z[j];
      z[j] = zn;
    }
    if (a.traj && valid && g == 0) {
#pragma unroll
      for (int j = 0; j < D; ++j) a.traj[((int64_t)(i + 1) * a.n + p) * D + j] = z[j];
    }
    pbeta = beta; peps = eps; pinv2s2 = inv2s2; pcst = cst;
  }
  w += logp;  // + log p(z_K)   mcdboundingmachine.py:178
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

// ------------------------------------------------------------------------------------------
// 5. finalize: merge the per-wave statistics in a fixed order (deterministic).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void stats_merge(double* a, const double* b) {
  a[0] += b[0];
  a[1] += b[1];
  a[2] += b[2];
  const double m = fmax(a[3], b[3]);
  const double sa = (a[3] > -INFINITY && m < INFINITY) ? a[4] * exp(a[3] - m) : (a[3] == m ? a[4] : 0.0);
  const double sb = (b[3] > -INFINITY && m < INFINITY) ? b[4] * exp(b[3] - m) : (b[3] == m ? b[4] : 0.0);
  a[3] = m;
  a[4] = sa + sb;
}

// Two passes over the records (L2-resident): the global maximum first, then every record scaled to it ONCE — one
// double-precision exp per record instead of two per level of a pairwise merge tree (5.2 -> ~3 us for 250 records).
// Same edge semantics as stats_merge (records of all-inf tiles, +inf maxima); fixed summation order.
// cmcd_bound_forward_prepared: `stamp_slot` (non-null) points at the stamp the prep launch of the call that FORMED the tables
// left in the workspace; when it is not the stamp of this call's (desc, layout, n, n_params, n_target) the caller broke the
// entry point's contract and the five statistics go out as NaN ("diverged": loud, and no host synchronisation needed).
__global__ __launch_bounds__(256) void finalize_kernel(const double* partials, int32_t n_waves, double* out,
                                                       const uint32_t* stamp_slot = nullptr, uint32_t stamp_expect = 0) {
  __shared__ double shm[256];
  __shared__ double sh[256][4];
  // contiguous chunks keep the merge order independent of blockDim-strided races
  const int per = (n_waves + 255) / 256;
  const int lo = threadIdx.x * per, hi = min(n_waves, lo + per);
  double m = -INFINITY;
  for (int i = lo; i < hi; ++i) m = fmax(m, partials[(int64_t)i * CMCD_NSTATS + 3]);
  shm[threadIdx.x] = m;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) shm[threadIdx.x] = fmax(shm[threadIdx.x], shm[threadIdx.x + s]);
    __syncthreads();
  }
  const double M = shm[0];
  double acc[4] = {0, 0, 0, 0};
  for (int i = lo; i < hi; ++i) {
    const double* p = partials + (int64_t)i * CMCD_NSTATS;
    acc[0] += p[0];
    acc[1] += p[1];
    acc[2] += p[2];
    acc[3] += (p[3] > -INFINITY && M < INFINITY) ? p[4] * exp(p[3] - M) : (p[3] == M ? p[4] : 0.0);
  }
  for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] = acc[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s)
      for (int k = 0; k < 4; ++k) sh[threadIdx.x][k] += sh[threadIdx.x + s][k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    out[0] = sh[0][0];
    out[1] = sh[0][1];
    out[2] = sh[0][2];
    out[3] = M;
    out[4] = sh[0][3];
    if (stamp_slot && *stamp_slot != stamp_expect)
      for (int k = 0; k < CMCD_NSTATS; ++k) out[k] = __builtin_nan("");
  }
}

// omega_n = d clip(var(l, ddof=0), +-1e7) / d w_n = -(2 / N)(l_n - mean l) from the merged statistics {.., sum l, sum l^2, ..};
// zero where the clip of mcdboundingmachine.py:231 is active (jnp.clip passes no gradient outside its bounds); a NaN
// variance (an infinite loss in the batch) gives NaN weights, as jax.grad does.
__global__ void vargrad_weights_kernel(const float* loss, const double* stats, int64_t n, int64_t n_total,
                                       float* omega) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double mean = stats[1] / (double)n_total;
  const double var = stats[2] / (double)n_total - mean * mean;
  const bool clipped = var > 1e7 || var < -1e7;
  // NaN variance (inf - inf): reverse mode multiplies the zero cotangent of the clip into infinite / NaN partials, every
  // entry of jax's gradient is NaN; the closed form alone would give +-inf for the finite losses, which Adam's clip
  // would turn into full-size steps
  omega[i] = (var != var) ? __builtin_nanf("") : (clipped ? 0.0f : (float)(-2.0 / (double)n_total * ((double)loss[i] - mean)));
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
typedef void (*traj_fn)(TrajArgs);

template <int TARGET, int ARCH, int D>
static traj_fn pick_T(int T) {
  switch (T) {
    case 2: return traj_kernel<TARGET, ARCH, D, 2>;
    case 4: return traj_kernel<TARGET, ARCH, D, 4>;
    case 9: return traj_kernel<TARGET, ARCH, D, 9>;
    default: return nullptr;
  }
}

// the instances on eval_net_tail4: the geffner net of the 2-d targets whose ninth tile holds at most 4 real neurons (2 + 130 = 132)
static bool traj_tail4(const cmcd_desc& d, int T) {
  const int in = net_in_dim(d) + d.emb_dim;
  return CMCD_TRAJ_TAIL4 != 0 && d.arch == CMCD_ARCH_GEFFNER && d.dim == 2 && T == 9 && in > 128 && in <= 132 &&
         (d.target == CMCD_TARGET_MANY_GMM || d.target == CMCD_TARGET_GMM);
}

static traj_fn pick_kernel(const cmcd_desc& d, int T) {
  if (traj_tail4(d, T)) {
    constexpr bool pf = traj_pf(CMCD_ARCH_GEFFNER, 2, 9);
    return d.target == CMCD_TARGET_MANY_GMM ? traj_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2, 9, pf, true>
                                            : traj_kernel<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2, 9, pf, true>;
  }
  if (d.arch == CMCD_ARCH_DDS) {
    if (T != 4) return nullptr;
    if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return traj_kernel<CMCD_TARGET_MANY_GMM, CMCD_ARCH_DDS, 2, 4>;
    if (d.target == CMCD_TARGET_GMM && d.dim == 2) return traj_kernel<CMCD_TARGET_GMM, CMCD_ARCH_DDS, 2, 4>;
    if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return traj_kernel<CMCD_TARGET_FUNNEL, CMCD_ARCH_DDS, 10, 4>;
    return nullptr;
  }
  if (d.target == CMCD_TARGET_MANY_GMM && d.dim == 2) return pick_T<CMCD_TARGET_MANY_GMM, CMCD_ARCH_GEFFNER, 2>(T);
  if (d.target == CMCD_TARGET_GMM && d.dim == 2) return pick_T<CMCD_TARGET_GMM, CMCD_ARCH_GEFFNER, 2>(T);
  if (d.target == CMCD_TARGET_FUNNEL && d.dim == 10) return pick_T<CMCD_TARGET_FUNNEL, CMCD_ARCH_GEFFNER, 10>(T);
  return nullptr;
}

// What a set of prepared tables was formed from, as far as the library can know it without reading device memory: FNV-1a
// over the descriptor (minus the kernel-variant field, which selects a kernel and not a table), the layout and the sizes.
static uint32_t tables_stamp(const cmcd_desc& d, const cmcd_layout& lay, int64_t n, int64_t n_params, int64_t n_target) {
  uint32_t h = 2166136261u;
  auto eat = [&](const void* p, size_t len) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < len; ++i) { h ^= b[i]; h *= 16777619u; }
  };
  cmcd_desc dd = d;
  dd.reserved = 0;
  eat(&dd, sizeof dd); eat(&lay, sizeof lay); eat(&n, sizeof n); eat(&n_params, sizeof n_params); eat(&n_target, sizeof n_target);
  return h ? h : 1u;
}

static void launch_prep(const cmcd_desc& d, const cmcd_layout& layr, const WsLayout& w, const float* params,
                        const float* target_consts, int n_mix, float* ws, hipStream_t stream, uint32_t stamp) {
  const cmcd_layout* lay = &layr;
  // D here = the state inputs of the network: z, or concat(z, rho) for the momentum mode
  const int64_t K = d.nbridges, D = net_in_dim(d), E = d.emb_dim, IN = D + E;
  PrepArgs pa{};
  const bool ula_mode = d.mode == CMCD_MODE_ULA || d.mode == CMCD_MODE_ULA_SN;
  // MCD_CAIS_UHA_sn: the cos^2 schedule is part of the function body (mcd_under_lp_a_cais.py:33-40,48)
  const int sched = ula_mode ? CMCD_EPS_CONST : (d.mode == CMCD_MODE_CAIS_UHA_SN ? CMCD_EPS_COS_SQ : d.eps_schedule);
  pa.sched = SchedArgs{params, ws, *lay, w, (int32_t)K, d.ngrid, sched, -1, -1};
  PackArgs& pk = pa.pack;
  pk.stamp = stamp;
  pa.sched.stamp = stamp;
  pk.params = params; pk.tgt = target_consts; pk.ws = ws; pk.w = w;
  pk.D = (int32_t)D; pk.DO = d.dim; pk.target = d.target; pk.n_mix = n_mix;
  if (d.mode == CMCD_MODE_ULA) {
    pk.o_w1 = pk.o_w2 = pk.o_b2 = pk.o_w3 = pk.o_b3 = pk.o_factor = -1; pk.IN = 0;
  } else if (d.arch == CMCD_ARCH_DDS) {
    pa.dds = DdsPrepArgs{params, ws, *lay, w, (int32_t)D};
    pk.o_w1 = lay->d_sw1; pk.o_w2 = lay->d_sw2; pk.o_b2 = lay->d_sb2; pk.o_w3 = lay->d_sw3;
    pk.o_b3 = lay->d_sb3; pk.o_factor = -1; pk.IN = 64;
  } else {
    pa.gef = GefPrepArgs{params, ws, *lay, w, (int32_t)D, (int32_t)E, (int32_t)K};
    pk.o_w1 = lay->g_w1; pk.o_w2 = lay->g_w2; pk.o_b2 = lay->g_b2; pk.o_w3 = lay->g_w3;
    pk.o_b3 = lay->g_b3; pk.o_factor = lay->g_factor; pk.IN = (int32_t)IN;
  }
  pa.K = (int32_t)K; pa.arch = d.mode == CMCD_MODE_ULA ? -1 : d.arch;
  pa.npack = (w.HP * w.HP + 255) / 256;
  if (pa.npack > 64) pa.npack = 64;
  hipLaunchKernelGGL(prep_fused_kernel, dim3((unsigned)(K + 2 + pa.npack)), dim3(256), 0, stream, pa);


}

int fail_msg(int code, const char* msg) { return fail(code, "%s", msg); }

int launch_finalize(const double* partials, int32_t count, double* out5, void* stream) {
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), partials, count, out5,
                     (const uint32_t*)nullptr, 0u);
  return hipGetLastError() == hipSuccess ? CMCD_OK : fail(CMCD_ERR_HIP, "finalize launch failed%s");
}

static int check_desc(const cmcd_desc* d) {
  if (!d) return fail(CMCD_ERR_BAD_ARG, "null desc%s");
  if (d->mode < CMCD_MODE_CAIS_SN || d->mode > CMCD_MODE_CAIS_UHA_SN)
    return fail(CMCD_ERR_UNSUPPORTED, "Mode not implemented.%s");
  if (d->mode == CMCD_MODE_ULA && d->arch != CMCD_ARCH_DDS)
    return fail(CMCD_ERR_BAD_ARG, "MCD_ULA has no network: pass arch = CMCD_ARCH_DDS as the placeholder%s");
  if (d->arch != CMCD_ARCH_DDS && d->arch != CMCD_ARCH_GEFFNER)
    return fail(CMCD_ERR_UNSUPPORTED, "nn_arch not implemented%s");
  if (d->nbridges < 1) return fail(CMCD_ERR_BAD_ARG, "nbridges must be >= 1%s");
  if (d->ngrid < 1 || d->ngrid > 32) return fail(CMCD_ERR_BAD_ARG, "ngrid must be in [1, 32]%s");
  if (d->eps_schedule == CMCD_EPS_LINEAR && d->nbridges < 2)
    return fail(CMCD_ERR_BAD_ARG, "linear eps schedule needs nbridges >= 2%s");
  int HP;
  if (!hidden_width(*d, HP)) return fail(CMCD_ERR_BAD_ARG, "bad emb_dim%s");
  if (d->target == CMCD_TARGET_LGCP) {
    if ((d->arch != CMCD_ARCH_GEFFNER && d->mode != CMCD_MODE_ULA) || d->dim < 4 || d->dim > 4096)
      return fail(CMCD_ERR_UNSUPPORTED, "lgcp runs with the geffner net only%s");
    return CMCD_OK;
  }
  if (d->mode == CMCD_MODE_CAIS_UHA_SN) {
    if (!uha_available(*d, HP / 16))
      return fail(CMCD_ERR_UNSUPPORTED, "no MCD_CAIS_UHA_sn kernel instance for this (target, dim, arch, width=%s%lld)", "", HP);
    return CMCD_OK;
  }
  if (!pick_kernel(*d, HP / 16))
    return fail(CMCD_ERR_UNSUPPORTED, "no kernel instance for this (target, dim, arch, width=%s%lld)", "", HP);
  return CMCD_OK;
}

}  // namespace cmcd

using namespace cmcd;

extern "C" {

int cmcd_version(void) { return CMCD_ABI_VERSION; }
const char* cmcd_last_error(void) { return g_err; }
#ifndef CMCD_NO_DIAG_HOOKS   // include/cmcd_hip_diag.h: compiled out of a boundary-only build
const char* cmcd_last_kernel_name(void) { return g_kernel_name; }
#endif

int64_t cmcd_target_floats(const cmcd_desc* desc, int32_t n_mixes) {
  if (!desc) return -1;
  switch (desc->target) {
    case CMCD_TARGET_GMM:
    case CMCD_TARGET_FUNNEL: return 0;
    case CMCD_TARGET_MANY_GMM: return 1 + 2 * (int64_t)n_mixes;
    case CMCD_TARGET_LGCP: return (int64_t)desc->dim * desc->dim + desc->dim + 3;
    default: return -1;
  }
}

int64_t cmcd_workspace_bytes(const cmcd_desc* desc, int64_t n) {
  if (check_desc(desc) != CMCD_OK || n < 1) return 0;
  WsLayout w;
  if (desc->target == CMCD_TARGET_LGCP) {
    make_ws_lgcp(*desc, n, w);
    return lgcp_workspace_floats(*desc, n, w.total_floats) * 4;
  }
  // size for the largest target-constant block this target can stage (64 mixtures)
  const int64_t nt = desc->target == CMCD_TARGET_MANY_GMM ? 1 + 2 * 64 : 0;
  if (!make_ws(*desc, n, nt, w)) return 0;
  return w.total_floats * 4;
}

#define CMCD_HIP_CHECK(expr)                                                                   \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(CMCD_ERR_HIP, "HIP error: %s (code %lld)", hipGetErrorString(e_), (long long)e_); \
  } while (0)

static int forward_impl(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                        const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                        void* workspace, int64_t workspace_bytes, float* out_loss, float* out_z,
                        double* out_stats, float* traj, void* stream_, bool tables_ready = false) {
  const NoiseCapture cap = g_capture;   // armed by cmcd_debug_capture_noise: this call consumes it, whatever happens
  g_capture = NoiseCapture{};
  int rc = check_desc(desc);
  if (rc != CMCD_OK) return rc;
  if ((cap.bits || cap.keys) && desc->target == CMCD_TARGET_LGCP)
    return fail(CMCD_ERR_UNSUPPORTED, "cmcd_debug_capture_noise: trajectory kernels only (not the lgcp launch sequence)%s");
  if (!lay || !seeds || !params || !workspace || !out_loss || !out_z || !out_stats)
    return fail(CMCD_ERR_BAD_ARG, "null pointer argument%s");
  if (n < 1 || n > (int64_t)1 << 31) return fail(CMCD_ERR_BAD_ARG, "n out of range%s");
  const cmcd_desc& d = *desc;
  const int64_t K = d.nbridges, D = d.dim, E = d.emb_dim, DIN = net_in_dim(d), IN = DIN + E;
  const bool uha = d.mode == CMCD_MODE_CAIS_UHA_SN;

  // every leaf this configuration reads must lie inside params_flat
  auto need = [&](int64_t off, int64_t len) { return off >= 0 && off + len <= n_params; };
  bool ok = need(lay->vd_mean, D) && need(lay->vd_logdiag, D) && need(lay->eps, 1) &&
            need(lay->mgridref_y, d.ngrid + 1) && (!uha || need(lay->gamma, 1));
  if (d.mode == CMCD_MODE_ULA) {
    // no network leaves
  } else if (d.arch == CMCD_ARCH_GEFFNER)
    ok = ok && need(lay->g_emb, K * E) && need(lay->g_factor, 1) && need(lay->g_w1, IN * IN) &&
         need(lay->g_b1, IN) && need(lay->g_w2, IN * IN) && need(lay->g_b2, IN) && need(lay->g_w3, IN * D) &&
         need(lay->g_b3, D);
  else
    ok = ok && need(lay->d_phase, 64) && need(lay->d_tw1, 128 * 64) && need(lay->d_tb1, 64) &&
         need(lay->d_tw2, 64 * 64) && need(lay->d_tb2, 64) && need(lay->d_sw1, (DIN + 64) * 64) &&
         need(lay->d_sb1, 64) && need(lay->d_sw2, 64 * 64) && need(lay->d_sb2, 64) &&
         need(lay->d_sw3, 64 * D) && need(lay->d_sb3, D);
  if (!ok) return fail(CMCD_ERR_BAD_ARG, "layout offset missing or outside params_flat%s");

  int n_mix = 0;
  if (d.target == CMCD_TARGET_MANY_GMM) {
    if (!target_consts || n_target < 3 || (n_target - 1) % 2 != 0 || (n_target - 1) / 2 > 64)
      return fail(CMCD_ERR_BAD_ARG, "many_gmm needs target_consts = {scale, means[n_mixes<=64][2]}%s");
    n_mix = int((n_target - 1) / 2);
  } else if (d.target == CMCD_TARGET_LGCP) {
    if (!target_consts || n_target != D * D + D + 3)
      return fail(CMCD_ERR_BAD_ARG, "lgcp needs target_consts = {Kinv[d,d], counts[d], mu0, a, lognorm}%s");
    WsLayout lw;
    make_ws_lgcp(d, n, lw);
    const int64_t need = lgcp_workspace_floats(d, n, lw.total_floats) * 4;
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
      return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
    hipStream_t st = static_cast<hipStream_t>(stream_);
    float* wsf = static_cast<float*>(workspace);
    // the overdamped baselines: constant eps, no clipping (the reference's dispatcher passes neither, mcd_utils.py:35-58)
    cmcd_desc dl = d;
    if (d.mode == CMCD_MODE_ULA || d.mode == CMCD_MODE_ULA_SN) { dl.eps_schedule = CMCD_EPS_CONST; dl.grad_clipping = 0; }
    if (uha) { dl.eps_schedule = CMCD_EPS_COS_SQ; dl.grad_clipping = 1; }   // fixed by the function body (mcd_under_lp_a_cais.py:23-48)
    // (cmcd_bound_forward_prepared: the schedule table, the first-layer bias table and the packed weights are still in the
    // workspace; the 2nd-order sequence has no prepared form)
    const bool lgcp_ready = tables_ready && !uha;
    SchedArgs sa{params, wsf, *lay, lw, (int32_t)K, d.ngrid, dl.eps_schedule, -1, -1};
    const uint32_t lstamp = tables_stamp(d, *lay, n, n_params, n_target);
    sa.stamp = lstamp;
    if (!lgcp_ready) hipLaunchKernelGGL(prep_sched_kernel, dim3(1), dim3(256), 0, st, sa);
    double* partials = nullptr;
    snprintf(g_kernel_name, sizeof(g_kernel_name), "%s", lgcp_use_wide(dl, n, traj != nullptr)
                 ? "lgcp wide-batch sequence (32x128-tile fp32 GEMM launches)"
                 : "lgcp launch sequence (skinny GEMMs + state kernels)");
    // gradient calls (traj set) place the gradient workspace right in front of the kept trajectory: the forward's consumers
    // keep their activations in its tables, so that the reverse sweep does not recompute them (cmcd_lgcp.hip: lgcp_keep)
    float* keep_gws = traj ? traj - align4(lgcp_grad_workspace_floats(dl, n)) : nullptr;
    rc = lgcp_forward(dl, *lay, lw, seeds, n, params, target_consts, wsf, out_loss, out_z, &partials, traj, stream_, lgcp_ready,
                      keep_gws);
    if (rc != CMCD_OK) return fail(rc, "lgcp launch sequence failed%s");
    // (prepared form: the stamp the forming call's schedule launch left in sched[0][7] must be this call's)
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, st, partials, (int32_t)n, out_stats,
                       lgcp_ready ? reinterpret_cast<const uint32_t*>(wsf + lw.sched + 7) : nullptr, lstamp);
    CMCD_HIP_CHECK(hipGetLastError());
    return CMCD_OK;
  }

  WsLayout w;
  if (!make_ws(d, n, n_target, w)) return fail(CMCD_ERR_BAD_ARG, "bad descriptor%s");
  if (workspace_bytes < w.total_floats * 4 || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "",
                w.total_floats * 4);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* ws = static_cast<float*>(workspace);

  // cmcd_bound_forward_prepared: the caller vouches that the workspace still holds the tables a previous call formed from the
  // SAME (desc, layout, params, target constants, n): the prep launch (4.8 us + a kernel boundary per call) is skipped
  const uint32_t stamp = tables_stamp(d, *lay, n, n_params, n_target);
  if (!tables_ready) launch_prep(d, *lay, w, params, target_consts, n_mix, ws, stream, stamp);
  // prepared form: whoever writes the statistics compares the stamp left in b3[13] with this call's (finalize_kernel)
  const uint32_t* stamp_slot = tables_ready ? reinterpret_cast<const uint32_t*>(ws + w.b3 + 13) : nullptr;

  if (uha) {   // 2nd-order CMCD: its own trajectory kernel (cmcd_uha.hip), same prep tables and statistics merge
    TrajArgs tu{seeds, params, ws, reinterpret_cast<double*>(ws + w.partials), out_loss, out_z, *lay, w, n,
                (int32_t)K, 0, 1, traj, 0};
    tu.dbg_bits = cap.bits; tu.dbg_keys = cap.keys; tu.dbg_noise = cap.noise;
    int n_records = w.n_waves;
    rc = uha_forward_launch(d, tu, stream, &n_records);
    snprintf(g_kernel_name, sizeof(g_kernel_name), "%s", uha_last_kernel_name());
    if (rc != CMCD_OK) return fail(rc, "MCD_CAIS_UHA_sn launch failed%s");
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, stream,
                       reinterpret_cast<const double*>(ws + w.partials), (int32_t)n_records, out_stats, stamp_slot, stamp);
    CMCD_HIP_CHECK(hipGetLastError());
    return CMCD_OK;
  }

  TrajArgs ta{seeds, params, ws, reinterpret_cast<double*>(ws + w.partials), out_loss, out_z, *lay, w, n,
              (int32_t)K, d.mode == CMCD_MODE_CAIS_VAR_SN ? 1 : 0,
              (d.mode == CMCD_MODE_ULA || d.mode == CMCD_MODE_ULA_SN) ? 0 : d.grad_clipping, traj,
              d.mode == CMCD_MODE_ULA ? 1 : (d.mode == CMCD_MODE_ULA_SN ? 2 : 0)};
  ta.dbg_bits = cap.bits; ta.dbg_keys = cap.keys; ta.dbg_noise = cap.noise;
  // Kernel variant (desc.reserved: 0 auto, 1 wave-per-tile, 2 CU-cooperative, 3 cooperative on 16-particle tiles,
  // 4 cooperative on 8-particle tiles).  Auto: the cooperative kernel while the batch cannot fill the chip with one
  // wave per tile, on 8-particle tiles while those still get a CU each (n <= 8 x 256).
  const bool coop_ok = coop_available(d, w.T) && d.mode != CMCD_MODE_ULA;
  const bool forced = d.reserved >= 2 && d.reserved <= 5;
  bool use_coop = forced ? coop_ok : (d.reserved == 1 ? false : (coop_ok && w.n_waves <= coop_max_tiles(d, w.T)));
  if (forced && !coop_ok) return fail(CMCD_ERR_UNSUPPORTED, "no cooperative kernel instance%s");
  const bool half_ok = coop_half_available(d, w.T);
  if ((d.reserved == 4 || d.reserved == 5) && !half_ok) return fail(CMCD_ERR_UNSUPPORTED, "no 8-particle-tile cooperative instance%s");
  const bool half = d.reserved == 4 || d.reserved == 5 || (d.reserved != 3 && half_ok && n <= 8 * 256);
  const bool wide8 = half && d.reserved != 5 && coop_wide8_available(d, w.T);   // d = 10: the dealt-coordinates kernel (5 = the narrow form, A / B)
  if (use_coop) {
    const bool prof = g_prof.on && g_prof.used < ProfileState::kMax;
    if (prof) {
      if (g_prof.used >= g_prof.created) {
        CMCD_HIP_CHECK(hipEventCreate(&g_prof.ev[g_prof.created][0]));
        CMCD_HIP_CHECK(hipEventCreate(&g_prof.ev[g_prof.created][1]));
        ++g_prof.created;
      }
      CMCD_HIP_CHECK(hipEventRecord(g_prof.ev[g_prof.used][0], stream));
    }
    snprintf(g_kernel_name, sizeof(g_kernel_name), "%s<%d-particle tiles%s>", wide8 ? "coop_wide8_kernel" : "coop_kernel",
             half ? 8 : 16, w.T == 9 ? ", 132-wide net" : "");
    // Small grids (<= 64 workgroups: the launch-bound configurations — gmm / funnel at N = 300 are 38 workgroups): the
    // statistics are merged by the last workgroup to arrive (its counter: the free slot 14 of the b3 row, zeroed by the prep
    // launch of this call) and the finalize launch is dropped: gmm N = 300, K = 8 0.0266 -> 0.0245 ms per call.  Larger
    // grids keep the finalize launch: at the named batch's 250 workgroups the merge tail costs the trajectory kernel what
    // the launch saves (per call 0.2026 vs 0.2024 ms).  Same five doubles bit for bit either way.
    const int64_t coop_wgs = half ? (n + 7) / 8 : w.n_waves;
    const bool fused_merge = coop_wgs <= 64;
    if (fused_merge) {
      ta.fin_out = out_stats;
      ta.fin_counter = reinterpret_cast<int32_t*>(ws + w.b3 + 14);
      ta.stamp_slot = stamp_slot;
      ta.stamp_expect = stamp;
    }
    rc = coop_launch(d, ta, half, stream, !wide8);
    if (rc != CMCD_OK) return fail(rc, "cooperative launch failed%s");
    if (prof) {
      CMCD_HIP_CHECK(hipEventRecord(g_prof.ev[g_prof.used][1], stream));
      ++g_prof.used;
    }
    if (!fused_merge)
      hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, stream,
                         reinterpret_cast<const double*>(ws + w.partials), (int32_t)coop_wgs, out_stats, stamp_slot, stamp);
    CMCD_HIP_CHECK(hipGetLastError());
    return CMCD_OK;
  }
  traj_fn fn = pick_kernel(d, w.T);
  // waves per workgroup: one wave per CU until every CU has one, then grow (weights are
  // staged once per workgroup, so bigger groups amortise the LDS fill).
  const int64_t tiles = w.n_waves;
  const size_t lds_bytes = size_t(w.HP * w.HP + 2 * D * w.HP + w.HP + 16 + w.tgt_floats) * 4;
  if (lds_bytes > 160 * 1024) return fail(CMCD_ERR_UNSUPPORTED, "network too wide for LDS%s");
  // Waves per workgroup.  Up to 1024 tiles: one wave per workgroup spreads over all SIMDs.  Beyond
  // that single-wave workgroups pile onto the same SIMD of a CU (measured: 2048 x 1 wave ran 2.4x
  // longer than 1024 x 1), so use 4-wave workgroups (one wave per SIMD), 8 for very large batches
  // or when LDS limits the CU to one resident workgroup.
  int64_t per_cu = (160 * 1024) / (int64_t)lds_bytes;
  int nw = tiles <= 1024 ? 1 : (tiles <= 8192 ? 4 : 8);
  if (per_cu < 2 && tiles > 256) nw = tiles <= 1024 ? 4 : 8;
  CMCD_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  const unsigned blocks = unsigned((tiles + nw - 1) / nw);
  const bool prof = g_prof.on && g_prof.used < ProfileState::kMax;
  if (prof) {
    if (g_prof.used >= g_prof.created) {
      CMCD_HIP_CHECK(hipEventCreate(&g_prof.ev[g_prof.created][0]));
      CMCD_HIP_CHECK(hipEventCreate(&g_prof.ev[g_prof.created][1]));
      ++g_prof.created;
    }
    CMCD_HIP_CHECK(hipEventRecord(g_prof.ev[g_prof.used][0], stream));
  }
  snprintf(g_kernel_name, sizeof(g_kernel_name), "traj_kernel");
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(64 * nw), lds_bytes, stream, ta);
  if (prof) {
    CMCD_HIP_CHECK(hipEventRecord(g_prof.ev[g_prof.used][1], stream));
    ++g_prof.used;
  }
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, stream,
                     reinterpret_cast<const double*>(ws + w.partials), w.n_waves, out_stats, stamp_slot, stamp);
  CMCD_HIP_CHECK(hipGetLastError());
  return CMCD_OK;
}

#ifndef CMCD_NO_DIAG_HOOKS
int cmcd_debug_capture_noise(uint32_t* bits, uint32_t* gen_keys, float* noise) {
  if ((bits == nullptr) != (noise == nullptr)) return fail(CMCD_ERR_BAD_ARG, "bits and noise go together%s");
  g_capture.bits = bits; g_capture.keys = gen_keys; g_capture.noise = noise;
  return CMCD_OK;
}
#endif

int cmcd_bound_forward(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                       const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                       void* workspace, int64_t workspace_bytes, float* out_loss, float* out_z,
                       double* out_stats, void* stream_) {
  return forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, workspace_bytes,
                      out_loss, out_z, out_stats, nullptr, stream_);
}

int cmcd_bound_forward_prepared(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                                const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                                void* workspace, int64_t workspace_bytes, float* out_loss, float* out_z,
                                double* out_stats, void* stream_) {
  const bool ready = desc != nullptr;
  return forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, workspace_bytes,
                      out_loss, out_z, out_stats, nullptr, stream_, ready);
}

int64_t cmcd_bound_grad_workspace_bytes(const cmcd_desc* desc, int64_t n) {
  if (check_desc(desc) != CMCD_OK || n < 1) return 0;
  if (desc->target == CMCD_TARGET_LGCP) {
    if (desc->mode == CMCD_MODE_CAIS_VAR_SN) {
      fail(CMCD_ERR_UNSUPPORTED, "MCD_CAIS_var_sn: cmcd_grad_workspace_bytes / cmcd_bound_var_forward / cmcd_bound_var_grad_kept%s");
      return 0;
    }
    WsLayout lw;
    make_ws_lgcp(*desc, n, lw);
    return (align4(lgcp_workspace_floats(*desc, n, lw.total_floats)) + align4(lgcp_grad_workspace_floats(*desc, n)) +
            kept_traj_floats(*desc, n)) * 4;
  }
  WsLayout w;
  const int64_t nt = desc->target == CMCD_TARGET_MANY_GMM ? 1 + 2 * 64 : 0;
  if (!make_ws(*desc, n, nt, w)) return 0;
  if (desc->mode == CMCD_MODE_ULA) {
    if (!ula_grad_available(*desc)) { fail(CMCD_ERR_UNSUPPORTED, "no MCD_ULA gradient instance for this target%s"); return 0; }
    return (align4(w.total_floats) + align4(ula_grad_workspace_floats(*desc, n)) +
            (int64_t)(desc->nbridges + 1) * n * desc->dim) * 4;
  }
  if (desc->mode == CMCD_MODE_CAIS_UHA_SN) {
    if (!uha_grad_available(*desc, w.T)) { fail(CMCD_ERR_UNSUPPORTED, "no MCD_CAIS_UHA_sn gradient instance for this (target, dim, arch, width)%s"); return 0; }
    return (align4(w.total_floats) + align4(uha_grad_workspace_floats(*desc, w.HP, n)) + uha_traj_floats(*desc, n)) * 4;
  }
  if ((desc->mode != CMCD_MODE_CAIS_SN && desc->mode != CMCD_MODE_ULA_SN) || !bptt_available(*desc, w.T)) {
    fail(CMCD_ERR_UNSUPPORTED, "no reparameterised-gradient kernel instance for this (mode, target, dim, arch, width)%s");
    return 0;
  }
  return (align4(w.total_floats) + align4(grad_workspace_floats(*desc, w.HP, n)) +
          align4((int64_t)(desc->nbridges + 1) * n * desc->dim) +
          (grad_item_mode(*desc, w.T, n) ? bptt_item_floats(*desc, n) : 0)) * 4;
}

int cmcd_bound_grad(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                    const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                    float omega, void* workspace, int64_t workspace_bytes, float* out_loss, float* out_z,
                    double* out_stats, float* grad, void* stream_) {
  int rc = check_desc(desc);
  if (rc != CMCD_OK) return rc;
  if (!grad) return fail(CMCD_ERR_BAD_ARG, "null pointer argument%s");
  if (desc->mode != CMCD_MODE_CAIS_SN && desc->mode != CMCD_MODE_ULA_SN && desc->mode != CMCD_MODE_ULA &&
      desc->mode != CMCD_MODE_CAIS_UHA_SN)
    return fail(CMCD_ERR_UNSUPPORTED, "the reparameterised gradient exists for MCD_CAIS_sn, MCD_CAIS_UHA_sn and MCD_ULA[_sn] (MCD_CAIS_var_sn: cmcd_bound_var_grad)%s");
  // MCD_ULA_sn: the reference's dispatcher passes neither eps_schedule nor grad_clipping (mcd_utils.py:35-58)
  cmcd_desc dd = *desc;
  if (dd.mode == CMCD_MODE_ULA_SN || dd.mode == CMCD_MODE_ULA) { dd.eps_schedule = CMCD_EPS_CONST; dd.grad_clipping = 0; }
  const cmcd_desc& d = dd;
  if (d.target == CMCD_TARGET_LGCP) {
    // d = 1600: launch-sequence forward (trajectory kept) + launch-sequence reverse sweep (cmcd_lgcp.hip)
    WsLayout lw;
    make_ws_lgcp(d, n, lw);
    const int64_t fwd = align4(lgcp_workspace_floats(d, n, lw.total_floats));
    const int64_t gfl = align4(lgcp_grad_workspace_floats(d, n));
    const int64_t need = (fwd + gfl + kept_traj_floats(d, n)) * 4;
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
      return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
    float* ws = static_cast<float*>(workspace);
    float* traj = ws + fwd + gfl;
    rc = forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, fwd * 4, out_loss,
                      out_z, out_stats, traj, stream_);
    if (rc != CMCD_OK) return rc;
    rc = lgcp_grad(d, *lay, lw, n, params, n_params, target_consts, ws, traj, ws + fwd, omega, nullptr, true, grad, stream_);
    if (rc != CMCD_OK) return fail(rc, "lgcp gradient launch sequence failed%s");
    return CMCD_OK;
  }
  WsLayout w;
  if (!make_ws(d, n, n_target, w)) return fail(CMCD_ERR_BAD_ARG, "bad descriptor%s");
  if (d.mode == CMCD_MODE_ULA) {   // no network: forward with the trajectory kept + the network-free reverse sweep
    if (!ula_grad_available(d)) return fail(CMCD_ERR_UNSUPPORTED, "no MCD_ULA gradient instance for this target%s");
    const int64_t fwd = align4(w.total_floats), gfl = align4(ula_grad_workspace_floats(d, n));
    const int64_t need = (fwd + gfl + (int64_t)(d.nbridges + 1) * n * d.dim) * 4;
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
      return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
    float* ws = static_cast<float*>(workspace);
    float* traj = ws + fwd + gfl;
    rc = forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, fwd * 4, out_loss,
                      out_z, out_stats, traj, stream_);
    if (rc != CMCD_OK) return rc;
    rc = ula_grad_launch(d, *lay, w, n, params, n_params, ws, traj, ws + fwd, omega, grad, stream_);
    if (rc != CMCD_OK) return fail(rc, "gradient launch failed%s");
    return CMCD_OK;
  }
  if (d.mode == CMCD_MODE_CAIS_UHA_SN) {   // 2nd-order CMCD: forward with (z, rho, rho') kept + its reverse sweep (cmcd_uha.hip)
    if (!uha_grad_available(d, w.T)) return fail(CMCD_ERR_UNSUPPORTED, "no MCD_CAIS_UHA_sn gradient instance for this (target, dim, arch, width)%s");
    const int64_t fwd = align4(w.total_floats), gfl = align4(uha_grad_workspace_floats(d, w.HP, n));
    const int64_t need = (fwd + gfl + uha_traj_floats(d, n)) * 4;
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
      return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
    float* ws = static_cast<float*>(workspace);
    float* traj = ws + fwd + gfl;
    rc = forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, fwd * 4, out_loss,
                      out_z, out_stats, traj, stream_);
    if (rc != CMCD_OK) return rc;
    rc = uha_grad_launch(d, *lay, w, n, params, n_params, ws, traj, ws + fwd, omega, grad, stream_);
    if (rc != CMCD_OK) return fail(rc, "gradient launch failed%s");
    return CMCD_OK;
  }
  if (!bptt_available(d, w.T)) return fail(CMCD_ERR_UNSUPPORTED, "no reparameterised-gradient kernel instance for this (target, dim, arch, width)%s");
  const int64_t fwd = align4(w.total_floats), gfl = align4(grad_workspace_floats(d, w.HP, n));
  const int64_t tfl = align4((int64_t)(d.nbridges + 1) * n * d.dim);
  const bool item = grad_item_mode(d, w.T, n);
  const int64_t need = (fwd + gfl + tfl + (item ? bptt_item_floats(d, n) : 0)) * 4;
  if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
  float* ws = static_cast<float*>(workspace);
  float* traj = ws + fwd + gfl;
  rc = forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, fwd * 4, out_loss,
                    out_z, out_stats, traj, stream_);
  if (rc != CMCD_OK) return rc;
  rc = grad_launch(d, *lay, w, seeds, n, params, n_params, ws, nullptr, omega, true, item, traj,
                   item ? traj + tfl : nullptr, ws + fwd, grad, stream_);
  if (rc != CMCD_OK) return fail(rc, "gradient launch failed%s");
  return CMCD_OK;
}

int64_t cmcd_grad_workspace_bytes(const cmcd_desc* desc, int64_t n) {
  if (check_desc(desc) != CMCD_OK || n < 1) return 0;
  if (desc->target == CMCD_TARGET_LGCP) {   // launch-sequence forward + reverse sweep + the kept trajectory
    WsLayout lw;
    make_ws_lgcp(*desc, n, lw);
    return (align4(lgcp_workspace_floats(*desc, n, lw.total_floats)) + align4(lgcp_grad_workspace_floats(*desc, n)) +
            kept_traj_floats(*desc, n)) * 4;
  }
  WsLayout w;
  const int64_t nt = desc->target == CMCD_TARGET_MANY_GMM ? 1 + 2 * 64 : 0;
  if (!make_ws(*desc, n, nt, w)) return 0;
  if (!grad_available(*desc, w.T)) {
    fail(CMCD_ERR_UNSUPPORTED, "no gradient kernel instance for this (target, dim, arch, width)%s");
    return 0;
  }
  int64_t fl = align4(w.total_floats) + align4(grad_workspace_floats(*desc, w.HP, n));
  if (grad_item_mode(*desc, w.T, n))   // trajectory + scratch outputs of the internal forward pass
    fl += align4((int64_t)(desc->nbridges + 1) * n * desc->dim) + align4(n) + align4(n * desc->dim) + 16;
  return fl * 4;
}

int cmcd_vargrad_weights(const float* loss, const double* stats, int64_t n, int64_t n_total, float* omega,
                         void* stream_) {
  if (!loss || !stats || !omega || n < 1 || n_total < n) return fail(CMCD_ERR_BAD_ARG, "bad argument%s");
  hipLaunchKernelGGL(vargrad_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream_), loss, stats, n, n_total, omega);
  CMCD_HIP_CHECK(hipGetLastError());
  return CMCD_OK;
}

static int var_grad_impl(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                         const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                         const float* omega, void* workspace, int64_t workspace_bytes, float* grad, bool kept,
                         void* stream_) {
  int rc = check_desc(desc);
  if (rc != CMCD_OK) return rc;
  if (!lay || !seeds || !params || !omega || !workspace || !grad) return fail(CMCD_ERR_BAD_ARG, "null pointer argument%s");
  if (desc->mode != CMCD_MODE_CAIS_VAR_SN)
    return fail(CMCD_ERR_UNSUPPORTED, "the local (stop_gradient) gradient exists for MCD_CAIS_var_sn only%s");
  const cmcd_desc& d = *desc;
  if (d.target == CMCD_TARGET_LGCP) {
    // d = 1600: the reverse launch sequence of cmcd_lgcp.hip with z detached, on the trajectory cmcd_bound_var_forward left
    if (!kept)
      return fail(CMCD_ERR_UNSUPPORTED, "lgcp: call cmcd_bound_var_forward, then cmcd_bound_var_grad_kept on the same workspace%s");
    WsLayout lw;
    make_ws_lgcp(d, n, lw);
    const int64_t fwd = align4(lgcp_workspace_floats(d, n, lw.total_floats));
    const int64_t gfl = align4(lgcp_grad_workspace_floats(d, n));
    const int64_t need = (fwd + gfl + (int64_t)(d.nbridges + 1) * n * d.dim) * 4;
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
      return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
    float* ws = static_cast<float*>(workspace);
    rc = lgcp_grad(d, *lay, lw, n, params, n_params, target_consts, ws, ws + fwd + gfl, ws + fwd, 0.f, omega, false, grad,
                   stream_);
    if (rc != CMCD_OK) return fail(rc, "lgcp gradient launch sequence failed%s");
    return CMCD_OK;
  }
  int n_mix = 0;
  if (d.target == CMCD_TARGET_MANY_GMM) {
    if (!target_consts || n_target < 3 || (n_target - 1) % 2 != 0 || (n_target - 1) / 2 > 64)
      return fail(CMCD_ERR_BAD_ARG, "many_gmm needs target_consts = {scale, means[n_mixes<=64][2]}%s");
    n_mix = int((n_target - 1) / 2);
  }
  WsLayout w;
  if (!make_ws(d, n, n_target, w)) return fail(CMCD_ERR_BAD_ARG, "bad descriptor%s");
  if (!grad_available(d, w.T)) return fail(CMCD_ERR_UNSUPPORTED, "no gradient kernel instance for this (target, dim, arch, width)%s");
  const int64_t need = cmcd_grad_workspace_bytes(desc, n);
  if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* ws = static_cast<float*>(workspace);
  const int64_t fwd = align4(w.total_floats), gfl = align4(grad_workspace_floats(d, w.HP, n));
  const bool item = grad_item_mode(d, w.T, n);
  float* traj = nullptr;
  if (item && kept) {
    traj = ws + fwd + gfl;   // left there, with the prep tables, by cmcd_bound_var_forward
  } else if (item) {
    // the work-item path reads the trajectory: run the forward launch sequence once more, keeping z_0..z_K
    traj = ws + fwd + gfl;
    float* sl = traj + align4((int64_t)(d.nbridges + 1) * n * d.dim);
    float* sz = sl + align4(n);
    double* sst = reinterpret_cast<double*>(sz + align4(n * d.dim));
    rc = forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, fwd * 4, sl, sz, sst,
                      traj, stream_);
    if (rc != CMCD_OK) return rc;
  } else if (!kept) {
    launch_prep(d, *lay, w, params, target_consts, n_mix, ws, stream, tables_stamp(d, *lay, n, n_params, n_target));
  }
  rc = grad_launch(d, *lay, w, seeds, n, params, n_params, ws, omega, 0.f, false, item, traj, nullptr, ws + fwd, grad,
                   stream_);
  if (rc != CMCD_OK) return fail(rc, "gradient launch failed%s");
  return CMCD_OK;
}

int cmcd_bound_var_grad(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                        const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                        const float* omega, void* workspace, int64_t workspace_bytes, float* grad, void* stream_) {
  return var_grad_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, omega, workspace,
                       workspace_bytes, grad, false, stream_);
}

int cmcd_bound_var_forward(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                           const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                           void* workspace, int64_t workspace_bytes, float* out_loss, float* out_z,
                           double* out_stats, void* stream_) {
  int rc = check_desc(desc);
  if (rc != CMCD_OK) return rc;
  if (desc->mode != CMCD_MODE_CAIS_VAR_SN)
    return fail(CMCD_ERR_UNSUPPORTED, "the local (stop_gradient) gradient exists for MCD_CAIS_var_sn only%s");
  const int64_t need = cmcd_grad_workspace_bytes(desc, n);
  if (need <= 0) return CMCD_ERR_UNSUPPORTED;
  if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return fail(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned (need %s%lld bytes)", "", need);
  if (desc->target == CMCD_TARGET_LGCP) {
    WsLayout lw;
    make_ws_lgcp(*desc, n, lw);
    const int64_t fwd = align4(lgcp_workspace_floats(*desc, n, lw.total_floats));
    const int64_t gfl = align4(lgcp_grad_workspace_floats(*desc, n));
    return forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, fwd * 4, out_loss,
                        out_z, out_stats, static_cast<float*>(workspace) + fwd + gfl, stream_);
  }
  WsLayout w;
  if (!make_ws(*desc, n, n_target, w)) return fail(CMCD_ERR_BAD_ARG, "bad descriptor%s");
  const int64_t fwd = align4(w.total_floats), gfl = align4(grad_workspace_floats(*desc, w.HP, n));
  float* traj = grad_item_mode(*desc, w.T, n) ? static_cast<float*>(workspace) + fwd + gfl : nullptr;
  return forward_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, workspace, fwd * 4, out_loss,
                      out_z, out_stats, traj, stream_);
}

int cmcd_bound_var_grad_kept(const cmcd_desc* desc, const cmcd_layout* lay, const int32_t* seeds, int64_t n,
                             const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                             const float* omega, void* workspace, int64_t workspace_bytes, float* grad,
                             void* stream_) {
  return var_grad_impl(desc, lay, seeds, n, params, n_params, target_consts, n_target, omega, workspace,
                       workspace_bytes, grad, true, stream_);
}

int cmcd_stats_merge_device(const double* rows, int32_t count, double* out5, void* stream_) {
  if (!rows || !out5 || count < 1) return fail(CMCD_ERR_BAD_ARG, "null pointer argument%s");
  hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream_), rows, count, out5,
                     (const uint32_t*)nullptr, 0u);
  CMCD_HIP_CHECK(hipGetLastError());
  return CMCD_OK;
}

#ifndef CMCD_NO_DIAG_HOOKS
int cmcd_debug_grad_item(int mode) {
  if (mode < -1 || mode > 1) return fail(CMCD_ERR_BAD_ARG, "mode must be -1, 0 or 1%s");
  set_grad_item_override(mode);
  return CMCD_OK;
}

int cmcd_profile_enable(int on) {
  g_prof.on = on != 0;
  g_prof.used = 0;
  // the first 512 event pairs are created here, outside any timed region (a 20-step measurement would otherwise pay two
  // hipEventCreate calls inside every one of its steps)
  for (; on && g_prof.created < 512; ++g_prof.created) {
    CMCD_HIP_CHECK(hipEventCreate(&g_prof.ev[g_prof.created][0]));
    CMCD_HIP_CHECK(hipEventCreate(&g_prof.ev[g_prof.created][1]));
  }
  return CMCD_OK;
}

int cmcd_profile_collect(double* total_ms, int64_t* launches) {
  if (!total_ms || !launches) return fail(CMCD_ERR_BAD_ARG, "null pointer argument%s");
  double tot = 0.0;
  for (int i = 0; i < g_prof.used; ++i) {
    float ms = 0.f;
    CMCD_HIP_CHECK(hipEventSynchronize(g_prof.ev[i][1]));
    CMCD_HIP_CHECK(hipEventElapsedTime(&ms, g_prof.ev[i][0], g_prof.ev[i][1]));
    tot += ms;
  }
  *total_ms = tot;
  *launches = g_prof.used;
  g_prof.used = 0;
  return CMCD_OK;
}
#endif   // CMCD_NO_DIAG_HOOKS

int cmcd_stats_merge(const double* stats, const int64_t* n_per, int32_t count, double* merged5, double* out3) {
  if (!stats || !n_per || count < 1 || !merged5 || !out3) return fail(CMCD_ERR_BAD_ARG, "null pointer argument%s");
  double acc[CMCD_NSTATS] = {0, 0, 0, -INFINITY, 0};
  int64_t n = 0;
  for (int i = 0; i < count; ++i) {
    const double* b = stats + (int64_t)i * CMCD_NSTATS;
    acc[0] += b[0]; acc[1] += b[1]; acc[2] += b[2];
    const double m = fmax(acc[3], b[3]);
    const double sa = (acc[3] > -INFINITY && m < INFINITY) ? acc[4] * exp(acc[3] - m) : (acc[3] == m ? acc[4] : 0.0);
    const double sb = (b[3] > -INFINITY && m < INFINITY) ? b[4] * exp(b[3] - m) : (b[3] == m ? b[4] : 0.0);
    acc[3] = m; acc[4] = sa + sb;
    n += n_per[i];
  }
  if (n < 1) return fail(CMCD_ERR_BAD_ARG, "no particles%s");
  memcpy(merged5, acc, sizeof(acc));
  const double mean = acc[1] / (double)n;
  out3[0] = mean;
  out3[1] = acc[2] / (double)n - mean * mean;        // var(ddof=0); inf - inf = NaN like the reference
  out3[2] = acc[3] + log(acc[4]) - log((double)n);   // logsumexp(-l) - log n
  return CMCD_OK;
}

}  // extern "C"
