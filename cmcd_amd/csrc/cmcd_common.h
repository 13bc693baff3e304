// Structs shared by the translation units of libcmcd_hip.so.
#pragma once
#include <stdint.h>

#include "cmcd_hip.h"

namespace cmcd {

// ------------------------------------------------------------------------------------------
// workspace carve-up (floats unless noted)
// ------------------------------------------------------------------------------------------
struct WsLayout {
  int64_t beta, eps, sig, logsig;  // [K] each
  int64_t sched;                   // [K][8] {beta, eps, sig, logsig + log sqrt(2pi), 1/(2 sig^2), eps*beta, eps*(1-beta), 0}
  int64_t bias1;                   // [K+1][HP]
  int64_t utab;                    // [K+1][HP]   (geffner only, else aliases bias1)
  int64_t w1z;                     // [D][HP]
  int64_t w2;                      // [T][T][64][4]
  int64_t w2t;                     // [T][T][64][4]  A fragments of W2^T (gradient kernel)
  int64_t w2q;                     // [T][HP/2][64]  A operands of the 4x4x1 MFMA (cooperative kernel, 8-particle tiles)
  int64_t b2;                      // [HP]
  int64_t w3t;                     // [D][HP]
  int64_t b3;                      // [16]  (b3[D] then factor)
  int64_t tgt;                     // target constants staged for LDS
  int64_t tgt_floats;
  int64_t partials;                // doubles: [n_waves][5]  (offset in floats, 8-byte aligned)
  int64_t total_floats;
  int32_t HP, T, n_waves;
};


// arguments of the trajectory kernels (both variants)
struct TrajArgs {
  const int32_t* seeds;
  const float* params;
  const float* ws;
  double* partials;
  float* out_loss;
  float* out_z;
  cmcd_layout lay;
  WsLayout w;
  int64_t n;
  int32_t K, var_mode, grad_clipping;
  float* traj;  // optional [K+1][n][D] trajectory z_0..z_K (the reparameterised gradient's reverse sweep reads it)
  int32_t ula;  // 0: CAIS; 1: MCD_ULA (no network); 2: MCD_ULA_sn (network in the backward kernel only, index i)
  int32_t prio = 0;  // cooperative kernel: s_setprio level per role, 2 bits each {MLP, TGT, RNG, ACC} from bit 0
  int32_t tail = 0;  // cooperative kernel, 8-particle tiles of the 9-tile net: the ninth MLP wave runs the 4-neuron form (coop_tail4)
  // cooperative kernels: statistics merged by the LAST workgroup to arrive (no finalize launch): out[5] and an int32 arrival
  // counter the prep launch zeroes on every call (WsLayout::b3 slot 14).  Both null: the caller launches finalize_kernel.
  double* fin_out = nullptr;
  int32_t* fin_counter = nullptr;
  // cmcd_bound_forward_prepared with the fused merge: where the forming call left its stamp, and this call's (finalize_kernel)
  const uint32_t* stamp_slot = nullptr;
  uint32_t stamp_expect = 0;
  // cmcd_debug_capture_noise (tests): the PRNG path of THIS launch, written next to the arithmetic that consumes it.
  // Stage 0 = the draw of z_0, stage i + 1 = bridge i.  All nullable.
  uint32_t* dbg_bits = nullptr;   // [K+1][n][D]  the random words that become the deviates (jax random_bits)
  uint32_t* dbg_keys = nullptr;   // [K+1][n][2]  gen_0 .. gen_K: the chain key entering bridge i (mcd_cais.py:66,87,94)
  float* dbg_noise = nullptr;     // [K+1][n][D]  the deviates (jax.random.normal)
};

// The 132-wide net (BASELINE configuration 4) is 8 neuron tiles + 4 neurons: on 8-particle tiles its ninth MLP wave
// contracts those 4 neurons over EIGHT slices of the 144 inputs (20 matrix instructions) instead of carrying 12 zero
// neurons through 72.  One rule for the prep launch (operand packing of that wave) and the launch (TrajArgs::tail).
__host__ __device__ inline bool coop_tail4(int T, int real_width) {
#ifdef CMCD_COOP_NO_TAIL   // A / B builds (tools/probes/t9_tail_ab.sh)
  return false;
#else
  return T == 9 && real_width > 128 && real_width <= 132;
#endif
}
// slice s of the tail wave's contraction: inputs [coop_tail_start(s), + coop_tail_len(s)), 16-byte aligned starts
__host__ __device__ inline int coop_tail_start(int s) { return s < 4 ? 20 * s : 80 + 16 * (s - 4); }
__host__ __device__ inline int coop_tail_len(int s) { return s < 4 ? 20 : 16; }

// cmcd_coop.hip: the CU-cooperative variant (one workgroup per 16-particle tile).
// Returns nullptr-equivalent (false) when no instance exists for this (target, arch, dim, T).
bool coop_available(const cmcd_desc& d, int T);
bool coop_half_available(const cmcd_desc& d, int T);   // the 8-particle-tile instances (batches of <= 2048 particles)
// narrow: keep a wide state (d > 8) on coop_kernel's 8-particle instance instead of coop_wide8_kernel (kernel variant 5: A / B, tests)
int coop_launch(const cmcd_desc& d, const TrajArgs& ta, bool half, void* stream, bool narrow = false);
// cmcd_coop_wide.hip: 8-particle tiles for wide states (the funnel, d = 10): every per-coordinate job dealt to the lanes of
// its particle.  lds_claim_min: dynamic LDS to claim at least (the CU-exclusive claim of coop_launch), 0 = none.
bool coop_wide8_available(const cmcd_desc& d, int T);
int coop_wide8_launch(const cmcd_desc& d, const TrajArgs& ta, size_t lds_claim_min, void* stream);


// cmcd_uha.hip: MCD_CAIS_UHA_sn (2nd-order CMCD) on the wave-per-tile mapping.  The kept trajectory is
// [3 K + 2][n][dim]: z_0..z_K, rho_0..rho_K, rho'_0..rho'_{K-1}.
int net_in_dim(const cmcd_desc& d);                    // dim, or 2 dim when the network takes concat(z, rho)
bool uha_available(const cmcd_desc& d, int T);
int64_t uha_traj_floats(const cmcd_desc& d, int64_t n);
int uha_forward_launch(const cmcd_desc& d, const TrajArgs& ta, void* stream, int* n_records);   // n_records: statistics records written
const char* uha_last_kernel_name();                    // "uha_traj_kernel" | "uha_coop_kernel<N-particle tiles>" (this host thread's last launch)
// reparameterised gradient: reverse sweep over the kept trajectory
bool uha_grad_available(const cmcd_desc& d, int T);
int64_t uha_grad_workspace_floats(const cmcd_desc& d, int HP, int64_t n);
int uha_grad_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, int64_t n, const float* params,
                    int64_t n_params, const float* ws_fwd, const float* traj, float* gws, float omega, float* grad,
                    void* stream);
// cmcd_grad.hip: schedule + network tails for a network with `din` state inputs
int launch_net_tails(const cmcd_desc& d, int din, int eps_schedule, const cmcd_layout& lay, const WsLayout& w,
                     const float* params, const float* gtab, int64_t o_S, int64_t o_S2, int64_t o_gbeta, int64_t o_geps,
                     int HP, float* dds_tail, float* grad, void* stream);

// cmcd_lgcp.hip: the d = 1600 path (per-bridge launch sequence)
int64_t lgcp_workspace_floats(const cmcd_desc& d, int64_t n, int64_t base);
// tables_ready (cmcd_bound_forward_prepared): the workspace still holds the first-layer bias table and the packed weight copies
// of the same parameters — those launches are skipped (the per-call zeroing of the operands is not)
int lgcp_forward(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const int32_t* seeds, int64_t n,
                 const float* params, const float* tc, float* ws, float* out_loss, float* out_z,
                 double** partials_out, float* traj, void* stream, bool tables_ready = false, float* keep_gws = nullptr);
// cmcd_lgcp_wide.hip: forward-only calls on wide batches (>= kLgcpWideMin particles): whole-batch launches of a real fp32
// GEMM body (32 x 128 tiles over the whole contraction, no split-K seam) instead of 32-row weight-streaming passes
constexpr int64_t kLgcpWideMin = 225;   // measured crossover against the 32-row passes on four lanes (K = 128, profiles/r05_h_lgcp_wide_valu_trims.txt: seven passes 11.2 ms against 12.3 ms at 224, eight passes 15.7 against 12.3 at 256)
bool lgcp_wide_supported(const cmcd_desc& d);
bool lgcp_use_wide(const cmcd_desc& d, int64_t n, bool keeps_trajectory);    // the one selection rule (workspace query + launch)
int64_t lgcp_wide_workspace_floats(const cmcd_desc& d, int64_t n, int64_t base);
int lgcp_wide_forward(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, const int32_t* seeds, int64_t n,
                      const float* params, const float* tc, float* ws, float* out_loss, float* out_z,
                      double** partials_out, void* stream, bool tables_ready = false);
// per-bridge first-layer bias table b1 + emb[min(i, K-1)] W1[d:, :] -> bias1[K+1][IN] (cmcd_lgcp.hip's prep launch)
int lgcp_launch_prep(const cmcd_desc& d, const cmcd_layout& lay, const float* params, float* bias1, void* stream);
// reverse sweep of the reparameterised gradient on the d = 1600 path (cmcd_lgcp.hip); traj as left by lgcp_forward
int64_t lgcp_grad_workspace_floats(const cmcd_desc& d, int64_t n);
int lgcp_grad(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& sw, int64_t n, const float* params,
              int64_t n_params, const float* tc, float* ws, const float* traj, float* gws, float omega,
              const float* omega_vec, bool bptt, float* grad, void* stream);
// cmcd_grad.hip: MCD_ULA (no network) reverse sweep
bool ula_grad_available(const cmcd_desc& d);
int64_t ula_grad_workspace_floats(const cmcd_desc& d, int64_t n);
int ula_grad_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, int64_t n, const float* params,
                    int64_t n_params, const float* ws_fwd, const float* traj, float* gws, float omega, float* grad,
                    void* stream);
// cmcd_grad.hip: the particle-independent tails of a geffner net's gradient from the S / S2 / beta / eps tables
int launch_geffner_tails(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, const float* params,
                         const float* gtab, int64_t o_S, int64_t o_S2, int64_t o_gbeta, int64_t o_geps, int HP,
                         float* grad, void* stream, bool with_net = true);

// mean-field VI on lgcp (cmcd_lgcp.hip) and the statistics merge launcher (cmcd_kernels.hip), used by cmcd_mfvi.hip
int64_t lgcp_mfvi_workspace_floats(int D, int64_t n, bool with_grad);
int lgcp_mfvi(int D, int64_t o_mean, int64_t o_logdiag, const int32_t* seeds, int64_t n, const float* params,
              const float* tc, float* ws, float* out_loss, float* out_z, double** partials_out, float** gbuf_out,
              bool with_grad, void* stream);
int launch_finalize(const double* partials, int32_t count, double* out5, void* stream);
int fail_msg(int code, const char* msg);

// cmcd_grad.hip: VarGrad gradient (widths <= 64)
bool grad_available(const cmcd_desc& d, int T);
int64_t grad_workspace_floats(const cmcd_desc& d, int HP, int64_t n);
// omega: per-particle weights (VarGrad), or nullptr with omega_scalar.  bptt: reparameterised gradient (reverse
// sweep over the stored trajectory `traj`) vs the local gradient.  item: the work-item path for small batches
// (needs `traj`; with bptt also `item_ws` of bptt_item_floats floats).
int grad_launch(const cmcd_desc& d, const cmcd_layout& lay, const WsLayout& w, const int32_t* seeds, int64_t n,
                const float* params, int64_t n_params, const float* ws_fwd, const float* omega, float omega_scalar,
                bool bptt, bool item, const float* traj, float* item_ws, float* gws, float* grad, void* stream);
bool bptt_available(const cmcd_desc& d, int T);
bool grad_item_mode(const cmcd_desc& d, int T, int64_t n);
int get_grad_item_override();
void set_grad_item_override(int v);   // -1: measured rule; 0 / 1: whole chains / work items (process-wide; diagnostics)
int64_t bptt_item_floats(const cmcd_desc& d, int64_t n);

}  // namespace cmcd
