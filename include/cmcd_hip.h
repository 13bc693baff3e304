/*
 * cmcd_hip.h — C ABI of libcmcd_hip.so: the MI355X (gfx950) implementation of CMCD's
 * annealed-Langevin bound evaluation (`MCD_CAIS_sn` / `MCD_CAIS_var_sn`).
 *
 * The reference is pure Python/JAX and has no FFI; the "interface each entry point
 * replaces" is therefore the jitted Python callable the reference builds around the path:
 *
 *   cmcd_bound_forward   <->  loss_fn = jax.jit(partial(mcdbm.compute_bound[_var], eps_schedule=…,
 *                             grad_clipping=…), static_argnums=(2,3,4))
 *                             /root/reference/src/main.py:161-177, called positionally as
 *                             f(seeds, params_flat, unflatten, params_fixed, log_prob_model) at
 *                             /root/reference/src/opt.py:97-99,102-104,188-190; body
 *                             /root/reference/src/mcdboundingmachine.py:126-231 ->
 *                             /root/reference/src/mcd_utils.py:134-161 ->
 *                             /root/reference/src/mcd_cais.py:6-99 / mcd_cais_var.py:7-112.
 *   cmcd_desc            <->  params_fixed = (dim, nbridges, mode, apply_fun_sn)
 *                             (/root/reference/src/mcdboundingmachine.py:121) + the static
 *                             eps_schedule / grad_clipping partial args (main.py:161-172) +
 *                             the config.model routing of load_model
 *                             (/root/reference/src/model_handler.py:30-43).
 *   cmcd_layout          <->  `unflatten` of jax.flatten_util.ravel_pytree
 *                             (/root/reference/src/mcdboundingmachine.py:122,141-143): where each
 *                             leaf of (params_train, params_notrain) sits inside params_flat.
 *   cmcd_stats_merge     <->  batch_log_elbos.mean() / .var(ddof=0) (mcdboundingmachine.py:205,231)
 *                             and logsumexp(-loss) - log n (/root/reference/src/utils.py:233-235),
 *                             in a form that merges across ranks.
 *
 * Ownership: every pointer marked [device] is caller-owned device memory (e.g. the data_ptr() of
 * a contiguous torch tensor).  The library allocates no device memory and keeps no state that a
 * result depends on: what persists between calls is read-only or per host thread — cached device
 * attributes (CU count, LDS opt-in, written once), the diagnostic wave-priority override of
 * cmcd_coop.hip (read from the environment once), the thread-local error string and profile hook,
 * and the lgcp gradient's thread-local side stream and events.  Calls are re-entrant as long as each
 * concurrent call has its own `workspace` (two calls sharing a workspace must be ordered on one
 * stream).  All work is enqueued asynchronously on `stream` (a hipStream_t, may be 0); nothing in
 * cmcd_bound_forward synchronises, so it can be captured into a hipGraph.
 * Errors: 0 on success, negative cmcd_status otherwise; message via cmcd_last_error()
 * (thread-local).  Nothing throws across this boundary.
 */
#ifndef CMCD_HIP_H
#define CMCD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMCD_ABI_VERSION 3   /* 2: cmcd_adam_step[_dev] take the divergence guard (losses, n_losses, diverged)
                                3: CMCD_MODE_CAIS_UHA_SN, cmcd_layout.gamma */

typedef enum cmcd_status {
  CMCD_OK = 0,
  CMCD_ERR_BAD_ARG = -1,       /* null pointer, bad size, layout offset missing          */
  CMCD_ERR_UNSUPPORTED = -2,   /* mode / arch / target / shape not implemented           */
  CMCD_ERR_WORKSPACE = -3,     /* workspace smaller than cmcd_workspace_bytes()          */
  CMCD_ERR_HIP = -4            /* a HIP runtime call failed                               */
} cmcd_status;

/* config.boundmode plugin switch (/root/reference/src/mcd_utils.py:34-190).  The two CAIS modes and
 * the two overdamped siblings on the same skeleton (MCD_ULA = Thin et al., MCD_ULA_sn = Doucet et al.,
 * /root/reference/src/mcd_over_orig.py:6-65) exist here; every other value is CMCD_ERR_UNSUPPORTED
 * ("Mode not implemented.").  MCD_ULA has no network: pass arch = CMCD_ARCH_DDS and network layout
 * offsets of -1; eps_schedule / grad_clipping are ignored for both ULA modes, as in the reference. */
enum { CMCD_MODE_CAIS_SN = 0, CMCD_MODE_CAIS_VAR_SN = 1, CMCD_MODE_ULA = 2, CMCD_MODE_ULA_SN = 3,
       /* 2nd-order CMCD (/root/reference/src/mcd_under_lp_a_cais.py:6-115 via mcd_utils.py:174-188): state (z, rho), the
        * score network takes concat(z, rho) (built with rho_dim = dim, mcdboundingmachine.py:82-98: geffner width
        * 2 dim + emb_dim, dds first layer [2 dim + 64, 64]), eta_aux = gamma * eps, one leap-frog step per bridge.  The
        * function body fixes the cos^2 step-size schedule and the 1e2 clip of grad log p: eps_schedule / grad_clipping
        * of the descriptor are ignored for this mode. */
       CMCD_MODE_CAIS_UHA_SN = 4 };
/* config.nn_arch (/root/reference/src/nn.py:21-39) */
enum { CMCD_ARCH_GEFFNER = 0, CMCD_ARCH_DDS = 1 };
/* config.model routing (/root/reference/src/model_handler.py:30-43) */
enum { CMCD_TARGET_GMM = 0, CMCD_TARGET_FUNNEL = 1, CMCD_TARGET_MANY_GMM = 2, CMCD_TARGET_LGCP = 3 };
/* config.eps_schedule (/root/reference/src/mcd_cais.py:54-59) */
enum { CMCD_EPS_CONST = 0, CMCD_EPS_LINEAR = 1, CMCD_EPS_COS_SQ = 2 };

typedef struct cmcd_desc {
  int32_t dim;            /* params_fixed[0]                                              */
  int32_t nbridges;       /* params_fixed[1]; >= 1                                        */
  int32_t mode;           /* params_fixed[2] as CMCD_MODE_*                               */
  int32_t arch;           /* which apply_fun_sn: CMCD_ARCH_*                              */
  int32_t emb_dim;        /* geffner: config.emb_dim (net width = dim + emb_dim); dds: 64 */
  int32_t target;         /* CMCD_TARGET_*                                                */
  int32_t eps_schedule;   /* CMCD_EPS_*                                                   */
  int32_t grad_clipping;  /* 0 / 1                                                        */
  int32_t ngrid;          /* len(mgridref_y) - 1  (mcdboundingmachine.py:107-112)         */
  int32_t reserved;       /* kernel variant (testing): 0 auto, 1 wave-per-tile, 2 CU-cooperative, 3 / 4 cooperative on 16- / 8-particle
                             tiles, 5 = 4 with a wide state (d = 10) kept on the narrow-state kernel (A / B); 2nd-order mode: 4 without the
                             4-neuron tail on the last MLP wave (A / B) */
} cmcd_desc;

/* Offsets (in floats) of each leaf inside params_flat; -1 = absent for this arch.
 * Dense weights are [in, out] row-major (stax Dense / haiku Linear: x @ W + b). */
typedef struct cmcd_layout {
  int64_t vd_mean, vd_logdiag;           /* [dim] each  (vardist/diag_gauss.py:6-7)       */
  int64_t eps;                           /* scalar                                        */
  int64_t mgridref_y;                    /* [ngrid+1]                                     */
  int64_t gamma;                         /* scalar friction (MCD_CAIS_UHA_sn: eta_aux = gamma * eps,
                                            /root/reference/src/mcd_under_lp_a_cais.py:50); -1 otherwise   */
  /* geffner (/root/reference/src/nn.py:42-72), in = dim + emb_dim */
  int64_t g_emb;                         /* [nbridges, emb_dim]                           */
  int64_t g_factor;                      /* scalar factor_sn                              */
  int64_t g_w1, g_b1, g_w2, g_b2;        /* [in,in],[in] x2                               */
  int64_t g_w3, g_b3;                    /* [in,dim],[dim]                                */
  /* dds PISNet (/root/reference/src/nn_dds.py:91-164), H = 64 */
  int64_t d_phase;                       /* timestep_phase [64]                           */
  int64_t d_tw1, d_tb1, d_tw2, d_tb2;    /* time coder [128,64],[64],[64,64],[64]         */
  int64_t d_sw1, d_sb1;                  /* [dim+64,64],[64]                              */
  int64_t d_sw2, d_sb2;                  /* [64,64],[64]                                  */
  int64_t d_sw3, d_sb3;                  /* LinearZero [64,dim],[dim]                     */
} cmcd_layout;

/* out_stats: 5 doubles, the mergeable batch statistics of the per-particle losses l_n:
 *   [0] number of finite l_n   [1] sum l_n   [2] sum l_n^2
 *   [3] M = max_n(-l_n)        [4] sum_n exp(-l_n - M)
 * +inf losses are legal (many_gmm floor, /root/reference/src/model_handler.py:279-280) and make
 * [1],[2] +inf exactly like the reference's mean/var; NaN anywhere means "diverged"
 * (/root/reference/src/opt.py:122). */
#define CMCD_NSTATS 5

int cmcd_version(void);
const char* cmcd_last_error(void);

/* Bytes of [device] scratch cmcd_bound_forward needs for n particles (0 on bad desc). */
int64_t cmcd_workspace_bytes(const cmcd_desc* desc, int64_t n);

/* Number of floats of target constants expected for desc->target:
 *   gmm: 0; funnel: 0; many_gmm: 1 + 2*n_mixes = {scale, means[n_mixes,2]} (n_target tells
 *   n_mixes); lgcp: dim*dim + dim + 3 = {Kinv[dim,dim], counts[dim], mu0, a, lognorm}. */
int64_t cmcd_target_floats(const cmcd_desc* desc, int32_t n_mixes);

/* One forward evaluation of the bound for n particles:
 *   seeds[n] int32 [device]  ->  out_loss[n] = -w_n, out_z[n*dim] = z_K  [device, float32],
 *   out_stats[5] [device, float64].  params[n_params] float32 [device] is params_flat.  */
int cmcd_bound_forward(const cmcd_desc* desc, const cmcd_layout* layout,
                       const int32_t* seeds, int64_t n,
                       const float* params, int64_t n_params,
                       const float* target_consts, int64_t n_target,
                       void* workspace, int64_t workspace_bytes,
                       float* out_loss, float* out_z, double* out_stats,
                       void* stream);

/* Evaluation loops on FIXED parameters (the reference's `opt.sample` calls its jitted loss_fn n_input_dist_seeds = 30 times
 * with the same params_flat, /root/reference/src/opt.py:185-190; XLA keeps nothing between those calls either, but its
 * executable has no per-call table-building stage): same arguments and results as cmcd_bound_forward, but the per-call PREP
 * launch (the beta / eps schedule, the time path of the score network folded into per-bridge first-layer biases, the
 * operand packing of the weights: everything that depends on params_flat and not on the particles) is skipped.
 * CONTRACT: `workspace` was last used by a cmcd_bound_forward / cmcd_bound_forward_prepared call with the SAME desc, layout,
 * n, params CONTENTS and target constants, and nothing wrote into it since.  The library cannot check this (it keeps no
 * state and never reads params back to the host): the caller owns the invalidation — the Python mirror takes this entry point
 * only inside an explicit `fixed_parameters()` context, where it keys the tables on (workspace, the parameter tensor OBJECT,
 * its data_ptr() and version counter, target constants, desc, layout, n) and bumps the version counter wherever it updates
 * parameters through a raw pointer (cmcd_adam_step).  On the d = 1600 (lgcp) launch sequences the prepared
 * form skips the schedule / bias-table launches and the re-packing of the weights (the per-call zeroing of the operand
 * buffers stays); the 2nd-order lgcp sequence ignores the hint and prepares every call. */
int cmcd_bound_forward_prepared(const cmcd_desc* desc, const cmcd_layout* lay,
                                const int32_t* seeds, int64_t n,
                                const float* params, int64_t n_params,
                                const float* target_consts, int64_t n_target,
                                void* workspace, int64_t workspace_bytes,
                                float* out_loss, float* out_z, double* out_stats,
                                void* stream);

/* (Measurement and diagnostic hooks — kernel-time events, the PRNG capture of the parity tests, the probes' switches — are NOT
 * part of this boundary: they are declared in include/cmcd_hip_diag.h and compiled out of the library by
 * -DCMCD_NO_DIAG_HOOKS; a deployment binds nothing of them.) */

/* ---- VarGrad gradient ("compute_log_var_grad"): d/d params_flat of compute_bound_var
 * (/root/reference/src/main.py:161-176 takes jax.grad of it; /root/reference/src/mcd_cais_var.py:59,79
 * detach z, which makes the gradient local per bridge).  Two calls after a cmcd_bound_forward on the
 * same seeds / params:
 *   cmcd_vargrad_weights  omega[n] = d var / d w_n = -(2/N)(l_n - mean l) from loss[n] and the (merged,
 *                         for multi-GPU) statistics; n_total = global particle count.  All zero while
 *                         |var| > 1e7: the bound is clip(var, +-1e7) (mcdboundingmachine.py:231).
 *   cmcd_bound_var_grad   grad[n_params] (overwritten; zeros for leaves without gradient) =
 *                         sum_n omega_n d w_n / d params_flat.  Across ranks: all-reduce(sum) of grad.
 * MCD_CAIS_var_sn only; kernel instances exist for the BASELINE nets (dds 64; geffner widths up to 144 on the
 * 2-d targets, 64 on funnel) and lgcp (geffner, any width: the launch-sequence reverse sweep with z detached —
 * through cmcd_bound_var_forward + cmcd_bound_var_grad_kept only); CMCD_ERR_UNSUPPORTED otherwise. */
int64_t cmcd_grad_workspace_bytes(const cmcd_desc* desc, int64_t n);
int cmcd_vargrad_weights(const float* loss, const double* stats, int64_t n, int64_t n_total, float* omega,
                         void* stream);
int cmcd_bound_var_grad(const cmcd_desc* desc, const cmcd_layout* layout, const int32_t* seeds, int64_t n,
                        const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                        const float* omega, void* workspace, int64_t workspace_bytes, float* grad, void* stream);
/* The same two steps without running the chain twice: cmcd_bound_var_forward is cmcd_bound_forward on the
 * GRADIENT workspace (cmcd_grad_workspace_bytes) and leaves the per-call tables — and, for batches small
 * enough for the work-item gradient path, the trajectory z_0..z_K (lgcp: also every evaluation's activations, which the
 * reverse launch sequence then reads instead of recomputing them) — there; cmcd_bound_var_grad_kept then
 * needs the same desc / seeds / params / workspace, untouched in between (weights from
 * cmcd_vargrad_weights as before). */
int cmcd_bound_var_forward(const cmcd_desc* desc, const cmcd_layout* layout, const int32_t* seeds, int64_t n,
                           const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                           void* workspace, int64_t workspace_bytes, float* out_loss, float* out_z,
                           double* out_stats, void* stream);
int cmcd_bound_var_grad_kept(const cmcd_desc* desc, const cmcd_layout* layout, const int32_t* seeds, int64_t n,
                             const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                             const float* omega, void* workspace, int64_t workspace_bytes, float* grad, void* stream);

/* ---- Reparameterised gradient of the mean bound: what jax.value_and_grad(compute_bound, 1, has_aux=True)
 * returns for MCD_CAIS_sn (/root/reference/src/main.py:174-176 over mcdboundingmachine.py:183-205 and
 * mcd_cais.py:46-89 — no stop_gradient, so the gradient flows back through every z_i: the net's input
 * Jacobian, the Hessians of log p and log q, and the step-size / beta schedules).
 * One call = forward (losses, z_K, statistics as cmcd_bound_forward; the trajectory z_0..z_K is kept in
 * the workspace) + reverse sweep.  grad[n_params] (overwritten) = omega * sum_n d loss_n / d params_flat;
 * omega = d value / d loss_n = 1 / N_total (across ranks: all-reduce(sum) of grad).
 * Also MCD_ULA_sn and MCD_ULA (/root/reference/src/mcd_over_orig.py), on every target.
 * Targets gmm / funnel / many_gmm with the BASELINE nets (dds 64; geffner widths up to 144 on the 2-d targets,
 * 64 on funnel), and lgcp (geffner, any width: launch-sequence reverse sweep); CMCD_ERR_UNSUPPORTED otherwise.
 * Repeated calls with the same arguments return the same bits (r04): every sum over particles is taken in a fixed order
 * — per-tile slots + a reduction launch; the workspace holds the slots — for gmm / funnel / many_gmm in every mode, up to a
 * slot table of 1 GB (about 245 000 particles at K = 256 with the 64-wide dds net); above it the overdamped modes'
 * bias-row and schedule sums fall back to float atomics and the last bits may differ from call to call.  (The
 * reference's own gradients are XLA reductions: deterministic on one device.) */
/* Workspace growth to plan for: beside the kept trajectory ((K + 1) n dim floats; 3 K + 2 rows of n dim for MCD_CAIS_UHA_sn) the
 * fixed-order sums keep one slot row per (16-particle tile, evaluation): K HP (2 | 4) / 16 floats per particle.  The overdamped
 * modes cap that table at 1 GB and fall back to atomics above it (see above); MCD_CAIS_UHA_sn has NO cap and no fallback — with the
 * 144-wide geffner net and K = 256 it is ~37 KB per particle: 0.6 GB at n = 16 000, 2.4 GB at n = 65 536 on top of the trajectory.
 * Query this function and split larger 2nd-order batches on the caller's side (gradients of particle batches add). */
int64_t cmcd_bound_grad_workspace_bytes(const cmcd_desc* desc, int64_t n);
int cmcd_bound_grad(const cmcd_desc* desc, const cmcd_layout* layout, const int32_t* seeds, int64_t n,
                    const float* params, int64_t n_params, const float* target_consts, int64_t n_target,
                    float omega, void* workspace, int64_t workspace_bytes,
                    float* out_loss, float* out_z, double* out_stats, float* grad, void* stream);

/* ---- Mean-field VI ("pretrain_mfvi"): the reference's plain bounding machine with nbridges = 0
 * (/root/reference/src/boundingmachine.py:73-111, bm.compute_bound :114-118), which
 * /root/reference/src/main.py:82-109 optimises with trainable = ("vd",) to obtain the q every CMCD run
 * starts from.  Per seed: z = mean + exp(logdiag) * normal(first(split(PRNGKey(seed)))) (the same z_0 as
 * cmcd_bound_forward draws), loss = log q(z) - log p(z).  out_loss / out_z / out_stats as cmcd_bound_forward.
 * grad (nullable: forward only) [n_params], overwritten: omega * sum_n d loss_n / d {vd.mean, vd.logdiag}
 * under the reparameterisation, zeros elsewhere; omega = 1 / N_total.
 * Targets: gmm, many_gmm (dim 2), funnel (dim 10), lgcp (dim = m^2). */
int64_t cmcd_mfvi_workspace_bytes(int32_t target, int32_t dim, int64_t n);
int cmcd_mfvi_bound_grad(int32_t target, int32_t dim, int64_t off_vd_mean, int64_t off_vd_logdiag,
                         const int32_t* seeds, int64_t n, const float* params, int64_t n_params,
                         const float* target_consts, int64_t n_target, float omega,
                         void* workspace, int64_t workspace_bytes,
                         float* out_loss, float* out_z, double* out_stats, float* grad, void* stream);

/* ---- Optimiser step of the training loop (/root/reference/src/opt.py:14-35,100-116) fused into one launch:
 * g = clip(grad, +-clip); Adam moments (optax.adam: bias correction 1 - b^step, eps outside the root);
 * params += -lr * m_hat / (sqrt(v_hat) + eps); projection of the listed ranges (opt.py:14-24);
 * optional ema = (1 - ema_step) ema + ema_step params (optax.incremental_update, opt.py:114-116).
 * step is the 1-based iteration count.  All pointers device, length n; ema may be NULL.
 * Non-finite input follows the reference: optax.clip passes a NaN gradient through (the parameter becomes NaN and the
 * next loss trips the check), +-inf is clipped.  Divergence guard (opt.py:122-124 `if isnan(mean(loss)): return` BEFORE
 * the update): with `losses` [n_losses] (device, nullable) the launch first decides whether mean(losses) is NaN — some
 * loss is NaN, or +inf and -inf both occur — and if so leaves params / moments / ema untouched and sets *diverged = 1
 * (device int32, sticky, nullable; once set every later guarded step is skipped too), so the caller may poll the flag
 * at leisure and still gets back the last parameters the reference would have returned. */
enum { CMCD_PROJECT_CLAMP = 0,      /* x -> min(max(x, lo), hi) */
       CMCD_PROJECT_RELU_FLOOR = 1  /* x -> relu(x - lo) + lo   (mgridref_y, opt.py:22-23) */ };
typedef struct cmcd_project_range {
  int64_t offset, length;
  int32_t kind, reserved;
  float lo, hi;
} cmcd_project_range;
int cmcd_adam_step(float* params, const float* grad, float* mu, float* nu, float* ema, int64_t n,
                   float lr, float b1, float b2, float eps, float clip, int64_t step, float ema_step,
                   const cmcd_project_range* ranges, int32_t n_ranges,
                   const float* losses, int64_t n_losses, int32_t* diverged, void* stream);
/* The same step with the iteration count on the device (*step_counter = completed steps, int64, incremented by the
 * call): every launch argument is then constant across iterations, so a whole training iteration (gradient call
 * + this) can be captured once in a hipGraph and replayed (cmcd_amd.opt.run does, for launch-bound configs). */
int cmcd_adam_step_dev(float* params, const float* grad, float* mu, float* nu, float* ema, int64_t n,
                       float lr, float b1, float b2, float eps, float clip, int64_t* step_counter, float ema_step,
                       const cmcd_project_range* ranges, int32_t n_ranges,
                       const float* losses, int64_t n_losses, int32_t* diverged, void* stream);

/* Device-side merge of `count` statistics vectors rows[count][5] (e.g. the result of an RCCL
 * all-gather of every rank's out_stats, in rank order) into out5[5], fixed order, one small kernel
 * on `stream`.  [device] pointers. */
int cmcd_stats_merge_device(const double* rows, int32_t count, double* out5, void* stream);

/* Host-side, no GPU: merge `count` stats vectors (e.g. one per rank, after an all-gather) in
 * the given fixed order, then produce mean, var(ddof=0), lnZ = logsumexp(-l) - log n_total.
 * n_per[i] = number of particles behind stats[i].  out3 = {mean, var, lnZ}. */
int cmcd_stats_merge(const double* stats, const int64_t* n_per, int32_t count,
                     double* merged5, double* out3);

#ifdef __cplusplus
}
#endif
#endif /* CMCD_HIP_H */
