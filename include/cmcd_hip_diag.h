/*
 * cmcd_hip_diag.h — measurement and diagnostic hooks of libcmcd_hip.so.  NOT part of the drop-in boundary
 * (include/cmcd_hip.h): nothing a result depends on goes through them.  They exist for bench.py (kernel time, kernel
 * name), the PRNG parity test and the probes under tools/probes, are per host thread unless stated, and are compiled OUT of
 * the library by -DCMCD_NO_DIAG_HOOKS (`CMCD_DIAG_HOOKS=0 python -m cmcd_amd.build`): that build exports the boundary only,
 * and the Python binding / bench.py run on it unchanged (kernel time then falls back to the per-step wall time).
 * The library reads TWO environment variables, once per process, for the probes: CMCD_COOP_PRIO, and CMCD_GRAD_ATOMICS=1,
 * which puts the overdamped gradients' sums over tiles back on the float atomics of rounds 1 - 3 (run-to-run differences in
 * the last bits) for A / B timing.  CMCD_GRAD_ITEM is NOT read by the library: only the Python binding forwards it, through
 * cmcd_debug_grad_item below — a C caller that sets the variable gets the measured batch-size rule.
 */
#ifndef CMCD_HIP_DIAG_H
#define CMCD_HIP_DIAG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Name of the trajectory kernel (or launch sequence) the last cmcd_bound_forward of this host thread enqueued, e.g.
 * "coop_kernel<8-particle tiles>", "traj_kernel", "uha_traj_kernel", "lgcp launch sequence"; "" before the first call.
 * bench.py reports it instead of re-deriving the library's selection rule. */
const char* cmcd_last_kernel_name(void);

/* Measurement hook (bench.py): while enabled (per host thread), every cmcd_bound_forward records a
 * hipEvent pair around its trajectory-kernel launch on the caller's stream.
 * cmcd_profile_collect synchronises those events, returns the summed kernel time and the number of
 * launches since the last enable/collect, and resets the counter.  Not for use under graph capture. */
int cmcd_profile_enable(int on);
int cmcd_profile_collect(double* total_ms, int64_t* launches);

/* Diagnostic (tests/test_gpu_prng.py): arm a capture of the PRNG path of the NEXT cmcd_bound_forward issued by this host
 * thread (gmm / funnel / many_gmm; either trajectory kernel writes the words next to the arithmetic that consumes
 * them; consumed and disarmed by that call).  [device] buffers, stage 0 = the draw of z_0, stage i + 1 = bridge i:
 *   bits     uint32 [nbridges+1][n][dim]  the random words that become deviates (jax random_bits of normal(key, (dim,)))
 *   gen_keys uint32 [nbridges+1][n][2]    the chain key entering bridge i, gen_0 .. gen_K (mcd_cais.py:66,87,94); nullable
 *   noise    float  [nbridges+1][n][dim]  the deviates (jax.random.normal)
 * MCD_CAIS_UHA_sn has one more draw in front of the loop (the initial momentum, mcd_under_lp_a_cais.py:92-93): bits / noise
 * are [nbridges+2][n][dim] with stage 1 = rho_0 and stage i + 2 = bridge i; gen_keys stays [nbridges+1][n][2].
 * Pass three NULLs to disarm. */
int cmcd_debug_capture_noise(uint32_t* bits, uint32_t* gen_keys, float* noise);

/* Diagnostic: pin the gradient calls of this process to whole chains (0) or to the work-item path (1); -1 returns to
 * the measured batch-size rule.  (Tests and tools/probes run every case through both; the Python binding forwards the
 * CMCD_GRAD_ITEM environment variable through this call, the library itself reads no environment per call.) */
int cmcd_debug_grad_item(int mode);

/* Diagnostic (tools/probes/uha_item_check.py): while `buf` is non-NULL the MCD_CAIS_UHA_sn gradient's sweep writes the
 * adjoint state it carries — (dL/dz_e, dL/drho_e, dL/drho''_e) entering point e — to buf, float [nbridges+1][3 dim][n]
 * [device]: the whole-chain sweep and the work-item path (whose chunks load that state from the scan launch) can be compared
 * point by point.  Process-wide; NULL disarms. */
void cmcd_debug_uha_xdump(float* buf);

/* Probe (tools/probes/prio_sweep.py): s_setprio levels of the cooperative kernels' roles, 2 bits each {MLP, TGT, RNG, ACC} from bit
 * 0; -1 = the library's table.  Process-wide. */
void cmcd_debug_set_coop_prio(int prio);

#ifdef __cplusplus
}
#endif
#endif /* CMCD_HIP_DIAG_H */
