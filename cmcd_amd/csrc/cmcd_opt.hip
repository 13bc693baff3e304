// Fused optimiser step of the reference's training loop (/root/reference/src/opt.py:14-35,100-116):
//   optax.chain(optax.clip(5.0), optax.adam(lr, b1, b2, eps))  ->  params += updates  ->  project(params)
//   [-> ema = optax.incremental_update(params, ema, 0.001)]
// as ONE elementwise launch over params_flat instead of ~15 framework kernels.  Adam as in optax: moments of
// the clipped gradient, bias correction by 1 - b^t, eps outside the square root.  The projection ranges
// (eps in [1e-7, 0.5], eta in [0, 0.99], gamma >= 1e-3, mgridref_y -> relu(x - 1e-3) + 1e-3) are passed as up
// to 8 (offset, length, kind, lo, hi) records in the launch arguments.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "cmcd_common.h"
#include "cmcd_hip.h"

namespace cmcd {

struct OptArgs {
  float* params;
  const float* grad;
  float* mu;
  float* nu;
  float* ema;          // nullable
  int64_t n;
  float lr, b1, b2, eps, clip, c1, c2, ema_step;   // c1 = 1 / (1 - b1^t), c2 = 1 / (1 - b2^t)
  const int64_t* step_dev;                          // non-null: t = *step_dev + 1 (graph replay: the count lives on the device)
  int32_t n_ranges;
  cmcd_project_range ranges[8];
  const float* losses;   // nullable: divergence guard on mean(losses)
  int64_t n_losses;
  int32_t* diverged;     // nullable, sticky
};

__global__ void adam_step_kernel(OptArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a.losses) {
    // opt.py:122-124: `if np.isnan(jnp.mean(loss)): return` before the update.  mean is NaN iff a loss is NaN or both
    // infinities occur.  Every block scans the (L2-resident) losses itself: no extra launch, no inter-block dependency.
    // The block that sets the flag is not ordered against blocks that read it in the SAME launch, so the decision of
    // this launch comes from the scan alone; the flag only carries it to later launches and to the host.
    int bits = (a.diverged && __hip_atomic_load(a.diverged, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ? 4 : 0;
    for (int64_t k = threadIdx.x; k < a.n_losses; k += blockDim.x) {
      const float l = a.losses[k];
      bits |= (l != l) ? 4 : (l == INFINITY ? 1 : (l == -INFINITY ? 2 : 0));
    }
    const int any_nan = __syncthreads_or(bits & 4), pos = __syncthreads_or(bits & 1), neg = __syncthreads_or(bits & 2);
    if (any_nan || (pos && neg)) {
      if (a.diverged && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(a.diverged, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      return;
    }
  }
  if (i >= a.n) return;
  const float graw = a.grad[i];
  // optax.clip = jnp.clip: +-inf is clipped, NaN passes through (and then poisons the parameter, as in the reference)
  const float g = (graw == graw) ? fminf(fmaxf(graw, -a.clip), a.clip) : graw;
  const float m = a.b1 * a.mu[i] + (1.0f - a.b1) * g;
  const float v = a.b2 * a.nu[i] + (1.0f - a.b2) * g * g;
  a.mu[i] = m;
  a.nu[i] = v;
  float c1 = a.c1, c2 = a.c2;
  if (a.step_dev) {
    const double t = (double)(*a.step_dev + 1);
    c1 = (float)(1.0 / (1.0 - pow((double)a.b1, t)));
    c2 = (float)(1.0 / (1.0 - pow((double)a.b2, t)));
  }
  float p = a.params[i] - a.lr * (m * c1) / (sqrtf(v * c2) + a.eps);
  for (int r = 0; r < a.n_ranges; ++r) {
    const cmcd_project_range q = a.ranges[r];
    // jnp.clip / jax.nn.relu propagate NaN (fminf / fmaxf would return the bound and hide a diverged parameter)
    if (i >= q.offset && i < q.offset + q.length && p == p)
      p = q.kind == CMCD_PROJECT_CLAMP ? fminf(fmaxf(p, q.lo), q.hi) : fmaxf(p - q.lo, 0.0f) + q.lo;
  }
  a.params[i] = p;
  if (a.ema) a.ema[i] = (1.0f - a.ema_step) * a.ema[i] + a.ema_step * p;
}

__global__ void step_counter_kernel(int64_t* counter) { *counter += 1; }

}  // namespace cmcd

using namespace cmcd;

static int adam_launch(OptArgs& a, int64_t step, const cmcd_project_range* ranges, int32_t n_ranges, void* stream) {
  if (n_ranges < 0 || n_ranges > 8 || (n_ranges > 0 && !ranges)) return fail_msg(CMCD_ERR_BAD_ARG, "at most 8 projection ranges");
  a.c1 = (float)(1.0 / (1.0 - pow((double)a.b1, (double)step)));
  a.c2 = (float)(1.0 / (1.0 - pow((double)a.b2, (double)step)));
  a.n_ranges = n_ranges;
  for (int r = 0; r < n_ranges; ++r) {
    if (ranges[r].offset < 0 || ranges[r].length < 0 || ranges[r].offset + ranges[r].length > a.n)
      return fail_msg(CMCD_ERR_BAD_ARG, "projection range outside params_flat");
    a.ranges[r] = ranges[r];
  }
  hipLaunchKernelGGL(adam_step_kernel, dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
  return hipGetLastError() == hipSuccess ? CMCD_OK : fail_msg(CMCD_ERR_HIP, "launch failed");
}

// The same step with the iteration count kept on the device (*step_counter = completed steps; incremented by a
// one-thread launch after the update), so that the whole training iteration can be captured in a hipGraph and
// replayed: no launch argument changes between iterations.
extern "C" int cmcd_adam_step_dev(float* params, const float* grad, float* mu, float* nu, float* ema, int64_t n,
                                  float lr, float b1, float b2, float eps, float clip, int64_t* step_counter,
                                  float ema_step, const cmcd_project_range* ranges, int32_t n_ranges,
                                  const float* losses, int64_t n_losses, int32_t* diverged, void* stream) {
  if (!params || !grad || !mu || !nu || !step_counter || n < 1 || (losses && n_losses < 1))
    return fail_msg(CMCD_ERR_BAD_ARG, "bad argument");
  OptArgs a{};
  a.params = params; a.grad = grad; a.mu = mu; a.nu = nu; a.ema = ema; a.n = n;
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.clip = clip; a.ema_step = ema_step; a.step_dev = step_counter;
  a.losses = losses; a.n_losses = n_losses; a.diverged = diverged;
  int rc = adam_launch(a, 1, ranges, n_ranges, stream);
  if (rc != CMCD_OK) return rc;
  hipLaunchKernelGGL(step_counter_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), step_counter);
  return hipGetLastError() == hipSuccess ? CMCD_OK : fail_msg(CMCD_ERR_HIP, "launch failed");
}

extern "C" int cmcd_adam_step(float* params, const float* grad, float* mu, float* nu, float* ema, int64_t n,
                              float lr, float b1, float b2, float eps, float clip, int64_t step, float ema_step,
                              const cmcd_project_range* ranges, int32_t n_ranges,
                              const float* losses, int64_t n_losses, int32_t* diverged, void* stream) {
  if (!params || !grad || !mu || !nu || n < 1 || step < 1 || (losses && n_losses < 1))
    return fail_msg(CMCD_ERR_BAD_ARG, "bad argument");
  OptArgs a{};
  a.params = params; a.grad = grad; a.mu = mu; a.nu = nu; a.ema = ema; a.n = n;
  a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.clip = clip; a.ema_step = ema_step; a.step_dev = nullptr;
  a.losses = losses; a.n_losses = n_losses; a.diverged = diverged;
  return adam_launch(a, step, ranges, n_ranges, stream);
}
