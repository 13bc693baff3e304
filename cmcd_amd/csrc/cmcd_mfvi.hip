// Mean-field VI bound and its gradient: the reference's plain bounding machine with nbridges = 0
// (/root/reference/src/boundingmachine.py:73-111: z = sample_rep(q), w = -log q(z) + log p(z), loss = -w;
// bm.compute_bound = mean over seeds), which /root/reference/src/main.py:82-109 optimises
// ("pretrain_mfvi", trainable = ("vd",)) to obtain the q that the CMCD runs start from.
//
// Under the reparameterisation z = mean + std e:  log q(z) = -|e|^2/2 - sum logdiag - const, so
//   d loss / d mean_j = -grad_j log p(z),     d loss / d logdiag_j = -1 - grad_j log p(z) std_j e_j.
// One wave per 16-particle tile (same lane layout and key chain as the trajectory kernels: the z this
// kernel draws for a seed is the z_0 cmcd_bound_forward draws for it); per-tile statistics and gradient
// rows, merged in a fixed order.  lgcp (d = 1600) goes through the skinny-GEMM path of cmcd_lgcp.hip.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "cmcd_common.h"
#include "cmcd_device.h"
#include "cmcd_hip.h"

namespace cmcd {

struct MfviArgs {
  const int32_t* seeds;
  const float* params;
  const float* tc;          // target_consts as handed to the C ABI ({scale, means} for many_gmm)
  float* out_loss;
  float* out_z;
  double* partials;         // [tiles][5]
  float* gpart;             // [tiles][2 D] or nullptr
  int64_t o_mean, o_logdiag, n;
  int32_t n_mix;
  float omega;
};

template <int TARGET, int D>
__global__ __launch_bounds__(256) void mfvi_kernel(MfviArgs a) {
  __shared__ __attribute__((aligned(16))) float lds_tgt[4 + 2 * 64];
  if (TARGET == CMCD_TARGET_MANY_GMM) {   // {scale, means} -> {1/scale, c2, n_mix bits, c0, means} (log2 units)
    const float s = a.tc[0];
    for (int idx = threadIdx.x; idx < 4 + 2 * a.n_mix; idx += blockDim.x) {
      float v;
      if (idx == 0) v = 1.0f / s;
      else if (idx == 1) v = -0.5f * 1.44269504088896340736f / (s * s);
      else if (idx == 2) v = __int_as_float(a.n_mix);
      else if (idx == 3) v = 1.44269504088896340736f * (-2.0f * (logf(s) + kHalfLog2Pi) - logf((float)a.n_mix));
      else v = a.tc[1 + (idx - 4)];
      lds_tgt[idx] = v;
    }
  }
  __syncthreads();
  constexpr int Hh = (D + 1) / 2;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wv;
  const int64_t p = tile * 16 + c;
  if (tile * 16 >= a.n) return;
  const bool valid = p < a.n;
  const int32_t seed = a.seeds[valid ? p : a.n - 1];

  // A = first(split(PRNGKey(seed))); e = normal(A, (D,))          boundingmachine.py:84-87, diag_gauss.py:49-55
  const int gb = g & 1;
  uint32_t x0 = gb, x1 = 2 + gb;
  threefry2x32(0u, (uint32_t)seed, x0, x1);
  uint32_t a0, a1;
  rows01(x0, a0, a1);
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
  float z[D], dz[D], gp[D], logp;
  float w = 0.f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float mean = a.params[a.o_mean + j], ld = a.params[a.o_logdiag + j];
    const float sd = expf(ld);
    z[j] = sd * nz[j] + mean;
    dz[j] = z[j] - mean;
    w -= -(dz[j] * dz[j]) / (2.0f * sd * sd) - logf(sd) - kHalfLog2Pi;   // w = -log q(z)
  }
  Target<TARGET, D>::eval(z, g, lds_tgt, logp, gp);
  w += logp;
  const float loss = -w;
  if (valid && g == 0) {
    a.out_loss[p] = loss;
#pragma unroll
    for (int j = 0; j < D; ++j) a.out_z[p * D + j] = z[j];
  }
  if (a.gpart) {
    const float om = valid ? a.omega : 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float gm = row_sum16(-om * gp[j]);
      const float gl = row_sum16(om * (-1.0f - gp[j] * dz[j]));
      if (lane == 0) {
        a.gpart[tile * (2 * D) + j] = gm;
        a.gpart[tile * (2 * D) + D + j] = gl;
      }
    }
  }
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
    double* o = a.partials + tile * CMCD_NSTATS;
    o[0] = cnt; o[1] = sm; o[2] = sq; o[3] = mx; o[4] = ex;
  }
}

// grad[o_mean + j], grad[o_logdiag + j] = fixed-order sums of the per-tile (or per-particle, scaled) rows
struct MfviReduceArgs {
  const float* rows;     // [count][2 D]
  float* grad;
  int64_t count, o_mean, o_logdiag;
  int32_t D;
  float scale;
};

__global__ void mfvi_reduce_kernel(MfviReduceArgs a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= 2 * a.D) return;
  float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
  int64_t r = 0;
  for (; r + 4 <= a.count; r += 4) {
    v0 += a.rows[(r + 0) * 2 * a.D + j];
    v1 += a.rows[(r + 1) * 2 * a.D + j];
    v2 += a.rows[(r + 2) * 2 * a.D + j];
    v3 += a.rows[(r + 3) * 2 * a.D + j];
  }
  for (; r < a.count; ++r) v0 += a.rows[r * 2 * a.D + j];
  const float v = ((v0 + v1) + (v2 + v3)) * a.scale;
  if (j < a.D) a.grad[a.o_mean + j] = v;
  else a.grad[a.o_logdiag + (j - a.D)] = v;
}

typedef void (*mfvi_fn)(MfviArgs);
static mfvi_fn pick_mfvi(int target, int dim) {
  if (target == CMCD_TARGET_GMM && dim == 2) return mfvi_kernel<CMCD_TARGET_GMM, 2>;
  if (target == CMCD_TARGET_MANY_GMM && dim == 2) return mfvi_kernel<CMCD_TARGET_MANY_GMM, 2>;
  if (target == CMCD_TARGET_FUNNEL && dim == 10) return mfvi_kernel<CMCD_TARGET_FUNNEL, 10>;
  return nullptr;
}

static inline int64_t al4(int64_t x) { return (x + 3) & ~int64_t(3); }

}  // namespace cmcd

using namespace cmcd;

extern "C" {

int64_t cmcd_mfvi_workspace_bytes(int32_t target, int32_t dim, int64_t n) {
  if (n < 1 || dim < 1) return 0;
  if (target == CMCD_TARGET_LGCP) return lgcp_mfvi_workspace_floats(dim, n, true) * 4;
  if (!pick_mfvi(target, dim)) {
    fail_msg(CMCD_ERR_UNSUPPORTED, "no mean-field VI kernel instance for this (target, dim)");
    return 0;
  }
  const int64_t tiles = (n + 15) / 16;
  return (al4(tiles * CMCD_NSTATS * 2) + al4(tiles * 2 * dim)) * 4;
}

int cmcd_mfvi_bound_grad(int32_t target, int32_t dim, int64_t off_mean, int64_t off_logdiag, const int32_t* seeds,
                         int64_t n, const float* params, int64_t n_params, const float* target_consts,
                         int64_t n_target, float omega, void* workspace, int64_t workspace_bytes, float* out_loss,
                         float* out_z, double* out_stats, float* grad, void* stream_) {
  if (!seeds || !params || !workspace || !out_loss || !out_z || !out_stats)
    return fail_msg(CMCD_ERR_BAD_ARG, "null pointer argument");
  if (n < 1 || n > (int64_t)1 << 31 || dim < 1) return fail_msg(CMCD_ERR_BAD_ARG, "n or dim out of range");
  if (off_mean < 0 || off_logdiag < 0 || off_mean + dim > n_params || off_logdiag + dim > n_params)
    return fail_msg(CMCD_ERR_BAD_ARG, "layout offset missing or outside params_flat");
  const int64_t need = cmcd_mfvi_workspace_bytes(target, dim, n);
  if (need <= 0) return CMCD_ERR_UNSUPPORTED;
  if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15))
    return fail_msg(CMCD_ERR_WORKSPACE, "workspace too small or not 16-byte aligned");
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  float* ws = static_cast<float*>(workspace);
  if (grad && hipMemsetAsync(grad, 0, sizeof(float) * n_params, stream) != hipSuccess) return fail_msg(CMCD_ERR_HIP, "memset failed");

  if (target == CMCD_TARGET_LGCP) {
    if (!target_consts || n_target != (int64_t)dim * dim + dim + 3)
      return fail_msg(CMCD_ERR_BAD_ARG, "lgcp needs target_consts = {Kinv[d,d], counts[d], mu0, a, lognorm}");
    double* partials = nullptr;
    float* gbuf = nullptr;
    int rc = lgcp_mfvi(dim, off_mean, off_logdiag, seeds, n, params, target_consts, ws, out_loss, out_z, &partials,
                       &gbuf, grad != nullptr, stream_);
    if (rc != CMCD_OK) return fail_msg(rc, "lgcp mean-field launch sequence failed");
    rc = launch_finalize(partials, (int32_t)n, out_stats, stream_);
    if (rc != CMCD_OK) return rc;
    if (grad) {
      MfviReduceArgs ra{gbuf, grad, n, off_mean, off_logdiag, dim, omega};
      hipLaunchKernelGGL(mfvi_reduce_kernel, dim3((2 * dim + 255) / 256), dim3(256), 0, stream, ra);
    }
    return hipGetLastError() == hipSuccess ? CMCD_OK : fail_msg(CMCD_ERR_HIP, "launch failed");
  }

  mfvi_fn fn = pick_mfvi(target, dim);
  int n_mix = 0;
  if (target == CMCD_TARGET_MANY_GMM) {
    if (!target_consts || n_target < 3 || (n_target - 1) % 2 != 0 || (n_target - 1) / 2 > 64)
      return fail_msg(CMCD_ERR_BAD_ARG, "many_gmm needs target_consts = {scale, means[n_mixes<=64][2]}");
    n_mix = int((n_target - 1) / 2);
  }
  const int64_t tiles = (n + 15) / 16;
  double* partials = reinterpret_cast<double*>(ws);
  float* gpart = ws + al4(tiles * CMCD_NSTATS * 2);
  MfviArgs ma{seeds, params, target_consts, out_loss, out_z, partials, grad ? gpart : nullptr,
              off_mean, off_logdiag, n, n_mix, omega};
  hipLaunchKernelGGL(fn, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, stream, ma);
  int rc = launch_finalize(partials, (int32_t)tiles, out_stats, stream_);
  if (rc != CMCD_OK) return rc;
  if (grad) {
    MfviReduceArgs ra{gpart, grad, tiles, off_mean, off_logdiag, dim, 1.0f};
    hipLaunchKernelGGL(mfvi_reduce_kernel, dim3((2 * dim + 255) / 256), dim3(256), 0, stream, ra);
  }
  return hipGetLastError() == hipSuccess ? CMCD_OK : fail_msg(CMCD_ERR_HIP, "launch failed");
}

}  // extern "C"
