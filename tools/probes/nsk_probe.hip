// Probe (r04): a NO-split-K skinny GEMM for the <= 32-particle lgcp pass — is a dependent launch shorter without the slab /
// ticket / last-arriver seam when every operand is PACKED so that a wave-load is 1 KB contiguous and a wave holds its whole
// share of the contraction in registers (one exposure of the load latency)?
//   workgroup = (16-column tile, 16-row half), 8 waves, the contraction split over the waves in interleaved 16-deep chunks
//   (13 per wave for K <= 1664), v_mfma_f32_16x16x4_f32, cross-wave sum through LDS, consumer on 256 threads, output
//   written in the packed operand layout of the NEXT launch.
// Timed: the launch chain A (x W1 -> u1), B (u1 W2 -> u2 | (x - mu0) K^-1 -> kr), C (u2 W3 -> state update stand-in) x 129.
//   hipcc --offload-arch=gfx950 -O3 -I cmcd_amd/csrc -I include tools/probes/nsk_probe.hip -o tools/probes/bin/nsk_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "cmcd_device.h"
using namespace cmcd;

constexpr int D = 1600, IN = 1620, NCH = 104, KP = NCH * 16;   // 104 chunks of 16 = 1664 >= 1620
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Seg { const float* A; const float* W; float* out; const float* u; int N, epi; float shift; };
struct Args { Seg seg[2]; int nt0; const float* bias; int M; };

__host__ __device__ inline int64_t packA(int r, int k) {   // [half][chunk][lane = r % 16 + 16 kq][s]
  return (((int64_t)(r / 16) * NCH + k / 16) * 64 + (r % 16) + 16 * ((k % 16) / 4)) * 4 + (k % 4);
}
__host__ __device__ inline int64_t packW(int k, int n) {   // [tile][chunk][lane = n % 16 + 16 kq][s]
  return (((int64_t)(n / 16) * NCH + k / 16) * 64 + (n % 16) + 16 * ((k % 16) / 4)) * 4 + (k % 4);
}

// MERGE: one workgroup per column tile does BOTH row halves (weights fetched once; the second half's operand rows beyond M
// are never loaded); SKIP: lanes of padding rows load nothing
template <int MERGE, int SKIP>
__global__ __launch_bounds__(512) void nsk(Args a) {
  __shared__ float red[8][256 * (MERGE ? 2 : 1)];
  const int sI = blockIdx.x >= a.nt0;
  const Seg sg = a.seg[sI];
  const int tile = blockIdx.x - (sI ? a.nt0 : 0), half = MERGE ? 0 : blockIdx.y;
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const f32x4v* Ap = reinterpret_cast<const f32x4v*>(sg.A) + ((int64_t)half * NCH + wv) * 64 + lane;
  const f32x4v* Wp = reinterpret_cast<const f32x4v*>(sg.W) + ((int64_t)tile * NCH + wv) * 64 + lane;
  f32x4v av[13], bv[13], a2[MERGE ? 13 : 1];
  const f32x4v zero4 = {0.f, 0.f, 0.f, 0.f};
  const bool live0 = !SKIP || half * 16 + (lane & 15) < a.M, live1 = 16 + (lane & 15) < a.M;
#ifdef NSK_INTERLEAVE   // chunk by chunk: the first matrix instruction waits for two loads, not fourteen
#pragma unroll
  for (int j = 0; j < 13; ++j) { bv[j] = Wp[j * 512]; av[j] = live0 ? Ap[j * 512] : zero4; }
#else
#pragma unroll
  for (int j = 0; j < 13; ++j) bv[j] = Wp[j * 512];     // weights first: independent of the previous launch
#pragma unroll
  for (int j = 0; j < 13; ++j) av[j] = live0 ? Ap[j * 512] : zero4;
#endif
  if (MERGE) {
#pragma unroll
    for (int j = 0; j < 13; ++j) a2[j] = live1 ? Ap[(int64_t)NCH * 64 + j * 512] : zero4;
  }
  __builtin_amdgcn_sched_barrier(0);      // all loads in flight before the first wait (the scheduler otherwise sinks them)
  f32x4v acc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) acc[s] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 13; ++j)
#pragma unroll
    for (int s = 0; s < 4; ++s) acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j][s] - sg.shift, bv[j][s], acc[s], 0, 0, 0);
  const f32x4v t = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) red[wv][r * 64 + lane] = t[r];
  if (MERGE) {
    f32x4v ac2[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) ac2[s] = zero4;
#pragma unroll
    for (int j = 0; j < 13; ++j)
#pragma unroll
      for (int s = 0; s < 4; ++s) ac2[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[j][s] - sg.shift, bv[j][s], ac2[s], 0, 0, 0);
    const f32x4v t2 = (ac2[0] + ac2[1]) + (ac2[2] + ac2[3]);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][256 + r * 64 + lane] = t2[r];
  }
  __syncthreads();
  if (!MERGE && threadIdx.x >= 256) return;
  const int l2 = threadIdx.x & 63, reg = (threadIdx.x >> 6) & 3, hb = MERGE ? threadIdx.x >> 8 : 0;
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) v += red[w][hb * 256 + reg * 64 + l2];
  const int cl = l2 & 15, rl = 4 * (l2 >> 4) + reg;        // D layout of 16x16x4: column = lane % 16, row = 4 (lane / 16) + reg
  const int n = tile * 16 + cl, row = (MERGE ? hb : half) * 16 + rl;
  if (n >= sg.N || row >= a.M) return;
  if (sg.epi == 0) {                                       // activation: u + softplus(v + bias)
    const float u = sg.u[packA(row, n)];
    sg.out[packA(row, n)] = u + softplus(v + a.bias[n]);
  } else if (sg.epi == 1) {                                // plain product
    sg.out[packA(row, n)] = v;
  } else {                                                 // stand-in for the state update: a Threefry block, a deviate, two exps
    uint32_t y0 = n, y1 = n + 800;
    threefry2x32(17u + row, 99u, y0, y1);
    const float z = sg.u[packA(row, n)];
    const float g = -v + 3.f - 0.000625f * expf(z);
    const float zn = z + 1e-5f * g + 0.0044f * bits_to_normal(y0);
    sg.out[packA(row, n)] = zn;
  }
}

int main() {
  const int M = 20;
  std::vector<float> hW((size_t)104 * NCH * 64 * 4, 0.f), hA((size_t)2 * NCH * 64 * 4, 0.f);
  srand(1);
  std::vector<float> Wn((size_t)IN * IN), An((size_t)32 * IN, 0.f);
  for (auto& w : Wn) w = (rand() / (float)RAND_MAX - 0.5f) * 0.05f;
  for (int r = 0; r < M; ++r) for (int k = 0; k < IN; ++k) An[(size_t)r * IN + k] = rand() / (float)RAND_MAX;
  for (int k = 0; k < IN; ++k) for (int n = 0; n < IN; ++n) hW[packW(k, n)] = Wn[(size_t)k * IN + n];
  for (int r = 0; r < M; ++r) for (int k = 0; k < IN; ++k) hA[packA(r, k)] = An[(size_t)r * IN + k];
  float *W1, *W2, *W3, *Ki, *x, *u1, *u2, *kr, *bias;
  const size_t wb = hW.size() * 4, ab = hA.size() * 4;
  hipMalloc(&W1, wb); hipMalloc(&W2, wb); hipMalloc(&W3, wb); hipMalloc(&Ki, wb);
  hipMalloc(&x, ab); hipMalloc(&u1, ab); hipMalloc(&u2, ab); hipMalloc(&kr, ab); hipMalloc(&bias, 8192);
  hipMemcpy(W1, hW.data(), wb, hipMemcpyHostToDevice); hipMemcpy(W2, hW.data(), wb, hipMemcpyHostToDevice);
  hipMemcpy(W3, hW.data(), wb, hipMemcpyHostToDevice); hipMemcpy(Ki, hW.data(), wb, hipMemcpyHostToDevice);
  hipMemcpy(x, hA.data(), ab, hipMemcpyHostToDevice); hipMemset(u1, 0, ab); hipMemset(u2, 0, ab); hipMemset(kr, 0, ab);
  hipMemset(bias, 0, 8192);
  const int tIN = (IN + 15) / 16, tD = D / 16;
  Args A{}; A.M = M; A.bias = bias;
  A.seg[0] = Seg{x, W1, u1, x, IN, 0, 0.f}; A.nt0 = tIN;
  Args B{}; B.M = M; B.bias = bias;
  B.seg[0] = Seg{u1, W2, u2, u1, IN, 0, 0.f}; B.seg[1] = Seg{x, Ki, kr, x, D, 1, 3.88f}; B.nt0 = tIN;
  Args C{}; C.M = M; C.bias = bias;
  C.seg[0] = Seg{u2, W3, x, x, D, 2, 0.f}; C.nt0 = tD;
  // correctness of one launch against the host
  nsk<1, 1><<<dim3(tIN, 1), 512>>>(A);
  std::vector<float> hu(hA.size());
  hipMemcpy(hu.data(), u1, ab, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int r = 0; r < M; r += 7) for (int n = 0; n < IN; n += 13) {
    double s = 0;
    for (int k = 0; k < IN; ++k) s += (double)An[(size_t)r * IN + k] * Wn[(size_t)k * IN + n];
    const double ref = An[(size_t)r * IN + n] + log1p(exp(s));
    worst = fmax(worst, fabs(ref - hu[packA(r, n)]));
  }
  printf("launch A against the host: worst abs error %.2e\n", worst);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* what, int launches, auto&& body) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      body();
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("%-60s %8.3f ms = %.2f us per launch\n", what, ms, ms * 1e3f / launches);
    }
  };
#define RUN(MG, SK, GY, tag)                                                                         \
  timeit(tag " chain A B C x 129", 387, [&] {                                                        \
    for (int i = 0; i < 129; ++i) {                                                                  \
      nsk<MG, SK><<<dim3(tIN, GY), 512>>>(A);                                                        \
      nsk<MG, SK><<<dim3(tIN + tD, GY), 512>>>(B);                                                   \
      nsk<MG, SK><<<dim3(tD, GY), 512>>>(C);                                                         \
    }                                                                                                \
  });                                                                                                \
  timeit(tag " A only x 387", 387, [&] { for (int i = 0; i < 387; ++i) nsk<MG, SK><<<dim3(tIN, GY), 512>>>(A); });      \
  timeit(tag " B only x 387", 387, [&] { for (int i = 0; i < 387; ++i) nsk<MG, SK><<<dim3(tIN + tD, GY), 512>>>(B); }); \
  timeit(tag " C only x 387", 387, [&] { for (int i = 0; i < 387; ++i) nsk<MG, SK><<<dim3(tD, GY), 512>>>(C); });
  RUN(0, 0, 2, "[split halves]")
  RUN(1, 0, 1, "[merged halves]")
  timeit("[A, C split halves; B merged] chain A B C x 129", 387, [&] {
    for (int i = 0; i < 129; ++i) {
      nsk<0, 0><<<dim3(tIN, 2), 512>>>(A);
      nsk<1, 0><<<dim3(tIN + tD, 1), 512>>>(B);
      nsk<0, 0><<<dim3(tD, 2), 512>>>(C);
    }
  });
  return 0;
}
