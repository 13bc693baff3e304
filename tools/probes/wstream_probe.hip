// Probe: how fast can 26 x 8 workgroups of 8 waves pull a 1620 x 1620 fp32 weight matrix (10.5 MB, resident in the
// Infinity Cache after the first launch) with the lgcp GEMM's access pattern, and does the layout matter?
//   mode 0: row-major [K][N] (row pitch 6480 B), a wave-load = 2 rows x 32 columns (2 x 128 B)        <- the GEMM today
//   mode 1: packed [column block][k slice][k][64]: a workgroup's 53 KB are contiguous, same 2 x 128 B wave-loads
//   mode 2: packed, 16 B per lane (a wave-load = 1 KB contiguous)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
constexpr int KD = 1620, ND = 1620, KS = 8, SL = 208, CB = 26;
template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ W, float* out) {
  const int cb = blockIdx.x, ks = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int half = wv & 1, q = wv >> 1, l31 = lane & 31, l5 = lane >> 5;
  float acc = 0.f;
  if (MODE == 0) {
    const int n = min(cb * 64 + half * 32 + l31, ND - 1);
    float w[26];
#pragma unroll
    for (int j = 0; j < 26; ++j) w[j] = W[(int64_t)min(ks * SL + q * 52 + 2 * j + l5, KD - 1) * ND + n];
#pragma unroll
    for (int j = 0; j < 26; ++j) acc += w[j];
  } else if (MODE == 1) {
    const float* base = W + (int64_t)(cb * KS + ks) * SL * 64;
    float w[26];
#pragma unroll
    for (int j = 0; j < 26; ++j) w[j] = base[(q * 52 + 2 * j + l5) * 64 + half * 32 + l31];
#pragma unroll
    for (int j = 0; j < 26; ++j) acc += w[j];
  } else {
    const float4* base = reinterpret_cast<const float4*>(W + (int64_t)(cb * KS + ks) * SL * 64) + wv * 416;   // 53248 B / 8 waves = 6656 B = 416 float4
    float4 w[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) w[j] = base[min(j * 64 + lane, 415)];
#pragma unroll
    for (int j = 0; j < 7; ++j) acc += w[j].x + w[j].y + w[j].z + w[j].w;
  }
  if (acc == 12345.678f) out[0] = acc;
}
int main() {
  float *W, *out;
  const size_t bytes = (size_t)CB * KS * SL * 64 * 4 + (1 << 20);
  hipMalloc(&W, bytes); hipMalloc(&out, 4096);
  hipMemset(W, 0, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 100; ++i) {
        if (mode == 0) k<0><<<dim3(CB, KS), 512>>>(W, out);
        else if (mode == 1) k<1><<<dim3(CB, KS), 512>>>(W, out);
        else k<2><<<dim3(CB, KS), 512>>>(W, out);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d: %.2f us per launch, %.2f TB/s\n", mode, ms * 10.f, 10.5e6 / (ms * 1e-5) / 1e12);
    }
  }
  return 0;
}
