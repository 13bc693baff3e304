// Issue cost of v_pk_fma_f32 vs v_fma_f32 on gfx950, one wave alone and with 1..4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 pk_probe.hip -o pk_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = 1.f, a2 = 2.f, a3 = 3.f, a4 = 4.f, a5 = 5.f, a6 = 6.f, a7 = 7.f;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  f2 q0 = p0, q1 = p1, q2 = p2, q3 = p3;
  const float c = 1.0001f, d = 0.5f;
  const f2 c2 = {c, c}, d2 = {d, d};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a1) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a2) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a3) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a4) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a5) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a6) : "v"(c), "v"(d));
        asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a7) : "v"(c), "v"(d));
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(c2), "v"(d2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(c2), "v"(d2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(c2), "v"(d2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(c2), "v"(d2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q0) : "v"(c2), "v"(d2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q1) : "v"(c2), "v"(d2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q2) : "v"(c2), "v"(d2));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(q3) : "v"(c2), "v"(d2));
      }
    }
  }
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1] + q0[0] + q1[1] + q2[0] + q3[1];
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1 << 16);
  const int iters = 200;
  for (int waves = 1; waves <= 16; waves *= 2) {      // waves per workgroup of ONE CU: 4 -> one per SIMD, 16 -> four per SIMD
    for (int mode = 0; mode < 2; ++mode) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      else hipLaunchKernelGGL(k<1>, dim3(1), dim3(64 * waves), 0, 0, out, cyc, iters);
      hipDeviceSynchronize();
      unsigned long long h[16];
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double mx = 0;
      for (int w = 0; w < waves; ++w) mx = h[w] > mx ? h[w] : mx;
      // s_memtime ticks at 100 MHz on gfx9: report ratio only, plus per-instruction in ticks
      printf("waves/CU %2d  %s: %8.0f ticks for %d instr/wave  -> %.4f ticks per instr per wave\n", waves,
             mode ? "v_pk_fma_f32" : "v_fma_f32   ", mx, iters * 64, mx / (iters * 64));
    }
  }
  return 0;
}
