// Probe: do fp32 MFMAs (v_mfma_f32_16x16x4_f32, 8 passes) and plain VALU work overlap on one SIMD of gfx950?
//  (a) two waves on one SIMD, one issuing independent MFMAs, the other independent FMAs: each alone, then together;
//  (b) ONE wave interleaving an MFMA with k independent FMAs.
// Prints s_memtime ticks (100 MHz constant clock on this part: compare ratios, not absolute cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define STAMP(t) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory")
template <int KF>
__global__ __launch_bounds__(512) void k(unsigned long long* out, unsigned* simd, float* buf, int mode, int reps) {
  const int wv = threadIdx.x >> 6;
  unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if ((threadIdx.x & 63) == 0) simd[wv] = hwid;
  float x0 = buf[threadIdx.x], x1 = x0 + 1.f, x2 = x0 + 2.f, x3 = x0 + 3.f;
  f32x4 a0 = {x0, x0, x0, x0}, a1 = a0, a2 = a0, a3 = a0;
  const float av = x0 * 0.5f, bv = x0 * 0.25f;
  unsigned long long t0 = 0, t1 = 0;
  __syncthreads();
  const bool do_mfma = (mode == 0 || mode == 2) && wv == 0;
  const bool do_valu = (mode == 1 || mode == 2) && wv == 4;
  const bool do_both = mode == 3 && wv == 0;
  if (do_mfma) {
    STAMP(t0);
    for (int i = 0; i < reps; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a3, 0, 0, 0);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    STAMP(t1);
  } else if (do_valu) {
    STAMP(t0);
    for (int i = 0; i < reps; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {   // 32 independent-ish FMAs per iteration = the issue slots of 4 MFMAs (4 x 8 passes)
        x0 = fmaf(x0, 1.0001f, 0.5f); x1 = fmaf(x1, 1.0001f, 0.5f); x2 = fmaf(x2, 1.0001f, 0.5f); x3 = fmaf(x3, 1.0001f, 0.5f);
      }
    }
    STAMP(t1);
  } else if (do_both) {
    STAMP(t0);
    for (int i = 0; i < reps; ++i) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        if (m == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a0, 0, 0, 0);
        if (m == 1) a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a1, 0, 0, 0);
        if (m == 2) a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a2, 0, 0, 0);
        if (m == 3) a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, a3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (u < KF) {
            if ((u & 3) == 0) x0 = fmaf(x0, 1.0001f, 0.5f);
            if ((u & 3) == 1) x1 = fmaf(x1, 1.0001f, 0.5f);
            if ((u & 3) == 2) x2 = fmaf(x2, 1.0001f, 0.5f);
            if ((u & 3) == 3) x3 = fmaf(x3, 1.0001f, 0.5f);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    STAMP(t1);
  }
  buf[threadIdx.x] = x0 + x1 + x2 + x3 + a0[0] + a1[1] + a2[2] + a3[3];
  if ((threadIdx.x & 63) == 0) out[wv] = t1 - t0;
}
int main() {
  unsigned long long* d; unsigned* s; float* b;
  hipMalloc(&d, 64); hipMalloc(&s, 64); hipMalloc(&b, 4096 * 4); hipMemset(b, 0, 4096 * 4);
  const int reps = 4000;
  const char* mn[] = {"mfma alone (wave 0)", "fma alone (wave 4)", "both, two waves", "one wave interleaved"};
  for (int mode = 0; mode < 4; ++mode)
    for (int kf = (mode == 3 ? 0 : 8); kf <= 8; kf += 2) {
      hipMemset(d, 0, 64);
      auto run = [&] {
        if (kf == 0) k<0><<<1, 512>>>(d, s, b, mode, reps);
        else if (kf == 2) k<2><<<1, 512>>>(d, s, b, mode, reps);
        else if (kf == 4) k<4><<<1, 512>>>(d, s, b, mode, reps);
        else if (kf == 6) k<6><<<1, 512>>>(d, s, b, mode, reps);
        else k<8><<<1, 512>>>(d, s, b, mode, reps);
      };
      run(); hipMemset(d, 0, 64); run();
      unsigned long long h[8]; unsigned hs[8];
      hipMemcpy(h, d, 64, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 32, hipMemcpyDeviceToHost);
      printf("%-24s kf=%d  wave0 %8llu ticks  wave4 %8llu ticks   (%d x 4 MFMA | %d x 32 FMA)  simd ids w0=%u w4=%u\n", mn[mode], kf, h[0], h[4],
             reps, reps, (hs[0] >> 4) & 3, (hs[4] >> 4) & 3);
    }
  return 0;
}
