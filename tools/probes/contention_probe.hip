// Probe: how much does wave 0's instruction stream slow down when other waves of the same
// workgroup run VALU / integer / LDS work at the same time?  Also prints each wave's SIMD id.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define KEEP() asm volatile("" : "+v"(x), "+v"(u)::"memory")
#define T0() KEEP(); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); KEEP(); __builtin_amdgcn_sched_barrier(0)
#define T1(slot) __builtin_amdgcn_sched_barrier(0); KEEP(); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); KEEP(); res[slot] = t1 - t0
__global__ void k(unsigned long long* out, unsigned* simd, float* buf, int mode, int nbusy, int mw) {
  unsigned long long t0, t1, res[4] = {0, 0, 0, 0};
  const int wv = threadIdx.x >> 6;
  float x = buf[threadIdx.x] + 1.5f; unsigned u = threadIdx.x * 2654435761u;
  unsigned hwid; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  if ((threadIdx.x & 63) == 0) simd[wv] = hwid;
  __syncthreads();
  if (wv == mw) {
    T0();
#pragma unroll
    for (int i = 0; i < 64; ++i) x = fmaf(x, 1.0001f, 0.5f);
    T1(0);
    T0();
#pragma unroll
    for (int i = 0; i < 32; ++i) { u += 0x9E3779B9u; u = (u << 13) | (u >> 19); u ^= 0x7F4A7C15u + i; }
    T1(1);   // 96 dependent integer ops (add, rotate, xor) — a Threefry-like chain
    if ((threadIdx.x & 63) == 0) { out[0] = res[0]; out[1] = res[1]; }
  } else if (wv < nbusy) {
    if (mode == 1) { for (int i = 0; i < 4000; ++i) x = fmaf(x, 1.0001f, 0.5f); }
    else if (mode == 2) { for (int i = 0; i < 1500; ++i) { u += 0x9E3779B9u; u = (u << 13) | (u >> 19); u ^= 0x7F4A7C15u; } }
    else if (mode == 3) { for (int i = 0; i < 600; ++i) x = __builtin_amdgcn_exp2f(x) * 0.25f; }
  }
  buf[threadIdx.x] = x + u;
}
int main() {
  unsigned long long* d; unsigned* s; float* b;
  hipMalloc(&d, 64); hipMalloc(&s, 64); hipMalloc(&b, 4096 * 4); hipMemset(b, 0, 4096 * 4);
  const char* mn[] = {"idle", "fma loop", "int loop", "exp2 loop"};
  for (int mw = 0; mw <= 4; mw += 4)
  for (int mode = 0; mode < 4; ++mode)
    for (int nbusy = (mode ? 4 : 8); nbusy <= 8; nbusy += 4) {
      k<<<1, 512>>>(d, s, b, mode, nbusy, mw); k<<<1, 512>>>(d, s, b, mode, nbusy, mw);
      unsigned long long h[2]; unsigned hs[8];
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(hs, s, sizeof(hs), hipMemcpyDeviceToHost);
      printf("measured wave %d | others: %-9s waves <%d busy |  64 dep fma = %4llu cyc, 96 dep int ops = %4llu cyc | simd of waves 0..7:", mw, mn[mode], nbusy, h[0], h[1]);
      for (int i = 0; i < 8; ++i) printf(" %u", (hs[i] >> 4) & 3);
      printf("\n");
    }
  return 0;
}
