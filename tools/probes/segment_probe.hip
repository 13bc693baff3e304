// Probe: cycle cost of the real device functions of the cooperative kernel, standalone.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "cmcd_device.h"
using namespace cmcd;
#define T0() asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); __builtin_amdgcn_sched_barrier(0)
#define T1(slot) __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); res[slot] = t1 - t0
__global__ void k(unsigned long long* out, float* buf, const float* tcg) {
  __shared__ float tc[128];
  __shared__ float xch[1024];
  unsigned long long t0, t1, res[12] = {0};
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 96; i += blockDim.x) tc[i] = tcg[i];
  __syncthreads();
  float z[2] = {buf[lane] + 1.0f, buf[lane + 64] - 2.0f};
  asm volatile("" : "+v"(z[0]), "+v"(z[1]));
  typename Target<CMCD_TARGET_MANY_GMM, 2>::State st;
  float lp, gp[2];
  T0();
  Target<CMCD_TARGET_MANY_GMM, 2>::pass1<8>(z, lane >> 3, tc, st);
  asm volatile("" : "+v"(st.dmin), "+v"(st.d2[0]), "+v"(st.d2[4]));
  T1(0);
  T0();
  Target<CMCD_TARGET_MANY_GMM, 2>::pass2<8>(z, lane >> 3, tc, st, lp, gp);
  asm volatile("" : "+v"(lp), "+v"(gp[0]), "+v"(gp[1]));
  T1(1);
  T0();
  Target<CMCD_TARGET_MANY_GMM, 2>::pass1<4>(z, lane >> 4, tc, st);
  Target<CMCD_TARGET_MANY_GMM, 2>::pass2<4>(z, lane >> 4, tc, st, lp, gp);
  asm volatile("" : "+v"(lp), "+v"(gp[0]), "+v"(gp[1]));
  T1(2);
  f32x4 h = {z[0], z[1], lp, gp[0]};
  T0();
#pragma unroll
  for (int r = 0; r < 4; ++r) h[r] = gelu_fast(h[r]);
  asm volatile("" : "+v"(h));
  T1(3);
  T0();
  float s = group_sum(h[0]), s2 = group_sum(h[1]);
  asm volatile("" : "+v"(s), "+v"(s2));
  T1(4);
  T0();
  float q = part_sum<8>(h[2]);
  asm volatile("" : "+v"(q));
  T1(5);
  T0();
  xch[lane * 4 + 0] = h[0]; 
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  float v = xch[((lane + 17) & 63) * 4];
  asm volatile("" : "+v"(v));
  T1(6);   // LDS write -> barrier -> LDS read round trip
  T0();
  uint32_t x0 = lane, x1 = 2 + lane;
  threefry2x32(12345u, 678u + lane, x0, x1);
  asm volatile("" : "+v"(x0), "+v"(x1));
  T1(7);   // one Threefry-2x32 block
  T0();
  float nrm = bits_to_normal(x0);
  asm volatile("" : "+v"(nrm));
  T1(8);   // bits -> normal
  T0();
  T1(9);   // empty
  buf[threadIdx.x] = s + s2 + q + v + nrm + h[3] + lp + gp[1];
  if (threadIdx.x == 0) for (int i = 0; i < 10; ++i) out[i] = res[i];
}
int main() {
  unsigned long long* d; float* b; float* tc;
  hipMalloc(&d, 128); hipMalloc(&b, 4096 * 4); hipMalloc(&tc, 512);
  float hb[1024]; for (int i = 0; i < 1024; ++i) hb[i] = (i % 37) * 0.37f - 5.0f; hipMemcpy(b, hb, sizeof(hb), hipMemcpyHostToDevice);
  float htc[128] = {0}; htc[0] = 1.343f; htc[1] = -1.302f; int nm = 40; memcpy(&htc[2], &nm, 4); htc[3] = -6.2f;
  for (int i = 0; i < 80; ++i) htc[4 + i] = (i * 7 % 80) - 40.0f;
  hipMemcpy(tc, htc, sizeof(htc), hipMemcpyHostToDevice);
  const char* n[] = {"many_gmm pass1<8> (5 comps)", "many_gmm pass2<8>", "many_gmm full eval<4> (10 comps)", "4 gelu_fast", "2 group_sum (swaps)", "part_sum<8>", "LDS write+barrier+read", "threefry block", "bits_to_normal", "empty"};
  for (int nw = 1; nw <= 8; nw += 7) {
    k<<<1, 64 * nw>>>(d, b, tc); k<<<1, 64 * nw>>>(d, b, tc);
    unsigned long long h[10]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("waves per block = %d\n", nw);
    for (int i = 0; i < 10; ++i) printf("  %-34s %5llu cycles\n", n[i], h[i]);
  }
  return 0;
}
