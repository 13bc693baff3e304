// Probe: single-wave instruction-chain costs on gfx950 (cycles via s_memtime), to calibrate the
// latency model of the cooperative kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define KEEP() asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w)::"memory")
#define T0() KEEP(); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory"); KEEP(); __builtin_amdgcn_sched_barrier(0)
#define T1(slot) __builtin_amdgcn_sched_barrier(0); KEEP(); asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory"); KEEP(); res[slot] = t1 - t0
__device__ __forceinline__ float gelu_fast(float x) {
  const float ax = fabsf(x); const float s = fminf(ax, 6.0f);
  float r = 5.626459558e-08f;
  r = fmaf(r, s, -1.389874702e-06f); r = fmaf(r, s, 1.521236383e-05f); r = fmaf(r, s, -9.455732447e-05f);
  r = fmaf(r, s, 3.240720773e-04f); r = fmaf(r, s, -6.315276129e-05f); r = fmaf(r, s, -6.896958595e-03f);
  r = fmaf(r, s, 5.242151140e-02f); r = fmaf(r, s, 4.592238824e-01f); r = fmaf(r, s, 1.151104120e+00f);
  const float e = __builtin_amdgcn_exp2f(-(s * r));
  return fmaf(-0.5f * ax, e, fmaxf(x, 0.0f));
}
__global__ void k(unsigned long long* out, float* buf, float seed) {
  unsigned long long t0, t1, res[16] = {0};
  __shared__ float lds[1024];
  float x = buf[threadIdx.x] + seed, y = x + 1, z = x + 2, w = x + 3;
  lds[threadIdx.x] = x; __syncthreads();
  T0();
#pragma unroll
  for (int i = 0; i < 64; ++i) x = fmaf(x, 1.0001f, 0.5f);
  T1(0);  // 64 dependent fma
  T0();
#pragma unroll
  for (int i = 0; i < 16; ++i) { x = fmaf(x, 1.0001f, 0.5f); y = fmaf(y, 1.0001f, 0.5f); z = fmaf(z, 1.0001f, 0.5f); w = fmaf(w, 1.0001f, 0.5f); }
  T1(1);  // 64 fma in 4 independent chains
  T0();
#pragma unroll
  for (int i = 0; i < 16; ++i) x = __builtin_amdgcn_exp2f(x) - 1.0f;
  T1(2);  // 16 dependent (exp2 + sub)
  T0();
  x = gelu_fast(x); y = gelu_fast(y); z = gelu_fast(z); w = gelu_fast(w);
  T1(3);  // 4 independent gelu
  T0();
  x = gelu_fast(x);
  T1(4);  // 1 gelu
  T0();
  int idx = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 8; ++i) idx = (int)lds[idx & 1023] & 1023;
  T1(5);  // 8 dependent LDS reads (+cvt/and)
  x += idx; y += 1;
  __syncthreads();
  T0();
  __syncthreads();
  T1(6);  // barrier, all waves arrive together
  T0();
  { unsigned a = __float_as_uint(x); auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false); x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    a = __float_as_uint(x); auto q = __builtin_amdgcn_permlane16_swap(a, a, false, false); x = __uint_as_float(q[0]) + __uint_as_float(q[1]); }
  T1(7);  // group_sum via swaps
  T0();
  x += __shfl_xor(x, 16); x += __shfl_xor(x, 32);
  T1(8);  // group_sum via ds_bpermute
  T0();
  float v = buf[(threadIdx.x * 17 + (int)x) & 1023];
  x += v;
  T1(9);  // one global load (L2) round trip
  T0();
  T1(10); // empty (stamp overhead)
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 acc = {x, y, z, w};
  T0();
#pragma unroll
  for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
  T1(11); // 8 dependent mfma 16x16x4
  x += acc[0] + acc[1];
  T0();
  x = 1.0f / x; y = x / y;
  T1(12); // 2 dependent IEEE divisions
  buf[threadIdx.x] = x + y + z + w;
  if (threadIdx.x == 0) for (int i = 0; i < 16; ++i) out[i] = res[i];
}
int main() {
  unsigned long long* d; float* b; hipMalloc(&d, 64 * 8); hipMalloc(&b, 4096 * 4); hipMemset(b, 0, 4096 * 4);
  for (int nw = 1; nw <= 6; nw += 5) {
    k<<<1, 64 * nw>>>(d, b, 0.25f); k<<<1, 64 * nw>>>(d, b, 0.25f);
    unsigned long long h[16]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* n[] = {"64 dep fma", "64 fma 4 chains", "16 dep exp2+sub", "4 indep gelu", "1 gelu", "8 dep lds reads", "barrier", "group_sum swap", "group_sum bpermute", "global load L2", "empty", "8 dep mfma16x16x4", "2 dep div"};
    printf("waves per block = %d\n", nw);
    for (int i = 0; i < 13; ++i) printf("  %-20s %llu cycles\n", n[i], h[i]);
  }
  return 0;
}
