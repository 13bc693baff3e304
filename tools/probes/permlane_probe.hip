// Probe: semantics of v_permlane16_swap / v_permlane32_swap on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[128 + threadIdx.x] = q[0]; o[192 + threadIdx.x] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"p32.r0", "p32.r1", "p16.r0", "p16.r1"};
  for (int v = 0; v < 4; ++v) { printf("%s:", names[v]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[v * 64 + i]); printf("\n"); }
  return 0;
}
