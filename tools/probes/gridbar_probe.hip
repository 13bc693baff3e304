// Cost of a grid-wide barrier on MI355X (256 workgroups, one per CU): atomic arrive + bounded spin on a
// device-scope counter.  build: hipcc --offload-arch=gfx950 -O3 gridbar_probe.hip -o gridbar_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void k(unsigned* counter, unsigned* fail, int nbar, int work) {
  __shared__ float sink;
  float acc = threadIdx.x;
  for (int b = 0; b < nbar; ++b) {
    for (int i = 0; i < work; ++i) acc = acc * 1.0001f + 0.5f;   // some per-phase work
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      const unsigned target = (unsigned)(b + 1) * gridDim.x;
      int spins = 0;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000) { *fail = 1; break; }   // bounded: never hang the GPU
      }
      __threadfence();
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) sink = acc;
}

int main() {
  unsigned *counter, *fail;
  hipMalloc(&counter, 4); hipMalloc(&fail, 4);
  for (int grid : {64, 128, 256}) {
    for (int nbar : {200, 1000}) {
      hipMemset(counter, 0, 4); hipMemset(fail, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, counter, fail, nbar, 0);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      unsigned f; hipMemcpy(&f, fail, 4, hipMemcpyDeviceToHost);
      printf("grid %3d  %4d barriers: %.3f ms  -> %.2f us per barrier  fail=%u\n", grid, nbar, ms, ms * 1e3 / nbar, f);
    }
  }
  return 0;
}
