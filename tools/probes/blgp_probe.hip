// Probe: the B-operand lane-group pattern (BLGP) and the A-operand block broadcast (CBSZ / ABID) of
// v_mfma_f32_4x4x1_16b_f32 on gfx950 (run on the GPU box).  A = 1 everywhere and B = lane id, so accumulator
// register i of lane 4 b + j holds B_b[j] as the instruction saw it; then A = lane id, B = 1 for the A side.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int BLGP>
__device__ void one(float* o, int slot) {
  const float b = (float)threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, b, acc, 0, 0, BLGP);
  o[slot * 64 + threadIdx.x] = acc[0];
}
template <int CBSZ, int ABID>
__device__ void onea(float* o, int slot) {
  const float a = (float)threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, 1.0f, acc, CBSZ, ABID, 0);
  o[slot * 64 + threadIdx.x] = acc[1];   // A of lane 4 b + 1
}
__global__ void k(float* o) {
  one<0>(o, 0); one<1>(o, 1); one<2>(o, 2); one<3>(o, 3); one<4>(o, 4); one<5>(o, 5); one<6>(o, 6); one<7>(o, 7);
  onea<0, 0>(o, 8); onea<1, 0>(o, 9); onea<1, 1>(o, 10); onea<2, 0>(o, 11); onea<2, 3>(o, 12); onea<4, 0>(o, 13); onea<4, 5>(o, 14);
}
int main() {
  float* d; hipMalloc(&d, 15 * 64 * 4);
  k<<<1, 64>>>(d);
  float h[15 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int v = 0; v < 15; ++v) {
    if (v < 8) printf("blgp %d:", v); else printf("A bcast case %d:", v - 8);
    for (int i = 0; i < 64; i += 4) printf(" %2.0f", h[v * 64 + i]);
    printf("\n");
  }
  return 0;
}
