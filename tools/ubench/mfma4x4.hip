// layout probe for v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per wave64)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  int l = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float a = (float)((l & 3) + 1 + 4 * (l >> 2 & 1));   // 1..4 (5..8 in odd blocks)
  float b = 1.0f; for (int i = 0; i < (l & 3); i++) b *= 10.0f;  // 10^(lane%4)
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; r++) out[l * 4 + r] = acc[r];
}
int main() {
  float* d; (void)hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 2, 3, 4, 5, 6, 7}) {
    printf("lane %d:", l);
    for (int r = 0; r < 4; r++) printf("  reg%d = %6.0f", r, h[l * 4 + r]);
    printf("\n");
  }
  return 0;
}
