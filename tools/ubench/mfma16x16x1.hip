// layout probe for v_mfma_f32_16x16x1_4b_f32 (4 independent 16x16 outer products per wave64)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out) {
  int l = threadIdx.x;
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  float a = (float)((l & 15) + 1 + 100 * (l >> 4));   // A: 1..16 (+100*block)
  float b = (float)(1000 * ((l & 15) + 1));            // B: 1000..16000
  acc = __builtin_amdgcn_mfma_f32_16x16x1f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 16; r++) out[l * 16 + r] = acc[r];
}
int main() {
  float* d; (void)hipMalloc(&d, 1024 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[1024]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // value = a_i * b_j = (i+1+100*blk) * 1000*(j+1)  -> decode i (A lane), j (B lane)
  for (int l : {0, 1, 5, 15, 16, 17, 63}) {
    printf("lane %2d:", l);
    for (int r = 0; r < 16; r++) {
      double v = h[l * 16 + r] / 1000.0; int found = 0;
      for (int j = 1; j <= 16 && !found; j++) { double ai = v / j; if (ai == (int)ai) { int aii = (int)ai; int blk = aii / 100, i = aii % 100; if (i >= 1 && i <= 16 && blk < 4) { printf(" r%d=(A%d.%d,B%d)", r, blk, i - 1, j - 1); found = 1; } } }
      if (!found) printf(" r%d=?%.0f", r, h[l * 16 + r]);
    }
    printf("\n");
  }
  return 0;
}
