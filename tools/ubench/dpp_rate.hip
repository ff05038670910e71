// probe: issue cost of independent DPP ops vs plain VALU for one wave per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 512
__global__ void k(float* out) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  float y0, y1, y2, y3, y4, y5, y6, y7;
  long t0 = clock64();
  for (int i = 0; i < REP; i++) {
    asm volatile(
        "v_add_f32_dpp %0, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %1, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %2, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %3, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %4, %12, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %5, %13, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %6, %14, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "v_add_f32_dpp %7, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        : "=&v"(y0), "=&v"(y1), "=&v"(y2), "=&v"(y3), "=&v"(y4), "=&v"(y5), "=&v"(y6), "=&v"(y7)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7));
    x0 = y0 * 0.5f; x1 = y1 * 0.5f; x2 = y2 * 0.5f; x3 = y3 * 0.5f; x4 = y4 * 0.5f; x5 = y5 * 0.5f; x6 = y6 * 0.5f; x7 = y7 * 0.5f;
  }
  long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) out[64] = (float)(t1 - t0) / (REP * 16.0f);
  t0 = clock64();
  for (int i = 0; i < REP; i++) {
    y0 = x0 + x1; y1 = x1 + x2; y2 = x2 + x3; y3 = x3 + x4; y4 = x4 + x5; y5 = x5 + x6; y6 = x6 + x7; y7 = x7 + x0;
    x0 = y0 * 0.5f; x1 = y1 * 0.5f; x2 = y2 * 0.5f; x3 = y3 * 0.5f; x4 = y4 * 0.5f; x5 = y5 * 0.5f; x6 = y6 * 0.5f; x7 = y7 * 0.5f;
  }
  t1 = clock64();
  out[threadIdx.x] += x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0) out[65] = (float)(t1 - t0) / (REP * 16.0f);
}
int main() {
  float* d; (void)hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[128]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("ticks per instruction: 8x dpp-add + 8x mul interleaved: %.2f ; 8x add + 8x mul: %.2f\n", h[64], h[65]);
  return 0;
}
