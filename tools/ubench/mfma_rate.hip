// micro-benchmark: issue cost of fp32 / bf16 MFMAs for one wave per SIMD (4 independent accumulators, like
// csrc/policy_mlp.hip) and for 2 waves per SIMD.  Prints cycles per MFMA as seen by wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define REP 512
__global__ void k_f32_16(float* out, float a, float b, int slot) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x = a + threadIdx.x, y = b;
  long t0 = clock64();
  for (int i = 0; i < REP; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[t], 0, 0, 0);
  long t1 = clock64();
  float s = 0;
  for (int t = 0; t < 4; t++) s += acc[t][0] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 + slot] = (float)(t1 - t0) / (REP * 4);
}
__global__ void k_f32_32(float* out, float a, float b, int slot) {
  f32x16 acc[2];
  for (int t = 0; t < 2; t++) for (int k = 0; k < 16; k++) acc[t][k] = 0;
  float x = a + threadIdx.x, y = b;
  long t0 = clock64();
  for (int i = 0; i < REP; i++)
#pragma unroll
    for (int t = 0; t < 2; t++) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[t], 0, 0, 0);
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0][0] + acc[1][5];
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 + slot] = (float)(t1 - t0) / (REP * 2);
}
__global__ void k_bf16_16(float* out, float a, int slot) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  bf16x8 x, y;
  for (int k = 0; k < 8; k++) { x[k] = (__bf16)(a + k); y[k] = (__bf16)(0.5f); }
  long t0 = clock64();
  for (int i = 0; i < REP; i++)
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc[t], 0, 0, 0);
  long t1 = clock64();
  float s = 0;
  for (int t = 0; t < 4; t++) s += acc[t][0] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 + slot] = (float)(t1 - t0) / (REP * 4);
}
int main() {
  float* d; (void)hipMalloc(&d, 8192 * 4);
  for (int waves = 1; waves <= 2; waves++) {
    dim3 g(1), b(256 * waves);
    k_f32_16<<<g, b>>>(d, 1.0f, 0.5f, 0); k_f32_32<<<g, b>>>(d, 1.0f, 0.5f, 1); k_bf16_16<<<g, b>>>(d, 1.0f, 2);
    float h[3]; (void)hipMemcpy(h, d + 4096, 12, hipMemcpyDeviceToHost);
    printf("%d wave(s)/SIMD: v_mfma_f32_16x16x4_f32 %.1f cycles (2048 flop), v_mfma_f32_32x32x2_f32 %.1f cycles (4096 flop), "
           "v_mfma_f32_16x16x32_bf16 %.1f cycles (16384 flop)\n", waves, h[0], h[1], h[2]);
  }
  return 0;
}
