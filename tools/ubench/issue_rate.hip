// micro-benchmark: single-wave VALU issue behaviour on gfx950 (1 wave per SIMD, like k_step at N=4096)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp_(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
#define REP 4096
__global__ void k_indep(float* out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  long t0 = clock64();
  for (int i = 0; i < REP / 8; i++) {
    x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
    x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
  }
  long t1 = clock64();
  out[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[64] = (float)(t1 - t0) / REP;
}
__global__ void k_dep(float* out, float a, float b) {
  float x0 = threadIdx.x;
  long t0 = clock64();
  for (int i = 0; i < REP / 8; i++) {
    x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
    x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b); x0 = fmaf(x0, a, b);
  }
  long t1 = clock64();
  out[threadIdx.x] = x0;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[65] = (float)(t1 - t0) / REP;
}
__global__ void k_dep2(float* out, float a, float b) {   // two interleaved dependent chains
  float x0 = threadIdx.x, x1 = x0 + 1;
  long t0 = clock64();
  for (int i = 0; i < REP / 8; i++) {
    x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b);
    x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b);
  }
  long t1 = clock64();
  out[threadIdx.x] = x0 + x1;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[66] = (float)(t1 - t0) / REP;
}
__global__ void k_dpp(float* out, float a) {   // dependent quad-sum chain: add+dpp pairs
  float x0 = threadIdx.x;
  long t0 = clock64();
  for (int i = 0; i < REP / 8; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) { x0 = x0 * a + dpp_<0xB1>(x0); x0 = x0 * a + dpp_<0x4E>(x0); }
  }
  long t1 = clock64();
  out[threadIdx.x] = x0;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[67] = (float)(t1 - t0) / REP;
}
__global__ void k_rcp(float* out, float a) {   // dependent rcp chain
  float x0 = threadIdx.x + 1.5f;
  long t0 = clock64();
  for (int i = 0; i < REP / 8; i++) {
#pragma unroll
    for (int j = 0; j < 8; j++) x0 = __builtin_amdgcn_rcpf(x0) + a;
  }
  long t1 = clock64();
  out[threadIdx.x] = x0;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[68] = (float)(t1 - t0) / REP;
}
__global__ void k_lds(float* out, int stride) {   // dependent LDS read chain
  __shared__ int buf[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) buf[i] = (i + stride) & 4095;
  __syncthreads();
  int p = threadIdx.x;
  long t0 = clock64();
  for (int i = 0; i < 512; i++) p = buf[p];
  long t1 = clock64();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[69] = (float)(t1 - t0) / 512;
}
int main() {
  float* d; hipMalloc(&d, 128 * 4);
  for (int blocks : {1, 256}) {
    hipLaunchKernelGGL(k_indep, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f);
    hipLaunchKernelGGL(k_dep, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f);
    hipLaunchKernelGGL(k_dep2, dim3(blocks), dim3(64), 0, 0, d, 1.0001f, 0.5f);
    hipLaunchKernelGGL(k_dpp, dim3(blocks), dim3(64), 0, 0, d, 0.5f);
    hipLaunchKernelGGL(k_rcp, dim3(blocks), dim3(64), 0, 0, d, 0.5f);
    hipLaunchKernelGGL(k_lds, dim3(blocks), dim3(64), 0, 0, d, 64);
    float h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("blocks=%d  clock64 ticks per op: indep-fma %.2f  dep-fma %.2f  2-chain-fma %.2f  dep(mul+dpp-add) %.2f per pair  dep-rcp+add %.2f  dep-lds-read %.1f\n",
           blocks, h[64], h[65], h[66], h[67], h[68], h[69]);
  }
  // wall-clock calibration of clock64 (ticks per microsecond)
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); for (int i = 0; i < 50; i++) hipLaunchKernelGGL(k_dep, dim3(256), dim3(64), 0, 0, d, 1.0001f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("50 launches of k_dep(4096 dependent fma): %.1f us each\n", ms * 1000 / 50);
  return 0;
}
