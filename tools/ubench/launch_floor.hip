// micro-benchmark: what does a dependent kernel launch cost on this queue, independent of the kernel's work?
// Back-to-back launches on one stream of (a) an empty kernel, (b) a kernel that touches 512 registers and 17 KB of LDS
// like the step kernels, 1024 single-wave workgroups each; time per launch from one event pair around 2000 launches.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(64) k_empty(float* out) { if (out == nullptr) out[0] = 1.0f; }
__global__ void __launch_bounds__(64) k_fat(float* out, float a) {
  __shared__ float lds[66 * 64];
  float x[200];
#pragma unroll
  for (int i = 0; i < 200; i++) x[i] = a * i + threadIdx.x;
  lds[threadIdx.x] = x[7];
#pragma unroll
  for (int i = 0; i < 200; i++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[i]) : "v"(a));
  float s = lds[threadIdx.x];
#pragma unroll
  for (int i = 0; i < 200; i++) s += x[i];
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* d;
  (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int L = 2000;
  for (int which = 0; which < 2; which++) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
      (void)hipEventRecord(e0, 0);
      for (int i = 0; i < L; i++) {
        if (which == 0) k_empty<<<dim3(1024), dim3(64)>>>(d); else k_fat<<<dim3(1024), dim3(64)>>>(d, 0.5f);
      }
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%s: %.2f us per launch (1024 workgroups x 64 threads, back to back on one stream)\n",
           which ? "200-register + 17 KB LDS kernel with 200 FMAs per lane" : "empty kernel", best * 1e3 / L);
  }
  return 0;
}
