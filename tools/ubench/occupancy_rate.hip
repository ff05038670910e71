// micro-benchmark: whole-chip VALU throughput vs resident waves per SIMD (gfx950).
// Every wave runs the same stream of NI v_fma_f32 (8 independent chains); grid = 256 CUs x W blocks of 256
// threads, so W = waves per SIMD when the dispatcher spreads blocks evenly.  Reports kernel time and the
// achieved wave-instructions per SIMD-cycle at the nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 4096
template <int PK> __global__ void __launch_bounds__(256) k(float* out, float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  f2 y[4], av = {a, a};
  for (int i = 0; i < 4; i++) y[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
  for (int i = 0; i < REP; i++) {
    if (PK == 0)
      asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                   : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
    else
      asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\nv_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n"
                   : "+v"(y[0]), "+v"(y[1]), "+v"(y[2]), "+v"(y[3]) : "v"(av));
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  for (int i = 0; i < 4; i++) s += y[i].x + y[i].y;
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* d;
  (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int pk = 0; pk < 2; pk++)
    for (int w = 1; w <= 8; w *= 2) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        (void)hipEventRecord(e0, 0);
        if (pk) k<1><<<dim3(256 * w), dim3(256)>>>(d, 0.5f, 1.0f); else k<0><<<dim3(256 * w), dim3(256)>>>(d, 0.5f, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      double instr = 8.0 * REP;   // per wave
      printf("%s  %d wave(s)/SIMD: %8.1f us  -> %.2f cycles per wave-instruction (per wave), %.3f wave-instr per SIMD-cycle @2.4GHz\n",
             pk ? "v_pk_fma_f32" : "v_fma_f32   ", w, best * 1e3, best * 1e-3 * 2.4e9 / instr, instr * w / (best * 1e-3 * 2.4e9));
    }
  return 0;
}
