// micro-benchmark: how much VALU throughput does a SECOND (third, fourth) resident wave per SIMD add for the instruction
// MIX of the physics tick (profiles/r02_ab_experiments.txt: per ~100 instructions 47 fma/mul/add, 17 v_add_f32_dpp, 7
// v_fmac_f32_dpp, 5 v_mov_b32_dpp, 10 v_pk_fma/mul, 2 rcp/rsq, 2 s_nop) -- against the same count of plain v_fma_f32.
// grid = 1024 x W single-wave workgroups (the step kernels' shape), <= 64 VGPRs so that W waves fit a SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 2000
template <int MIX> __global__ void __launch_bounds__(64) k(float* out, float a, float b) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 0.001f + i;
  f2 y[2] = {f2{a, b}, f2{b, a}}, av = {a, a};
  for (int i = 0; i < REP; i++) {
    if (MIX == 0) {
      for (int r = 0; r < 6; r++)   // 48 plain FMAs
        asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
    } else {
      // 48 instructions in the tick's proportions: 22 fma/mul/add, 8 add_dpp, 4 fmac_dpp, 2 mov_dpp, 5 pk, 1 rcp, 1 rsq (+1 s_nop), 4 more fma
      asm volatile(
          "v_fma_f32 %0, %0, %10, %11\nv_fma_f32 %1, %1, %10, %11\nv_mul_f32 %2, %2, %10\nv_add_f32 %3, %3, %11\n"
          "v_fma_f32 %4, %4, %10, %11\nv_fma_f32 %5, %5, %10, %11\nv_mul_f32 %6, %6, %10\nv_add_f32 %7, %7, %11\n"
          "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_fma_f32 %2, %2, %10, %11\nv_fma_f32 %3, %3, %10, %11\nv_fmac_f32 %4, %10, %11\nv_fmac_f32 %5, %10, %11\n"
          "v_add_f32_dpp %6, %6, %6 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %7, %7, %7 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_pk_fma_f32 %8, %8, %12, %12\nv_pk_mul_f32 %9, %9, %12\n"
          "v_fmac_f32_dpp %0, %2, %3 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_fmac_f32_dpp %1, %3, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_fma_f32 %4, %4, %10, %11\nv_mul_f32 %5, %5, %10\nv_fma_f32 %6, %6, %10, %11\nv_sub_f32 %7, %7, %11\n"
          "v_mov_b32_dpp %2, %0 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %3, %3, %3 quad_perm:[3,0,1,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %4, %4, %4 quad_perm:[1,2,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_pk_fma_f32 %9, %9, %12, %12\nv_pk_fma_f32 %8, %8, %12, %12\n"
          "v_rcp_f32 %5, %5\nv_fma_f32 %6, %6, %10, %11\nv_fma_f32 %7, %7, %10, %11\n"
          "v_fmac_f32_dpp %2, %0, %1 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_fmac_f32_dpp %3, %1, %0 row_newbcast:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_fma_f32 %4, %4, %10, %11\nv_mul_f32 %0, %0, %10\nv_fma_f32 %1, %1, %10, %11\n"
          "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_add_f32_dpp %7, %7, %7 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_rsq_f32 %5, %5\ns_nop 1\n"
          "v_mov_b32_dpp %4, %2 row_newbcast:12 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
          "v_pk_mul_f32 %8, %8, %12\nv_fma_f32 %2, %2, %10, %11\nv_fma_f32 %3, %3, %10, %11\nv_max_f32 %5, %5, %11\n"
          "v_fma_f32 %6, %6, %10, %11\nv_fma_f32 %7, %7, %10, %11\nv_fma_f32 %0, %0, %10, %11\nv_fma_f32 %1, %1, %10, %11\n"
          : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(y[0]), "+v"(y[1])
          : "v"(a), "v"(b), "v"(av));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  s += y[0].x + y[0].y + y[1].x + y[1].y;
  if (s == 12345.678f) out[0] = s;
}
int main() {
  float* d;
  (void)hipMalloc(&d, 4096);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mix = 0; mix < 2; mix++)
    for (int w = 1; w <= 4; w++) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; rep++) {
        (void)hipEventRecord(e0, 0);
        if (mix) k<1><<<dim3(1024 * w), dim3(64)>>>(d, 0.5f, 1.0f); else k<0><<<dim3(1024 * w), dim3(64)>>>(d, 0.5f, 1.0f);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double instr = 48.0 * REP;   // VALU instructions per wave (the mix has one s_nop on top)
      printf("%s  %d wave(s)/SIMD: %8.1f us -> %.2f cycles per instruction seen by a wave, %.3f wave-instr per SIMD-cycle @2.4GHz (x%.2f of one wave)\n",
             mix ? "tick mix " : "v_fma_f32", w, best * 1e3, best * 1e-3 * 2.4e9 / instr, instr * w / (best * 1e-3 * 2.4e9), 0.0);
    }
  return 0;
}
