// micro-benchmark 2: single-wave VALU issue cost on gfx950 with hand-written instruction streams
// (the compiler packs independent FMAs into v_pk_fma_f32, which hid the real numbers in issue_rate.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 1024
#define R8(x) x x x x x x x x
// CH independent dependent-chains of v_fma_f32, round-robin
template <int CH> __global__ void k_chains(float* out, float a, float b, int slot) {
  float x[8];
  for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;
  long t0 = clock64();
  for (int i = 0; i < REP; i++) {
    if (CH == 1) asm volatile(R8("v_fma_f32 %0, %0, %8, %9\n") : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
    if (CH == 2) asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\n" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
    if (CH == 3) asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\n" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
    if (CH == 4) asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
    if (CH == 8) asm volatile("v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\nv_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b));
  }
  long t1 = clock64();
  float s = 0;
  for (int i = 0; i < 8; i++) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 + slot] = (float)(t1 - t0) / (REP * 8);
}
// other opcodes, 8 independent / 1 dependent
#define BODY8(OP) OP " %0, %0, %8\n" OP " %1, %1, %8\n" OP " %2, %2, %8\n" OP " %3, %3, %8\n" OP " %4, %4, %8\n" OP " %5, %5, %8\n" OP " %6, %6, %8\n" OP " %7, %7, %8\n"
#define BODY1(OP) R8(OP " %0, %0, %8\n")
#define KERN(NAME, BODY)                                                                                          \
  __global__ void NAME(float* out, float a, int slot) {                                                            \
    float x[8];                                                                                                    \
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x + i;                                                            \
    long t0 = clock64();                                                                                           \
    for (int i = 0; i < REP; i++)                                                                                  \
      asm volatile(BODY : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a) : "s20", "s21", "vcc"); \
    long t1 = clock64();                                                                                           \
    float s = 0;                                                                                                   \
    for (int i = 0; i < 8; i++) s += x[i];                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                                \
    if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 + slot] = (float)(t1 - t0) / (REP * 8);                     \
  }
KERN(k_mul_i, BODY8("v_mul_f32"))
KERN(k_mul_d, BODY1("v_mul_f32"))
KERN(k_add_i, BODY8("v_add_f32"))
KERN(k_dppadd_i, "v_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %4, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %5, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %6, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %7, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERN(k_dppadd_d, R8("s_nop 1\nv_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"))
KERN(k_mov_i, "v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n")
KERN(k_cndmask_i, "v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n")
KERN(k_rcp_i, "v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7\n")
KERN(k_cnd64_i, "s_mov_b64 s[20:21], 0x5555\nv_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\nv_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_cndmask_b32_e64 %5, %5, %8, s[20:21]\nv_cndmask_b32_e64 %6, %6, %8, s[20:21]\nv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n")
KERN(k_mul_sgpr, "s_mov_b32 s20, 0x3f000000\nv_mul_f32 %0, s20, %0\nv_mul_f32 %1, s20, %1\nv_mul_f32 %2, s20, %2\nv_mul_f32 %3, s20, %3\nv_mul_f32 %4, s20, %4\nv_mul_f32 %5, s20, %5\nv_mul_f32 %6, s20, %6\nv_mul_f32 %7, s20, %7\n")
KERN(k_cmp_i, "v_cmp_gt_f32 vcc, %0, %8\nv_cmp_gt_f32 vcc, %1, %8\nv_cmp_gt_f32 vcc, %2, %8\nv_cmp_gt_f32 vcc, %3, %8\nv_cmp_gt_f32 vcc, %4, %8\nv_cmp_gt_f32 vcc, %5, %8\nv_cmp_gt_f32 vcc, %6, %8\nv_cmp_gt_f32 vcc, %7, %8\n")
KERN(k_cmpcnd, "v_cmp_gt_f32 vcc, %0, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_gt_f32 vcc, %2, %8\nv_cndmask_b32 %3, %3, %8, vcc\nv_cmp_gt_f32 vcc, %4, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_gt_f32 vcc, %6, %8\nv_cndmask_b32 %7, %7, %8, vcc\n")
KERN(k_fmac_i, "v_fmac_f32 %0, %8, %8\nv_fmac_f32 %1, %8, %8\nv_fmac_f32 %2, %8, %8\nv_fmac_f32 %3, %8, %8\nv_fmac_f32 %4, %8, %8\nv_fmac_f32 %5, %8, %8\nv_fmac_f32 %6, %8, %8\nv_fmac_f32 %7, %8, %8\n")
KERN(k_fma_const, "v_fma_f32 %0, %0, 0.5, 1.0\nv_fma_f32 %1, %1, 0.5, 1.0\nv_fma_f32 %2, %2, 0.5, 1.0\nv_fma_f32 %3, %3, 0.5, 1.0\nv_fma_f32 %4, %4, 0.5, 1.0\nv_fma_f32 %5, %5, 0.5, 1.0\nv_fma_f32 %6, %6, 0.5, 1.0\nv_fma_f32 %7, %7, 0.5, 1.0\n")
KERN(k_nop0, R8("s_nop 0\n"))
KERN(k_nop1, R8("s_nop 1\n"))
KERN(k_mulnop, "v_mul_f32 %0, %0, %8\ns_nop 0\nv_mul_f32 %1, %1, %8\ns_nop 0\nv_mul_f32 %2, %2, %8\ns_nop 0\nv_mul_f32 %3, %3, %8\ns_nop 0\n")
KERN(k_dpp_mov_d, R8("s_nop 1\nv_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"))
KERN(k_dppadd_2ch, "v_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERN(k_dppadd_3ch, "v_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\nv_add_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
// packed: 4 independent v_pk_fma_f32 on register pairs (8 FMAs), and a dependent chain
__global__ void k_pk_i(float* out, float a, int slot) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x[4], av = {a, a};
  for (int i = 0; i < 4; i++) x[i] = f2{(float)threadIdx.x + i, (float)threadIdx.x - i};
  long t0 = clock64();
  for (int i = 0; i < REP; i++)
    asm volatile("v_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\nv_pk_fma_f32 %0, %0, %4, %4\nv_pk_fma_f32 %1, %1, %4, %4\nv_pk_fma_f32 %2, %2, %4, %4\nv_pk_fma_f32 %3, %3, %4, %4\n" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(av));
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x[0].x + x[1].y + x[2].x + x[3].y;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 + slot] = (float)(t1 - t0) / (REP * 8);
}
__global__ void k_pk_d(float* out, float a, int slot) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 x = {(float)threadIdx.x, 1.0f}, av = {a, a};
  long t0 = clock64();
  for (int i = 0; i < REP; i++) asm volatile(R8("v_pk_fma_f32 %0, %0, %1, %1\n") : "+v"(x) : "v"(av));
  long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x.x + x.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[4096 + slot] = (float)(t1 - t0) / (REP * 8);
}
int main() {
  float* d;
  hipMalloc(&d, 8192 * 4);
  typedef void (*KF)(float*, float, int);
  struct { const char* name; KF f; } extra[] = {
      {"mul independent", k_mul_i}, {"mul dependent", k_mul_d}, {"add independent", k_add_i}, {"add_dpp independent", k_dppadd_i},
      {"s_nop1+add_dpp dependent (per pair)", k_dppadd_d}, {"add_dpp 2 chains (no nop)", k_dppadd_2ch}, {"add_dpp 3 chains (no nop)", k_dppadd_3ch},
      {"s_nop1+mov_dpp dependent (per pair)", k_dpp_mov_d},
      {"mov independent", k_mov_i}, {"cndmask vcc independent", k_cndmask_i}, {"cndmask e64 s[20:21]", k_cnd64_i},
      {"mul with SGPR operand", k_mul_sgpr}, {"v_cmp -> vcc", k_cmp_i}, {"v_cmp+cndmask alternating", k_cmpcnd}, {"fmac independent", k_fmac_i},
      {"fma inline constants", k_fma_const}, {"s_nop 0", k_nop0}, {"s_nop 1", k_nop1}, {"mul + s_nop 0 (per pair)", k_mulnop}, {"rcp independent", k_rcp_i},
      {"pk_fma 4 chains (per instr)", k_pk_i}, {"pk_fma dependent (per instr)", k_pk_d}};
  setvbuf(stdout, nullptr, _IONBF, 0);
  for (int waves = 1; waves <= 4; waves++) {   // waves per SIMD: block of 256*waves threads on one CU
    dim3 g(1), b(256 * waves);
    hipMemset(d, 0, 8192 * 4);
    printf("== %d wave(s) per SIMD: cycles per instruction as seen by wave 0\n", waves);
    k_chains<1><<<g, b>>>(d, 0.5f, 1.0f, 0); k_chains<2><<<g, b>>>(d, 0.5f, 1.0f, 1); k_chains<8><<<g, b>>>(d, 0.5f, 1.0f, 2);
    float h3[3];
    hipMemcpy(h3, d + 4096, 12, hipMemcpyDeviceToHost);
    printf("  %-40s %6.2f\n  %-40s %6.2f\n  %-40s %6.2f\n", "fma dependent", h3[0], "fma 2 chains", h3[1], "fma 8 chains", h3[2]);
    for (unsigned i = 0; i < sizeof(extra) / sizeof(extra[0]); i++) {
      extra[i].f<<<g, b>>>(d, 0.5f, 3);
      float h = 0;
      hipMemcpy(&h, d + 4096 + 3, 4, hipMemcpyDeviceToHost);
      printf("  %-40s %6.2f\n", extra[i].name, h);
    }
  }
  return 0;
}
