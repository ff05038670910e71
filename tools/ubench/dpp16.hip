// probe: DPP row ops for 16-lane groups on gfx950 (row_half_mirror, row_mirror, row_newbcast) + cost
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp_(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
// DPP ctrl encodings (LLVM SIDefines.h): quad_perm 0x00-0xFF, row_shl 0x101-0x10F, row_shr 0x111-0x11F,
// row_ror 0x121-0x12F, row_mirror 0x140, row_half_mirror 0x141, row_bcast15 0x142, row_newbcast 0x150-0x15F
__device__ __forceinline__ float sum16(float x) {
  x += dpp_<0xB1>(x); x += dpp_<0x4E>(x); x += dpp_<0x141>(x); x += dpp_<0x140>(x); return x;
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  float v = (float)(1 << (l & 15)) + 65536.0f * (l >> 4);
  out[l] = sum16(v);                       // expect 65535 + 16*65536*row per row
  out[64 + l] = dpp_<0x155>((float)l);     // row_newbcast:5 -> lane 5 of each row
  out[128 + l] = dpp_<0x111>((float)l);    // row_shr:1 -> lane l-1 (0 for first lane of row with bound_ctrl=false -> old=0)
  // timing: 21 independent 16-lane reductions, repeated
  float a[21]; for (int i = 0; i < 21; i++) a[i] = v + i;
  long t0 = clock64();
  for (int rep = 0; rep < 64; rep++) {
#pragma unroll
    for (int i = 0; i < 21; i++) a[i] = sum16(a[i]) * 0.0625f;
  }
  long t1 = clock64();
  float s = 0; for (int i = 0; i < 21; i++) s += a[i];
  out[192 + l] = s;
  if (l == 0) out[256] = (float)(t1 - t0) / (64.0f * 21.0f);
}
int main() {
  float* d; (void)hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("sum16 lanes 0,15,16,63: %.0f %.0f %.0f %.0f (expect 65535, 65535, 65535+16*65536=1114111, 65535+16*3*65536)\n", h[0], h[15], h[16], h[63]);
  printf("row_newbcast:5 lanes 0,7,16,40: %.0f %.0f %.0f %.0f (expect 5 5 21 37)\n", h[64], h[71], h[80], h[104]);
  printf("row_shr:1 lanes 0,1,15,16,17: %.0f %.0f %.0f %.0f %.0f (expect 0 0 14 0 16)\n", h[128], h[129], h[143], h[144], h[145]);
  printf("ticks per 16-lane reduction (4 dpp-adds + 1 mul), 21 independent: %.1f\n", h[256]);
  return 0;
}
