#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp_(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
__global__ void k(float* out) {
  int l = threadIdx.x;
  out[l] = dpp_<0x124>((float)l);        // row_ror:4
  out[64 + l] = dpp_<0x12C>((float)l);   // row_ror:12
  out[128 + l] = dpp_<0xF9>((float)l);   // quad_perm [1,2,3,3]
  out[192 + l] = dpp_<0x90>((float)l);   // quad_perm [0,0,1,2]
  out[256 + l] = dpp_<0xD8>((float)l);   // quad_perm [0,2,1,3]
  out[320 + l] = dpp_<0x141>((float)l);  // row_half_mirror
  out[384 + l] = dpp_<0x140>((float)l);  // row_mirror
}
int main() {
  float* d; (void)hipMalloc(&d, 512 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[512]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"row_ror:4", "row_ror:12", "qp[1,2,3,3]", "qp[0,0,1,2]", "qp[0,2,1,3]", "half_mirror", "mirror"};
  for (int t = 0; t < 7; t++) { printf("%-12s lane0..19 <-", names[t]); for (int l = 0; l < 20; l++) printf(" %2.0f", h[64 * t + l]); printf("\n"); }
  return 0;
}
