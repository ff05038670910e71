// calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for THIS repo's access width (one dword per lane, coalesced SoA):
// a copy of a 512 MiB float array (larger than the 256 MiB Infinity Cache), known bytes = 512 MiB read + 512 MiB
// written per launch.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) and divide.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_calib_copy_dword(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) b[i] = a[i] + 1.0f;
}
int main() {
  const size_t n = (size_t)128 << 20;   // 128 Mi floats = 512 MiB
  float *a, *b;
  (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
  (void)hipMemset(a, 0, n * 4);
  for (int r = 0; r < 3; r++) k_calib_copy_dword<<<dim3((unsigned)((n + 255) / 256)), dim3(256)>>>(a, b, n);
  (void)hipDeviceSynchronize();
  printf("copied %zu MiB per launch\n", (n * 4) >> 20);
  return 0;
}
