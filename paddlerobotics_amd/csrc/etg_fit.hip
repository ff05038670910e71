// etg_fit.hip -- batched Opt_with_points / LS_sol (train.py:59-110) on the GPU.
//
// The reference fits every ES candidate's ETG weights on the host with two <=1000-step gradient
// descents on a 6x20 least-squares problem (train.py:100-104); at thousands of candidates per
// generation that is the step right before reset and it dwarfs the rollout.
// Here one lane owns one (candidate, dimension) problem: its 20 weights live in registers, the shared
// 6x20 feature matrix is broadcast from LDS, and the loop keeps LS_sol's stopping rule
// (`while err > precision and i < 1000`) per lane.  fp64 like the numpy reference.
#include <hip/hip_runtime.h>

#include "../../include/etgsim.h"

extern "C" void etg_set_last_error_(const char* msg);

namespace {
constexpr int H = ETG_RBF_H, NP = 6;

__global__ void __launch_bounds__(64) k_etg_fit(const double* __restrict__ pts, const double* __restrict__ feats,
                                                 const double* __restrict__ w0, double b0x, double b0z, double precision,
                                                 double alpha, double lamb, int max_iter, int nb, double* __restrict__ out_w,
                                                 double* __restrict__ out_b) {
  __shared__ double sA[NP * H];
  __shared__ double sW[2 * H];
  for (int i = threadIdx.x; i < NP * H; i += blockDim.x) sA[i] = feats[i];
  for (int i = threadIdx.x; i < 2 * H; i += blockDim.x) sW[i] = w0[i];
  __syncthreads();
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int cand = tid >> 1, dim = tid & 1;
  if (cand >= nb) return;
  const double b0 = dim ? b0z : b0x;
  double bv[NP], x[H];
#pragma unroll
  for (int p = 0; p < NP; p++) bv[p] = pts[((size_t)cand * NP + p) * 2 + dim] - b0;   // points - b (train.py:98)
#pragma unroll
  for (int h = 0; h < H; h++) x[h] = sW[dim * H + h];                                  // x = copy(w0)
  for (int it = 0; it <= max_iter; it++) {
    double r[NP], err = 0.0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      double s = -bv[p];
#pragma unroll
      for (int h = 0; h < H; h++) s = fma(sA[p * H + h], x[h], s);
      r[p] = s;
      err = fma(s, s, err);
    }
    if (!(err > precision) || it == max_iter) break;
#pragma unroll
    for (int h = 0; h < H; h++) {
      double g = lamb * (x[h] - sW[dim * H + h]);                                        // lamb * (x - w0)
#pragma unroll
      for (int p = 0; p < NP; p++) g = fma(sA[p * H + h], r[p], g);                      // A^T (A x - b)
      x[h] = fma(-alpha, g, x[h]);
    }
  }
  double* ow = out_w + (size_t)cand * 3 * H;
#pragma unroll
  for (int h = 0; h < H; h++) {
    ow[(dim ? 2 : 0) * H + h] = x[h];
    if (dim == 0) ow[H + h] = 0.0;                                                       // y row forced to 0 (train.py:108)
  }
  if (dim == 0) { out_b[(size_t)cand * 3 + 0] = b0x; out_b[(size_t)cand * 3 + 1] = 0.0; }
  else out_b[(size_t)cand * 3 + 2] = b0z;
}
}  // namespace

// points [nb,6,2], feats [6,20], w0 [2,20] (x row, z row), all float64 device pointers;
// out_w [nb,3,20], out_b [nb,3] float64.  Mirrors Opt_with_points(ETG, ETG_T, points, b0, w0, precision, lamb).
extern "C" int etg_fit_etg(const double* points, int nb, const double* feats, const double* w0, double b0x, double b0z,
                           double precision, double alpha, double lamb, int max_iter, double* out_w, double* out_b,
                           void* stream) {
  if (!points || !feats || !w0 || !out_w || !out_b || nb <= 0 || max_iter < 0) {
    etg_set_last_error_("etg_fit_etg: bad arguments");
    return ETG_ERR_BAD_ARG;
  }
  // launch on the device that owns the buffers (the caller's current device may be another one): the other entry
  // points bind their handle's device the same way
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, points) != hipSuccess || hipSetDevice(attr.device) != hipSuccess) {
    (void)hipGetLastError();
    etg_set_last_error_("etg_fit_etg: points is not a device pointer");
    return ETG_ERR_BAD_ARG;
  }
  const int threads = 2 * nb;
  hipLaunchKernelGGL(k_etg_fit, dim3((threads + 63) / 64), dim3(64), 0, (hipStream_t)stream, points, feats, w0, b0x, b0z,
                     precision, alpha, lamb, max_iter, nb, out_w, out_b);
  if (hipGetLastError() != hipSuccess) {
    etg_set_last_error_("etg_fit_etg: launch failed");
    return ETG_ERR_HIP;
  }
  return ETG_OK;
}
