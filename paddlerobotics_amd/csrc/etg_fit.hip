// etg_fit.hip -- batched Opt_with_points / LS_sol (train.py:59-110) on the GPU.
//
// The reference fits every ES candidate's ETG weights on the host with two <=1000-step gradient
// descents on a 6x20 least-squares problem (train.py:100-104); at thousands of candidates per
// generation that is the step right before reset and it dwarfs the rollout.
// Here a quad of lanes owns one (candidate, dimension) problem: lane q keeps weights 5q .. 5q+4 and its 6x5 slice of the
// shared feature matrix in registers, the 6 residuals are order-symmetric quad sums (bit-identical on the four lanes, so
// the quad takes LS_sol's stopping decision -- `while err > precision and i < 1000` -- together).  fp64 like the numpy
// reference.  (One lane per problem -- round 1 to 3 -- was 128 waves of serial gradient steps re-reading the matrix from LDS:
// 1.24 ms per 4096 candidates, 7 % of an ES generation.)
#include <hip/hip_runtime.h>

#include "../../include/etgsim.h"

extern "C" void etg_set_last_error_(const char* msg);

namespace {
constexpr int H = ETG_RBF_H, NP = 6;

constexpr int HQ = H / 4;   // weights per lane
static_assert(H % 4 == 0, "a quad splits the RBF weights evenly");

// v + (the value of lane ^ 1), then + (lane ^ 2): the quad's sum in the same order on every lane
__device__ __forceinline__ double quad_sum(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  double o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0xB1, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  v += o;
  lo = __double2loint(v); hi = __double2hiint(v);
  o = __hiloint2double(__builtin_amdgcn_mov_dpp(hi, 0x4E, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xf, 0xf, true));          // quad_perm [2,3,0,1]
  return v + o;
}

__global__ void __launch_bounds__(64) k_etg_fit(const double* __restrict__ pts, const double* __restrict__ feats,
                                                 const double* __restrict__ w0, double b0x, double b0z, double precision,
                                                 double alpha, double lamb, int max_iter, int nb, double* __restrict__ out_w,
                                                 double* __restrict__ out_b) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int prob = tid >> 2, q = tid & 3;
  const int cand = prob >> 1, dim = prob & 1;
  if (cand >= nb) return;                                    // (whole quads leave together)
  const double b0 = dim ? b0z : b0x;
  double bv[NP], a[NP][HQ], x[HQ], xa[HQ];
#pragma unroll
  for (int p = 0; p < NP; p++) {
    bv[p] = pts[((size_t)cand * NP + p) * 2 + dim] - b0;     // points - b (train.py:98)
#pragma unroll
    for (int j = 0; j < HQ; j++) a[p][j] = feats[p * H + HQ * q + j];
  }
#pragma unroll
  for (int j = 0; j < HQ; j++) xa[j] = x[j] = w0[dim * H + HQ * q + j];   // x = copy(w0); xa = the anchor
  for (int it = 0; it <= max_iter; it++) {
    double r[NP], err = 0.0;
#pragma unroll
    for (int p = 0; p < NP; p++) {
      double s = a[p][0] * x[0];
#pragma unroll
      for (int j = 1; j < HQ; j++) s = fma(a[p][j], x[j], s);
      r[p] = quad_sum(s) - bv[p];
      err = fma(r[p], r[p], err);
    }
    if (!(err > precision) || it == max_iter) break;
#pragma unroll
    for (int j = 0; j < HQ; j++) {
      double g = lamb * (x[j] - xa[j]);                      // lamb * (x - w0)
#pragma unroll
      for (int p = 0; p < NP; p++) g = fma(a[p][j], r[p], g);   // A^T (A x - b)
      x[j] = fma(-alpha, g, x[j]);
    }
  }
  double* ow = out_w + (size_t)cand * 3 * H;
#pragma unroll
  for (int j = 0; j < HQ; j++) {
    ow[(dim ? 2 : 0) * H + HQ * q + j] = x[j];
    if (dim == 0) ow[H + HQ * q + j] = 0.0;                  // y row forced to 0 (train.py:108)
  }
  if (q == 0) {
    if (dim == 0) { out_b[(size_t)cand * 3 + 0] = b0x; out_b[(size_t)cand * 3 + 1] = 0.0; }
    else out_b[(size_t)cand * 3 + 2] = b0z;
  }
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
  const int threads = 8 * nb;   // a quad per (candidate, dimension)
  hipLaunchKernelGGL(k_etg_fit, dim3((threads + 63) / 64), dim3(64), 0, (hipStream_t)stream, points, feats, w0, b0x, b0z,
                     precision, alpha, lamb, max_iter, nb, out_w, out_b);
  if (hipGetLastError() != hipSuccess) {
    etg_set_last_error_("etg_fit_etg: launch failed");
    return ETG_ERR_HIP;
  }
  return ETG_OK;
}
