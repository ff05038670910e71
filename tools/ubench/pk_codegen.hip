#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
struct W2 { f2 ax, ay, az; };   // (a.x,l.x), (a.y,l.y), (a.z,l.z)
__device__ __forceinline__ W2 axpy(float s, W2 x, W2 y) {
  f2 ss = {s, s};
  return {__builtin_elementwise_fma(ss, x.ax, y.ax), __builtin_elementwise_fma(ss, x.ay, y.ay), __builtin_elementwise_fma(ss, x.az, y.az)};
}
// LDL-like rank-1 update on pairs: s[j] -= l * t[j]
extern "C" __global__ void k(float* o, const float* in) {
  int i = threadIdx.x;
  W2 x = {{in[i], in[i + 64]}, {in[i + 128], in[i + 192]}, {in[i + 256], in[i + 320]}};
  W2 y = {{in[i + 384], in[i + 448]}, {in[i + 512], in[i + 576]}, {in[i + 640], in[i + 704]}};
  float s = in[i + 768], t = in[i + 832];
  W2 r = axpy(s, x, y);
  r = axpy(t, r, x);          // chained
  // mix: scalar from inside a pair (r.ax.y) times pair
  r = axpy(r.ax.y, y, r);
  // dot of 6: pairs then horizontal
  f2 d = r.ax * x.ax + r.ay * x.ay + r.az * x.az;
  o[i] = d.x + d.y; o[i + 64] = r.ay.x; o[i + 128] = r.az.y;
}
