// emu_lanes.h -- TEST-ONLY W-wide lane type for the host emulation of the HIP kernels' math.
#pragma once
#include <cmath>
#include <cstring>

#include "../../paddlerobotics_amd/csrc/etg_layout.h"

namespace etg {
template <int W> struct BW { bool v[W]; };
template <int W> struct FW {
  float v[W];
  FW() {}
  explicit FW(float s) { for (int i = 0; i < W; i++) v[i] = s; }
};
#define ETG_OP2(op) template <int W> inline FW<W> operator op(FW<W> a, FW<W> b) { FW<W> r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
ETG_OP2(+) ETG_OP2(-) ETG_OP2(*) ETG_OP2(/)
#undef ETG_OP2
template <int W> inline FW<W> operator-(FW<W> a) { FW<W> r; for (int i = 0; i < W; i++) r.v[i] = -a.v[i]; return r; }
#define ETG_CMP(op) template <int W> inline BW<W> operator op(FW<W> a, FW<W> b) { BW<W> r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
ETG_CMP(<) ETG_CMP(>) ETG_CMP(<=) ETG_CMP(>=)
#undef ETG_CMP
template <int W> inline BW<W> operator&&(BW<W> a, BW<W> b) { BW<W> r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
template <int W> inline BW<W> operator||(BW<W> a, BW<W> b) { BW<W> r; for (int i = 0; i < W; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
template <int W> inline BW<W> operator!(BW<W> a) { BW<W> r; for (int i = 0; i < W; i++) r.v[i] = !a.v[i]; return r; }
template <int W> inline FW<W> sel_(BW<W> c, FW<W> a, FW<W> b) { FW<W> r; for (int i = 0; i < W; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
#define ETG_FN1(name) template <int W> inline FW<W> name(FW<W> a) { FW<W> r; for (int i = 0; i < W; i++) r.v[i] = name(a.v[i]); return r; }
ETG_FN1(fabsf_) ETG_FN1(sqrt_) ETG_FN1(rsqrt_) ETG_FN1(rsqrt_hf_) ETG_FN1(rcp_) ETG_FN1(sin_) ETG_FN1(cos_) ETG_FN1(exp_) ETG_FN1(tanh_) ETG_FN1(acos_) ETG_FN1(asin_) ETG_FN1(wrap_pi_)
#undef ETG_FN1
template <int W> inline FW<W> fminf_(FW<W> a, FW<W> b) { FW<W> r; for (int i = 0; i < W; i++) r.v[i] = fminf(a.v[i], b.v[i]); return r; }
template <int W> inline FW<W> fmaxf_(FW<W> a, FW<W> b) { FW<W> r; for (int i = 0; i < W; i++) r.v[i] = fmaxf(a.v[i], b.v[i]); return r; }
template <int W> inline FW<W> atan2_(FW<W> a, FW<W> b) { FW<W> r; for (int i = 0; i < W; i++) r.v[i] = atan2f(a.v[i], b.v[i]); return r; }
template <int W> inline BW<W> isfinite_(FW<W> a) { BW<W> r; for (int i = 0; i < W; i++) r.v[i] = std::isfinite(a.v[i]); return r; }
template <int W> inline void sincos_(FW<W> a, FW<W>& s, FW<W>& c) { for (int i = 0; i < W; i++) { s.v[i] = sinf(a.v[i]); c.v[i] = cosf(a.v[i]); } }
template <int W> inline void sincos_tick_(FW<W> a, FW<W>& s, FW<W>& c) { sincos_(a, s, c); }
typedef FW<4> F4;
typedef BW<4> B4;
typedef FW<16> F16;
typedef BW<16> B16;
}  // namespace etg
