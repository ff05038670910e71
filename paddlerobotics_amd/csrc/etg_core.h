// etg_core.h -- per-robot math of the fused env.step()/reset() kernels.
//
// Mapping (DESIGN.md "kernel mapping"): ONE ROBOT = ONE QUAD = 4 ADJACENT LANES, lane l
// owns leg l (FR, FL, RR, RL); a wave64 carries 16 robots.  Everything a leg owns (its
// 3 joints, 3 links, foot contact, ETG/IK/PD) lives in that lane's registers; the
// floating base (pose, twist, 6x6 articulated inertia) is REPLICATED in the 4 lanes and
// kept bit-identical by combining per-leg contributions with order-symmetric quad
// reductions (DPP quad_perm on gfx950).
//
// The code is written against an abstract lane scalar F:
//   - HIP build (etg_kernels.hip): F = float, quad ops = DPP, this is the product.
//   - tests/emu build: F = a 4-wide struct executed on the host, used ONLY by tests to
//     debug this exact source against the oracle without a GPU.
//
// Model (same as oracle/etgsim_oracle.cpp, derived independently there with generic
// spatial algebra): floating base + 4 x (hip-x, thigh-y, calf-y) revolute chains, all
// quantities expressed in the base frame about the base origin (trunk COM);
//   M(q)[a_b; qdd] + C = [0; tau] + J^T f
// solved by block elimination on the arrow structure of M (leg blocks 3x3, base Schur
// complement 6x6), contacts by projected Gauss-Seidel over the 4 feet in lane order.
#pragma once

#include <type_traits>

#include "etg_layout.h"

namespace etg {

// ------------------------------------------------------------------ tiny vector algebra
template <class F> struct V3 { F x, y, z; };
template <class F> ETG_HD V3<F> operator+(V3<F> a, V3<F> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class F> ETG_HD V3<F> operator-(V3<F> a, V3<F> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class F> ETG_HD V3<F> operator*(F s, V3<F> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class F> ETG_HD F dot(V3<F> a, V3<F> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <class F> ETG_HD V3<F> cross(V3<F> a, V3<F> b) {
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class F> struct S3 { F xx, yy, zz, xy, xz, yz; };  // symmetric 3x3
template <class F> ETG_HD V3<F> mul(S3<F> s, V3<F> v) {
  return {s.xx * v.x + s.xy * v.y + s.xz * v.z, s.xy * v.x + s.yy * v.y + s.yz * v.z,
          s.xz * v.x + s.yz * v.y + s.zz * v.z};
}
template <class F> struct SV { V3<F> a, l; };  // spatial vector: (angular, linear) / (moment, force)
template <class F> ETG_HD SV<F> operator+(SV<F> p, SV<F> q) { return {p.a + q.a, p.l + q.l}; }
template <class F> ETG_HD SV<F> operator-(SV<F> p, SV<F> q) { return {p.a - q.a, p.l - q.l}; }
template <class F> ETG_HD SV<F> operator*(F s, SV<F> p) { return {s * p.a, s * p.l}; }
template <class F> ETG_HD F dot(SV<F> p, SV<F> q) { return dot(p.a, q.a) + dot(p.l, q.l); }
template <class F> ETG_HD SV<F> cmul(SV<F> p, SV<F> q) {  // component-wise product
  return {{p.a.x * q.a.x, p.a.y * q.a.y, p.a.z * q.a.z}, {p.l.x * q.l.x, p.l.y * q.l.y, p.l.z * q.l.z}};
}
template <class F> ETG_HD F comp(const SV<F>& p, int i) {  // compile-time i after unrolling
  return i == 0 ? p.a.x : i == 1 ? p.a.y : i == 2 ? p.a.z : i == 3 ? p.l.x : i == 4 ? p.l.y : p.l.z;
}
// motion x motion, motion x* force
template <class F> ETG_HD SV<F> crm(SV<F> v, SV<F> m) { return {cross(v.a, m.a), cross(v.a, m.l) + cross(v.l, m.a)}; }
template <class F> ETG_HD SV<F> crf(SV<F> v, SV<F> f) { return {cross(v.a, f.a) + cross(v.l, f.l), cross(v.a, f.l)}; }

#if defined(__HIPCC__)
// ---- packed fp32 forms of the spatial-vector algebra (gfx950 v_pk_{mul,add,fma}_f32: two lanes' worth of FMA per
// issue slot).  Every SV is treated as three (angular_k, linear_k) pairs -- ONE pairing everywhere, so the register
// allocator can keep each pair in an aligned register pair without copies; scalar factors are broadcast by the
// instruction's op_sel bits (tools/ubench/pk_codegen.hip).  Plain overloads for float: call sites are unchanged and
// the test emulator (F = lane struct) keeps using the scalar templates above; results differ by rounding only.
typedef float pk2 __attribute__((ext_vector_type(2)));
ETG_HD pk2 pk_(float a, float b) { pk2 r = {a, b}; return r; }
ETG_HD pk2 pkfma_(pk2 a, pk2 b, pk2 c) { return __builtin_elementwise_fma(a, b, c); }
ETG_HD SV<float> unpk_(pk2 x, pk2 y, pk2 z) { return {{x.x, y.x, z.x}, {x.y, y.y, z.y}}; }
ETG_HD SV<float> operator+(SV<float> p, SV<float> q) {
  return unpk_(pk_(p.a.x, p.l.x) + pk_(q.a.x, q.l.x), pk_(p.a.y, p.l.y) + pk_(q.a.y, q.l.y), pk_(p.a.z, p.l.z) + pk_(q.a.z, q.l.z));
}
ETG_HD SV<float> operator-(SV<float> p, SV<float> q) {
  return unpk_(pk_(p.a.x, p.l.x) - pk_(q.a.x, q.l.x), pk_(p.a.y, p.l.y) - pk_(q.a.y, q.l.y), pk_(p.a.z, p.l.z) - pk_(q.a.z, q.l.z));
}
ETG_HD SV<float> operator*(float s, SV<float> p) {
  const pk2 ss = pk_(s, s);
  return unpk_(ss * pk_(p.a.x, p.l.x), ss * pk_(p.a.y, p.l.y), ss * pk_(p.a.z, p.l.z));
}
ETG_HD SV<float> cmul(SV<float> p, SV<float> q) {
  return unpk_(pk_(p.a.x, p.l.x) * pk_(q.a.x, q.l.x), pk_(p.a.y, p.l.y) * pk_(q.a.y, q.l.y), pk_(p.a.z, p.l.z) * pk_(q.a.z, q.l.z));
}
ETG_HD float dot(SV<float> p, SV<float> q) {
  pk2 d = pk_(p.a.x, p.l.x) * pk_(q.a.x, q.l.x);
  d = pkfma_(pk_(p.a.y, p.l.y), pk_(q.a.y, q.l.y), d);
  d = pkfma_(pk_(p.a.z, p.l.z), pk_(q.a.z, q.l.z), d);
  return d.x + d.y;
}
// (v.a x m.a, v.a x m.l + v.l x m.a): the v.a x (m.a | m.l) part is packed, v.l x m.a lands on the linear halves
ETG_HD SV<float> crm(SV<float> v, SV<float> m) {
  const pk2 mx = pk_(m.a.x, m.l.x), my = pk_(m.a.y, m.l.y), mz = pk_(m.a.z, m.l.z);
  const pk2 vx = pk_(v.a.x, v.a.x), vy = pk_(v.a.y, v.a.y), vz = pk_(v.a.z, v.a.z);
  pk2 rx = pkfma_(vy, mz, -(vz * my)), ry = pkfma_(vz, mx, -(vx * mz)), rz = pkfma_(vx, my, -(vy * mx));
  rx.y += v.l.y * m.a.z - v.l.z * m.a.y;
  ry.y += v.l.z * m.a.x - v.l.x * m.a.z;
  rz.y += v.l.x * m.a.y - v.l.y * m.a.x;
  return unpk_(rx, ry, rz);
}
// (v.a x f.a + v.l x f.l, v.a x f.l): v.l x f.l lands on the angular halves
ETG_HD SV<float> crf(SV<float> v, SV<float> f) {
  const pk2 fx = pk_(f.a.x, f.l.x), fy = pk_(f.a.y, f.l.y), fz = pk_(f.a.z, f.l.z);
  const pk2 vx = pk_(v.a.x, v.a.x), vy = pk_(v.a.y, v.a.y), vz = pk_(v.a.z, v.a.z);
  pk2 rx = pkfma_(vy, fz, -(vz * fy)), ry = pkfma_(vz, fx, -(vx * fz)), rz = pkfma_(vx, fy, -(vy * fx));
  rx.x += v.l.y * f.l.z - v.l.z * f.l.y;
  ry.x += v.l.z * f.l.x - v.l.x * f.l.z;
  rz.x += v.l.x * f.l.y - v.l.y * f.l.x;
  return unpk_(rx, ry, rz);
}
#endif

// rigid-body inertia about the base origin, in base axes
template <class F> struct RBI { F m; V3<F> h; S3<F> I; };
template <class F> ETG_HD RBI<F> operator+(RBI<F> a, RBI<F> b) {
  return {a.m + b.m, a.h + b.h,
          {a.I.xx + b.I.xx, a.I.yy + b.I.yy, a.I.zz + b.I.zz, a.I.xy + b.I.xy, a.I.xz + b.I.xz, a.I.yz + b.I.yz}};
}
template <class F> ETG_HD SV<F> apply(const RBI<F>& I, SV<F> v) {
  return {mul(I.I, v.a) + cross(I.h, v.l), I.m * v.l - cross(I.h, v.a)};
}
// link frame given by its axes (ex, ey, ez) in base coordinates
template <class F> struct Fr { V3<F> ex, ey, ez; };
template <class F> ETG_HD V3<F> rot(const Fr<F>& R, V3<F> v) { return v.x * R.ex + v.y * R.ey + v.z * R.ez; }
template <class F> ETG_HD RBI<F> link_inertia(F m, V3<F> com, S3<F> Ic, const Fr<F>& R, V3<F> o) {
  V3<F> c = o + rot(R, com);
  // R Ic R^T
  V3<F> tx = Ic.xx * R.ex + Ic.xy * R.ey + Ic.xz * R.ez;
  V3<F> ty = Ic.xy * R.ex + Ic.yy * R.ey + Ic.yz * R.ez;
  V3<F> tz = Ic.xz * R.ex + Ic.yz * R.ey + Ic.zz * R.ez;
  S3<F> I;
  I.xx = R.ex.x * tx.x + R.ey.x * ty.x + R.ez.x * tz.x;
  I.yy = R.ex.y * tx.y + R.ey.y * ty.y + R.ez.y * tz.y;
  I.zz = R.ex.z * tx.z + R.ey.z * ty.z + R.ez.z * tz.z;
  I.xy = R.ex.x * tx.y + R.ey.x * ty.y + R.ez.x * tz.y;
  I.xz = R.ex.x * tx.z + R.ey.x * ty.z + R.ez.x * tz.z;
  I.yz = R.ex.y * tx.z + R.ey.y * ty.z + R.ez.y * tz.z;
  // parallel axis to the base origin
  I.xx = I.xx + m * (c.y * c.y + c.z * c.z);
  I.yy = I.yy + m * (c.x * c.x + c.z * c.z);
  I.zz = I.zz + m * (c.x * c.x + c.y * c.y);
  I.xy = I.xy - m * c.x * c.y;
  I.xz = I.xz - m * c.x * c.z;
  I.yz = I.yz - m * c.y * c.z;
  return {m, m * c, I};
}

// link_inertia16: what the 16-lane tick calls (one link per lane).  Generic F: the scalar routine above.
template <class F> ETG_HD RBI<F> link_inertia16(F m, V3<F> com, S3<F> Ic, const Fr<F>& R, V3<F> o) { return link_inertia(m, com, Ic, R, o); }
#if defined(__HIPCC__)
// float: the 3-vector chains inside work on packed (x, y) pairs + z; inputs and outputs are scalars, so the pairing
// stays local to this function (a global (x, y) pairing of all V3 algebra costs more register copies than it saves,
// and so does this routine in the 4-lane tick, where one lane carries three links: 50 -> 89 AGPR spill moves)
ETG_HD RBI<float> link_inertia16(float m, V3<float> com, S3<float> Ic, const Fr<float>& R, V3<float> o) {
  const pk2 ex = pk_(R.ex.x, R.ex.y), ey = pk_(R.ey.x, R.ey.y), ez = pk_(R.ez.x, R.ez.y);
  // c = o + R com
  pk2 cxy = pkfma_(pk_(com.x, com.x), ex, pk_(o.x, o.y));
  cxy = pkfma_(pk_(com.y, com.y), ey, cxy);
  cxy = pkfma_(pk_(com.z, com.z), ez, cxy);
  const float cz = o.z + com.x * R.ex.z + com.y * R.ey.z + com.z * R.ez.z;
  // t_k = Ic row k in base axes: (xy pair, z)
  pk2 txy = pkfma_(pk_(Ic.xz, Ic.xz), ez, pkfma_(pk_(Ic.xy, Ic.xy), ey, pk_(Ic.xx, Ic.xx) * ex));
  pk2 tyy = pkfma_(pk_(Ic.yz, Ic.yz), ez, pkfma_(pk_(Ic.yy, Ic.yy), ey, pk_(Ic.xy, Ic.xy) * ex));
  pk2 tzy = pkfma_(pk_(Ic.zz, Ic.zz), ez, pkfma_(pk_(Ic.yz, Ic.yz), ey, pk_(Ic.xz, Ic.xz) * ex));
  const float txz = Ic.xx * R.ex.z + Ic.xy * R.ey.z + Ic.xz * R.ez.z;
  const float tyz = Ic.xy * R.ex.z + Ic.yy * R.ey.z + Ic.yz * R.ez.z;
  const float tzz = Ic.xz * R.ex.z + Ic.yz * R.ey.z + Ic.zz * R.ez.z;
  // rows of R Ic R^T: row x -> (xx, xy), xz; row y -> (yx, yy), yz; zz
  const pk2 rx = pkfma_(pk_(R.ez.x, R.ez.x), tzy, pkfma_(pk_(R.ey.x, R.ey.x), tyy, pk_(R.ex.x, R.ex.x) * txy));
  const pk2 ry = pkfma_(pk_(R.ez.y, R.ez.y), tzy, pkfma_(pk_(R.ey.y, R.ey.y), tyy, pk_(R.ex.y, R.ex.y) * txy));
  S3<float> I;
  I.xx = rx.x; I.xy = rx.y; I.yy = ry.y;
  I.xz = R.ex.x * txz + R.ey.x * tyz + R.ez.x * tzz;
  I.yz = R.ex.y * txz + R.ey.y * tyz + R.ez.y * tzz;
  I.zz = R.ex.z * txz + R.ey.z * tyz + R.ez.z * tzz;
  const float cx = cxy.x, cy = cxy.y;
  I.xx = I.xx + m * (cy * cy + cz * cz);
  I.yy = I.yy + m * (cx * cx + cz * cz);
  I.zz = I.zz + m * (cx * cx + cy * cy);
  I.xy = I.xy - m * cx * cy;
  I.xz = I.xz - m * cx * cz;
  I.yz = I.yz - m * cy * cz;
  const pk2 hxy = pk_(m, m) * cxy;
  return {m, {hxy.x, hxy.y, m * cz}, I};
}
#endif

// ------------------------------------------------------------------ per-lane parameters / state
// Per-lane parameters are NOT held in registers: the context serves them from a
// lane-private LDS column (GPU) / the par array (emulator) at the point of use:
//   c.par(k) with k one of the PR_* field indices of etg_layout.h.
template <class F, class Ctx> ETG_HD V3<F> par3(const Ctx& c, int k) { return {c.par(k), c.par(k + 1), c.par(k + 2)}; }
template <class F, class Ctx> ETG_HD S3<F> par_s3(const Ctx& c, int k) {
  return {c.par(k), c.par(k + 1), c.par(k + 2), c.par(k + 3), c.par(k + 4), c.par(k + 5)};
}

template <class F> struct LaneState {
  V3<F> p;            // base origin, world
  F qx, qy, qz, qw;   // base orientation (xyzw, body -> world)
  V3<F> wb, vb;       // base twist in base coordinates
  F q[3], qd[3], lam[3];
  F lamb;             // the normal impulse of this leg's body contact in the last tick (its warm start: K.warmstart_b)
  F contact;          // 1 if this lane's foot carried load in the last tick
  F energy;           // sum |tau qd| dt of this lane's joints since the step began
  int sweeps;         // PGS sweeps this wave executed since the step began (wave-uniform; not stored)
};

template <class F, class Ctx> ETG_HD LaneState<F> load_state(const Ctx& c, const float* base, const float* leg) {
  LaneState<F> L;
  L.p.x = c.ld_env(base, BS_PX); L.p.y = c.ld_env(base, BS_PY); L.p.z = c.ld_env(base, BS_PZ);
  L.qx = c.ld_env(base, BS_QX); L.qy = c.ld_env(base, BS_QY); L.qz = c.ld_env(base, BS_QZ); L.qw = c.ld_env(base, BS_QW);
  L.wb.x = c.ld_env(base, BS_WX); L.wb.y = c.ld_env(base, BS_WY); L.wb.z = c.ld_env(base, BS_WZ);
  L.vb.x = c.ld_env(base, BS_VX); L.vb.y = c.ld_env(base, BS_VY); L.vb.z = c.ld_env(base, BS_VZ);
  for (int j = 0; j < 3; j++) {
    L.q[j] = c.ld_lane(leg, LG_Q + j); L.qd[j] = c.ld_lane(leg, LG_QD + j); L.lam[j] = c.ld_lane(leg, LG_LAM + j);
  }
  L.lamb = c.ld_lane(leg, LG_LAMB);
  L.contact = c.ld_lane(leg, LG_CONTACT);
  L.energy = F(0.0f);
  L.sweeps = 0;
  return L;
}
template <class F, class Ctx> ETG_HD void store_state(const Ctx& c, float* base, float* leg, const LaneState<F>& L) {
  c.st_env(base, BS_PX, L.p.x); c.st_env(base, BS_PY, L.p.y); c.st_env(base, BS_PZ, L.p.z);
  c.st_env(base, BS_QX, L.qx); c.st_env(base, BS_QY, L.qy); c.st_env(base, BS_QZ, L.qz); c.st_env(base, BS_QW, L.qw);
  c.st_env(base, BS_WX, L.wb.x); c.st_env(base, BS_WY, L.wb.y); c.st_env(base, BS_WZ, L.wb.z);
  c.st_env(base, BS_VX, L.vb.x); c.st_env(base, BS_VY, L.vb.y); c.st_env(base, BS_VZ, L.vb.z);
  for (int j = 0; j < 3; j++) {
    c.st_lane(leg, LG_Q + j, L.q[j]); c.st_lane(leg, LG_QD + j, L.qd[j]); c.st_lane(leg, LG_LAM + j, L.lam[j]);
  }
  c.st_lane(leg, LG_LAMB, L.lamb);
  c.st_lane(leg, LG_CONTACT, L.contact);
}

// rows of the base rotation matrix R (body -> world): r0 = R^T x_w etc.
template <class F> struct Rows { V3<F> r0, r1, r2; };
template <class F> ETG_HD Rows<F> quat_rows(F x, F y, F z, F w) {
  F two(2.0f), one(1.0f);
  Rows<F> R;
  R.r0 = {one - two * (y * y + z * z), two * (x * y - z * w), two * (x * z + y * w)};
  R.r1 = {two * (x * y + z * w), one - two * (x * x + z * z), two * (y * z - x * w)};
  R.r2 = {two * (x * z - y * w), two * (y * z + x * w), one - two * (x * x + y * y)};
  return R;
}

// in-place LDL^T of a symmetric 6x6 stored as lower triangle s[i*(i+1)/2+j]:
// on return s holds unit-lower L (strictly lower part), dinv[j] = 1/d_j, dsq[j] = 1/sqrt(d_j)
template <class F> ETG_HD void ldl6(F* s, F* dinv, F* dsq) {
  F d[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    // t[k] = L_jk d_k once per (j, k): every term of row j and of the column below it is then a single FMA
    F t[6];
    F dj = s[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; k++) {
      t[k] = s[j * (j + 1) / 2 + k] * d[k];
      dj = dj - s[j * (j + 1) / 2 + k] * t[k];
    }
    d[j] = dj;
    dinv[j] = rcp_(dj);
    dsq[j] = rsqrt_(dj);
#pragma unroll
    for (int i = j + 1; i < 6; i++) {
      F v = s[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; k++) v = v - s[i * (i + 1) / 2 + k] * t[k];
      s[i * (i + 1) / 2 + j] = v * dinv[j];
    }
  }
}
template <class F> ETG_HD void fwd6(const F* L, F* b) {  // b <- L^-1 b
#pragma unroll
  for (int i = 1; i < 6; i++)
#pragma unroll
    for (int k = 0; k < i; k++) b[i] = b[i] - L[i * (i + 1) / 2 + k] * b[k];
}
template <class F> ETG_HD void bwd6(const F* L, F* b) {  // b <- L^-T b
#pragma unroll
  for (int i = 4; i >= 0; i--)
#pragma unroll
    for (int k = i + 1; k < 6; k++) b[i] = b[i] - L[k * (k + 1) / 2 + i] * b[k];
}

// per-lane constants of a physics tick, fetched from the LDS parameter column ONCE per kernel (the compiler
// parks them in AGPRs): a lone wave per SIMD cannot hide the LDS latency of re-reading them every tick
template <class F> struct TickPar4 { F kp[3], kd[3], qd_des[3], tau_ff[3], str[3], sy, m0, mu, link[30]; V3<F> o1, gw; S3<F> I0s; };
template <class F, class Ctx> ETG_HD TickPar4<F> load_tick_par4(const Ctx& c) {
  // tpar: straight from the HBM parameter array into registers (no LDS hop)
  TickPar4<F> t;
  for (int j = 0; j < 3; j++) {
    t.kp[j] = c.tpar(PR_KP + j); t.kd[j] = c.tpar(PR_KD + j); t.qd_des[j] = F(0.0f); t.tau_ff[j] = F(0.0f);
    t.str[j] = Ctx::kPlain ? F(1.0f) : c.tpar(PR_STR + j);   // motor strength ratios (laikago_motor.py:67-76), 1 unless set
  }
  t.sy = c.tpar(PR_SY); t.m0 = c.tpar(PR_M0); t.mu = c.tpar(PR_MU);
  for (int k = 0; k < 30; k++) t.link[k] = c.tpar(PR_LINK + k);
  t.o1 = {c.tpar(PR_O1), c.tpar(PR_O1 + 1), c.tpar(PR_O1 + 2)};
  t.gw = {c.tpar(PR_G), c.tpar(PR_G + 1), c.tpar(PR_G + 2)};
  t.I0s = {c.tpar(PR_I0), c.tpar(PR_I0 + 1), c.tpar(PR_I0 + 2), c.tpar(PR_I0 + 3), c.tpar(PR_I0 + 4), c.tpar(PR_I0 + 5)};
  return t;
}

// ------------------------------------------------------------------ one physics tick
// stepSimulation() + ApplyAction + ReceiveObservation of minitaur.py:242-246 for one quad.
template <class F, class Ctx>
ETG_HD void physics_tick(const Ctx& c, const KCfg& K, const TickPar4<F>& tp, LaneState<F>& L, const F* qdes,
                         const V3<F>& fext_w, bool torque_cmd = false,
                         const F* pd = nullptr,     // pd[0..2] angles, pd[3..5] velocities the PD law reads (EtgConfig.pd_latency), pd[6..8] the angles the command clip refers to
                         F live = F(1.0f)) {        // 0 on the lanes of a robot whose episode has ended (fused rollouts, KCfg.stop_at_done): no rows (physics_tick16)
  typedef V3<F> V;
  typedef SV<F> W;
  const F dt(K.dt), zero(0.0f), one(1.0f);

  // ---- PD motor model (laikago_motor.py:165-173, pd_latency = 0)
  F tau[3];
#pragma unroll
  for (int j = 0; j < 3; j++) {
    F cmd = qdes[j];
    if (!Ctx::kPlain) {   // a1.py:439-457 around GetMotorAngles() = the delayed reading pd[6 + j]; branch-free: an option that is off clamps at +-1e30 (see physics_tick16)
      const F clipv((K.clip_cmd > 0.0f && !torque_cmd) ? K.clip_cmd : 1e30f);
      const F qref = pd ? pd[6 + j] : L.q[j];
      cmd = fminf_(fmaxf_(cmd, qref - clipv), qref + clipv);
    }
    const F qm = (!Ctx::kPlain && pd) ? pd[j] : L.q[j], qdm = (!Ctx::kPlain && pd) ? pd[3 + j] : L.qd[j];   // minitaur.py:1195-1199
    // laikago_motor.py:103-175: the law, x strength ratio, then the clip to +-torque_limit; TORQUE mode: ratio x command, no clip
    F t = Ctx::kPlain ? -(tp.kp[j] * (L.q[j] - cmd)) - tp.kd[j] * L.qd[j]
                      : (torque_cmd ? tp.str[j] * cmd : tp.str[j] * ((-(tp.kp[j] * (qm - cmd)) - tp.kd[j] * (qdm - tp.qd_des[j])) + tp.tau_ff[j]));
    if (!Ctx::kPlain) {
      const F tlim((K.torque_limit > 0.0f && !torque_cmd) ? K.torque_limit : 1e30f);
      t = fminf_(fmaxf_(t, -tlim), tlim);
    }
    tau[j] = t;
  }

  // ---- leg kinematics in the base frame
  F sa, ca, sh, ch, sk, ck;
  sincos_tick_(L.q[0], sa, ca);
  sincos_tick_(L.q[1], sh, ch);
  sincos_tick_(L.q[2], sk, ck);
  const F shk = sh * ck + ch * sk, chk = ch * ck - sh * sk;  // angle addition instead of a third sincos
  Fr<F> R1 = {{one, zero, zero}, {zero, ca, sa}, {zero, -sa, ca}};
  Fr<F> R2 = {{ch, sa * sh, -(ca * sh)}, {zero, ca, sa}, {sh, -(sa * ch), ca * ch}};
  Fr<F> R3 = {{chk, sa * shk, -(ca * shk)}, {zero, ca, sa}, {shk, -(sa * chk), ca * chk}};
  V yax = {zero, ca, sa};
  V xax = {one, zero, zero};
  V o1 = tp.o1;
  V o2 = o1 + tp.sy * yax;
  V o3 = o2 - F(K.upper_len) * R2.ez;
  V pf = o3 - F(K.lower_len) * R3.ez;

  RBI<F> I1 = link_inertia(tp.link[0], V{tp.link[1], tp.link[2], tp.link[3]},
                           S3<F>{tp.link[4], tp.link[5], tp.link[6], tp.link[7], tp.link[8], tp.link[9]}, R1, o1);
  RBI<F> I2 = link_inertia(tp.link[10], V{tp.link[11], tp.link[12], tp.link[13]},
                           S3<F>{tp.link[14], tp.link[15], tp.link[16], tp.link[17], tp.link[18], tp.link[19]}, R2, o2);
  RBI<F> I3 = link_inertia(tp.link[20], V{tp.link[21], tp.link[22], tp.link[23]},
                           S3<F>{tp.link[24], tp.link[25], tp.link[26], tp.link[27], tp.link[28], tp.link[29]}, R3, o3);
  W S1 = {xax, cross(o1, xax)}, S2 = {yax, cross(o2, yax)}, S3_ = {yax, cross(o3, yax)};

  c.phase(0);
  // ---- velocities, bias accelerations (qdd = 0, a_base = -g), bias forces (RNEA)
  Rows<F> Rw = quat_rows(L.qx, L.qy, L.qz, L.qw);
  // contact detection works on the start-of-tick pose: on a heightfield the foot point and the four corner loads of its
  // terrain cell are issued HERE and consumed in phase 4 (a lone wave would otherwise sit out their latency every tick)
  V fw;
  F tap[6];
  if (!Ctx::kFlat) {
    fw = {L.p.x + dot(Rw.r0, pf), L.p.y + dot(Rw.r1, pf), L.p.z + dot(Rw.r2, pf)};
    c.terrain_fetch(K, fw.x, fw.y, tap);
  }
  V gb;  // R^T g: columns of R are (r0.x, r1.x, r2.x) ...
  V gw = tp.gw;
  gb.x = Rw.r0.x * gw.x + Rw.r1.x * gw.y + Rw.r2.x * gw.z;
  gb.y = Rw.r0.y * gw.x + Rw.r1.y * gw.y + Rw.r2.y * gw.z;
  gb.z = Rw.r0.z * gw.x + Rw.r1.z * gw.y + Rw.r2.z * gw.z;
  W V0 = {L.wb, L.vb};
  W V1 = V0 + L.qd[0] * S1;
  W V2 = V1 + L.qd[1] * S2;
  W V3_ = V2 + L.qd[2] * S3_;
  W a0 = {{zero, zero, zero}, {-gb.x, -gb.y, -gb.z}};
  W a1 = a0 + L.qd[0] * crm(V0, S1);
  W a2 = a1 + L.qd[1] * crm(V1, S2);
  W a3 = a2 + L.qd[2] * crm(V2, S3_);
  W f3 = apply(I3, a3) + crf(V3_, apply(I3, V3_));
  W f2 = apply(I2, a2) + crf(V2, apply(I2, V2)) + f3;
  W f1 = apply(I1, a1) + crf(V1, apply(I1, V1)) + f2;
  F C1 = dot(S1, f1), C2 = dot(S2, f2), C3 = dot(S3_, f3);
  const F m0 = tp.m0;
  const S3<F> I0s = tp.I0s;
  RBI<F> I0 = {m0, {zero, zero, zero}, I0s};
  W f0 = apply(I0, a0) + crf(V0, apply(I0, V0));

  c.phase(1);
  // ---- composite inertias, leg block H (3x3), base coupling Fm (6x3) (CRBA)
  RBI<F> Ic2 = I2 + I3;
  RBI<F> Ic1 = I1 + Ic2;
  W F1 = apply(Ic1, S1), F2 = apply(Ic2, S2), F3 = apply(I3, S3_);
  F H11 = dot(S1, F1), H12 = dot(S1, F2), H13 = dot(S1, F3);
  F H22 = dot(S2, F2), H23 = dot(S2, F3), H33 = dot(S3_, F3);
  // H^-1 by cofactors
  F cA = H22 * H33 - H23 * H23, cB = H13 * H23 - H12 * H33, cC = H12 * H23 - H13 * H22;
  F cD = H11 * H33 - H13 * H13, cE = H12 * H13 - H11 * H23, cF = H11 * H22 - H12 * H12;
  F idet = rcp_(H11 * cA + H12 * cB + H13 * cC);
  F Hi11 = cA * idet, Hi12 = cB * idet, Hi13 = cC * idet, Hi22 = cD * idet, Hi23 = cE * idet, Hi33 = cF * idet;
  // P = Fm H^-1 (columns)
  W P1 = Hi11 * F1 + Hi12 * F2 + Hi13 * F3;
  W P2 = Hi12 * F1 + Hi22 * F2 + Hi23 * F3;
  W P3 = Hi13 * F1 + Hi23 * F2 + Hi33 * F3;
  F rl1 = tau[0] - C1, rl2 = tau[1] - C2, rl3 = tau[2] - C3;
  W pb = rl1 * P1 + rl2 * P2 + rl3 * P3;

  c.phase(2);
  // ---- base Schur complement: S = Mbb - sum_legs P Fm^T, rhs = -(f0 + sum f1) - sum pb
  F s[21];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++)
      s[i * (i + 1) / 2 + j] =
          c.qsum(comp(P1, i) * comp(F1, j) + comp(P2, i) * comp(F2, j) + comp(P3, i) * comp(F3, j));
  RBI<F> Ib;
  Ib.m = m0 + c.qsum(Ic1.m);
  Ib.h = {c.qsum(Ic1.h.x), c.qsum(Ic1.h.y), c.qsum(Ic1.h.z)};
  Ib.I = {I0s.xx + c.qsum(Ic1.I.xx), I0s.yy + c.qsum(Ic1.I.yy), I0s.zz + c.qsum(Ic1.I.zz),
          I0s.xy + c.qsum(Ic1.I.xy), I0s.xz + c.qsum(Ic1.I.xz), I0s.yz + c.qsum(Ic1.I.yz)};
  // Mbb lower triangle: rows/cols (wx wy wz vx vy vz); [w,w] = I, [v,w] = -[h]x, [v,v] = m
  F mbb[21];
  mbb[0] = Ib.I.xx;
  mbb[1] = Ib.I.xy; mbb[2] = Ib.I.yy;
  mbb[3] = Ib.I.xz; mbb[4] = Ib.I.yz; mbb[5] = Ib.I.zz;
  mbb[6] = zero;     mbb[7] = Ib.h.z;   mbb[8] = -Ib.h.y;  mbb[9] = Ib.m;
  mbb[10] = -Ib.h.z; mbb[11] = zero;    mbb[12] = Ib.h.x;  mbb[13] = zero; mbb[14] = Ib.m;
  mbb[15] = Ib.h.y;  mbb[16] = -Ib.h.x; mbb[17] = zero;    mbb[18] = zero; mbb[19] = zero; mbb[20] = Ib.m;
#pragma unroll
  for (int i = 0; i < 21; i++) s[i] = mbb[i] - s[i];
  F rb[6];
#pragma unroll
  for (int i = 0; i < 6; i++) rb[i] = -(comp(f0, i) + c.qsum(comp(f1, i))) - c.qsum(comp(pb, i));
  if (!Ctx::kPlain) {  // external force on the trunk COM (world frame; zero unless set) -> base frame: R^T f
    rb[3] = rb[3] + Rw.r0.x * fext_w.x + Rw.r1.x * fext_w.y + Rw.r2.x * fext_w.z;
    rb[4] = rb[4] + Rw.r0.y * fext_w.x + Rw.r1.y * fext_w.y + Rw.r2.y * fext_w.z;
    rb[5] = rb[5] + Rw.r0.z * fext_w.x + Rw.r1.z * fext_w.y + Rw.r2.z * fext_w.z;
  }
  F dinv[6], sq[6];
  ldl6(s, dinv, sq);
  const W sqv = {{sq[0], sq[1], sq[2]}, {sq[3], sq[4], sq[5]}};
  fwd6(s, rb);
  {
    const W rbd = cmul(W{{rb[0], rb[1], rb[2]}, {rb[3], rb[4], rb[5]}}, W{{dinv[0], dinv[1], dinv[2]}, {dinv[3], dinv[4], dinv[5]}});
#pragma unroll
    for (int i = 0; i < 6; i++) rb[i] = comp(rbd, i);
  }
  bwd6(s, rb);
  W ab = {{rb[0], rb[1], rb[2]}, {rb[3], rb[4], rb[5]}};
  F qdd1 = Hi11 * rl1 + Hi12 * rl2 + Hi13 * rl3 - dot(P1, ab);
  F qdd2 = Hi12 * rl1 + Hi22 * rl2 + Hi23 * rl3 - dot(P2, ab);
  F qdd3 = Hi13 * rl1 + Hi23 * rl2 + Hi33 * rl3 - dot(P3, ab);

  c.phase(3);
  // ---- unconstrained velocity
  V wbs = L.wb + dt * ab.a, vbs = L.vb + dt * ab.l;
  F qds1 = L.qd[0] + dt * qdd1, qds2 = L.qd[1] + dt * qdd2, qds3 = L.qd[2] + dt * qdd3;

  c.phase(4);
  // ---- foot contact (sphere vs ground)
  if (Ctx::kFlat) fw = {L.p.x + dot(Rw.r0, pf), L.p.y + dot(Rw.r1, pf), L.p.z + dot(Rw.r2, pf)};
  F phi;
  V dn, d1, d2;
  if (Ctx::kFlat) {
    // plane z = 0: n = z_w, t1 = x_w, t2 = y_w -> in base coordinates the rows of R
    phi = fw.z - F(K.foot_radius);
    dn = Rw.r2; d1 = Rw.r0; d2 = Rw.r1;
  } else {
    F hgt, nwx, nwy, nwz;
    c.terrain_finish(K, tap, hgt, nwx, nwy, nwz);
    phi = (fw.z - hgt) * nwz - F(K.foot_radius);
    // contact frame in world: n, t1 = normalised (x_w - (x_w.n) n), t2 = n x t1; then to base coords
    // with |n| = 1: |x_w - nx n|^2 = 1 - nx^2 and n x t1 = (0, nz, -ny) / |..|
    const F it1 = rsqrt_hf_(one - nwx * nwx);
    const V nw = {nwx, nwy, nwz};
    const V t1w = {it1 * (one - nwx * nwx), -(it1 * (nwx * nwy)), -(it1 * (nwx * nwz))};
    const F t2y = it1 * nwz, t2z = -(it1 * nwy);
    dn = {Rw.r0.x * nw.x + Rw.r1.x * nw.y + Rw.r2.x * nw.z, Rw.r0.y * nw.x + Rw.r1.y * nw.y + Rw.r2.y * nw.z,
          Rw.r0.z * nw.x + Rw.r1.z * nw.y + Rw.r2.z * nw.z};
    d1 = {Rw.r0.x * t1w.x + Rw.r1.x * t1w.y + Rw.r2.x * t1w.z, Rw.r0.y * t1w.x + Rw.r1.y * t1w.y + Rw.r2.y * t1w.z,
          Rw.r0.z * t1w.x + Rw.r1.z * t1w.y + Rw.r2.z * t1w.z};
    d2 = {Rw.r1.x * t2y + Rw.r2.x * t2z, Rw.r1.y * t2y + Rw.r2.y * t2z, Rw.r1.z * t2y + Rw.r2.z * t2z};
  }
  const auto lives = live > F(0.5f);
  auto act = (phi < F(K.margin)) && lives;
  F actf = sel_(act, one, zero);
  V rc = pf - F(K.foot_radius) * dn;
  V k1 = cross(xax, rc - o1), k2 = cross(yax, rc - o2), k3 = cross(yax, rc - o3);
  // ---- body rows (EtgConfig.body_contacts; the instantiations with Ctx::kBody > 0): rows of this leg on spheres of knee_radius,
  // solved after the feet's rows of the same kind.  kBody = 1 -- ONE contact per leg with a normal row and two friction rows
  // (rows 3, 4, 5 of the lane): body_contacts 1: a sphere at the knee (calf joint origin, carried by the thigh: the calf joint
  // does not move it); body_contacts 2: the DEEPEST of knee / shin midpoint (moved by all three joints) / trunk corner next to
  // this leg's hip (moved by none), ties to the earlier candidate.  kBody = 3 -- body_contacts 3: all three spheres collide at
  // once, a FRICTIONLESS normal row each (rows 3, 4, 5), in that order.  Same model as the oracle (no warm start).
  constexpr int NB = Ctx::kBody;
  constexpr bool knee = NB > 0;
  constexpr bool bfric = NB == 1;              // the leg's one body contact has friction rows
  constexpr int NP = NB;                       // body contact points per leg
  constexpr int NBR = bfric ? 3 : NB;          // body rows per leg
  constexpr int NRW = 3 + NBR;
  constexpr int NBA = NBR > 0 ? NBR : 1;       // array extents (no zero-length arrays)
  constexpr int NPA = NP > 0 ? NP : 1;
  F phib[NPA], actbf[NPA];
  V dnb[NPA], rcb[NPA], kb1[NPA], kb2[NPA], kb3[NPA], nwb[NPA];
  decltype(act) actb[NPA];
  if constexpr (knee) {
    // depth of a body point along the terrain normal (without the radius) and the normal there, world frame
    auto depth = [&](const V& q, V& nw) -> F {
      const V w = {L.p.x + dot(Rw.r0, q), L.p.y + dot(Rw.r1, q), L.p.z + dot(Rw.r2, q)};
      if (Ctx::kFlat) { nw = {zero, zero, one}; return w.z; }
      F hgt, nwx, nwy, nwz;
      c.terrain(K, w.x, w.y, hgt, nwx, nwy, nwz);
      nw = {nwx, nwy, nwz};
      return (w.z - hgt) * nwz;
    };
    // Every body point is a weighted mean of sphere centres (weights one-hot unless EtgConfig.body_blend > 0, see physics_tick16):
    // q12 = w_knee o3 + w_shin ps and s12 = w_knee + w_shin (the spheres hip and thigh move), w_shin (the one the calf moves)
    V pbd[NPA], q12[NPA];
    F dep[NPA], s12[NPA], wsh[NPA];
    V psh = o3;                                  // the shin midpoint
    pbd[0] = o3; q12[0] = o3; s12[0] = one; wsh[0] = zero;
    dep[0] = depth(pbd[0], nwb[0]);
    if (NB == 3 || K.knee >= 2) {
      const V ps = o3 - F(0.5f * K.lower_len) * R3.ez;
      const V pt = {sel_(o1.x > zero, F(K.trunk_half[0]), F(-K.trunk_half[0])),
                    sel_(o1.y > zero, F(K.trunk_half[1]), F(-K.trunk_half[1])), F(-K.trunk_half[2])};
      psh = ps;
      V nws, nwt;
      const F ds = depth(ps, nws), dtk = depth(pt, nwt);
      if constexpr (NB == 3) {
        pbd[NB - 2] = ps; nwb[NB - 2] = nws; dep[NB - 2] = ds; q12[NB - 2] = ps; s12[NB - 2] = one; wsh[NB - 2] = one;
        pbd[NB - 1] = pt; nwb[NB - 1] = nwt; dep[NB - 1] = dtk; q12[NB - 1] = {zero, zero, zero}; s12[NB - 1] = zero; wsh[NB - 1] = zero;
      } else {
        const F d0 = dep[0];
        const auto ms = ds < d0;
        const F best = sel_(ms, ds, d0);
        const auto mt = dtk < best;
        F w0 = sel_(mt, zero, sel_(ms, zero, one)), w1 = sel_(mt, zero, sel_(ms, one, zero)), w2 = sel_(mt, one, zero);
        if (K.blend_inv > 0.0f) {
          const F dmin = sel_(mt, dtk, best);
          const F e0 = exp_((dmin - d0) * F(K.blend_inv)), e1 = exp_((dmin - ds) * F(K.blend_inv)), e2 = exp_((dmin - dtk) * F(K.blend_inv));
          const F wi = rcp_(e0 + e1 + e2);
          w0 = e0 * wi; w1 = e1 * wi; w2 = e2 * wi;
        }
        q12[0] = w0 * o3 + w1 * ps;
        s12[0] = w0 + w1;
        wsh[0] = w1;
        pbd[0] = q12[0] + w2 * pt;
        dep[0] = depth(pbd[0], nwb[0]);          // the ground under the weighted centre (one-hot weights: the picked sphere's own)
      }
    }
#pragma unroll
    for (int b = 0; b < NP; b++) {
      phib[b] = dep[b] - F(K.knee_radius);
      if (Ctx::kFlat) dnb[b] = Rw.r2;
      else dnb[b] = {Rw.r0.x * nwb[b].x + Rw.r1.x * nwb[b].y + Rw.r2.x * nwb[b].z, Rw.r0.y * nwb[b].x + Rw.r1.y * nwb[b].y + Rw.r2.y * nwb[b].z,
                     Rw.r0.z * nwb[b].x + Rw.r1.z * nwb[b].y + Rw.r2.z * nwb[b].z};
      actb[b] = (phib[b] < F(K.margin)) && lives;
      actbf[b] = sel_(actb[b], one, zero);
      const V rn = F(K.knee_radius) * dnb[b];
      rcb[b] = pbd[b] - rn;
      const V a12 = q12[b] - s12[b] * rn;
      kb1[b] = cross(xax, a12 - s12[b] * o1);
      kb2[b] = cross(yax, a12 - s12[b] * o2);
      kb3[b] = wsh[b] * cross(yax, (psh - rn) - o3);
    }
  }
  V dir[NRW];
  dir[0] = dn; dir[1] = d1; dir[2] = d2;
  if constexpr (bfric) {
    // the body contact's frame, as for a foot: n, t1 = x_w projected on the tangent plane, t2 = n x t1 (flat ground: the foot's)
    dir[3] = dnb[0];
    if (Ctx::kFlat) { dir[4] = d1; dir[5] = d2; }
    else {
      const V nb = nwb[0];
      const F it1 = rsqrt_hf_(one - nb.x * nb.x);
      const V t1w = {it1 * (one - nb.x * nb.x), -(it1 * (nb.x * nb.y)), -(it1 * (nb.x * nb.z))};
      const F t2y = it1 * nb.z, t2z = -(it1 * nb.y);
      dir[4] = {Rw.r0.x * t1w.x + Rw.r1.x * t1w.y + Rw.r2.x * t1w.z, Rw.r0.y * t1w.x + Rw.r1.y * t1w.y + Rw.r2.y * t1w.z,
                Rw.r0.z * t1w.x + Rw.r1.z * t1w.y + Rw.r2.z * t1w.z};
      dir[5] = {Rw.r1.x * t2y + Rw.r2.x * t2z, Rw.r1.y * t2y + Rw.r2.y * t2z, Rw.r1.z * t2y + Rw.r2.z * t2z};
    }
  } else {
#pragma unroll
    for (int b = 0; b < NB; b++) dir[3 + b] = dnb[b];
  }
  F Jl[NRW][3];   // [row d][joint]
  F HJ[NRW][3];   // H^-1 Jl^T, [row d][joint]
  W Z[NRW];       // D^-1/2 L^-1 G_d
#pragma unroll
  for (int d = 0; d < NRW; d++) {
    const bool body = d >= 3;
    const int bi = (body && !bfric) ? d - 3 : 0;   // the body POINT of row d (kBody = 1: one point, three rows)
    const F af = body ? actbf[bi] : actf;
    const V rcd = body ? rcb[bi] : rc;
    Jl[d][0] = af * dot(dir[d], body ? kb1[bi] : k1);
    Jl[d][1] = af * dot(dir[d], body ? kb2[bi] : k2);
    Jl[d][2] = af * dot(dir[d], body ? kb3[bi] : k3);
    HJ[d][0] = Hi11 * Jl[d][0] + Hi12 * Jl[d][1] + Hi13 * Jl[d][2];
    HJ[d][1] = Hi12 * Jl[d][0] + Hi22 * Jl[d][1] + Hi23 * Jl[d][2];
    HJ[d][2] = Hi13 * Jl[d][0] + Hi23 * Jl[d][1] + Hi33 * Jl[d][2];
    W Jb = af * W{cross(rcd, dir[d]), dir[d]};
    W G = Jb - (Jl[d][0] * P1 + Jl[d][1] * P2 + Jl[d][2] * P3);
    F g6[6] = {G.a.x, G.a.y, G.a.z, G.l.x, G.l.y, G.l.z};
    fwd6(s, g6);
    Z[d] = cmul(W{{g6[0], g6[1], g6[2]}, {g6[3], g6[4], g6[5]}}, sqv);   // packed on the GPU
  }
  c.phase(5);
  // Delassus blocks A[j][d][e] = Z_mine,d . Z_j,e (+ local leg compliance on the own block).
  // Every block is a quad-wide outer product over the 4 lanes, contracted over the 6 base
  // coordinates: on the GPU that is 6 accumulating v_mfma_f32_4x4x1_16b_f32 per (d, e) pair
  // (one 4x4 block per robot, 16 robots per instruction) instead of 18 DPP broadcasts + 54 FMAs.
  F A[4][NRW][NRW];
#pragma unroll
  for (int d = 0; d < NRW; d++)
#pragma unroll
    for (int e = 0; e < NRW; e++) {
      F acc[4] = {zero, zero, zero, zero};
#pragma unroll
      for (int k = 0; k < 6; k++) c.quad_outer(comp(Z[e], k), comp(Z[d], k), acc);  // acc[j] += Z_j,e[k] * Z_mine,d[k]
      F loc = Jl[d][0] * HJ[e][0] + Jl[d][1] * HJ[e][1] + Jl[d][2] * HJ[e][2];
#pragma unroll
      for (int j = 0; j < 4; j++) A[j][d][e] = acc[j] + sel_(c.lane_is(j), loc, zero);
    }
  // own diagonal inverses
  F Aown[NRW][NRW];
#pragma unroll
  for (int d = 0; d < NRW; d++)
#pragma unroll
    for (int e = 0; e < NRW; e++)
      Aown[d][e] = sel_(c.lane_is(0), A[0][d][e], sel_(c.lane_is(1), A[1][d][e], sel_(c.lane_is(2), A[2][d][e], A[3][d][e])));
  F iA0 = sel_(act, rcp_(Aown[0][0]), zero), iA1 = sel_(act, rcp_(Aown[1][1]), zero), iA2 = sel_(act, rcp_(Aown[2][2]), zero);
  F iAb[NBA];
#pragma unroll
  for (int b = 0; b < NBR; b++) iAb[b] = sel_(actb[bfric ? 0 : b], rcp_(Aown[3 + b][3 + b]), zero);
  c.phase(6);
  // contact-point velocity under the unconstrained motion
  V vc = vbs + cross(wbs, rc) + qds1 * k1 + qds2 * k2 + qds3 * k3;
  F u0 = actf * dot(dn, vc), u1 = actf * dot(d1, vc), u2 = actf * dot(d2, vc);
  const F idt(1.0f / K.dt);
  const F pen = phi + F(K.slop);                                  // Bullet: penetration = distance + m_linearSlop
  F tgt = sel_(pen > zero, -(pen * idt), -(F(K.erp) * pen * idt));
  if (!Ctx::kPlain) {
    // EtgConfig.foot_restitution: a foot that approaches faster than 0.2 m/s at the START of the tick bounces (see physics_tick16)
    const V vc0 = L.vb + cross(L.wb, rc) + L.qd[0] * k1 + L.qd[1] * k2 + L.qd[2] * k3;
    const F un0 = actf * dot(dn, vc0);
    tgt = tgt + sel_(un0 < F(-0.2f), -(F(K.restitution) * un0), zero);
  }
  // warm start: the normal impulse x K.warmstart, the friction impulses x K.warmstart_t (Bullet's multibody solver restarts
  // them from zero); inactive feet forget their impulse
  F l0 = actf * F(K.warmstart) * L.lam[0], l1 = actf * F(K.warmstart_t) * L.lam[1], l2 = actf * F(K.warmstart_t) * L.lam[2];
  // the body rows: their own points' velocities, Baumgarte / speculative target like the foot's normal row; the normal row of
  // the leg's ONE body contact (kBody = 1) starts from K.warmstart_b x its impulse of the tick before (physics_tick16), the rest at 0;
  // kbf[b][x]: A[3+b][3+x] / A[3+b][3+b] for the body rows x < b solved earlier in the leg's turn
  F ub[NBA], lb[NBA], cb[NBA], kbf[NBA][NBA];
#pragma unroll
  for (int b = 0; b < NBR; b++) {
    const int pt = bfric ? 0 : b;
    const V vcb = vbs + cross(wbs, rcb[pt]) + qds1 * kb1[pt] + qds2 * kb2[pt] + qds3 * kb3[pt];
    ub[b] = actbf[pt] * dot(dir[3 + b], vcb);
    lb[b] = zero;
    const F penb = phib[pt] + F(K.slop);
    const F tgtb = sel_(penb > zero, -(penb * idt), -(F(K.erp) * penb * idt));
    cb[b] = (bfric && b > 0) ? zero : tgtb * iAb[b];         // only normal rows have a target
#pragma unroll
    for (int x = 0; x < b; x++) kbf[b][x] = Aown[3 + b][3 + x] * iAb[b];
  }
#pragma unroll
  for (int j = 0; j < 4; j++) {
    F b0 = c.qbcast(l0, j), b1 = c.qbcast(l1, j), b2 = c.qbcast(l2, j);
    u0 = u0 + A[j][0][0] * b0 + A[j][0][1] * b1 + A[j][0][2] * b2;
    u1 = u1 + A[j][1][0] * b0 + A[j][1][1] * b1 + A[j][1][2] * b2;
    u2 = u2 + A[j][2][0] * b0 + A[j][2][1] * b1 + A[j][2][2] * b2;
#pragma unroll
    for (int b = 0; b < NBR; b++) ub[b] = ub[b] + A[j][3 + b][0] * b0 + A[j][3 + b][1] * b1 + A[j][3 + b][2] * b2;
  }
  if constexpr (bfric) {   // the body normal's warm start enters every row's velocity like the feet's
    lb[0] = actbf[0] * F(K.warmstart_b) * L.lamb;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const F bb0 = c.qbcast(lb[0], j);
      u0 = u0 + A[j][0][3] * bb0; u1 = u1 + A[j][1][3] * bb0; u2 = u2 + A[j][2][3] * bb0;
#pragma unroll
      for (int b = 0; b < NBR; b++) ub[b] = ub[b] + A[j][3 + b][3] * bb0;
    }
  }
  c.phase(7);
  // ---- projected Gauss-Seidel in the order of Bullet's btMultiBodyConstraintSolver::solveSingleIteration (the oracle's
  // physics_tick states it): per sweep (1) the joint-limit rows, (2) the NORMAL rows -- feet in lane order, then the body
  // rows, leg by leg --, (3) the friction rows of the feet in lane order: both candidates of a foot from the same
  // velocities, the pair projected on the disc mu ln (friction_model 1: each clamped on its own), skipped while the foot's
  // normal impulse is not positive.  Lane j's raw deltas are broadcast unmasked: an inactive foot has iA = 0 and l = 0, which
  // makes its deltas exact zeros.
  // The sweeps read iA*, c0, mu through the variables below: under the residual stopping rule a converged robot is
  // FROZEN for the sweeps its wave neighbours still need (iA = c0 = 0: every candidate is the current impulse, exact
  // zero deltas; mu = 1e30: the cone projection is the identity) -- see physics_tick16.
  F mu = tp.mu;
  F mub(K.body_mu);      // friction coefficient of the body contact (kBody = 1)
  F c0 = tgt * iA0;
  F own[4];
#pragma unroll
  for (int j = 0; j < 4; j++) own[j] = sel_(c.lane_is(j), one, zero);
  // ---- joint-limit rows (EtgConfig.joint_limits; see physics_tick16): this lane owns the rows of ITS three joints
  bool anyj = false;
  F jactf[3] = {zero, zero, zero};
  if (K.jlim) {   // (compiled into the PLAIN instantiations too: the stops are on by default)
    auto anyhit = c.lane_is(0) && !c.lane_is(0);   // all-false mask
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const auto hit = ((L.q[j] >= F(K.jhi[j])) || (L.q[j] <= F(K.jlo[j]))) && lives;
      jactf[j] = sel_(hit, one, zero);
      anyhit = anyhit || hit;
    }
    anyj = c.any(anyhit);
  }
  // solve + apply, instantiated with and without the joint rows: without them their variables are compile-time zeros
  auto finish_tick = [&](auto joints_tag) {
  constexpr bool joints = decltype(joints_tag)::value;
  const F Hinv[3][3] = {{Hi11, Hi12, Hi13}, {Hi12, Hi22, Hi23}, {Hi13, Hi23, Hi33}};
  F sgn[3] = {zero, zero, zero}, lamq[3] = {zero, zero, zero}, iAq[3] = {zero, zero, zero}, c0q[3] = {zero, zero, zero};
  W zj[3];
  if (joints) {
    const W Pc[3] = {P1, P2, P3};
#pragma unroll
    for (int j = 0; j < 3; j++) {
      F z6[6] = {Pc[j].a.x, Pc[j].a.y, Pc[j].a.z, Pc[j].l.x, Pc[j].l.y, Pc[j].l.z};
      fwd6(s, z6);
      zj[j] = cmul(W{{z6[0], z6[1], z6[2]}, {z6[3], z6[4], z6[5]}}, sqv);
      const F lo(K.jlo[j]), hi(K.jhi[j]);
      sgn[j] = jactf[j] * sel_(L.q[j] >= hi, -one, one);
      const F viol = fmaxf_(L.q[j] - hi, lo - L.q[j]);
      iAq[j] = jactf[j] * rcp_(Hinv[j][j] + dot(zj[j], zj[j]));
      c0q[j] = (F(K.erp) * viol * idt) * iAq[j];
    }
  }
  // instantiated per friction model (a compile-time constant inside the sweeps): see physics_tick16
  auto solve = [&](auto pyramid_tag) {
    constexpr bool pyramid = decltype(pyramid_tag)::value;
    F iAqe[3] = {iAq[0], iAq[1], iAq[2]}, c0qe[3] = {c0q[0], c0q[1], c0q[2]};
    // which of the 12 joint rows exist for SOME robot of the wave: one wave-uniform bit each, made once per tick (etg_core16.h)
    unsigned jrows = 0u;
    if (joints) {
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 3; i++)
          if (c.any(own[j] * jactf[i] > F(0.5f))) jrows |= 1u << (3 * j + i);
      jrows = c.uniform_bits(jrows);
    }
    auto joint_phase = [&]() {
      // ZL = sum over the robot's rows of lam_r Z_r (the joint rows' Z-vectors are -sgn zj)
      W zs = l0 * Z[0] + l1 * Z[1] + l2 * Z[2];
#pragma unroll
      for (int b = 0; b < NBR; b++) zs = zs + lb[b] * Z[3 + b];
#pragma unroll
      for (int i = 0; i < 3; i++) zs = zs - (sgn[i] * lamq[i]) * zj[i];
      F zl[6] = {c.qsum(zs.a.x), c.qsum(zs.a.y), c.qsum(zs.a.z), c.qsum(zs.l.x), c.qsum(zs.l.y), c.qsum(zs.l.z)};
      const W zl0 = {{zl[0], zl[1], zl[2]}, {zl[3], zl[4], zl[5]}};
      F qc[3] = {qds1, qds2, qds3};          // unconstrained joint velocities + the contact impulses' leg part (fixed during the phase)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        qc[i] = qc[i] + HJ[0][i] * l0 + HJ[1][i] * l1 + HJ[2][i] * l2;
#pragma unroll
        for (int b = 0; b < NBR; b++) qc[i] = qc[i] + HJ[3 + b][i] * lb[b];
      }
      const F lamq0[3] = {lamq[0], lamq[1], lamq[2]};
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < 3; i++) {
          if (!(jrows & (1u << (3 * j + i)))) continue;              // no robot of the wave has this joint at a stop
          const F qj = qc[i] + Hinv[i][0] * (sgn[0] * lamq[0]) + Hinv[i][1] * (sgn[1] * lamq[1]) + Hinv[i][2] * (sgn[2] * lamq[2]);
          const W zlw = {{zl[0], zl[1], zl[2]}, {zl[3], zl[4], zl[5]}};
          const F uq = sgn[i] * (qj - dot(zj[i], zlw));
          const F dq = own[j] * fmaxf_(-lamq[i], c0qe[i] - uq * iAqe[i]);
          lamq[i] = lamq[i] + dq;
          const F sd = -(sgn[i] * dq);
#pragma unroll
          for (int k = 0; k < 6; k++) zl[k] = zl[k] + c.qbcast(sd * comp(zj[i], k), j);
        }
      // the contact rows see the joint impulses' changes: A[c][q] d lam_q = Z_c . dZL + HJ_c[joint] sgn d lam_q
      const W dZ = W{{zl[0], zl[1], zl[2]}, {zl[3], zl[4], zl[5]}} - zl0;
      const F w0 = sgn[0] * (lamq[0] - lamq0[0]), w1 = sgn[1] * (lamq[1] - lamq0[1]), w2 = sgn[2] * (lamq[2] - lamq0[2]);
      u0 = u0 + (dot(Z[0], dZ) + (HJ[0][0] * w0 + HJ[0][1] * w1 + HJ[0][2] * w2));
      u1 = u1 + (dot(Z[1], dZ) + (HJ[1][0] * w0 + HJ[1][1] * w1 + HJ[1][2] * w2));
      u2 = u2 + (dot(Z[2], dZ) + (HJ[2][0] * w0 + HJ[2][1] * w1 + HJ[2][2] * w2));
#pragma unroll
      for (int b = 0; b < NBR; b++) ub[b] = ub[b] + (dot(Z[3 + b], dZ) + (HJ[3 + b][0] * w0 + HJ[3 + b][1] * w1 + HJ[3 + b][2] * w2));
    };
    auto pgs_sweep = [&]() {
      if (joints) joint_phase();
      // (2) normal rows of the feet: ln = max(0, l0 - (u0 - tgt)/A00), as a change
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const F e0 = fmaxf_(-l0, c0 - u0 * iA0);
        const F b0 = c.qbcast(e0, j);
        u0 = u0 + A[j][0][0] * b0; u1 = u1 + A[j][1][0] * b0; u2 = u2 + A[j][2][0] * b0;
#pragma unroll
        for (int b = 0; b < NBR; b++) ub[b] = ub[b] + A[j][3 + b][0] * b0;
        l0 = l0 + own[j] * e0;                                           // the owner commits
      }
      if constexpr (bfric) {
        // the body contacts' normal rows, leg by leg
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const F eb0 = fmaxf_(-lb[0], cb[0] - ub[0] * iAb[0]);
          const F bb0 = c.qbcast(eb0, j);
          u0 = u0 + A[j][0][3] * bb0; u1 = u1 + A[j][1][3] * bb0; u2 = u2 + A[j][2][3] * bb0;
#pragma unroll
          for (int b = 0; b < NBR; b++) ub[b] = ub[b] + A[j][3 + b][3] * bb0;
          lb[0] = lb[0] + own[j] * eb0;
        }
      } else if constexpr (knee) {
        // the body rows (normal rows too), leg by leg; within a leg each row sees the changes of the earlier ones (kbf)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          F eb[NB], bb[NB];
#pragma unroll
          for (int b = 0; b < NB; b++) {
            F fold = zero;
#pragma unroll
            for (int x = 0; x < b; x++) fold = fold + kbf[b][x] * eb[x];
            eb[b] = fmaxf_(-lb[b], (cb[b] - ub[b] * iAb[b]) - fold);
            bb[b] = c.qbcast(eb[b], j);
          }
#pragma unroll
          for (int b = 0; b < NB; b++) {
            u0 = u0 + A[j][0][3 + b] * bb[b];
            u1 = u1 + A[j][1][3 + b] * bb[b];
            u2 = u2 + A[j][2][3 + b] * bb[b];
            F acc = zero;
#pragma unroll
            for (int x = 0; x < NB; x++) acc = acc + A[j][3 + b][3 + x] * bb[x];
            ub[b] = ub[b] + acc;
            lb[b] = lb[b] + own[j] * eb[b];
          }
        }
      }
      // (3) friction rows
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const F lt1 = l1 - u1 * iA1, lt2 = l2 - u2 * iA2;
        const F lim = mu * l0;
        F e1, e2;
        if (pyramid) {
          e1 = fminf_(fmaxf_(lt1, -lim), lim) - l1;
          e2 = fminf_(fmaxf_(lt2, -lim), lim) - l2;
        } else {
          const F sc = fminf_(one, lim * rsqrt_((lt1 * lt1 + F(1e-30f)) + (lt2 * lt2 + F(1e-30f))));   // (as the 16-lane sweep: each row adds its 1e-30)
          e1 = lt1 * sc - l1; e2 = lt2 * sc - l2;
        }
        const auto grip = l0 > zero;                                     // Bullet: `if (totalImpulse > 0)`
        e1 = sel_(grip, e1, zero); e2 = sel_(grip, e2, zero);
        const F b1 = c.qbcast(e1, j), b2 = c.qbcast(e2, j);
        u0 = u0 + A[j][0][1] * b1 + A[j][0][2] * b2;
        u1 = u1 + A[j][1][1] * b1 + A[j][1][2] * b2;
        u2 = u2 + A[j][2][1] * b1 + A[j][2][2] * b2;
#pragma unroll
        for (int b = 0; b < NBR; b++) ub[b] = ub[b] + A[j][3 + b][1] * b1 + A[j][3 + b][2] * b2;
        l1 = l1 + own[j] * e1; l2 = l2 + own[j] * e2;
      }
      if constexpr (bfric) {
        // (4) the friction pairs of the body contacts, after the feet's: the same rule with the coefficient K.body_mu
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const F lt1 = lb[1] - ub[1] * iAb[1], lt2 = lb[2] - ub[2] * iAb[2];
          const F lim = mub * lb[0];
          F e1, e2;
          if (pyramid) {
            e1 = fminf_(fmaxf_(lt1, -lim), lim) - lb[1];
            e2 = fminf_(fmaxf_(lt2, -lim), lim) - lb[2];
          } else {
            const F sc = fminf_(one, lim * rsqrt_((lt1 * lt1 + F(1e-30f)) + (lt2 * lt2 + F(1e-30f))));
            e1 = lt1 * sc - lb[1]; e2 = lt2 * sc - lb[2];
          }
          const auto grip = lb[0] > zero;
          e1 = sel_(grip, e1, zero); e2 = sel_(grip, e2, zero);
          const F b1 = c.qbcast(e1, j), b2 = c.qbcast(e2, j);
          u0 = u0 + A[j][0][4] * b1 + A[j][0][5] * b2;
          u1 = u1 + A[j][1][4] * b1 + A[j][1][5] * b2;
          u2 = u2 + A[j][2][4] * b1 + A[j][2][5] * b2;
#pragma unroll
          for (int b = 0; b < NBR; b++) ub[b] = ub[b] + A[j][3 + b][4] * b1 + A[j][3 + b][5] * b2;
          lb[1] = lb[1] + own[j] * e1; lb[2] = lb[2] + own[j] * e2;
        }
      }
    };
    if (K.res_thr > 0.0f) {
      // EtgConfig.solver_residual: sweep until the robot's largest squared row residual is <= the threshold (see
      // physics_tick16): |l - l at the start of the sweep| > sqrt(thr) / A_rr per row, tolerances from the unfrozen inverses
      const F tol0 = F(K.res_sqrt) * iA0, tol1 = F(K.res_sqrt) * iA1, tol2 = F(K.res_sqrt) * iA2;
      F tolb[NBA];
#pragma unroll
      for (int b = 0; b < NBR; b++) tolb[b] = F(K.res_sqrt) * iAb[b];
      const F tolq[3] = {F(K.res_sqrt) * iAq[0], F(K.res_sqrt) * iAq[1], F(K.res_sqrt) * iAq[2]};
      int it = 0;
      bool more;
      auto sweep_and_test = [&]() {
        const F s0 = l0, s1 = l1, s2 = l2;
        F sb[NBA];
#pragma unroll
        for (int b = 0; b < NBR; b++) sb[b] = lb[b];
        const F sq_[3] = {lamq[0], lamq[1], lamq[2]};
        pgs_sweep();
        it++;
        auto moved = (fabsf_(l0 - s0) > tol0) || (fabsf_(l1 - s1) > tol1) || (fabsf_(l2 - s2) > tol2);
#pragma unroll
        for (int b = 0; b < NBR; b++) moved = moved || (fabsf_(lb[b] - sb[b]) > tolb[b]);
        if (joints) {
#pragma unroll
          for (int i = 0; i < 3; i++) moved = moved || (fabsf_(lamq[i] - sq_[i]) > tolq[i]);
        }
        const auto live = c.robot_any(moved);
        iA0 = sel_(live, iA0, zero); iA1 = sel_(live, iA1, zero); iA2 = sel_(live, iA2, zero);
        c0 = sel_(live, c0, zero);
        mu = sel_(live, mu, F(1e30f));
        if (bfric) mub = sel_(live, mub, F(1e30f));
#pragma unroll
        for (int b = 0; b < NBR; b++) {
          iAb[b] = sel_(live, iAb[b], zero); cb[b] = sel_(live, cb[b], zero);
#pragma unroll
          for (int x = 0; x < b; x++) kbf[b][x] = sel_(live, kbf[b][x], zero);
        }
        if (joints) {
#pragma unroll
          for (int i = 0; i < 3; i++) { iAqe[i] = sel_(live, iAqe[i], zero); c0qe[i] = sel_(live, c0qe[i], zero); }
        }
        more = c.wave_any(live) && it < K.iters;
      };
      if (Ctx::kPlain && !joints) {   // nested forward exits instead of a loop for the first sweeps: see physics_tick16
        sweep_and_test();
        if (__builtin_expect(more, 1)) { sweep_and_test();
        if (__builtin_expect(more, 1)) { sweep_and_test();
        if (__builtin_expect(more, 1)) { sweep_and_test();
        if (more) { sweep_and_test();
        if (more) { sweep_and_test();
        if (more) { sweep_and_test();
        if (more) { sweep_and_test();
          while (more) sweep_and_test();
        }}}}}}}
      } else {
        do sweep_and_test(); while (more);
      }
      L.sweeps += it;
    } else if (K.iters == 2 && !joints) {   // a fixed pair of sweeps, straight-line (see physics_tick16)
      pgs_sweep();
      pgs_sweep();
      L.sweeps += 2;
    } else {
      for (int it = 0; it < K.iters; it++) pgs_sweep();
      L.sweeps += K.iters;
    }
  };
  if constexpr (Ctx::kPlain) {
    solve(std::false_type{});
  } else {
    if (K.fric_pyramid) solve(std::true_type{});
    else solve(std::false_type{});
  }
  c.phase(8);
  // ---- apply impulses: base via the Schur factor, leg via H^-1
  W zs = l0 * Z[0] + l1 * Z[1] + l2 * Z[2];
#pragma unroll
  for (int b = 0; b < NBR; b++) zs = zs + lb[b] * Z[3 + b];
  if (joints) {
#pragma unroll
    for (int i = 0; i < 3; i++) zs = zs - (sgn[i] * lamq[i]) * zj[i];
  }
  const W dbs = cmul(W{{c.qsum(zs.a.x), c.qsum(zs.a.y), c.qsum(zs.a.z)}, {c.qsum(zs.l.x), c.qsum(zs.l.y), c.qsum(zs.l.z)}}, sqv);
  F db[6] = {dbs.a.x, dbs.a.y, dbs.a.z, dbs.l.x, dbs.l.y, dbs.l.z};
  bwd6(s, db);
  W dB = {{db[0], db[1], db[2]}, {db[3], db[4], db[5]}};
  L.wb = wbs + dB.a;
  L.vb = vbs + dB.l;
  L.qd[0] = qds1 + HJ[0][0] * l0 + HJ[1][0] * l1 + HJ[2][0] * l2 - dot(P1, dB);
  L.qd[1] = qds2 + HJ[0][1] * l0 + HJ[1][1] * l1 + HJ[2][1] * l2 - dot(P2, dB);
  L.qd[2] = qds3 + HJ[0][2] * l0 + HJ[1][2] * l1 + HJ[2][2] * l2 - dot(P3, dB);
#pragma unroll
  for (int b = 0; b < NBR; b++) {
    L.qd[0] = L.qd[0] + HJ[3 + b][0] * lb[b]; L.qd[1] = L.qd[1] + HJ[3 + b][1] * lb[b]; L.qd[2] = L.qd[2] + HJ[3 + b][2] * lb[b];
  }
  if (joints) {
    const F sl0 = sgn[0] * lamq[0], sl1 = sgn[1] * lamq[1], sl2 = sgn[2] * lamq[2];
#pragma unroll
    for (int i = 0; i < 3; i++) L.qd[i] = L.qd[i] + (Hinv[i][0] * sl0 + Hinv[i][1] * sl1 + Hinv[i][2] * sl2);
  }
  L.lam[0] = l0; L.lam[1] = l1; L.lam[2] = l2;
  L.lamb = bfric ? lb[0] : zero;
  L.contact = sel_(act && (l0 > zero), one, zero);
  };   // finish_tick
  if (anyj) finish_tick(std::true_type{});
  else finish_tick(std::false_type{});

  c.phase(9);
  // ---- semi-implicit Euler on positions
#pragma unroll
  for (int j = 0; j < 3; j++) L.q[j] = L.q[j] + dt * L.qd[j];
  L.p.x = L.p.x + dt * dot(Rw.r0, L.vb);
  L.p.y = L.p.y + dt * dot(Rw.r1, L.vb);
  L.p.z = L.p.z + dt * dot(Rw.r2, L.vb);
  V th = dt * L.wb;
  F a2_ = dot(th, th);
  // sin(a/2)/a and cos(a/2) by series (|a| <= ~0.1 per tick)
  F sh2 = F(0.5f) - a2_ * (F(1.0f / 48.0f) - a2_ * F(1.0f / 3840.0f));
  F ch2 = one - a2_ * (F(0.125f) - a2_ * (F(1.0f / 384.0f) - a2_ * F(1.0f / 46080.0f)));
  F dx = th.x * sh2, dy = th.y * sh2, dz = th.z * sh2, dw = ch2;
  F nx = L.qw * dx + L.qx * dw + L.qy * dz - L.qz * dy;
  F ny = L.qw * dy - L.qx * dz + L.qy * dw + L.qz * dx;
  F nz = L.qw * dz + L.qx * dy - L.qy * dx + L.qz * dw;
  F nw_ = L.qw * dw - L.qx * dx - L.qy * dy - L.qz * dz;
  F inv = rsqrt_(nx * nx + ny * ny + nz * nz + nw_ * nw_);
  L.qx = nx * inv; L.qy = ny * inv; L.qz = nz * inv; L.qw = nw_ * inv;

  L.energy = L.energy + (fabsf_(tau[0] * L.qd[0]) + fabsf_(tau[1] * L.qd[1]) + fabsf_(tau[2] * L.qd[2])) * dt;
}

// ------------------------------------------------------------------ latency ring (minitaur.py:1142-1193)
// per lane 8 floats per tick: q3 qd3 + two base words (lane0: qx qy, lane1: qz qw,
// lane2: wx wy, lane3: wz 0)
template <class F, class Ctx> ETG_HD void ring_push(const Ctx& c, float* ring, int slot, const LaneState<F>& L) {
  F b0 = sel_(c.lane_is(0), L.qx, sel_(c.lane_is(1), L.qz, sel_(c.lane_is(2), L.wb.x, L.wb.z)));
  F b1 = sel_(c.lane_is(0), L.qy, sel_(c.lane_is(1), L.qw, sel_(c.lane_is(2), L.wb.y, F(0.0f))));
  c.st_ring(ring, slot, 0, L.q[0]); c.st_ring(ring, slot, 1, L.q[1]); c.st_ring(ring, slot, 2, L.q[2]);
  c.st_ring(ring, slot, 3, L.qd[0]); c.st_ring(ring, slot, 4, L.qd[1]); c.st_ring(ring, slot, 5, L.qd[2]);
  c.st_ring(ring, slot, 6, b0); c.st_ring(ring, slot, 7, b1);
}
template <class F> struct Delayed { F q[3], qd[3]; F qx, qy, qz, qw; V3<F> w; };
ETG_HD const float* ring_of_tick4(const KCfg& K, const float* ring, int t) { return (K.cring != nullptr && t <= K.settle_ticks) ? K.cring : ring; }
// the PD law's reading under EtgConfig.pd_latency (see pd_reading16)
template <class F, class Ctx> ETG_HD void pd_reading(const Ctx& c, const KCfg& K, const float* ring, int tick, bool live, F* pd) {
  const int ta = tick - K.pd_n < 0 ? 0 : tick - K.pd_n, tb = tick - K.pd_n - 1 < 0 ? 0 : tick - K.pd_n - 1;
  const float *ra = live ? ring : ring_of_tick4(K, ring, ta), *rb_ = live ? ring : ring_of_tick4(K, ring, tb);
  const int sa = ta & (RING - 1), sb = tb & (RING - 1);
  const F a(K.pd_a), oma(1.0f - K.pd_a);
#pragma unroll
  for (int k = 0; k < 6; k++) pd[k] = oma * c.ld_ring(ra, sa, k) + a * c.ld_ring(rb_, sb, k);
}
// GetMotorAngles() for A1._ClipMotorCommands: see clip_reading16
template <class F, class Ctx> ETG_HD void clip_reading(const Ctx& c, const KCfg& K, const float* ring, int tick, bool live, const F* now, F* out) {
  const int n = c.uniform_int(c.par(PR_LAT_N));
  if (n < 0) {
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = wrap_pi_(now[k]);
    return;
  }
  const F alpha = c.par(PR_LAT_A);
  const int ta = tick - n < 0 ? 0 : tick - n, tb = tick - n - 1 < 0 ? 0 : tick - n - 1;
  const float *ra = live ? ring : ring_of_tick4(K, ring, ta), *rb_ = live ? ring : ring_of_tick4(K, ring, tb);
#pragma unroll
  for (int k = 0; k < 3; k++) out[k] = wrap_pi_((F(1.0f) - alpha) * c.ld_ring(ra, ta & (RING - 1), k) + alpha * c.ld_ring(rb_, tb & (RING - 1), k));
}
template <class F, class Ctx>
ETG_HD Delayed<F> ring_read(const Ctx& c, const KCfg& K, const float* ring, int tick) {
  // n_steps_ago / blend_alpha are env-uniform; lat_n < 0 encodes latency <= 0.  Readings of ticks up to the reset tick come
  // from the settle cache's ring (KCfg.cring), later ones from the live ring -- see etg_layout.h
  F v[8];
  int n = c.uniform_int(c.par(PR_LAT_N));
  const F alpha = c.par(PR_LAT_A);
  if (n < 0) {
    const float* r0 = (K.cring != nullptr && tick <= K.settle_ticks) ? K.cring : ring;
#pragma unroll
    for (int k = 0; k < 8; k++) v[k] = c.ld_ring(r0, tick & (RING - 1), k);
  } else {
    int sa = (tick - n) & (RING - 1), sb = (tick - n - 1) & (RING - 1);
    const float* ra = (K.cring != nullptr && tick - n <= K.settle_ticks) ? K.cring : ring;
    const float* rb_ = (K.cring != nullptr && tick - n - 1 <= K.settle_ticks) ? K.cring : ring;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      F a = c.ld_ring(ra, sa, k), b = c.ld_ring(rb_, sb, k);
      v[k] = (F(1.0f) - alpha) * a + alpha * b;
    }
  }
  Delayed<F> D;
  for (int k = 0; k < 3; k++) { D.q[k] = v[k]; D.qd[k] = v[3 + k]; }
  D.qx = c.qbcast(v[6], 0); D.qy = c.qbcast(v[7], 0);
  D.qz = c.qbcast(v[6], 1); D.qw = c.qbcast(v[7], 1);
  D.w = {c.qbcast(v[6], 2), c.qbcast(v[7], 2), c.qbcast(v[6], 3)};
  return D;
}

// ------------------------------------------------------------------ ETG + IK
// a1.py:97-110.  The reference detects an unreachable target through the NaN that arccos/arcsin
// return outside [-1,1]; here the domain test is explicit (same predicate, no reliance on NaN
// propagation) and the arguments are clamped so the functions stay in-domain.
template <class F, class B> ETG_HD void leg_ik(V3<F> foot, F sign, F* ang, B& valid) {
  const F l_up(0.2f), l_low(0.2f), one(1.0f);
  F l_hip = F(0.08505f) * sign;
  F x = foot.x, y = foot.y, z = foot.z;
  F ck = (x * x + y * y + z * z - l_hip * l_hip - l_low * l_low - l_up * l_up) / (F(2.0f) * l_low * l_up);
  F ckc = fminf_(fmaxf_(ck, -one), one);
  F theta_knee = -acos_(ckc);
  F l = sqrt_(fmaxf_(l_up * l_up + l_low * l_low + F(2.0f) * l_up * l_low * cos_(theta_knee), F(1e-12f)));
  F sarg = -x / l;
  F sargc = fminf_(fmaxf_(sarg, -one), one);
  F theta_hip = asin_(sargc) - theta_knee * F(0.5f);
  F cc = cos_(theta_hip + theta_knee * F(0.5f));
  F c1 = l_hip * y - l * cc * z;
  F s1 = l * cc * y + l_hip * z;
  ang[0] = atan2_(s1, c1);
  ang[1] = theta_hip;
  ang[2] = theta_knee;
  valid = (fabsf_(ck) <= one) && (fabsf_(sarg) <= one) && isfinite_(ck) && isfinite_(sarg);
}
// joint-space ETG action of this lane's leg at time t (minus pose_ori)
template <class F, class Ctx>
ETG_HD void etg_action(const Ctx& c, const KCfg& K, const float* etgp, float t, F* act) {
  // legs 0,3 use r(t), legs 1,2 use r(t + T2*T) (trot)
  F tl = sel_(c.lane_is(0) || c.lane_is(3), F(t), F(t + K.etg_T2 * K.etg_T));
  F x0 = F(K.etg_amp) * sin_(F(K.etg_phase0) + tl * F(K.etg_omega));
  F x1 = F(K.etg_amp) * sin_(F(K.etg_phase1) + tl * F(K.etg_omega));
  F ax = c.ld_env(etgp, EP_B + 0), az = c.ld_env(etgp, EP_B + 2);
  F ay = c.ld_env(etgp, EP_B + 1);
  const F isig(1.0f / K.etg_sigma_sq);
#pragma unroll 4
  for (int h = 0; h < ETG_RBF_H; h++) {
    F d0 = x0 - F(K.etg_u[h][0]), d1 = x1 - F(K.etg_u[h][1]);
    F r = exp_(-((d0 * d0 + d1 * d1) * isig));
    ax = ax + c.ld_env(etgp, EP_W + h) * r;
    ay = ay + c.ld_env(etgp, EP_W + ETG_RBF_H + h) * r;
    az = az + c.ld_env(etgp, EP_W + 2 * ETG_RBF_H + h) * r;
  }
  // IK with the 0.95 shrink guard against unreachable targets
  F scale(1.0f);
  const V3<F> posev = par3<F>(c, PR_POSE), bfoot = par3<F>(c, PR_BASE_FOOT), o1 = par3<F>(c, PR_O1);
  const F hipsign = c.par(PR_HIPSIGN);
  F ang[3] = {posev.x, posev.y, posev.z};
  auto pending = c.lane_is(0) || !c.lane_is(0);  // all-true mask
  {   // the target as commanded: all the common path executes (see etg_action16)
    V3<F> foot = {bfoot.x + ax - o1.x, bfoot.y + ay - o1.y, bfoot.z + az - o1.z};
    F a[3];
    auto ok = pending;
    leg_ik(foot, hipsign, a, ok);
    ang[0] = sel_(ok, a[0], ang[0]); ang[1] = sel_(ok, a[1], ang[1]); ang[2] = sel_(ok, a[2], ang[2]);
    pending = !ok;
  }
  if (c.any(pending)) {
    scale = F(0.95f);
    for (int it = 1; it < 200; it++) {
      V3<F> foot = {bfoot.x + ax * scale - o1.x, bfoot.y + ay * scale - o1.y, bfoot.z + az * scale - o1.z};
      F a[3];
      auto ok = pending;
      leg_ik(foot, hipsign, a, ok);
      auto take = pending && ok;
      ang[0] = sel_(take, a[0], ang[0]); ang[1] = sel_(take, a[1], ang[1]); ang[2] = sel_(take, a[2], ang[2]);
      pending = pending && !ok;
      scale = scale * F(0.95f);
      if (!c.any(pending)) break;
    }
  }
  act[0] = ang[0] - posev.x; act[1] = ang[1] - posev.y; act[2] = ang[2] - posev.z;
}

// roll-pitch-yaw (ZYX) of a quaternion (getEulerFromQuaternion, minitaur.py:620,633)
template <class F> ETG_HD V3<F> quat_rpy(F x, F y, F z, F w) {
  F n = x * x + y * y + z * z + w * w;
  F s = F(2.0f) / n;
  F r20 = s * (x * z - y * w), r21 = s * (y * z + x * w), r22 = F(1.0f) - s * (x * x + y * y);
  F r10 = s * (x * y + z * w), r00 = F(1.0f) - s * (y * y + z * z);
  F sp = fminf_(fmaxf_(-r20, F(-1.0f)), F(1.0f));
  return {atan2_(r21, r22), asin_(sp), atan2_(r10, r00)};
}

template <class F> ETG_HD F c_prec(F v, F t, F m) {
  // atanh(sqrt(0.95)) = 2.1783...
  F w = F(2.178343806f) / m;
  F x = (v - t) * w;
  return tanh_(x * x);
}

// ------------------------------------------------------------------ observation (EnvWrapper.py:60-109)
// sorted keys: BaseDisplacement(3) FootContactSensor(4) IMU(6) MotorAngleAcc(24) + ETG(12) = 49
template <class F, class Ctx>
ETG_HD void write_obs(const Ctx& c, const KCfg& K, const LaneState<F>& L, const Delayed<F>& D, F f0, F f1, F f2,
                      const F* etg, F lbx, F lby, F lbz, float* obs, F* imu) {
  V3<F> rpy = quat_rpy(D.qx, D.qy, D.qz, D.qw);
  const bool nrm = K.obs_normal != 0;
  const float cdt = K.dt * (float)K.action_repeat;
  F sdis(nrm ? 1.0f / cdt : 1.0f), srpy(nrm ? 10.0f : 1.0f), sdr(nrm ? 2.0f : 1.0f), sq(nrm ? 10.0f : 1.0f);
  imu[0] = rpy.x - f0; imu[1] = rpy.y - f1; imu[2] = rpy.z - f2;
  imu[3] = D.w.x; imu[4] = D.w.y; imu[5] = D.w.z;
  if (obs) {
    c.st_row_env(obs, ETG_OBS_DIM, 0, (L.p.x - lbx) * sdis);
    c.st_row_env(obs, ETG_OBS_DIM, 1, (L.p.y - lby) * sdis);
    c.st_row_env(obs, ETG_OBS_DIM, 2, (L.p.z - lbz) * sdis);
    c.st_row_lane(obs, ETG_OBS_DIM, 3, 1, L.contact);
    for (int k = 0; k < 3; k++) c.st_row_env(obs, ETG_OBS_DIM, 7 + k, imu[k] * srpy);
    for (int k = 0; k < 3; k++) c.st_row_env(obs, ETG_OBS_DIM, 10 + k, imu[3 + k] * sdr);
    for (int j = 0; j < 3; j++) {
      const F posej = c.par(PR_POSE + j), emj = c.par(PR_EMEAN + j), esj = c.par(PR_ESTD + j);
      // MapToMinusPiToPi (minitaur.py:67-83)
      F a = wrap_pi_(D.q[j]);
      c.st_row_lane(obs, ETG_OBS_DIM, 13 + j, 3, nrm ? (a - posej) * sq : a);
      c.st_row_lane(obs, ETG_OBS_DIM, 25 + j, 3, D.qd[j]);
      c.st_row_lane(obs, ETG_OBS_DIM, 37 + j, 3, nrm ? (etg[j] - emj) / esj : etg[j]);
    }
  }
}

// world x of this lane's foot centre and base-frame foot z / knee height over ground
template <class F> struct FootKin { F fwx, fbz, knee_h; };
template <class F, class Ctx>
ETG_HD FootKin<F> foot_kin(const Ctx& c, const KCfg& K, const LaneState<F>& L) {
  F sa, ca, sh, ch, shk, chk;
  sincos_(L.q[0], sa, ca);
  sincos_(L.q[1], sh, ch);
  sincos_(L.q[1] + L.q[2], shk, chk);
  F Lu(K.upper_len), Ll(K.lower_len);
  V3<F> yax = {F(0.0f), ca, sa};
  V3<F> o2 = par3<F>(c, PR_O1) + c.par(PR_SY) * yax;
  V3<F> e2 = {sh, -(sa * ch), ca * ch}, e3 = {shk, -(sa * chk), ca * chk};
  V3<F> o3 = o2 - Lu * e2;
  V3<F> pf = o3 - Ll * e3;
  Rows<F> Rw = quat_rows(L.qx, L.qy, L.qz, L.qw);
  FootKin<F> k;
  k.fwx = L.p.x + dot(Rw.r0, pf);
  k.fbz = pf.z;
  F kx = L.p.x + dot(Rw.r0, o3), ky = L.p.y + dot(Rw.r1, o3), kz = L.p.z + dot(Rw.r2, o3);
  if (Ctx::kFlat) {
    k.knee_h = kz;
  } else {
    F hgt, nx, ny, nz;
    c.terrain(K, kx, ky, hgt, nx, ny, nz);
    k.knee_h = kz - hgt;
  }
  return k;
}

// ------------------------------------------------------------------ one control step (env.step)
// returns reward / done for the quad; obs/info optional.
// The control-loop variables of a robot that live across steps: loaded once per kernel, kept in registers over
// one step (env.step) or many (open-loop rollout), stored once.
template <class F> struct StepCtl4 {
  int step_count, tick, has_last;
  F last[3], lbx, lby, lbz, last_fwx;   // last position command, last base position, last foot x (world)
  F ret, len, alive;                    // episode accumulators
  F r0, r1, r2;                         // first rpy reading after reset (EnvWrapper.py:79-84)
  F fx0[3], fx1[3], fy0[3], fy1[3];     // action filter history (only with K.enable_filter)
};
template <class F, class Ctx>
ETG_HD StepCtl4<F> load_ctl4(const Ctx& c, const KCfg& K, const float* ctl, const int* ictl, const float* legctl) {
  StepCtl4<F> S;
  S.step_count = c.ld_env_i(ictl, IC_STEP);
  S.tick = c.ld_env_i(ictl, IC_TICK);
  S.has_last = c.ld_env_i(ictl, IC_HAS_LAST);
  for (int j = 0; j < 3; j++) S.last[j] = c.ld_lane(legctl, LC_LAST_QDES + j);
  S.lbx = c.ld_env(ctl, CT_LAST_BASE + 0); S.lby = c.ld_env(ctl, CT_LAST_BASE + 1); S.lbz = c.ld_env(ctl, CT_LAST_BASE + 2);
  S.last_fwx = c.ld_lane(legctl, LC_LAST_FOOT_X);
  S.ret = c.ld_env(ctl, CT_RET); S.len = c.ld_env(ctl, CT_LEN); S.alive = c.ld_env(ctl, CT_ALIVE);
  S.r0 = c.ld_env(ctl, CT_FIRST_RPY + 0); S.r1 = c.ld_env(ctl, CT_FIRST_RPY + 1); S.r2 = c.ld_env(ctl, CT_FIRST_RPY + 2);
  for (int j = 0; j < 3; j++) {
    S.fx0[j] = S.fx1[j] = S.fy0[j] = S.fy1[j] = F(0.0f);
    if (!Ctx::kPlain && K.enable_filter) {
      S.fx0[j] = c.ld_lane(legctl, LC_FX0 + j); S.fx1[j] = c.ld_lane(legctl, LC_FX1 + j);
      S.fy0[j] = c.ld_lane(legctl, LC_FY0 + j); S.fy1[j] = c.ld_lane(legctl, LC_FY1 + j);
    }
  }
  return S;
}
template <class F, class Ctx>
ETG_HD void store_ctl4(const Ctx& c, const KCfg& K, const StepCtl4<F>& S, float* ctl, int* ictl, float* legctl) {
  c.st_env_i(ictl, IC_STEP, S.step_count);
  c.st_env_i(ictl, IC_TICK, S.tick);
  c.st_env_i(ictl, IC_HAS_LAST, S.has_last);
  for (int j = 0; j < 3; j++) c.st_lane(legctl, LC_LAST_QDES + j, S.last[j]);
  c.st_env(ctl, CT_LAST_BASE + 0, S.lbx); c.st_env(ctl, CT_LAST_BASE + 1, S.lby); c.st_env(ctl, CT_LAST_BASE + 2, S.lbz);
  c.st_lane(legctl, LC_LAST_FOOT_X, S.last_fwx);
  c.st_env(ctl, CT_RET, S.ret); c.st_env(ctl, CT_LEN, S.len); c.st_env(ctl, CT_ALIVE, S.alive);
  if (!Ctx::kPlain && K.enable_filter)
    for (int j = 0; j < 3; j++) {
      c.st_lane(legctl, LC_FX0 + j, S.fx0[j]); c.st_lane(legctl, LC_FX1 + j, S.fx1[j]);
      c.st_lane(legctl, LC_FY0 + j, S.fy0[j]); c.st_lane(legctl, LC_FY1 + j, S.fy1[j]);
    }
}

// one env.step on the register-resident control state S and tick constants tp
template <class F, class Ctx>
ETG_HD void control_step_core(const Ctx& c, const KCfg& K, TickPar4<F>& tp, const V3<F>& fext, LaneState<F>& L, StepCtl4<F>& S,
                              float* ring, const float* etgp, const F* action, F donef, float* obs, F& reward, F& done,
                              float* info, const F* hyb = nullptr,     // hyb[4*j + (kp, qd_des, kd, tau_ff)], HYBRID mode
                              bool want_obs = true,                    // false: inner steps of the open-loop rollout (row unread)
                              float* rec_q = nullptr, float* rec_imu = nullptr,     // action-tape rollouts: see control_step16_core
                              bool skip_dead = false,                  // fused rollouts under KCfg.stop_at_done: see control_step16_core
                              float* obs_end = nullptr) {              // skip_dead: also receives the last row of a robot whose episode ends here
  const F live = skip_dead ? S.alive : F(1.0f);
  const auto is_live = live > F(0.5f);
  if (skip_dead) c.set_gate(is_live);
  int step_count = S.step_count;
  int tick = S.tick;
  const int has_last = S.has_last;
  // ETG at t = (k+1) dt (fixture convention of gait_action_list_ETG_exp.npy)
  F etg[3], qdes[3];
  if (Ctx::kPlain || K.etg_on) etg_action(c, K, etgp, (float)(step_count + 1) * K.etg_dt, etg);
  else etg[0] = etg[1] = etg[2] = F(0.0f);   // EtgConfig.enable_etg = 0: the command is pose_ori + action
  const bool torque_cmd = !Ctx::kPlain && K.motor_mode == 1;
  const bool hybrid_cmd = !Ctx::kPlain && K.motor_mode == 2 && hyb != nullptr;
#pragma unroll
  for (int j = 0; j < 3; j++) qdes[j] = (torque_cmd || hybrid_cmd) ? action[j] : c.par(PR_POSE + j) + etg[j] + action[j];
  if (!Ctx::kPlain && K.enable_filter) {  // action_filter.py:111-120, order 2
#pragma unroll
    for (int j = 0; j < 3; j++) {
      F y = F(K.fb[0]) * qdes[j] + F(K.fb[1]) * S.fx0[j] + F(K.fb[2]) * S.fx1[j] - F(K.fa[1]) * S.fy0[j] - F(K.fa[2]) * S.fy1[j];
      if (skip_dead) {   // (a finished robot's filter history stays)
        S.fx1[j] = sel_(is_live, S.fx0[j], S.fx1[j]); S.fx0[j] = sel_(is_live, qdes[j], S.fx0[j]);
        S.fy1[j] = sel_(is_live, S.fy0[j], S.fy1[j]); S.fy0[j] = sel_(is_live, y, S.fy0[j]);
      } else {
        S.fx1[j] = S.fx0[j]; S.fx0[j] = qdes[j];
        S.fy1[j] = S.fy0[j]; S.fy0[j] = y;
      }
      qdes[j] = y;
    }
  }
  const F last[3] = {S.last[0], S.last[1], S.last[2]};
  const F lbx = S.lbx, lby = S.lby, lbz = S.lbz, last_fwx = S.last_fwx;
  L.energy = F(0.0f);
  L.sweeps = 0;
  const bool interp = !Ctx::kPlain && K.enable_interp && has_last;
  // Only the two readings a later observation will blend (minitaur.py:1185-1193: ticks T-n and
  // T-n-1 of some step end T) have to reach the ring: (i+1+n) mod R in {0, R-1}, i.e. i == ia or i == ib.
  const int n_lat = c.uniform_int(c.par(PR_LAT_N));
  const int R_ = K.action_repeat;
  const float inv_repeat = 1.0f / (float)K.action_repeat;
  if (hybrid_cmd)
    for (int j = 0; j < 3; j++) { tp.kp[j] = hyb[4 * j]; tp.qd_des[j] = hyb[4 * j + 1]; tp.kd[j] = hyb[4 * j + 2]; tp.tau_ff[j] = hyb[4 * j + 3]; }
  const int mlat = n_lat < 0 ? 0 : n_lat % R_;
  const int ia = R_ - 1 - mlat;
  const int ib = n_lat < 0 ? ia : (ia == 0 ? R_ - 1 : ia - 1);
  const bool pdl = !Ctx::kPlain && K.pd_n >= 0;
  const bool cl = !Ctx::kPlain && K.clip_cmd > 0.0f && !torque_cmd;   // the command clip reads the delayed angles of every tick
  for (int i = 0; i < K.action_repeat; i++) {  // minitaur.py:254-258
    F proc[3];
    float lerp = (float)(i + 1) * inv_repeat;
#pragma unroll
    for (int j = 0; j < 3; j++) proc[j] = interp ? last[j] + F(lerp) * (qdes[j] - last[j]) : qdes[j];
    F pd[9] = {L.q[0], L.q[1], L.q[2], L.qd[0], L.qd[1], L.qd[2], L.q[0], L.q[1], L.q[2]};   // EtgConfig.pd_latency: see control_step16_core
    if (pdl) pd_reading(c, K, ring, tick, false, pd);
    if (cl) clip_reading(c, K, ring, tick, false, L.q, pd + 6);
    physics_tick(c, K, tp, L, proc, fext, torque_cmd, Ctx::kPlain ? (const F*)nullptr : pd, live);
    tick++;
    if (pdl || cl || i == ia || i == ib) ring_push(c, ring, tick & (RING - 1), L);
  }
  // (skip_dead: the control variables of a finished robot stay what they were when it finished: control_step16_core)
  const int li = skip_dead ? c.sel_i(is_live, 1, 0) : 1;
#pragma unroll
  for (int j = 0; j < 3; j++) S.last[j] = (skip_dead && !Ctx::kPlain) ? sel_(is_live, qdes[j], last[j]) : qdes[j];
  step_count += li;
  S.step_count = step_count; S.tick = tick - (1 - li) * K.action_repeat; S.has_last |= li;
  tick = S.tick;
  c.ring_fence();

  // ---- reward / termination (this repo's definitions)
  const float cdt = K.dt * (float)K.action_repeat;
  Rows<F> Rw = quat_rows(L.qx, L.qy, L.qz, L.qw);
  V3<F> rpy = quat_rpy(L.qx, L.qy, L.qz, L.qw);
  FootKin<F> fk = foot_kin(c, K, L);
  F vx = (L.p.x - lbx) * F(1.0f / cdt);
  F torso = fminf_(vx, F(K.vel_d));
  F up = (F(1.0f) - c_prec(rpy.x, F(0.0f), F(0.5f))) * (F(1.0f) - c_prec(rpy.y, F(0.0f), F(0.5f)));
  F feet = c.qsum((fk.fwx - last_fwx) * F(0.25f)) * F(1.0f / cdt);
  feet = fminf_(feet, F(K.vel_d));
  F energy = c.qsum(L.energy);
  F lost = c.qsum(F(1.0f) - L.contact);
  F bad = c.qsum(sel_(fk.knee_h < F(0.03f), F(1.0f), F(0.0f)));
  F footcontact = -fmaxf_(lost - F(2.0f), F(0.0f));
  F fz_mean = c.qsum(fk.fbz) * F(0.25f);
  F fz_max = c.qmax(fk.fbz);
  auto fin = isfinite_(L.p.x) && isfinite_(L.p.z) && isfinite_(c.qbcast(L.q[0], 0));
  auto term = (Rw.r2.z < F(0.5f)) || (fz_mean > F(-0.1f)) || (fz_max > F(0.0f)) || (fabsf_(rpy.z) > F(0.6f)) || !fin;
  F termf = sel_(term, F(1.0f), F(0.0f));
  F terms[8] = {F(K.rw[0]) * torso, F(K.rw[1]) * feet, F(K.rw[2]) * up, F(K.rw[3]) * (-energy), F(0.0f),
                F(K.rw[5]) * (-bad), F(K.rw[6]) * footcontact, F(K.rw[7]) * (-termf)};
  F sum = terms[0];
#pragma unroll
  for (int k = 1; k < 8; k++) sum = sum + terms[k];
  reward = F(K.reward_p) * sum;
  done = sel_(term || (donef > F(0.5f)), F(1.0f), F(0.0f));
  if (skip_dead) {   // a finished robot reports what the reference's loop would see if it looked again: nothing new
    reward = sel_(is_live, reward, F(0.0f));
    done = sel_(is_live, done, F(1.0f));
  }
  // the observation row: wanted by the caller, needed by info -- or the LAST one of a robot whose episode ends in this step
  const bool ending = skip_dead && c.any(is_live && (done > F(0.5f)));
  F imu[6] = {F(0.0f), F(0.0f), F(0.0f), F(0.0f), F(0.0f), F(0.0f)};
  if (want_obs || info || ending) {
    const auto Dl = ring_read<F>(c, K, ring, tick);
    write_obs(c, K, L, Dl, S.r0, S.r1, S.r2, etg, lbx, lby, lbz, obs, imu);
    if (ending && obs_end && obs_end != obs) {
      c.set_gate(is_live && (done > F(0.5f)));
      write_obs(c, K, L, Dl, S.r0, S.r1, S.r2, etg, lbx, lby, lbz, obs_end, imu);
      c.set_gate(is_live);
    }
  }
  if (rec_q)
    for (int j = 0; j < 3; j++) c.st_row_lane(rec_q, ETG_ACT_DIM, j, 3, L.q[j]);
  if (rec_imu)
    for (int k = 0; k < 6; k++) c.st_row_env(rec_imu, 6, k, imu[k]);
  if (info) {
    for (int k = 0; k < 8; k++) c.st_row_env(info, ETG_INFO_DIM, k, terms[k]);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_VELX, vx);
    for (int j = 0; j < 3; j++) {
      c.st_row_lane(info, ETG_INFO_DIM, ETG_INFO_ETG_ACT + j, 3, etg[j]);
      c.st_row_lane(info, ETG_INFO_DIM, ETG_INFO_JOINT_ANGLE + j, 3, L.q[j]);
      c.st_row_lane(info, ETG_INFO_DIM, ETG_INFO_REAL_ACTION + j, 3, qdes[j]);
    }
    for (int k = 0; k < 6; k++) c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_OBS_IMU + k, imu[k]);
    c.st_row_lane(info, ETG_INFO_DIM, ETG_INFO_FOOT_CONTACT, 1, L.contact);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_BASE + 0, L.p.x);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_BASE + 1, L.p.y);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_BASE + 2, L.p.z);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_RPY + 0, rpy.x);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_RPY + 1, rpy.y);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_RPY + 2, rpy.z);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_ENERGY, energy);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_STEPS, F((float)step_count));
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_SWEEPS, F((float)L.sweeps));
  }
  S.lbx = L.p.x; S.lby = L.p.y; S.lbz = L.p.z;
  S.last_fwx = fk.fwx;
  // episode accumulators with alive masking (the batched counterpart of the callers' loops,
  // train.py:213-249 / pretrain.py:129-154: return and length stop growing after `done`)
  S.ret = S.ret + S.alive * reward;
  S.len = S.len + S.alive;
  S.alive = sel_(done > F(0.5f), F(0.0f), S.alive);
  if (skip_dead) {
    c.open_gate();
    S.lbx = sel_(is_live, S.lbx, lbx); S.lby = sel_(is_live, S.lby, lby); S.lbz = sel_(is_live, S.lbz, lbz);
    S.last_fwx = sel_(is_live, S.last_fwx, last_fwx);
  }
}

// env.step for one robot quad: load the control state, one step, store it
template <class F, class Ctx>
ETG_HD void control_step(const Ctx& c, const KCfg& K, LaneState<F>& L, float* ring, float* ctl,
                         int* ictl, float* legctl, const float* etgp, const F* action, F donef, float* obs, F& reward, F& done,
                         float* info, const F* hyb = nullptr) {
  StepCtl4<F> S = load_ctl4<F>(c, K, ctl, ictl, legctl);
  TickPar4<F> tp = load_tick_par4<F>(c);
  V3<F> fext = {F(0.0f), F(0.0f), F(0.0f)};
  if (!Ctx::kPlain && K.ext_force) fext = {c.ld_env(ctl, CT_FEXT + 0) + c.ld_env(ctl, CT_PUSH + 0), c.ld_env(ctl, CT_FEXT + 1) + c.ld_env(ctl, CT_PUSH + 1),
                                           c.ld_env(ctl, CT_FEXT + 2) + c.ld_env(ctl, CT_PUSH + 2)};   // set force + random push
  control_step_core(c, K, tp, fext, L, S, ring, etgp, action, donef, obs, reward, done, info, hyb);
  store_ctl4(c, K, S, ctl, ictl, legctl);
}

// fused rollouts under KCfg.stop_at_done: see rollout_dead_store16 / rollout_store16 (etg_core16.h), one leg per lane here
template <class F, class Ctx>
ETG_HD void rollout_dead_store(const Ctx& c, const KCfg& K, const LaneState<F>& L, F was_alive, F done, bool last_step, bool noise_ending,
                               int obs_call, float* base, float* leg, int* ictl) {
  const auto was = was_alive > F(0.5f);
  const auto ended = was && (done > F(0.5f));
  if (c.any(ended)) {
    c.set_gate(ended);
    store_state(c, base, leg, L);
    c.open_gate();
  }
  if (K.noise_on && (last_step || noise_ending)) {
    const auto wrote = last_step ? was : ended;
    if (c.any(wrote)) {
      c.set_gate(wrote);
      c.st_env_i(ictl, IC_OBS_CALL, obs_call);
      c.open_gate();
    }
  }
}
template <class F, class Ctx>
ETG_HD void rollout_store(const Ctx& c, const KCfg& K, const LaneState<F>& L, F alive, float* base, float* leg) {
  if (K.stop_at_done) c.set_gate(alive > F(0.5f));
  store_state(c, base, leg, L);
  c.open_gate();
}

// open-loop rollout (pretrain.py:129-154): n_steps env.steps with zero residual action in one kernel
template <class F, class Ctx>
ETG_HD void rollout_steps(const Ctx& c, const KCfg& K, LaneState<F>& L, float* base, float* leg, float* ring, float* ctl, int* ictl,
                          float* legctl, const float* etgp, int n_steps, float* obs) {
  StepCtl4<F> S = load_ctl4<F>(c, K, ctl, ictl, legctl);
  TickPar4<F> tp = load_tick_par4<F>(c);
  V3<F> fext = {F(0.0f), F(0.0f), F(0.0f)};
  if (!Ctx::kPlain && K.ext_force) fext = {c.ld_env(ctl, CT_FEXT + 0) + c.ld_env(ctl, CT_PUSH + 0), c.ld_env(ctl, CT_FEXT + 1) + c.ld_env(ctl, CT_PUSH + 1),
                                           c.ld_env(ctl, CT_FEXT + 2) + c.ld_env(ctl, CT_PUSH + 2)};   // set force + random push
  const F zero3[3] = {F(0.0f), F(0.0f), F(0.0f)};
  F reward, done;
  const bool skip = K.stop_at_done != 0;
  for (int s = 0; s < n_steps; s++) {   // only the last observation of the rollout is ever read (see rollout_steps16)
    if (skip && !c.any(S.alive > F(0.5f))) break;     // every robot of the wave has finished
    const F was_alive = S.alive;
    control_step_core(c, K, tp, fext, L, S, ring, etgp, zero3, F(0.0f), obs, reward, done, (float*)nullptr, (const F*)nullptr,
                      s == n_steps - 1, (float*)nullptr, (float*)nullptr, skip);
    if (skip) rollout_dead_store(c, K, L, was_alive, done, s == n_steps - 1, true, (int)K.noise_call + s, base, leg, ictl);
  }
  store_ctl4(c, K, S, ctl, ictl, legctl);
  rollout_store(c, K, L, S.alive, base, leg);
}


// ------------------------------------------------------------------ reset (minitaur.py:403-445, a1.py:289-349)
template <class F, class Ctx>
ETG_HD void reset_settle(const Ctx& c, const KCfg& K, LaneState<F>& L, float* ring, F offx = F(0.0f), F offy = F(0.0f)) {
  L.p = {F(K.init_pos[0]) + offx, F(K.init_pos[1]) + offy, F(K.init_pos[2])};   // (offx, offy): etg_set_reset_offsets
  L.qx = F(0.0f); L.qy = F(0.0f); L.qz = F(0.0f); L.qw = F(1.0f);
  L.wb = {F(0.0f), F(0.0f), F(0.0f)};
  L.vb = {F(0.0f), F(0.0f), F(0.0f)};
  F pose[3] = {c.par(PR_POSE), c.par(PR_POSE + 1), c.par(PR_POSE + 2)};
  for (int j = 0; j < 3; j++) { L.q[j] = pose[j]; L.qd[j] = F(0.0f); L.lam[j] = F(0.0f); }
  L.lamb = F(0.0f);
  L.contact = F(0.0f);
  L.energy = F(0.0f);
  L.sweeps = 0;
  // ReceiveObservation before settling (a1.py:290): seed every ring slot with the initial reading
  for (int sl = 0; sl < RING; sl++) ring_push(c, ring, sl, L);
  int tick = 0;
  const TickPar4<F> tp = load_tick_par4<F>(c);
  for (int i = 0; i < K.settle_ticks; i++) {  // a1.py:294-297
    F pd[9] = {L.q[0], L.q[1], L.q[2], L.qd[0], L.qd[1], L.qd[2], L.q[0], L.q[1], L.q[2]};
    if (!Ctx::kPlain && K.pd_n >= 0) pd_reading(c, K, ring, tick, true, pd);
    if (!Ctx::kPlain && K.clip_cmd > 0.0f) clip_reading(c, K, ring, tick, true, L.q, pd + 6);
    physics_tick(c, K, tp, L, pose, V3<F>{F(0.0f), F(0.0f), F(0.0f)}, false, Ctx::kPlain ? (const F*)nullptr : pd);
    tick++;
    ring_push(c, ring, tick & (RING - 1), L);
  }
  c.ring_fence();
  L.energy = F(0.0f);
}
// the part of a reset after the settle (L and the ring freshly settled or restored from the settle cache)
template <class F, class Ctx>
ETG_HD void reset_finish(const Ctx& c, const KCfg& K, LaneState<F>& L, float* ring, float* ctl, int* ictl, float* legctl,
                         const float* etgp, float* obs) {
  F pose[3] = {c.par(PR_POSE), c.par(PR_POSE + 1), c.par(PR_POSE + 2)};
  const int tick = K.settle_ticks;
  L.energy = F(0.0f);
  c.st_env_i(ictl, IC_STEP, 0);
  c.st_env_i(ictl, IC_TICK, tick);
  c.st_env_i(ictl, IC_HAS_LAST, 0);
  c.st_env(ctl, CT_RET, F(0.0f)); c.st_env(ctl, CT_LEN, F(0.0f)); c.st_env(ctl, CT_ALIVE, F(1.0f));
  c.st_env(ctl, CT_LAST_BASE + 0, L.p.x); c.st_env(ctl, CT_LAST_BASE + 1, L.p.y); c.st_env(ctl, CT_LAST_BASE + 2, L.p.z);
  for (int j = 0; j < 3; j++) {
    c.st_lane(legctl, LC_LAST_QDES + j, pose[j]);
    c.st_lane(legctl, LC_FX0 + j, pose[j]); c.st_lane(legctl, LC_FX1 + j, pose[j]);  // init_history, action_filter.py:122-126
    c.st_lane(legctl, LC_FY0 + j, pose[j]); c.st_lane(legctl, LC_FY1 + j, pose[j]);
  }
  FootKin<F> fk = foot_kin(c, K, L);
  c.st_lane(legctl, LC_LAST_FOOT_X, fk.fwx);
  F etg[3], imu[6];
  if (Ctx::kPlain || K.etg_on) etg_action(c, K, etgp, 0.0f, etg);
  else etg[0] = etg[1] = etg[2] = F(0.0f);
  // the first reading after reset defines the rpy reference (EnvWrapper.py:79-84)
  const Delayed<F> D0 = ring_read<F>(c, K, ring, tick);
  const V3<F> rpy0 = quat_rpy(D0.qx, D0.qy, D0.qz, D0.qw);
  c.st_env(ctl, CT_FIRST_RPY + 0, rpy0.x); c.st_env(ctl, CT_FIRST_RPY + 1, rpy0.y); c.st_env(ctl, CT_FIRST_RPY + 2, rpy0.z);
  write_obs(c, K, L, D0, rpy0.x, rpy0.y, rpy0.z, etg, L.p.x, L.p.y, L.p.z, obs, imu);
}
template <class F, class Ctx>
ETG_HD void reset_quad(const Ctx& c, const KCfg& K, LaneState<F>& L, float* ring, float* ctl,
                       int* ictl, float* legctl, const float* etgp, float* obs, F offx = F(0.0f), F offy = F(0.0f)) {
  reset_settle(c, K, L, ring, offx, offy);
  reset_finish(c, K, L, ring, ctl, ictl, legctl, etgp, obs);
}

// ------------------------------------------------------------------ state access (parity tests)
// external state row [37]: pos3 quat4(xyzw) linvel3 angvel3 (world frame, pybullet convention) q12 qd12
template <class F, class Ctx> ETG_HD void get_state_quad(const Ctx& c, const LaneState<F>& L, float* st) {
  Rows<F> R = quat_rows(L.qx, L.qy, L.qz, L.qw);
  c.st_row_env(st, ETG_STATE_DIM, 0, L.p.x); c.st_row_env(st, ETG_STATE_DIM, 1, L.p.y); c.st_row_env(st, ETG_STATE_DIM, 2, L.p.z);
  c.st_row_env(st, ETG_STATE_DIM, 3, L.qx); c.st_row_env(st, ETG_STATE_DIM, 4, L.qy);
  c.st_row_env(st, ETG_STATE_DIM, 5, L.qz); c.st_row_env(st, ETG_STATE_DIM, 6, L.qw);
  c.st_row_env(st, ETG_STATE_DIM, 7, dot(R.r0, L.vb)); c.st_row_env(st, ETG_STATE_DIM, 8, dot(R.r1, L.vb));
  c.st_row_env(st, ETG_STATE_DIM, 9, dot(R.r2, L.vb));
  c.st_row_env(st, ETG_STATE_DIM, 10, dot(R.r0, L.wb)); c.st_row_env(st, ETG_STATE_DIM, 11, dot(R.r1, L.wb));
  c.st_row_env(st, ETG_STATE_DIM, 12, dot(R.r2, L.wb));
  for (int j = 0; j < 3; j++) {
    c.st_row_lane(st, ETG_STATE_DIM, 13 + j, 3, L.q[j]);
    c.st_row_lane(st, ETG_STATE_DIM, 25 + j, 3, L.qd[j]);
  }
}
template <class F, class Ctx>
ETG_HD void set_state_quad(const Ctx& c, const float* st, LaneState<F>& L, float* ring, float* ctl, int* ictl, int tick0 = 0) {
  L.p = {c.ld_row_env(st, ETG_STATE_DIM, 0), c.ld_row_env(st, ETG_STATE_DIM, 1), c.ld_row_env(st, ETG_STATE_DIM, 2)};
  F x = c.ld_row_env(st, ETG_STATE_DIM, 3), y = c.ld_row_env(st, ETG_STATE_DIM, 4);
  F z = c.ld_row_env(st, ETG_STATE_DIM, 5), w = c.ld_row_env(st, ETG_STATE_DIM, 6);
  F inv = rsqrt_(x * x + y * y + z * z + w * w);
  L.qx = x * inv; L.qy = y * inv; L.qz = z * inv; L.qw = w * inv;
  Rows<F> R = quat_rows(L.qx, L.qy, L.qz, L.qw);
  V3<F> vw = {c.ld_row_env(st, ETG_STATE_DIM, 7), c.ld_row_env(st, ETG_STATE_DIM, 8), c.ld_row_env(st, ETG_STATE_DIM, 9)};
  V3<F> ww = {c.ld_row_env(st, ETG_STATE_DIM, 10), c.ld_row_env(st, ETG_STATE_DIM, 11), c.ld_row_env(st, ETG_STATE_DIM, 12)};
  // R^T v
  L.vb = {R.r0.x * vw.x + R.r1.x * vw.y + R.r2.x * vw.z, R.r0.y * vw.x + R.r1.y * vw.y + R.r2.y * vw.z,
          R.r0.z * vw.x + R.r1.z * vw.y + R.r2.z * vw.z};
  L.wb = {R.r0.x * ww.x + R.r1.x * ww.y + R.r2.x * ww.z, R.r0.y * ww.x + R.r1.y * ww.y + R.r2.y * ww.z,
          R.r0.z * ww.x + R.r1.z * ww.y + R.r2.z * ww.z};
  for (int j = 0; j < 3; j++) {
    L.q[j] = c.ld_row_lane(st, ETG_STATE_DIM, 13 + j, 3);
    L.qd[j] = c.ld_row_lane(st, ETG_STATE_DIM, 25 + j, 3);
    L.lam[j] = F(0.0f);
  }
  L.lamb = F(0.0f);
  L.contact = F(0.0f);
  for (int sl = 0; sl < RING; sl++) ring_push(c, ring, sl, L);  // re-seed the latency ring
  // tick0: the HIP library passes settle_ticks + RING, so that every later reading is newer than the reset tick and comes
  // from the re-seeded LIVE ring, not from the settle cache (ring_read)
  c.st_env_i(ictl, IC_TICK, tick0);
  c.st_env(ctl, CT_LAST_BASE + 0, L.p.x); c.st_env(ctl, CT_LAST_BASE + 1, L.p.y); c.st_env(ctl, CT_LAST_BASE + 2, L.p.z);
}

}  // namespace etg
