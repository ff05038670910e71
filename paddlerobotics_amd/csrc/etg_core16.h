// etg_core16.h -- the same env.step()/reset() math as etg_core.h, mapped ONE ROBOT = ONE DPP ROW.
//
// Why a second mapping: with one wave per busy SIMD a wave issues one VALU instruction per
// ~4-5 cycles, so a control step costs (instructions per lane) x ~4.5 cycles, and at the headline size
// (4096 robots) the 4-lanes-per-robot kernel only occupies 256 of the chip's 1024 SIMDs
// (DESIGN.md "what bounds the kernel").  Here a robot is the 16 lanes of one DPP row:
//     lane r = 4*leg + sub,   leg = the quad inside the row,   sub = 0,1,2,3
//     sub 0..2 : joint/link `sub` of the leg (hip, thigh, calf)  AND contact row `sub` (n, t1, t2)
//     sub 3    : auxiliary lane (all its link/row contributions are zero; carries base ring words)
// 4096 robots -> 1024 waves = one per SIMD, and each lane executes ~1/2 of the instructions.
// Chain recursions (velocity/acceleration prefix, composite force/inertia suffix) are 2-step
// quad_perm scans, leg-level gathers are quad_perm broadcasts, robot-level reductions are 4-step
// row butterflies (quad xor1, xor2, row_half_mirror, row_mirror), the sequential contact solve
// broadcasts one row's impulse change with row_newbcast, and the Delassus row of a lane is 72
// fused broadcast-FMAs (v_fmac_f32_dpp row_newbcast) against the row's Z vectors.
// State lives in the same HBM arrays as the 4-lane kernel (etg_layout.h), so both kernels are
// interchangeable on one handle.
#pragma once

#include <type_traits>

#include "etg_core.h"

namespace etg {

template <class F> struct State16 {
  V3<F> p;
  F qx, qy, qz, qw;
  V3<F> wb, vb;
  F q, qd;       // this lane's joint (0 on the aux lane)
  F lam;         // this lane's contact row impulse (row = sub: n, t1, t2)
  F contact;     // leg-level flag, replicated in the quad
  F energy;
  int sweeps;    // PGS sweeps this wave executed since the step began (wave-uniform; not part of the stored state)
};

template <class F, class Ctx> ETG_HD State16<F> load_state16(const Ctx& c, const float* base, const float* leg) {
  State16<F> L;
  L.p = {c.ld_env(base, BS_PX), c.ld_env(base, BS_PY), c.ld_env(base, BS_PZ)};
  L.qx = c.ld_env(base, BS_QX); L.qy = c.ld_env(base, BS_QY); L.qz = c.ld_env(base, BS_QZ); L.qw = c.ld_env(base, BS_QW);
  L.wb = {c.ld_env(base, BS_WX), c.ld_env(base, BS_WY), c.ld_env(base, BS_WZ)};
  L.vb = {c.ld_env(base, BS_VX), c.ld_env(base, BS_VY), c.ld_env(base, BS_VZ)};
  const F mj = c.jointf();
  L.q = mj * c.ld_joint(leg, LG_Q);
  L.qd = mj * c.ld_joint(leg, LG_QD);
  L.lam = c.ld_quad(leg, LG_LAM);            // joint lanes: the foot's (n, t1, t2); aux lane: the body contact's normal (LG_LAMB)
  L.contact = c.ld_legf(leg, LG_CONTACT);
  L.energy = F(0.0f);
  L.sweeps = 0;
  return L;
}
template <class F, class Ctx> ETG_HD void store_state16(const Ctx& c, float* base, float* leg, const State16<F>& L) {
  c.st_env(base, BS_PX, L.p.x); c.st_env(base, BS_PY, L.p.y); c.st_env(base, BS_PZ, L.p.z);
  c.st_env(base, BS_QX, L.qx); c.st_env(base, BS_QY, L.qy); c.st_env(base, BS_QZ, L.qz); c.st_env(base, BS_QW, L.qw);
  c.st_env(base, BS_WX, L.wb.x); c.st_env(base, BS_WY, L.wb.y); c.st_env(base, BS_WZ, L.wb.z);
  c.st_env(base, BS_VX, L.vb.x); c.st_env(base, BS_VY, L.vb.y); c.st_env(base, BS_VZ, L.vb.z);
  c.st_joint(leg, LG_Q, L.q); c.st_joint(leg, LG_QD, L.qd); c.st_quad(leg, LG_LAM, L.lam);
  c.st_legf(leg, LG_CONTACT, L.contact);
}

// scans along the 3-link chain held by sub-lanes 0..2 of a quad (the aux lane holds zeros)
// prefix: the inputs are ZERO on the aux lane (they carry a factor qd, masked there), so the quad_perms
// [3,0,1,2] and [3,3,0,1] shift zeros in from the aux lane and no lane mask is needed (two v_add_f32_dpp)
template <class F, class Ctx> ETG_HD F chain_prefix(const Ctx& c, F x, F, F) { return x + c.qdn1(x) + c.qdn2(x); }
template <class F, class Ctx> ETG_HD F chain_suffix(const Ctx& c, F x) { return x + c.qup1(x) + c.qup2(x); }
template <class F, class Ctx> ETG_HD SV<F> chain_prefix(const Ctx& c, SV<F> v, F m1, F m2) {
  return {{chain_prefix(c, v.a.x, m1, m2), chain_prefix(c, v.a.y, m1, m2), chain_prefix(c, v.a.z, m1, m2)},
          {chain_prefix(c, v.l.x, m1, m2), chain_prefix(c, v.l.y, m1, m2), chain_prefix(c, v.l.z, m1, m2)}};
}
template <class F, class Ctx> ETG_HD SV<F> chain_suffix(const Ctx& c, SV<F> v) {
  return {{chain_suffix(c, v.a.x), chain_suffix(c, v.a.y), chain_suffix(c, v.a.z)},
          {chain_suffix(c, v.l.x), chain_suffix(c, v.l.y), chain_suffix(c, v.l.z)}};
}
template <class F, class Ctx> ETG_HD SV<F> quad_bcast(const Ctx& c, SV<F> v, int j) {
  return {{c.qb(v.a.x, j), c.qb(v.a.y, j), c.qb(v.a.z, j)}, {c.qb(v.l.x, j), c.qb(v.l.y, j), c.qb(v.l.z, j)}};
}

// per-lane constants of a physics tick, read from the LDS parameter column ONCE per kernel and kept in
// registers over the 13 (step) / 500 (settle) ticks: a lone wave per SIMD cannot hide the LDS latency of
// re-reading them at the top of every tick (phase profile: +~1000 cycles per tick)
template <class F> struct TickPar { F kp, kd, qd_des, tau_ff, str, sy, m0, mu, link[10]; V3<F> o1, gw, fext; S3<F> I0s; };
template <class F, class Ctx> ETG_HD TickPar<F> load_tick_par(const Ctx& c) {
  TickPar<F> t;
  // tpar*: straight from the HBM parameter array into registers (no LDS hop) -- issued at kernel start, consumed in
  // the tick loop
  t.kp = c.tpar_joint(PR_KP); t.kd = c.tpar_joint(PR_KD); t.sy = c.tpar(PR_SY); t.m0 = c.tpar(PR_M0); t.mu = c.tpar(PR_MU);
  for (int k = 0; k < 10; k++) t.link[k] = c.tpar_link(k);
  t.o1 = {c.tpar(PR_O1), c.tpar(PR_O1 + 1), c.tpar(PR_O1 + 2)};
  t.gw = {c.tpar(PR_G), c.tpar(PR_G + 1), c.tpar(PR_G + 2)};
  t.I0s = {c.tpar(PR_I0), c.tpar(PR_I0 + 1), c.tpar(PR_I0 + 2), c.tpar(PR_I0 + 3), c.tpar(PR_I0 + 4), c.tpar(PR_I0 + 5)};
  t.fext = {F(0.0f), F(0.0f), F(0.0f)};   // external trunk force (world frame); control_step16 fills it in
  t.qd_des = F(0.0f); t.tau_ff = F(0.0f); // HYBRID motor commands only (laikago_motor.py:152-167)
  t.str = Ctx::kPlain ? F(1.0f) : c.tpar_joint(PR_STR);   // motor strength ratio (laikago_motor.py:67-76), 1 unless set
  return t;
}

// leg geometry shared by every lane of the quad
template <class F> struct LegGeo { F sa, ca, sh, ch, shk, chk; V3<F> yax, o1, o2, o3, pf, ez2, ez3; };
template <class F, class Ctx> ETG_HD LegGeo<F> leg_geometry(const Ctx& c, const KCfg& K, const V3<F>& o1, F sy, F q_own) {
  LegGeo<F> g;
  F sq, cq;
  sincos_tick_(q_own, sq, cq);  // one sincos per lane (its own joint), shared through the quad
  g.sa = c.qb(sq, 0); g.ca = c.qb(cq, 0);
  g.sh = c.qb(sq, 1); g.ch = c.qb(cq, 1);
  const F sk = c.qb(sq, 2), ck = c.qb(cq, 2);
  g.shk = g.sh * ck + g.ch * sk;
  g.chk = g.ch * ck - g.sh * sk;
  const F zero(0.0f);
  g.yax = {zero, g.ca, g.sa};
  g.ez2 = {g.sh, -(g.sa * g.ch), g.ca * g.ch};
  g.ez3 = {g.shk, -(g.sa * g.chk), g.ca * g.chk};
  g.o1 = o1;
  g.o2 = g.o1 + sy * g.yax;
  g.o3 = g.o2 - F(K.upper_len) * g.ez2;
  g.pf = g.o3 - F(K.lower_len) * g.ez3;
  return g;
}

// ------------------------------------------------------------------ one physics tick, 16 lanes per robot
template <class F, class Ctx>
ETG_HD void physics_tick16(const Ctx& c, const KCfg& K, const TickPar<F>& tp, State16<F>& L, F qdes, bool torque_cmd = false,
                           const F* pd = nullptr,     // pd: (angle, velocity) the PD law reads instead of the true ones (pd_latency), pd[2]: the angle the command clip refers to
                           F live = F(1.0f)) {        // 0 on the lanes of a robot whose episode has ended (fused rollouts, KCfg.stop_at_done): it gets
                                                      // no contact and no joint-stop rows, so it neither sends its wave down the body / joint paths nor
                                                      // holds it in the sweeps; what its registers do from then on is never stored (control_step16_core)
  typedef V3<F> V;
  typedef SV<F> W;
  const F dt(K.dt), zero(0.0f), one(1.0f);
  const F mj = c.jointf();                    // 1 on joint lanes (sub 0..2), 0 on the aux lane
  const auto s0 = c.sub_is(0), s1 = c.sub_is(1);
  const F f0 = sel_(s0, one, zero), f1 = sel_(s1, one, zero), f2 = sel_(c.sub_is(2), one, zero);
  const F m1 = one - f0;                      // sub >= 1 (aux lane results are discarded)
  const F m2 = m1 - f1;                       // sub >= 2

  // ---- PD motor model (laikago_motor.py:165-173), this lane's joint
  // The option code of the all-options instantiations is branch-free inside the tick (a uniform branch splits the scheduler's
  // block: +160 / +200 cycles per tick in the PD and Schur phases): an option that is off clamps at +-1e30 / adds a zero force.
  if (!Ctx::kPlain) {   // a1.py:439-457: the command is clipped to +-clip_cmd around GetMotorAngles(), the delayed reading (pd[2])
    const F clipv((K.clip_cmd > 0.0f && !torque_cmd) ? K.clip_cmd : 1e30f);
    const F qref = pd ? pd[2] : L.q;
    qdes = fminf_(fmaxf_(qdes, qref - clipv), qref + clipv);
  }
  const F qm = (!Ctx::kPlain && pd) ? pd[0] : L.q, qdm = (!Ctx::kPlain && pd) ? pd[1] : L.qd;   // _GetPDObservation, minitaur.py:1195-1199
  // laikago_motor.py:103-175: the law, x strength ratio, then the clip to +-torque_limit; TORQUE mode: ratio x command, no clip
  F tau = Ctx::kPlain ? mj * (-(tp.kp * (L.q - qdes)) - tp.kd * L.qd)
                      : (torque_cmd ? mj * (tp.str * qdes) : mj * (tp.str * ((-(tp.kp * (qm - qdes)) - tp.kd * (qdm - tp.qd_des)) + tp.tau_ff)));
  if (!Ctx::kPlain) {
    const F tlim((K.torque_limit > 0.0f && !torque_cmd) ? K.torque_limit : 1e30f);
    tau = fminf_(fmaxf_(tau, -tlim), tlim);
  }

  // ---- leg geometry, this lane's link frame R_s = Rx(a) Ry(theta_s), theta = (0, h, h+k)
  const LegGeo<F> g = leg_geometry(c, K, tp.o1, tp.sy, L.q);
  const F ct = sel_(s0, one, sel_(s1, g.ch, g.chk)), st = sel_(s0, zero, sel_(s1, g.sh, g.shk));
  Fr<F> R = {{ct, g.sa * st, -(g.ca * st)}, g.yax, {st, -(g.sa * ct), g.ca * ct}};
  V os = {sel_(s0, g.o1.x, sel_(s1, g.o2.x, g.o3.x)), sel_(s0, g.o1.y, sel_(s1, g.o2.y, g.o3.y)),
          sel_(s0, g.o1.z, sel_(s1, g.o2.z, g.o3.z))};
  V zs = {f0, m1 * g.ca, m1 * g.sa};          // joint axis: x for the hip, y' for thigh and calf
  W S = {zs, cross(os, zs)};
  RBI<F> I = link_inertia16(tp.link[0], V{tp.link[1], tp.link[2], tp.link[3]},
                          S3<F>{tp.link[4], tp.link[5], tp.link[6], tp.link[7], tp.link[8], tp.link[9]}, R, os);
  c.phase(0);
  // ---- RNEA along the chain (prefix scans), bias forces (suffix scan)
  Rows<F> Rw = quat_rows(L.qx, L.qy, L.qz, L.qw);
  // ---- contact point of this lane's row: contact detection works on the start-of-tick pose, so on a heightfield the
  // point and the four corner loads of its terrain cell are issued HERE and consumed in phase 4 -- a lone wave would
  // otherwise sit out the load latency once per tick (flat ground: computed in phase 4, nothing to fetch).
  // This lane owns contact row `sub` (n, t1, t2) of its leg.  With body contacts (EtgConfig.body_contacts, the KNEE
  // instantiations) the aux lane owns a 4th row of the leg: the NORMAL row of the leg's body contact -- a sphere of knee_radius at
  // the knee (calf joint origin, carried by the thigh; body_contacts 1) or the deepest of knee / shin midpoint / trunk corner
  // (body_contacts 2).  The contact's two FRICTION rows are a second row set on the leg's t1 / t2 lanes (finish_tick).
  constexpr bool knee = Ctx::kKnee;   // compile-time: the toe-spheres-only kernels do not contain the rows
  const bool bodies = knee && K.knee != 0;   // (a KNEE instantiation also serves body_contacts = 0 of the all-options layer)
  const auto s3 = c.sub_is(3);
  const F f3 = sel_(s3, one, zero);
  V pc = g.pf, fw;
  F rad(K.foot_radius);
  // The leg's body contact: a sphere of knee_radius at the knee (body_contacts 1) or at the WEIGHTED MEAN of three sphere
  // centres -- knee, shin midpoint, trunk corner next to the hip (body_contacts 2) -- with weights bw0, bw1, bw2 (replicated in
  // the quad): one-hot on the deepest sphere (EtgConfig.body_blend = 0; ties to the earlier candidate, as in the oracle), or
  // exp(-(d_i - d_min) / body_blend), normalised: the contact's impulse is distributed over the spheres, a shin lying along the
  // ground is carried at both ends instead of hopping between them (etgsim.h: body_blend).  Joints: hip and thigh move the
  // knee and the shin midpoint, the calf joint the shin midpoint only, none the trunk corner -- the joint columns of the
  // contact's rows are the weighted sums:  axis_j x (q12 - s12 (r n + o_j))  for hip / thigh,  bw1 axis x (ps - r n - o3)  for
  // the calf, with q12 = bw0 o3 + bw1 ps, s12 = bw0 + bw1.
  V pb = g.o3;               // centre of the leg's body contact sphere
  V ps_ = g.o3;              // the shin midpoint (body_contacts 2)
  F bw0 = one, bw1 = zero, bw2 = zero;
  V q12 = g.o3;              // bw0 o3 + bw1 ps
  F s12 = one;               // bw0 + bw1
  F tap[6];
  auto contact_point = [&]() {
    if (bodies) {
      if (K.knee >= 2) {
        const V ps = g.o3 - F(0.5f * K.lower_len) * g.ez3;
        const V pt = {sel_(g.o1.x > zero, F(K.trunk_half[0]), F(-K.trunk_half[0])),
                      sel_(g.o1.y > zero, F(K.trunk_half[1]), F(-K.trunk_half[1])), F(-K.trunk_half[2])};
        F d0, d1, d2;
        if (Ctx::kFlat) {
          d0 = L.p.z + dot(Rw.r2, g.o3); d1 = L.p.z + dot(Rw.r2, ps); d2 = L.p.z + dot(Rw.r2, pt);
        } else {
          // ONE terrain query per lane: sub-lane j looks up candidate j (the aux lane repeats the knee), the three depths are
          // shared through the quad
          const auto s2 = c.sub_is(2);
          const V q = {sel_(s1, ps.x, sel_(s2, pt.x, g.o3.x)), sel_(s1, ps.y, sel_(s2, pt.y, g.o3.y)), sel_(s1, ps.z, sel_(s2, pt.z, g.o3.z))};
          const V w = {L.p.x + dot(Rw.r0, q), L.p.y + dot(Rw.r1, q), L.p.z + dot(Rw.r2, q)};
          F hgt, nwx, nwy, nwz;
          c.terrain(K, w.x, w.y, hgt, nwx, nwy, nwz);
          const F d = (w.z - hgt) * nwz;
          d0 = c.qb(d, 0); d1 = c.qb(d, 1); d2 = c.qb(d, 2);
        }
        const auto ms = d1 < d0;
        const F best = sel_(ms, d1, d0);
        const auto mt = d2 < best;
        bw0 = sel_(mt, zero, sel_(ms, zero, one));
        bw1 = sel_(mt, zero, sel_(ms, one, zero));
        bw2 = sel_(mt, one, zero);
        if (K.blend_inv > 0.0f) {
          const F dmin = sel_(mt, d2, best);
          const F e0 = exp_((dmin - d0) * F(K.blend_inv)), e1 = exp_((dmin - d1) * F(K.blend_inv)), e2 = exp_((dmin - d2) * F(K.blend_inv));
          const F wi = rcp_(e0 + e1 + e2);
          bw0 = e0 * wi; bw1 = e1 * wi; bw2 = e2 * wi;
        }
        ps_ = ps;
        q12 = bw0 * g.o3 + bw1 * ps;
        s12 = bw0 + bw1;
        pb = q12 + bw2 * pt;
      }
      pc = {sel_(s3, pb.x, g.pf.x), sel_(s3, pb.y, g.pf.y), sel_(s3, pb.z, g.pf.z)};
      rad = sel_(s3, F(K.knee_radius), F(K.foot_radius));
    }
    fw = {L.p.x + dot(Rw.r0, pc), L.p.y + dot(Rw.r1, pc), L.p.z + dot(Rw.r2, pc)};
  };
  if (!Ctx::kFlat) {
    contact_point();
    c.terrain_fetch(K, fw.x, fw.y, tap);
  }
  const V gw = tp.gw;
  V gb = {Rw.r0.x * gw.x + Rw.r1.x * gw.y + Rw.r2.x * gw.z, Rw.r0.y * gw.x + Rw.r1.y * gw.y + Rw.r2.y * gw.z,
          Rw.r0.z * gw.x + Rw.r1.z * gw.y + Rw.r2.z * gw.z};
  W V0 = {L.wb, L.vb};
  W vJ = L.qd * S;
  W Vs = V0 + chain_prefix(c, vJ, m1, m2);
  W Vpar = Vs - vJ;
  W a0 = {{zero, zero, zero}, {-gb.x, -gb.y, -gb.z}};
  W as = a0 + chain_prefix(c, L.qd * crm(Vpar, S), m1, m2);
  W f = apply(I, as) + crf(Vs, apply(I, Vs));
  const F m0 = tp.m0;
  const S3<F> I0s = tp.I0s;
  W fc = chain_suffix(c, f);
  F C = dot(S, fc);
  RBI<F> I0 = {m0, {zero, zero, zero}, I0s};
  W fb0 = apply(I0, a0) + crf(V0, apply(I0, V0));
  c.phase(1);
  // ---- CRBA: composite inertia (suffix scan), this joint's column F = Ic S, leg block H (replicated)
  RBI<F> Ic;
  Ic.m = chain_suffix(c, I.m);
  Ic.h = {chain_suffix(c, I.h.x), chain_suffix(c, I.h.y), chain_suffix(c, I.h.z)};
  Ic.I = {chain_suffix(c, I.I.xx), chain_suffix(c, I.I.yy), chain_suffix(c, I.I.zz),
          chain_suffix(c, I.I.xy), chain_suffix(c, I.I.xz), chain_suffix(c, I.I.yz)};
  W Fs = mj * apply(Ic, S);
  W Fj[3] = {quad_bcast(c, Fs, 0), quad_bcast(c, Fs, 1), quad_bcast(c, Fs, 2)};
  V xax = {one, zero, zero};
  W S0 = {xax, cross(g.o1, xax)}, S1 = {g.yax, cross(g.o2, g.yax)}, S2 = {g.yax, cross(g.o3, g.yax)};
  F H11 = dot(S0, Fj[0]), H12 = dot(S0, Fj[1]), H13 = dot(S0, Fj[2]);
  F H22 = dot(S1, Fj[1]), H23 = dot(S1, Fj[2]), H33 = dot(S2, Fj[2]);
  F cA = H22 * H33 - H23 * H23, cB = H13 * H23 - H12 * H33, cC = H12 * H23 - H13 * H22;
  F cD = H11 * H33 - H13 * H13, cE = H12 * H13 - H11 * H23, cF = H11 * H22 - H12 * H12;
  F idet = rcp_(H11 * cA + H12 * cB + H13 * cC);
  F Hi11 = cA * idet, Hi12 = cB * idet, Hi13 = cC * idet, Hi22 = cD * idet, Hi23 = cE * idet, Hi33 = cF * idet;
  // row `sub` of H^-1 (zero on the aux lane)
  F h0 = f0 * Hi11 + f1 * Hi12 + f2 * Hi13, h1 = f0 * Hi12 + f1 * Hi22 + f2 * Hi23, h2 = f0 * Hi13 + f1 * Hi23 + f2 * Hi33;
  W P = h0 * Fj[0] + h1 * Fj[1] + h2 * Fj[2];          // column `sub` of P = Fm H^-1
  F rl = tau - C;                                      // (aux lane: tau = 0, C = 0 since S = 0 there)
  c.dpp_ready(&rl, 1);                                 // broadcast source of the fused DPP-FMAs below
  c.phase(2);
  // ---- base Schur complement: every term is a sum over the robot's 16 lanes
  // S = M_bb - Fm H^-1 Fm^T with M_bb = trunk + sum of the links' inertias about the base origin: each lane
  // subtracts its P F^T term from ITS link's 6x6 inertia before the one 16-lane reduction (21 sums)
  const F lm[21] = {I.I.xx,
                    I.I.xy, I.I.yy,
                    I.I.xz, I.I.yz, I.I.zz,
                    zero, I.h.z, -I.h.y, I.m,
                    -I.h.z, zero, I.h.x, zero, I.m,
                    I.h.y, -I.h.x, zero, zero, zero, I.m};
  F s[21];
#pragma unroll
  for (int i = 0; i < 6; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) s[i * (i + 1) / 2 + j] = lm[i * (i + 1) / 2 + j] - comp(P, i) * comp(Fs, j);
  // Same arithmetic either way (per-entry butterflies).  On flat ground the scheduler interleaves the 21 chains by
  // itself and shaves 16 instructions off the explicit form; in the heightfield kernels (more registers live) it
  // serialises them -- 4 dependent DPP adds with an s_nop 1 between each -- unless they are written stage by stage.
  if (Ctx::kFlat) {
#pragma unroll
    for (int i = 0; i < 21; i++) s[i] = c.sum16(s[i]);
  } else {
    c.sum16xn(s, 21);
  }
  s[0] = s[0] + I0s.xx;
  s[1] = s[1] + I0s.xy; s[2] = s[2] + I0s.yy;
  s[3] = s[3] + I0s.xz; s[4] = s[4] + I0s.yz; s[5] = s[5] + I0s.zz;
  s[9] = s[9] + m0; s[14] = s[14] + m0; s[20] = s[20] + m0;
  F rb[6];
#pragma unroll
  for (int i = 0; i < 6; i++) rb[i] = comp(f, i) + rl * comp(P, i);
  c.sum16x6(rb);
#pragma unroll
  for (int i = 0; i < 6; i++) rb[i] = -comp(fb0, i) - rb[i];
  if (!Ctx::kPlain) {  // external force on the trunk COM (world frame; zero unless set) -> base frame: R^T f
    rb[3] = rb[3] + Rw.r0.x * tp.fext.x + Rw.r1.x * tp.fext.y + Rw.r2.x * tp.fext.z;
    rb[4] = rb[4] + Rw.r0.y * tp.fext.x + Rw.r1.y * tp.fext.y + Rw.r2.y * tp.fext.z;
    rb[5] = rb[5] + Rw.r0.z * tp.fext.x + Rw.r1.z * tp.fext.y + Rw.r2.z * tp.fext.z;
  }
  F dinv[6], sq[6];
  ldl6(s, dinv, sq);
  fwd6(s, rb);
  {
    const W rbd = cmul(W{{rb[0], rb[1], rb[2]}, {rb[3], rb[4], rb[5]}}, W{{dinv[0], dinv[1], dinv[2]}, {dinv[3], dinv[4], dinv[5]}});
#pragma unroll
    for (int i = 0; i < 6; i++) rb[i] = comp(rbd, i);
  }
  bwd6(s, rb);
  W ab = {{rb[0], rb[1], rb[2]}, {rb[3], rb[4], rb[5]}};
  F qdd = -dot(P, ab);                                 // + row `sub` of H^-1 times (tau - C) of the leg
  c.fmac_qb(qdd, rl, h0, 0); c.fmac_qb(qdd, rl, h1, 1); c.fmac_qb(qdd, rl, h2, 2);   // rl is a phase old: no DPP hazard
  c.phase(3);
  // ---- unconstrained velocity
  V wbs = L.wb + dt * ab.a, vbs = L.vb + dt * ab.l;
  F qds = L.qd + dt * qdd;
  c.phase(4);
  // ---- foot contact (the row's contact point: see `contact_point` above)
  if (Ctx::kFlat) contact_point();
  F phi;
  V dn, dir;   // contact normal and this lane's row direction (n, t1, t2 by sub-lane; n on the aux lane), base coordinates
  V nwo = {zero, zero, one};   // terrain normal under this lane's row point, world frame (the aux lane: under the body sphere)
  if (Ctx::kFlat) {
    phi = fw.z - rad;
    dn = Rw.r2;
    const V d1 = Rw.r0, d2 = Rw.r1;
    dir = {sel_(s0, dn.x, sel_(s1, d1.x, d2.x)), sel_(s0, dn.y, sel_(s1, d1.y, d2.y)), sel_(s0, dn.z, sel_(s1, d1.z, d2.z))};
  } else {
    F hgt, nwx, nwy, nwz;
    c.terrain_finish(K, tap, hgt, nwx, nwy, nwz);
    phi = (fw.z - hgt) * nwz - rad;
    // contact frame in world coordinates: n, t1 = (x_w - (x_w.n) n) / |..|, t2 = n x t1.  With |n| = 1:
    // |x_w - nx n|^2 = 1 - nx^2 and n x t1 = (0, nz, -ny) / |..|.  Each lane rotates only ITS row's direction (and n,
    // which every lane needs for the contact point) into base coordinates.
    const F it1 = rsqrt_hf_(one - nwx * nwx);
    const V nw = {nwx, nwy, nwz};
    nwo = nw;
    const V t1w = {it1 * (one - nwx * nwx), -(it1 * (nwx * nwy)), -(it1 * (nwx * nwz))};
    const V t2w = {zero, it1 * nwz, -(it1 * nwy)};
    const auto s2 = c.sub_is(2);
    const V dw = {sel_(s1, t1w.x, sel_(s2, t2w.x, nw.x)), sel_(s1, t1w.y, sel_(s2, t2w.y, nw.y)), sel_(s1, t1w.z, sel_(s2, t2w.z, nw.z))};
    dn = {Rw.r0.x * nw.x + Rw.r1.x * nw.y + Rw.r2.x * nw.z, Rw.r0.y * nw.x + Rw.r1.y * nw.y + Rw.r2.y * nw.z,
          Rw.r0.z * nw.x + Rw.r1.z * nw.y + Rw.r2.z * nw.z};
    dir = {Rw.r0.x * dw.x + Rw.r1.x * dw.y + Rw.r2.x * dw.z, Rw.r0.y * dw.x + Rw.r1.y * dw.y + Rw.r2.y * dw.z,
           Rw.r0.z * dw.x + Rw.r1.z * dw.y + Rw.r2.z * dw.z};
  }
  auto act = phi < F(K.margin);
  const F rowf = (bodies ? one : mj) * sel_(act, one, zero) * live;   // 1 on the rows of an active foot (/ body sphere)
  V rc = pc - rad * dn;
  V k1 = cross(xax, rc - g.o1), k2 = cross(g.yax, rc - g.o2), k3 = cross(g.yax, rc - g.o3);
  if (knee) {   // the body row pushes along the normal; its joint columns are the weighted sums over the spheres (see above)
    if (Ctx::kFlat) dir = {sel_(s3, dn.x, dir.x), sel_(s3, dn.y, dir.y), sel_(s3, dn.z, dir.z)};
    const V rn = rad * dn;
    const V a12 = q12 - s12 * rn, a3 = ps_ - rn;
    const V b1 = cross(xax, a12 - s12 * g.o1), b2 = cross(g.yax, a12 - s12 * g.o2), b3 = bw1 * cross(g.yax, a3 - g.o3);
    k1 = {sel_(s3, b1.x, k1.x), sel_(s3, b1.y, k1.y), sel_(s3, b1.z, k1.z)};
    k2 = {sel_(s3, b2.x, k2.x), sel_(s3, b2.y, k2.y), sel_(s3, b2.z, k2.z)};
    k3 = {sel_(s3, b3.x, k3.x), sel_(s3, b3.y, k3.y), sel_(s3, b3.z, k3.z)};
  }
  // wave-uniform: does ANY robot of the wave have a body sphere inside the margin this tick?  If not, the tick finishes on the
  // toe-spheres path (hand-scheduled sweeps, no body columns); a robot's result does not depend on which path its wave takes.
  const bool anyb = bodies && c.any_body((rowf * f3) > F(0.5f));
  F Jl0 = rowf * dot(dir, k1), Jl1 = rowf * dot(dir, k2), Jl2 = rowf * dot(dir, k3);
  F HJ0 = Hi11 * Jl0 + Hi12 * Jl1 + Hi13 * Jl2;
  F HJ1 = Hi12 * Jl0 + Hi22 * Jl1 + Hi23 * Jl2;
  F HJ2 = Hi13 * Jl0 + Hi23 * Jl1 + Hi33 * Jl2;
  W Jb = rowf * W{cross(rc, dir), dir};
  W G = Jb - (HJ0 * Fj[0] + HJ1 * Fj[1] + HJ2 * Fj[2]);        // J_b^T - Fm H^-1 J_l^T
  F g6[6] = {G.a.x, G.a.y, G.a.z, G.l.x, G.l.y, G.l.z};
  fwd6(s, g6);
  const W sqv = {{sq[0], sq[1], sq[2]}, {sq[3], sq[4], sq[5]}};
  const W Zv = cmul(W{{g6[0], g6[1], g6[2]}, {g6[3], g6[4], g6[5]}}, sqv);
  F Z[6] = {Zv.a.x, Zv.a.y, Zv.a.z, Zv.l.x, Zv.l.y, Zv.l.z};
  c.phase(5);
  // ---- Delassus row of this lane: A[l'][e] = Z_own . Z_(l',e) over the 6 base coordinates, every term one
  // fused broadcast-FMA (v_fmac_f32_dpp row_newbcast); rows of the own leg add the leg compliance
  // J_l H^-1 J_l^T.  (The 4-lane kernel contracts the same products on the matrix pipe; with one row per
  // lane the DPP form needs no accumulator shuffles and no MFMA latency padding.)
  // warm start (defined here: DPP source below): the normal row x K.warmstart, the friction rows x K.warmstart_t (Bullet's
  // multibody solver restarts friction rows from zero); the aux lane's body normal x K.warmstart_b (= warmstart when the body
  // contact is one persistent point, 0 otherwise: etg_layout.h), the body contact's friction rows start at 0
  F lam = rowf * sel_(s0, F(K.warmstart), sel_(c.sub_is(3), F(K.warmstart_b), F(K.warmstart_t))) * L.lam;
  F hj[3] = {HJ0, HJ1, HJ2};
  c.dpp_ready10(Z, hj, &lam);                                   // one fence for all broadcast sources of this phase
  F A[4][3];
#pragma unroll
  for (int lp = 0; lp < 4; lp++)
#pragma unroll
    for (int e = 0; e < 3; e++) A[lp][e] = c.rbcast(Z[0], 4 * lp + e) * Z[0];
#pragma unroll
  for (int k = 1; k < 6; k++)
#pragma unroll
    for (int lp = 0; lp < 4; lp++)
#pragma unroll
      for (int e = 0; e < 3; e++) c.fmac_rbcast(A[lp][e], Z[k], Z[k], 4 * lp + e);
  F ownl[4];
#pragma unroll
  for (int lp = 0; lp < 4; lp++) ownl[lp] = sel_(c.leg_is(lp), one, zero);
  // (back-to-back DPP-FMAs into ONE accumulator cost a wait state each: the three sums advance side by side)
  F own[3];                                                     // own[e] = sum_j HJ_e[j] Jl_sub[j]
#pragma unroll
  for (int e = 0; e < 3; e++) own[e] = c.qb(hj[0], e) * Jl0;
#pragma unroll
  for (int e = 0; e < 3; e++) c.fmac_qb(own[e], hj[1], Jl1, e);
#pragma unroll
  for (int e = 0; e < 3; e++) c.fmac_qb(own[e], hj[2], Jl2, e);
#pragma unroll
  for (int e = 0; e < 3; e++)
#pragma unroll
    for (int lp = 0; lp < 4; lp++) A[lp][e] = A[lp][e] + ownl[lp] * own[e];
  F Add = hj[0] * Jl0 + hj[1] * Jl1 + hj[2] * Jl2;              // own diagonal
  Add = Add + dot(W{{Z[0], Z[1], Z[2]}, {Z[3], Z[4], Z[5]}}, W{{Z[0], Z[1], Z[2]}, {Z[3], Z[4], Z[5]}});
  const F iA = sel_(rowf > F(0.5f), rcp_(Add), zero);
  c.phase(6);
  // ---- row velocity under the unconstrained motion, warm start
  const F qs0 = c.qb(qds, 0), qs1 = c.qb(qds, 1), qs2 = c.qb(qds, 2);
  V vc = vbs + cross(wbs, rc) + qs0 * k1 + qs1 * k2 + qs2 * k3;
  F u = rowf * dot(dir, vc);
  const F idt(1.0f / K.dt);
  const F pen = phi + F(K.slop);                                // Bullet: penetration = distance + m_linearSlop
  F tgt = (knee ? f0 + f3 : f0) * sel_(pen > zero, -(pen * idt), -(F(K.erp) * pen * idt));   // only normal rows have a target
  if (!Ctx::kPlain) {
    // EtgConfig.foot_restitution: a foot that approaches faster than 0.2 m/s at the START of the tick bounces (branch-free: e = 0 adds 0)
    const V vc0 = L.vb + cross(L.wb, rc) + c.qb(L.qd, 0) * k1 + c.qb(L.qd, 1) * k2 + c.qb(L.qd, 2) * k3;
    const F un0 = rowf * dot(dir, vc0);
    tgt = tgt + f0 * sel_(un0 < F(-0.2f), -(F(K.restitution) * un0), zero);
  }
  c.fmac_rbcast12(u, lam, &A[0][0]);                            // u += sum_(lp,e) lam@row(4 lp + e) * A[lp][e], one asm block
  c.phase(7);
  // ---- projected Gauss-Seidel in the order of Bullet's btMultiBodyConstraintSolver::solveSingleIteration (the oracle's
  // physics_tick states it): per sweep (1) the joint-limit rows, (2) the NORMAL rows of the four feet FR, FL, RR, RL, then the
  // body rows, (3) the friction rows of the four feet.  The owner lane's candidate is broadcast over the row with
  // row_newbcast and applied by every lane.
  // Under the residual stopping rule a robot that has converged is FROZEN for the sweeps the other robots of its wave still
  // need, so its result does not depend on its wave neighbours (batch invariance) and equals the oracle's, which stops per
  // robot.  The default does it lazily (sweep_and_test below: sweep on, then put the values at convergence back); the sweeps
  // still read their per-lane constants through `iAe`, `c0e`, `mue` for the ETG_EAGER_FREEZE build variant, which zeroes them
  // instead (iAe = c0e = 0: every candidate equals the current impulse; mue = 1e30: the cone projection is the identity).
  F iAe = iA, c0e = tgt * iA, mue = tp.mu;
  const F tangf = f1 + f2;
  // owner masks of the three rows of every leg, hoisted out of the sweeps (1 on the lane that owns row e of leg lp)
  F mk0[4], mt[4], mk3[4];
#pragma unroll
  for (int lp = 0; lp < 4; lp++) {
    mk0[lp] = ownl[lp] * f0;
    mt[lp] = ownl[lp] * tangf;
    mk3[lp] = ownl[lp] * f3;
  }
  // ---- joint-limit rows (EtgConfig.joint_limits, bounds of a1.py:186-195 = the URDF limits Bullet enforces with
  // btMultiBodyJointLimitConstraint rows): a joint at or beyond a bound gets a unilateral row along its coordinate, pushing
  // back into the range with the velocity target erp * violation / dt, solved inside the sweeps before the contact rows.
  // Rare: everything about them sits behind ONE wave-uniform test per tick (`anyj`); the common tick runs the sweeps below
  // without them.  This lane owns the row of ITS joint: Z-vector -sgn * zj (zj = D^-1/2 L^-1 P, P = column `sub` of Fm H^-1),
  // leg part sgn * (row `sub` of H^-1).  The row's velocity is rebuilt from the current impulses (no Delassus columns are
  // kept for these rows): u_q = sgn (qd* + [H^-1 J_l^T lam]_sub - zj . ZL), ZL = sum over the robot's rows of lam_r Z_r.
  bool anyj = false;
  F jactf = zero, jlo = zero, jhi = zero;
  if (K.jlim) {   // (compiled into the PLAIN instantiations too: the stops are on by default)
    jlo = sel_(s0, F(K.jlo[0]), sel_(s1, F(K.jlo[1]), F(K.jlo[2])));
    jhi = sel_(s0, F(K.jhi[0]), sel_(s1, F(K.jhi[1]), F(K.jhi[2])));
    jactf = mj * sel_((L.q >= jhi) || (L.q <= jlo), one, zero) * live;
    anyj = c.any(jactf > F(0.5f));
  }
  // solve + apply, instantiated with and without the joint rows: without them their variables are compile-time zeros
  auto finish_tick = [&](auto joints_tag, auto body_tag) {
  constexpr bool joints = decltype(joints_tag)::value;
  constexpr bool body = knee && decltype(body_tag)::value;      // some robot of the wave has a body sphere inside the margin
  // ---- body normal rows (body == true only): the columns of the four body normal rows (aux lanes).  (Measured and dropped:
  // making these columns and solving the rows only in sweeps where some body normal row of the wave would change its impulse --
  // the per-leg scalar tests cost a lone wave more than the four rows they skip: profiles/r05_ab_experiments.txt section 2.)
  F Ak[4] = {zero, zero, zero, zero};                           // column of the body normal row of leg lp
  if (body) {
#pragma unroll
    for (int lp = 0; lp < 4; lp++) {
      Ak[lp] = c.rbcast(Z[0], 4 * lp + 3) * Z[0];
#pragma unroll
      for (int k = 1; k < 6; k++) c.fmac_rbcast(Ak[lp], Z[k], Z[k], 4 * lp + 3);
    }
    F ownk = c.qb(hj[0], 3) * Jl0;
    c.fmac_qb(ownk, hj[1], Jl1, 3);
    c.fmac_qb(ownk, hj[2], Jl2, 3);
#pragma unroll
    for (int lp = 0; lp < 4; lp++) Ak[lp] = Ak[lp] + ownl[lp] * ownk;
    // the body normals' warm start (K.warmstart_b; `lam` of the aux lanes) enters every first row's velocity, as the feet's did above
#pragma unroll
    for (int lp = 0; lp < 4; lp++) c.fmac_rbcast(u, lam, Ak[lp], 4 * lp + 3);
    c.phase_p(10);
  }
  // ---- the SECOND row set (body == true only): the two friction rows of the leg's body contact, t1 on sub-lane 1, t2 on
  // sub-lane 2 (nothing on sub-lane 0 and the aux lane).  Bullet solves a contact's friction rows only while its normal impulse
  // is positive: the friction PHASE of these rows is behind a wave-uniform test in every sweep, but the rows themselves are
  // built here, at the start of the tail.  (Built lazily inside the sweeps they cost the average wave less -- a third of the
  // ticks with a sphere inside the margin ever load it -- but everything the build reads, ~100 registers of leg geometry,
  // Schur factor and unconstrained velocities, then stays alive through the sweeps: the closed-loop kernels spilled 140 dwords
  // per lane to scratch memory, and a launch lasts as long as its SLOWEST wave, whose ticks all build the rows anyway:
  // profiles/r05_ab_experiments.txt section 3.)  The rows' velocity `u2` is made once, before the first sweep, and then TRACKED:
  // every impulse change of every phase is applied to it as well (one more broadcast-FMA per row, in a wait state the row's own
  // broadcast needs anyway: GpuCtx16::pgs_normals_body2).  Rebuilding it at every friction phase instead (24 broadcast-FMAs per
  // sweep) cost the slowest wave 12 % of its sweep: profiles/r05_ab_experiments.txt section 11.
  F Z2[6] = {zero, zero, zero, zero, zero, zero}, hj2[3] = {zero, zero, zero};
  // The 32 Delassus columns of the second rows -- BA[lp][e]: second row x (n, t1, t2, body n) of leg lp; BB[lp][t]: second row x
  // second rows of leg lp; AT[lp][t]: FIRST row x second rows of leg lp -- live from the build to the end of the tick.  In the
  // open-loop and step kernels they stay in registers (`sb`); the closed-loop kernels, whose policy tile shares the 512-register
  // budget with the tick, park them in free slots of the lane's LDS parameter column (Ctx::kSlotBLds) and read them back at the
  // start of the solve (it separates the build's register pressure from the sweeps'; the kernels' scratch use sits around
  // the policy tile, not in the tick, and did not move: profiles/r05_ab_experiments.txt section 4).
  constexpr bool b_lds = Ctx::kSlotBLds;
  F sb[32];
#pragma unroll
  for (int k = 0; k < 32; k++) sb[k] = zero;
  F u2s = zero, lam2 = zero, iA2 = zero;
  if constexpr (body) {
    F BA[4][4], BB[4][2], AT[4][2];
    const F rowf2 = tangf * c.qb(rowf, 3);                       // t1 / t2 lane of a leg whose body sphere is inside the margin
    V dnb, dir2;
    if (Ctx::kFlat) {
      dnb = Rw.r2;                                               // flat ground: the body contact's frame is the foot's (world axes)
      dir2 = dir;
    } else {
      // the terrain normal under the body sphere is the aux lane's row normal; frame as for a foot (t1 = x_w projected, t2 = n x t1)
      const V nb = {c.qb(nwo.x, 3), c.qb(nwo.y, 3), c.qb(nwo.z, 3)};
      const F it1 = rsqrt_hf_(one - nb.x * nb.x);
      const V t1w = {it1 * (one - nb.x * nb.x), -(it1 * (nb.x * nb.y)), -(it1 * (nb.x * nb.z))};
      const V t2w = {zero, it1 * nb.z, -(it1 * nb.y)};
      const V dw = {sel_(s1, t1w.x, t2w.x), sel_(s1, t1w.y, t2w.y), sel_(s1, t1w.z, t2w.z)};
      dnb = {Rw.r0.x * nb.x + Rw.r1.x * nb.y + Rw.r2.x * nb.z, Rw.r0.y * nb.x + Rw.r1.y * nb.y + Rw.r2.y * nb.z,
             Rw.r0.z * nb.x + Rw.r1.z * nb.y + Rw.r2.z * nb.z};
      dir2 = {Rw.r0.x * dw.x + Rw.r1.x * dw.y + Rw.r2.x * dw.z, Rw.r0.y * dw.x + Rw.r1.y * dw.y + Rw.r2.y * dw.z,
              Rw.r0.z * dw.x + Rw.r1.z * dw.y + Rw.r2.z * dw.z};
    }
    const V rn2 = F(K.knee_radius) * dnb;
    const V rc2 = pb - rn2;
    const V a12b = q12 - s12 * rn2;
    const V kb1 = cross(xax, a12b - s12 * g.o1), kb2 = cross(g.yax, a12b - s12 * g.o2), kb3 = bw1 * cross(g.yax, (ps_ - rn2) - g.o3);
    const F Jb0 = rowf2 * dot(dir2, kb1), Jb1 = rowf2 * dot(dir2, kb2), Jb2_ = rowf2 * dot(dir2, kb3);
    hj2[0] = Hi11 * Jb0 + Hi12 * Jb1 + Hi13 * Jb2_;
    hj2[1] = Hi12 * Jb0 + Hi22 * Jb1 + Hi23 * Jb2_;
    hj2[2] = Hi13 * Jb0 + Hi23 * Jb1 + Hi33 * Jb2_;
    const W Jbb = rowf2 * W{cross(rc2, dir2), dir2};
    const W G2 = Jbb - (hj2[0] * Fj[0] + hj2[1] * Fj[1] + hj2[2] * Fj[2]);
    F g2[6] = {G2.a.x, G2.a.y, G2.a.z, G2.l.x, G2.l.y, G2.l.z};
    fwd6(s, g2);
    const W Z2v = cmul(W{{g2[0], g2[1], g2[2]}, {g2[3], g2[4], g2[5]}}, sqv);
#pragma unroll
    for (int k = 0; k < 6; k++) Z2[k] = comp(Z2v, k);
    const V vc2 = vbs + cross(wbs, rc2) + qs0 * kb1 + qs1 * kb2 + qs2 * kb3;
    u2s = rowf2 * dot(dir2, vc2);
    c.dpp_ready(Z2, 6);
    c.dpp_ready(hj2, 3);
#pragma unroll
    for (int lp = 0; lp < 4; lp++) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        BA[lp][e] = c.rbcast(Z[0], 4 * lp + e) * Z2[0];
#pragma unroll
        for (int k = 1; k < 6; k++) c.fmac_rbcast(BA[lp][e], Z[k], Z2[k], 4 * lp + e);
      }
#pragma unroll
      for (int t = 0; t < 2; t++) {
        BB[lp][t] = c.rbcast(Z2[0], 4 * lp + 1 + t) * Z2[0];
        AT[lp][t] = c.rbcast(Z2[0], 4 * lp + 1 + t) * Z[0];
#pragma unroll
        for (int k = 1; k < 6; k++) {
          c.fmac_rbcast(BB[lp][t], Z2[k], Z2[k], 4 * lp + 1 + t);
          c.fmac_rbcast(AT[lp][t], Z2[k], Z[k], 4 * lp + 1 + t);
        }
      }
    }
    // rows of the own leg add the leg compliance J_l H^-1 J_l^T
    F oa[4], ob[2], ot[2];
#pragma unroll
    for (int e = 0; e < 4; e++) oa[e] = c.qb(hj[0], e) * Jb0 + c.qb(hj[1], e) * Jb1 + c.qb(hj[2], e) * Jb2_;
#pragma unroll
    for (int t = 0; t < 2; t++) {
      ob[t] = c.qb(hj2[0], 1 + t) * Jb0 + c.qb(hj2[1], 1 + t) * Jb1 + c.qb(hj2[2], 1 + t) * Jb2_;
      ot[t] = c.qb(hj2[0], 1 + t) * Jl0 + c.qb(hj2[1], 1 + t) * Jl1 + c.qb(hj2[2], 1 + t) * Jl2;
    }
#pragma unroll
    for (int lp = 0; lp < 4; lp++) {
#pragma unroll
      for (int e = 0; e < 4; e++) BA[lp][e] = BA[lp][e] + ownl[lp] * oa[e];
#pragma unroll
      for (int t = 0; t < 2; t++) { BB[lp][t] = BB[lp][t] + ownl[lp] * ob[t]; AT[lp][t] = AT[lp][t] + ownl[lp] * ot[t]; }
    }
    F Add2 = hj2[0] * Jb0 + hj2[1] * Jb1 + hj2[2] * Jb2_;
    Add2 = Add2 + dot(Z2v, Z2v);
    iA2 = sel_(rowf2 > F(0.5f), rcp_(Add2), zero);
#pragma unroll
    for (int lp = 0; lp < 4; lp++) {
#pragma unroll
      for (int e = 0; e < 4; e++) { if (b_lds) c.slotb_st(4 * lp + e, BA[lp][e]); else sb[4 * lp + e] = BA[lp][e]; }
#pragma unroll
      for (int t = 0; t < 2; t++) {
        if (b_lds) { c.slotb_st(16 + 2 * lp + t, BB[lp][t]); c.slotb_st(24 + 2 * lp + t, AT[lp][t]); }
        else { sb[16 + 2 * lp + t] = BB[lp][t]; sb[24 + 2 * lp + t] = AT[lp][t]; }
      }
    }
    c.phase_p(12);
  }
  F sgn = zero, lamq = zero, iAq = zero, c0q = zero;
  F zj[6] = {zero, zero, zero, zero, zero, zero};
  if (joints) {
    F z6[6] = {P.a.x, P.a.y, P.a.z, P.l.x, P.l.y, P.l.z};
    fwd6(s, z6);
    const W zv = cmul(W{{z6[0], z6[1], z6[2]}, {z6[3], z6[4], z6[5]}}, sqv);
#pragma unroll
    for (int k = 0; k < 6; k++) zj[k] = comp(zv, k);
    sgn = jactf * sel_(L.q >= jhi, -one, one);
    const F viol = fmaxf_(L.q - jhi, jlo - L.q);
    const F Aqq = (f0 * Hi11 + f1 * Hi22 + f2 * Hi33) + dot(zv, zv) + (one - mj);    // (M^-1)_jj; 1 on the aux lane
    iAq = jactf * rcp_(Aqq);
    c0q = (F(K.erp) * viol * idt) * iAq;
  }
  // The whole solve is instantiated per friction model (PYRAMID is a compile-time constant inside): a runtime test per foot
  // inside the sweeps' serial chain cost the all-options kernels 800 cycles per tick (phase profile: 2327 against 1504);
  // and with / without the joint rows.
  auto solve = [&](auto pyramid_tag) {
    constexpr bool pyramid = decltype(pyramid_tag)::value;
    F iAqe = iAq, c0qe = c0q;
    // the second rows' columns as the sweeps use them -- Bn: second row x the 8 normal rows (feet, then body); Bt: x the feet's
    // friction rows (t1, t2 of leg 0, ...); BBr / ATr: the body pairs' own -- and the second rows' velocity under the impulses
    // the sweeps start from (warm-started foot normals; lam2 = 0)
    F Bn[8], Bt[8], BBr[4][2], ATr[4][2], u2 = zero;
#pragma unroll
    for (int k = 0; k < 8; k++) Bn[k] = Bt[k] = zero;
#pragma unroll
    for (int lp = 0; lp < 4; lp++) BBr[lp][0] = BBr[lp][1] = ATr[lp][0] = ATr[lp][1] = zero;
    if constexpr (body) {
      F BA[16];
#pragma unroll
      for (int k = 0; k < 16; k++) BA[k] = b_lds ? c.slotb_ld(k) : sb[k];
#pragma unroll
      for (int lp = 0; lp < 4; lp++) {
        Bn[lp] = BA[4 * lp]; Bn[4 + lp] = BA[4 * lp + 3];
        Bt[2 * lp] = BA[4 * lp + 1]; Bt[2 * lp + 1] = BA[4 * lp + 2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
          BBr[lp][t] = b_lds ? c.slotb_ld(16 + 2 * lp + t) : sb[16 + 2 * lp + t];
          ATr[lp][t] = b_lds ? c.slotb_ld(24 + 2 * lp + t) : sb[24 + 2 * lp + t];
        }
      }
      u2 = c.row2_velocity(u2s, lam, BA, lam2, &BBr[0][0]);
    }
    // which of the 12 joint rows exist for SOME robot of the wave: one wave-uniform bit each, made once per tick (the sweeps
    // test a scalar bit; twelve lane-mask tests per sweep cost ~240 cycles of a 840-cycle sweep)
    unsigned jrows = 0u;
    if (joints) {
#pragma unroll
      for (int lp = 0; lp < 4; lp++)
#pragma unroll
        for (int e = 0; e < 3; e++) {
          const F me = ownl[lp] * (e == 0 ? f0 : (e == 1 ? f1 : f2));
          if (c.any(me * jactf > F(0.5f))) jrows |= 1u << (3 * lp + e);
        }
      jrows = c.uniform_bits(jrows);
    }
    auto joint_phase = [&]() {
      F zl[6], zl0[6];
      const F slq = sgn * lamq;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        F t = lam * Z[k];
        c.opaque(t);          // (see the impulse application below: the same bits whichever instantiation the wave runs)
        if (body) t = t + lam2 * Z2[k];
        zl[k] = t - slq * zj[k];
      }
      c.sum16x6(zl);
#pragma unroll
      for (int k = 0; k < 6; k++) zl0[k] = zl[k];
      F dl[3] = {hj[0] * lam, hj[1] * lam, hj[2] * lam};
      c.opaque3(dl);
      if (body) { dl[0] = dl[0] + hj2[0] * lam2; dl[1] = dl[1] + hj2[1] * lam2; dl[2] = dl[2] + hj2[2] * lam2; }
      const F dj0 = c.qsum(dl[0]), dj1 = c.qsum(dl[1]), dj2 = c.qsum(dl[2]);   // the contact impulses do not change here
      const F qc = qds + (f0 * dj0 + f1 * dj1 + f2 * dj2);
      const F lamq0 = lamq;
#pragma unroll
      for (int lp = 0; lp < 4; lp++)
#pragma unroll
        for (int e = 0; e < 3; e++) {
          if (!(jrows & (1u << (3 * lp + e)))) continue;               // no robot of the wave has this joint at a stop
          const F me = ownl[lp] * (e == 0 ? f0 : (e == 1 ? f1 : f2));
          const F sl = sgn * lamq;
          const F qj = qc + (h0 * c.qb(sl, 0) + h1 * c.qb(sl, 1) + h2 * c.qb(sl, 2));
          F zz = zj[0] * zl[0];
#pragma unroll
          for (int k = 1; k < 6; k++) zz = zz + zj[k] * zl[k];
          const F uq = sgn * (qj - zz);
          const F dq = me * fmaxf_(-lamq, c0qe - uq * iAqe);
          lamq = lamq + dq;
          const F sd = -(sgn * dq);
#pragma unroll
          for (int k = 0; k < 6; k++) zl[k] = zl[k] + c.rbcast(sd * zj[k], 4 * lp + e);
        }
      // the contact rows see the joint impulses' changes: A[c][q] d lam_q = Z_c . dZL + HJ_c[joint] sgn d lam_q
      const F w = sgn * (lamq - lamq0);
      F du = hj[0] * c.qb(w, 0) + hj[1] * c.qb(w, 1) + hj[2] * c.qb(w, 2);
#pragma unroll
      for (int k = 0; k < 6; k++) du = du + Z[k] * (zl[k] - zl0[k]);
      u = u + du;
      if (body) {   // and so do the second rows
        F du2 = hj2[0] * c.qb(w, 0) + hj2[1] * c.qb(w, 1) + hj2[2] * c.qb(w, 2);
#pragma unroll
        for (int k = 0; k < 6; k++) du2 = du2 + Z2[k] * (zl[k] - zl0[k]);
        u2 = u2 + du2;
      }
    };
    // (4) the friction pairs of the body contacts, after the feet's (Bullet: every normal row, then every friction row): the
    // same rule on the second row set, with the coefficient K.body_mu and the leg's body normal impulse (aux lane).  Skipped
    // while no body normal of the wave carries load.
    auto body_friction = [&](F lbn) {     // lbn: the leg's body normal impulse (aux lane) on its four lanes
      c.phase_p(8);
      const auto grip2 = lbn > zero;
      // which legs carry a loaded body contact for SOME robot of the wave (the pairs of the other legs would change nothing for
      // any robot: skipped)
      const auto gm = c.body_mask(grip2);
      if (!c.mask_any(gm)) return;
      const F lim2 = F(K.body_mu) * lbn;
      if constexpr (Ctx::kAsmSweep && !pyramid) {
        // the device build's hand-scheduled pair (GpuCtx16::pgs_pair_body): the same arithmetic as the C++ below, the grip
        // condition folded into the per-lane constants as in the feet's friction phase
        const F iA2g = sel_(grip2, iA2, zero), lim2g = sel_(grip2, lim2, F(1e30f));
#pragma unroll
        for (int lp = 0; lp < 4; lp++) {
          if (!c.mask_leg(gm, lp)) continue;
          c.pgs_pair_body(lam2, u2, u, iA2g, lim2g, ATr[lp][0], ATr[lp][1], BBr[lp][0], BBr[lp][1], mt[lp], lp);
        }
        c.phase_p(11);
        return;
      }
#pragma unroll
      for (int lp = 0; lp < 4; lp++) {
        if (!c.mask_leg(gm, lp)) continue;
        const F lc = lam2 - u2 * iA2;
        F dl;
        if (pyramid) {
          dl = fminf_(fmaxf_(lc, -lim2), lim2) - lam2;
        } else {
          const F sq1 = lc * lc + F(1e-30f);
          const F sc = fminf_(one, lim2 * rsqrt_(sq1 + c.qswap12(sq1)));
          dl = lc * sc - lam2;
        }
        dl = sel_(grip2, dl, zero);
        const F b1 = c.rbcast(dl, 4 * lp + 1), b2 = c.rbcast(dl, 4 * lp + 2);
        u = u + ATr[lp][0] * b1 + ATr[lp][1] * b2;
        u2 = u2 + BBr[lp][0] * b1 + BBr[lp][1] * b2;
        lam2 = lam2 + mt[lp] * dl;
      }
      c.phase_p(11);
    };
    auto pgs_sweep = [&]() {
      if (joints) joint_phase();
      if constexpr (Ctx::kAsmSweep && !pyramid) {
        // the device build's hand-scheduled sweep (GpuCtx16::pgs_normals / pgs_tangents_disc state why): the same arithmetic
        // as the C++ below; the friction skip is folded into the per-lane constants of the friction phase
        if constexpr (body) {
          F lnq, lbn;
          c.pgs_normals_body2(lam, u, u2, iAe, c0e, A, Ak, mk0, mk3, Bn, lnq, lbn);
          const auto grip = lnq > zero;
          c.pgs_tangents_disc2(lam, u, u2, sel_(grip, iAe, zero), sel_(grip, mue * lnq, F(1e30f)), A, mt, Bt);
          body_friction(lbn);
        } else {
          c.pgs_normals(lam, u, iAe, c0e, A, mk0);
          const F lnq = c.qb(lam, 0);
          const auto grip = lnq > zero;
          c.pgs_tangents_disc(lam, u, sel_(grip, iAe, zero), sel_(grip, mue * lnq, F(1e30f)), A, mt);
        }
        return;
      }
      // (2) normal rows: ln = max(0, lam - (u - tgt)/A), as a change: max(-lam, (tgt - u)/A); the owner's lam update sits
      // between the candidate and its broadcast, where the DPP read needs two wait states anyway
#pragma unroll
      for (int lp = 0; lp < 4; lp++) {
        F dln = fmaxf_(-lam, c0e - u * iAe);                          // = max(0, lam + c0 - u / A) - lam
        lam = lam + mk0[lp] * dln;
        const F b = c.rbcast(dln, 4 * lp);
        u = u + A[lp][0] * b;
        if (body) u2 = u2 + Bn[lp] * b;
      }
      if (body) {   // the body normal rows: lk = max(0, lk - (u - tgt)/A)
#pragma unroll
        for (int lp = 0; lp < 4; lp++) {
          F dlk = fmaxf_(-lam, c0e - u * iAe);
          lam = lam + mk3[lp] * dlk;
          F bk = c.rbcast(dlk, 4 * lp + 3);
          u = u + Ak[lp] * bk;
          u2 = u2 + Bn[4 + lp] * bk;
        }
      }
      // (3) friction rows t1, t2 of a foot as ONE block: each of the two lanes computes its row's candidate from the same
      // velocities in the same instruction, the pair is projected on the friction disc mu ln (Bullet's implicit cone,
      // resolveConeFrictionConstraintRows; friction_model 1: each clamped on its own), and the two total changes are
      // broadcast once.  A foot whose normal impulse is not positive keeps its friction impulses (`if (totalImpulse > 0)`).
      const F lnq = c.qb(lam, 0);
      const F lim = mue * lnq;
      const auto grip = lnq > zero;
#pragma unroll
      for (int lp = 0; lp < 4; lp++) {
        const F lc = lam - u * iAe;                                   // this lane's candidate (meaningful on the tangent lanes)
        F dl;
        if (pyramid) {
          dl = fminf_(fmaxf_(lc, -lim), lim) - lam;
        } else {
          const F sq1 = lc * lc + F(1e-30f);                          // (the 1e-30 keeps 0 * rsq(0) off the table)
          const F sc = fminf_(one, lim * rsqrt_(sq1 + c.qswap12(sq1)));     // both lanes add it, as in pgs_tangents_disc
          dl = lc * sc - lam;
        }
        dl = sel_(grip, dl, zero);
        const F b1 = c.rbcast(dl, 4 * lp + 1), b2 = c.rbcast(dl, 4 * lp + 2);
        u = u + A[lp][1] * b1 + A[lp][2] * b2;
        if (body) u2 = u2 + Bt[2 * lp] * b1 + Bt[2 * lp + 1] * b2;
        lam = lam + mt[lp] * dl;
      }
      if constexpr (body) body_friction(c.qb(lam, 3));
    };
    if (K.res_thr > 0.0f) {
      // EtgConfig.solver_residual (etgsim.h): sweep until the robot's largest squared row residual
      // ((lam - lam at the start of the sweep) * A_rr)^2 is <= the threshold, K.iters sweeps at most.
      // ((lam - lam0) A_rr)^2 > thr  <=>  |lam - lam0| > sqrt(thr) / A_rr: one subtraction and one compare per lane and sweep
      // against a tolerance made once per tick (inactive rows: iA = 0, no change, 0 > 0 is false); "any row of my robot"
      // comes from the compare's wave mask (robot_any), not from a 4-stage lane reduction
      // (the votes are taken compare by compare and OR-ed as wave masks: a vote on `a || b` goes through a 0/1 register and a
      // second compare -- three issue slots per sweep)
      F tol = F(K.res_sqrt) * iA, tol2 = F(K.res_sqrt) * iA2;
      if (body) { c.opaque(tol); c.opaque(tol2); }                    // (kept, not re-multiplied in every sweep)
      const F tolq = F(K.res_sqrt) * iAq;
      int it = 0;
      bool more;
      auto frozen = lam < lam;                                        // (all false) robots that have converged
      auto sweep_and_test = [&]() {
#ifdef ETG_EAGER_FREEZE   // A/B build variant: the round-3 form (constants zeroed before the next sweep)
        constexpr bool lazy = false;
#else
        constexpr bool lazy = true;
#endif
        if constexpr (lazy) {
          // A converged robot is frozen LAZILY: it sweeps on with its real constants and is put back to its values at
          // convergence afterwards (two selects, three with joint rows) -- the same result bit for bit as zeroed constants,
          // but the per-robot mask (ballot -> shift -> and -> compare: a 6-deep chain) is only needed AFTER the next sweep,
          // so it leaves the sweep's critical path; the wave's exit test is the compare's wave mask alone.
          const F lam0 = lam, u0 = u, lamq0 = lamq, lam20 = lam2, u20 = u2;
          pgs_sweep();
          it++;
          lam = sel_(frozen, lam0, lam);
          u = sel_(frozen, u0, u);
          if (joints) lamq = sel_(frozen, lamq0, lamq);
          auto moved = c.vote(fabsf_(lam - lam0) > tol);
          if (joints) moved = c.vote_or(moved, c.vote(fabsf_(lamq - lamq0) > tolq));
          if (body) {   // the second rows (inactive rows: iA2 = 0, no change, 0 > 0 is false)
            lam2 = sel_(frozen, lam20, lam2);
            u2 = sel_(frozen, u20, u2);
            moved = c.vote_or(moved, c.vote(fabsf_(lam2 - lam20) > tol2));
          }
          if constexpr (body) {
            more = c.vote_wave(moved);
            if (more) { c.fence(); more = it < K.iters; }               // two scalar branches (merged into one condition the two tests
                                                                        // go through lane masks: 7 scalar instructions instead of 4;
                                                                        // the toe-spheres tick's unrolled exits lose 1 % with it)
          } else {
            more = c.vote_wave(moved) && it < K.iters;
          }
          frozen = !c.vote_robot(moved);
          return;
        }
        static_assert(!body || lazy, "the eager-freeze A/B variant predates the body friction rows");
        const F lam0 = lam, lamq0 = lamq;
        pgs_sweep();
        it++;
        const auto live = joints ? c.robot_any((fabsf_(lam - lam0) > tol) || (fabsf_(lamq - lamq0) > tolq))
                                 : c.robot_any(fabsf_(lam - lam0) > tol);
        iAe = sel_(live, iAe, zero);
        c0e = sel_(live, c0e, zero);
        mue = sel_(live, mue, F(1e30f));
        if (joints) { iAqe = sel_(live, iAqe, zero); c0qe = sel_(live, c0qe, zero); }
        more = c.wave_any(live) && it < K.iters;
      };
      if (Ctx::kPlain && !joints && !body) {
        // The default robot layer: the first 8 sweeps as nested forward exits (falling through costs nothing, the one taken
        // branch per tick is the exit); ticks that need more (a fraction of a percent) enter the loop at the bottom.
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
      } else {   // the all-options instantiations: one copy per friction model (the unrolled form is worth 0.5 %)
        do sweep_and_test(); while (more);
      }
      L.sweeps += it;
    } else if (K.iters == 2 && !joints && !body) {
      // a fixed pair of sweeps (the round-1/2 default) as straight-line code: no loop back-edge inside the tick
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
  // ---- apply impulses: base via the Schur factor, joints via H^-1
  F db[6];
#pragma unroll
  for (int k = 0; k < 6; k++) {
    // A robot's result must not depend on which instantiation of this tail its WAVE runs (with / without body rows, joint rows):
    // the terms the other instantiations add are exact zeros for a robot without such rows, but under -ffp-contract=fast the
    // compiler fuses a product into whatever add follows it, and that differs between the instantiations.  The product common to
    // all of them is therefore made opaque (no instruction: an empty asm) before anything is added to it.
    F t = lam * Z[k];
    c.opaque(t);
    db[k] = body ? t + lam2 * Z2[k] : t;
  }
  const F slq = sgn * lamq;                                     // the joint rows' impulses along the joint coordinates
  if (joints) {
#pragma unroll
    for (int k = 0; k < 6; k++) db[k] = db[k] - slq * zj[k];
  }
  c.sum16x6(db);                                                // the six reductions stage by stage (no DPP wait states)
  {
    const W dbs = cmul(W{{db[0], db[1], db[2]}, {db[3], db[4], db[5]}}, sqv);
#pragma unroll
    for (int k = 0; k < 6; k++) db[k] = comp(dbs, k);
  }
  bwd6(s, db);
  W dB = {{db[0], db[1], db[2]}, {db[3], db[4], db[5]}};
  L.wb = wbs + dB.a;
  L.vb = vbs + dB.l;
  // joint j of this leg receives sum_d HJ_d[j] lam_d
  F dl[3] = {hj[0] * lam, hj[1] * lam, hj[2] * lam};
  c.opaque3(dl);
  if (body) { dl[0] = dl[0] + hj2[0] * lam2; dl[1] = dl[1] + hj2[1] * lam2; dl[2] = dl[2] + hj2[2] * lam2; }
  const F dj0 = c.qsum(dl[0]), dj1 = c.qsum(dl[1]), dj2 = c.qsum(dl[2]);
  L.qd = mj * (qds + (f0 * dj0 + f1 * dj1 + f2 * dj2) - dot(P, dB));
  if (joints) L.qd = L.qd + mj * (h0 * c.qb(slq, 0) + h1 * c.qb(slq, 1) + h2 * c.qb(slq, 2));
  L.lam = lam;
  const F ln_leg = c.qb(lam, 0);
  L.contact = sel_(act && (ln_leg > zero), one, zero);
#ifdef ETG_TRACE_TICKS   // debugging build (tools/first_divergence.py): what this tick decided and solved, per lane
  {
    const F tv[10] = {rowf, phi, lam, lam2, jactf, lamq, bw1 + F(2.0f) * bw2, F((float)L.sweeps), L.q, L.qd};
    c.trace_tick(K, tv);
  }
#endif
  };   // finish_tick
  if (anyb) {
    if (anyj) finish_tick(std::true_type{}, std::true_type{});
    else finish_tick(std::false_type{}, std::true_type{});
  } else {
    if (anyj) finish_tick(std::true_type{}, std::false_type{});
    else finish_tick(std::false_type{}, std::false_type{});
  }
  c.phase(9);
  // ---- semi-implicit Euler
  L.q = L.q + dt * L.qd;
  L.p.x = L.p.x + dt * dot(Rw.r0, L.vb);
  L.p.y = L.p.y + dt * dot(Rw.r1, L.vb);
  L.p.z = L.p.z + dt * dot(Rw.r2, L.vb);
  V th = dt * L.wb;
  F a2_ = dot(th, th);
  F sh2 = F(0.5f) - a2_ * (F(1.0f / 48.0f) - a2_ * F(1.0f / 3840.0f));
  F ch2 = one - a2_ * (F(0.125f) - a2_ * (F(1.0f / 384.0f) - a2_ * F(1.0f / 46080.0f)));
  F dx = th.x * sh2, dy = th.y * sh2, dz = th.z * sh2, dw = ch2;
  F nx = L.qw * dx + L.qx * dw + L.qy * dz - L.qz * dy;
  F ny = L.qw * dy - L.qx * dz + L.qy * dw + L.qz * dx;
  F nz = L.qw * dz + L.qx * dy - L.qy * dx + L.qz * dw;
  F nw_ = L.qw * dw - L.qx * dx - L.qy * dy - L.qz * dz;
  F inv = rsqrt_(nx * nx + ny * ny + nz * nz + nw_ * nw_);
  L.qx = nx * inv; L.qy = ny * inv; L.qz = nz * inv; L.qw = nw_ * inv;
  L.energy = L.energy + fabsf_(tau * L.qd) * dt;
}

// ------------------------------------------------------------------ latency ring, same HBM layout as the 4-lane kernel
// slot k of leg-lane column: k = 0..2 q, 3..5 qd (joint lanes), 6,7 base words written by the aux lane
template <class F, class Ctx> ETG_HD void ring_push16(const Ctx& c, float* ring, int slot, const State16<F>& L) {
  const auto l0 = c.leg_is(0), l1 = c.leg_is(1), l2 = c.leg_is(2);
  F b0 = sel_(l0, L.qx, sel_(l1, L.qz, sel_(l2, L.wb.x, L.wb.z)));
  F b1 = sel_(l0, L.qy, sel_(l1, L.qw, sel_(l2, L.wb.y, F(0.0f))));
  c.st_ring_joint(ring, slot, 0, L.q);
  c.st_ring_joint(ring, slot, 3, L.qd);
  c.st_ring_aux(ring, slot, 6, b0);
  c.st_ring_aux(ring, slot, 7, b1);
}
template <class F> struct Delayed16 { F q, qd; F qx, qy, qz, qw; V3<F> w; };
// the PD law's reading under EtgConfig.pd_latency (minitaur.py:1195-1199): this lane's joint angle and velocity K.pd_n ticks
// ago, blended with the reading before it.  `tick` = ticks completed so far (the newest reading in the ring).  live: during the
// reset settle every reading comes from the ring being written (the cache's copy is the PREVIOUS settle).
template <class F, class Ctx> ETG_HD void pd_reading16(const Ctx& c, const KCfg& K, const float* ring, int tick, bool live, F* pd);
// readings of ticks up to the reset tick come from the settle cache's ring (KCfg.cring), later ones from the live ring
ETG_HD const float* ring_of_tick(const KCfg& K, const float* ring, int t) { return (K.cring != nullptr && t <= K.settle_ticks) ? K.cring : ring; }
template <class F, class Ctx> ETG_HD Delayed16<F> ring_read16(const Ctx& c, const KCfg& K, const float* ring, int tick) {
  F vq, vqd, v6, v7;
  int n = c.uniform_int(c.par(PR_LAT_N));
  const F alpha = c.par(PR_LAT_A);
  if (n < 0) {
    int sl = tick & (RING - 1);
    const float* r0 = ring_of_tick(K, ring, tick);
    vq = c.ld_ring_joint(r0, sl, 0); vqd = c.ld_ring_joint(r0, sl, 3);
    v6 = c.ld_ring_k(r0, sl, 6); v7 = c.ld_ring_k(r0, sl, 7);
  } else {
    int sa = (tick - n) & (RING - 1), sb = (tick - n - 1) & (RING - 1);
    const float *ra = ring_of_tick(K, ring, tick - n), *rb_ = ring_of_tick(K, ring, tick - n - 1);
    const F oma = F(1.0f) - alpha;
    vq = oma * c.ld_ring_joint(ra, sa, 0) + alpha * c.ld_ring_joint(rb_, sb, 0);
    vqd = oma * c.ld_ring_joint(ra, sa, 3) + alpha * c.ld_ring_joint(rb_, sb, 3);
    v6 = oma * c.ld_ring_k(ra, sa, 6) + alpha * c.ld_ring_k(rb_, sb, 6);
    v7 = oma * c.ld_ring_k(ra, sa, 7) + alpha * c.ld_ring_k(rb_, sb, 7);
  }
  Delayed16<F> D;
  D.q = vq; D.qd = vqd;
  // the base words of leg-lane l are the same on its 4 sub-lanes: broadcast from sub-lane 0 of each leg
  D.qx = c.rbcast(v6, 0); D.qy = c.rbcast(v7, 0);
  D.qz = c.rbcast(v6, 4); D.qw = c.rbcast(v7, 4);
  D.w = {c.rbcast(v6, 8), c.rbcast(v7, 8), c.rbcast(v6, 12)};
  return D;
}

template <class F, class Ctx> ETG_HD void pd_reading16(const Ctx& c, const KCfg& K, const float* ring, int tick, bool live, F* pd) {
  const int ta = tick - K.pd_n < 0 ? 0 : tick - K.pd_n, tb = tick - K.pd_n - 1 < 0 ? 0 : tick - K.pd_n - 1;   // (slot 0.. hold the initial reading)
  const float *ra = live ? ring : ring_of_tick(K, ring, ta), *rb_ = live ? ring : ring_of_tick(K, ring, tb);
  const int sa = ta & (RING - 1), sb = tb & (RING - 1);
  const F a(K.pd_a), oma(1.0f - K.pd_a);
  pd[0] = oma * c.ld_ring_joint(ra, sa, 0) + a * c.ld_ring_joint(rb_, sb, 0);
  pd[1] = oma * c.ld_ring_joint(ra, sa, 3) + a * c.ld_ring_joint(rb_, sb, 3);
}
// GetMotorAngles() for A1._ClipMotorCommands (a1.py:439-457; minitaur.py:753-764): this lane's joint angle as the control
// observation sees it -- the robot's control latency (PR_LAT_N / PR_LAT_A, the blend of minitaur.py:1172-1193) -- wrapped to
// [-pi, pi].  `now`: the true angle (latency 0).  Every tick's reading must be in the ring (the caller pushes every tick).
template <class F, class Ctx> ETG_HD F clip_reading16(const Ctx& c, const KCfg& K, const float* ring, int tick, bool live, F now) {
  const int n = c.uniform_int(c.par(PR_LAT_N));
  if (n < 0) return wrap_pi_(now);
  const F alpha = c.par(PR_LAT_A);
  const int ta = tick - n < 0 ? 0 : tick - n, tb = tick - n - 1 < 0 ? 0 : tick - n - 1;
  const float *ra = live ? ring : ring_of_tick(K, ring, ta), *rb_ = live ? ring : ring_of_tick(K, ring, tb);
  return wrap_pi_((F(1.0f) - alpha) * c.ld_ring_joint(ra, ta & (RING - 1), 0) + alpha * c.ld_ring_joint(rb_, tb & (RING - 1), 0));
}

// ------------------------------------------------------------------ ETG + IK: computed by every lane of the leg, each keeps its joint
template <class F, class Ctx>
ETG_HD F etg_action16(const Ctx& c, const KCfg& K, const float* etgp, float t) {
  F tl = sel_(c.leg_is(0) || c.leg_is(3), F(t), F(t + K.etg_T2 * K.etg_T));
  F x0 = F(K.etg_amp) * sin_(F(K.etg_phase0) + tl * F(K.etg_omega));
  F x1 = F(K.etg_amp) * sin_(F(K.etg_phase1) + tl * F(K.etg_omega));
  // the 20 RBF terms of the leg are split over its 4 lanes (sub-lane s takes h = 5 s .. 5 s + 4), then quad-summed
  const F isig(1.0f / K.etg_sigma_sq);
  const auto s0 = c.sub_is(0), s1 = c.sub_is(1), s2 = c.sub_is(2);
  F ax(0.0f), ay(0.0f), az(0.0f);
#pragma unroll
  for (int i = 0; i < ETG_RBF_H / 4; i++) {
    const int q = ETG_RBF_H / 4;
    const F u0 = sel_(s0, F(K.etg_u[i][0]), sel_(s1, F(K.etg_u[i + q][0]), sel_(s2, F(K.etg_u[i + 2 * q][0]), F(K.etg_u[i + 3 * q][0]))));
    const F u1 = sel_(s0, F(K.etg_u[i][1]), sel_(s1, F(K.etg_u[i + q][1]), sel_(s2, F(K.etg_u[i + 2 * q][1]), F(K.etg_u[i + 3 * q][1]))));
    F d0 = x0 - u0, d1 = x1 - u1;
    F r = exp_(-((d0 * d0 + d1 * d1) * isig));
    ax = ax + c.ld_env_sub(etgp, EP_W + i, q) * r;
    ay = ay + c.ld_env_sub(etgp, EP_W + ETG_RBF_H + i, q) * r;
    az = az + c.ld_env_sub(etgp, EP_W + 2 * ETG_RBF_H + i, q) * r;
  }
  ax = c.qsum(ax) + c.ld_env(etgp, EP_B + 0);
  ay = c.qsum(ay) + c.ld_env(etgp, EP_B + 1);
  az = c.qsum(az) + c.ld_env(etgp, EP_B + 2);
  F scale(1.0f);
  const V3<F> posev = par3<F>(c, PR_POSE), bfoot = par3<F>(c, PR_BASE_FOOT), o1 = par3<F>(c, PR_O1);
  const F hipsign = c.par(PR_HIPSIGN);
  F ang[3] = {posev.x, posev.y, posev.z};
  auto pending = c.leg_is(0) || !c.leg_is(0);
  {   // the target as commanded: reachable for every sane gait, so this is all the common path executes
    V3<F> foot = {bfoot.x + ax - o1.x, bfoot.y + ay - o1.y, bfoot.z + az - o1.z};
    F a[3];
    auto ok = pending;
    leg_ik(foot, hipsign, a, ok);
    ang[0] = sel_(ok, a[0], ang[0]); ang[1] = sel_(ok, a[1], ang[1]); ang[2] = sel_(ok, a[2], ang[2]);
    pending = !ok;
  }
  if (c.any(pending)) {   // the 0.95 shrink guard against unreachable targets (rare: a fall-through branch otherwise)
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
  F own = sel_(s0, ang[0] - posev.x, sel_(s1, ang[1] - posev.y, ang[2] - posev.z));
  return c.jointf() * own;
}

// foot kinematics of the leg (replicated in the quad)
template <class F> struct FootKin16 { F fwx, fbz, knee_h; };
template <class F, class Ctx> ETG_HD FootKin16<F> foot_kin16(const Ctx& c, const KCfg& K, const State16<F>& L) {
  const LegGeo<F> g = leg_geometry(c, K, par3<F>(c, PR_O1), c.par(PR_SY), L.q);
  Rows<F> Rw = quat_rows(L.qx, L.qy, L.qz, L.qw);
  FootKin16<F> k;
  k.fwx = L.p.x + dot(Rw.r0, g.pf);
  k.fbz = g.pf.z;
  F kx = L.p.x + dot(Rw.r0, g.o3), ky = L.p.y + dot(Rw.r1, g.o3), kz = L.p.z + dot(Rw.r2, g.o3);
  if (Ctx::kFlat) {
    k.knee_h = kz;
  } else {
    F hgt, nx, ny, nz;
    c.terrain(K, kx, ky, hgt, nx, ny, nz);
    k.knee_h = kz - hgt;
  }
  return k;
}

// observation (EnvWrapper.py:60-109), same 49-float row as write_obs()
template <class F, class Ctx>
ETG_HD void write_obs16(const Ctx& c, const KCfg& K, const State16<F>& L, const Delayed16<F>& D, F r0, F r1, F r2, F etg,
                        F lbx, F lby, F lbz, float* obs, F* imu) {
  V3<F> rpy = quat_rpy(D.qx, D.qy, D.qz, D.qw);
  const bool nrm = K.obs_normal != 0;
  const float cdt = K.dt * (float)K.action_repeat;
  F sdis(nrm ? 1.0f / cdt : 1.0f), srpy(nrm ? 10.0f : 1.0f), sdr(nrm ? 2.0f : 1.0f), sqn(nrm ? 10.0f : 1.0f);
  imu[0] = rpy.x - r0; imu[1] = rpy.y - r1; imu[2] = rpy.z - r2;
  imu[3] = D.w.x; imu[4] = D.w.y; imu[5] = D.w.z;
  {   // obs is never null on this path (every caller passes a row buffer: the caller's, the library's sink or the LDS tile)
    c.st_row_env(obs, ETG_OBS_DIM, 0, (L.p.x - lbx) * sdis);
    c.st_row_env(obs, ETG_OBS_DIM, 1, (L.p.y - lby) * sdis);
    c.st_row_env(obs, ETG_OBS_DIM, 2, (L.p.z - lbz) * sdis);
    c.st_row_leg(obs, ETG_OBS_DIM, 3, L.contact);
    for (int k = 0; k < 3; k++) c.st_row_env(obs, ETG_OBS_DIM, 7 + k, imu[k] * srpy);
    for (int k = 0; k < 3; k++) c.st_row_env(obs, ETG_OBS_DIM, 10 + k, imu[3 + k] * sdr);
    const F posej = c.par_joint(PR_POSE), emj = c.par_joint(PR_EMEAN), esj = c.par_joint(PR_ESTD);
    F a = wrap_pi_(D.q);
    c.st_row_joint(obs, ETG_OBS_DIM, 13, nrm ? (a - posej) * sqn : a);
    c.st_row_joint(obs, ETG_OBS_DIM, 25, D.qd);
    c.st_row_joint(obs, ETG_OBS_DIM, 37, nrm ? (etg - emj) / esj : etg);
  }
}

// ------------------------------------------------------------------ one control step (env.step), 16 lanes per robot
// The control-loop variables of a robot that live across steps.  A step kernel loads them once, runs one step
// (env.step) or several (open-loop rollout) on them in registers, and stores them once.
template <class F> struct StepCtl16 {
  int step_count, tick, has_last;
  F last, lbx, lby, lbz, last_fwx;   // last position command, last base position, last foot x (world)
  F ret, len, alive;                 // episode accumulators
  F r0, r1, r2;                      // first rpy reading after reset (EnvWrapper.py:79-84)
  F fx0, fx1, fy0, fy1;              // action filter history (only with K.enable_filter)
};
template <class F, class Ctx>
ETG_HD StepCtl16<F> load_ctl16(const Ctx& c, const KCfg& K, const float* ctl, const int* ictl, const float* legctl) {
  StepCtl16<F> S;
  S.step_count = c.ld_env_i(ictl, IC_STEP);
  S.tick = c.ld_env_i(ictl, IC_TICK);
  S.has_last = c.ld_env_i(ictl, IC_HAS_LAST);
  S.last = c.ld_joint(legctl, LC_LAST_QDES);
  S.lbx = c.ld_env(ctl, CT_LAST_BASE + 0); S.lby = c.ld_env(ctl, CT_LAST_BASE + 1); S.lbz = c.ld_env(ctl, CT_LAST_BASE + 2);
  S.last_fwx = c.ld_legf(legctl, LC_LAST_FOOT_X);
  S.ret = c.ld_env(ctl, CT_RET); S.len = c.ld_env(ctl, CT_LEN); S.alive = c.ld_env(ctl, CT_ALIVE);
  S.r0 = c.ld_env(ctl, CT_FIRST_RPY + 0); S.r1 = c.ld_env(ctl, CT_FIRST_RPY + 1); S.r2 = c.ld_env(ctl, CT_FIRST_RPY + 2);
  S.fx0 = S.fx1 = S.fy0 = S.fy1 = F(0.0f);
  if (!Ctx::kPlain && K.enable_filter) {
    S.fx0 = c.ld_joint(legctl, LC_FX0); S.fx1 = c.ld_joint(legctl, LC_FX1);
    S.fy0 = c.ld_joint(legctl, LC_FY0); S.fy1 = c.ld_joint(legctl, LC_FY1);
  }
  return S;
}
template <class F, class Ctx>
ETG_HD void store_ctl16(const Ctx& c, const KCfg& K, const StepCtl16<F>& S, float* ctl, int* ictl, float* legctl) {
  c.st_env_i(ictl, IC_STEP, S.step_count);
  c.st_env_i(ictl, IC_TICK, S.tick);
  c.st_env_i(ictl, IC_HAS_LAST, S.has_last);
  c.st_joint(legctl, LC_LAST_QDES, S.last);
  c.st_env(ctl, CT_LAST_BASE + 0, S.lbx); c.st_env(ctl, CT_LAST_BASE + 1, S.lby); c.st_env(ctl, CT_LAST_BASE + 2, S.lbz);
  c.st_legf(legctl, LC_LAST_FOOT_X, S.last_fwx);
  c.st_env(ctl, CT_RET, S.ret); c.st_env(ctl, CT_LEN, S.len); c.st_env(ctl, CT_ALIVE, S.alive);
  if (!Ctx::kPlain && K.enable_filter) {
    c.st_joint(legctl, LC_FX0, S.fx0); c.st_joint(legctl, LC_FX1, S.fx1);
    c.st_joint(legctl, LC_FY0, S.fy0); c.st_joint(legctl, LC_FY1, S.fy1);
  }
}

// external trunk force of the step: the set force plus the current random push (separate columns of ctl)
template <class F, class Ctx> ETG_HD V3<F> load_fext16(const Ctx& c, const float* ctl) {
  return {c.ld_env(ctl, CT_FEXT + 0) + c.ld_env(ctl, CT_PUSH + 0), c.ld_env(ctl, CT_FEXT + 1) + c.ld_env(ctl, CT_PUSH + 1),
          c.ld_env(ctl, CT_FEXT + 2) + c.ld_env(ctl, CT_PUSH + 2)};
}

// one env.step on the register-resident control state S and tick constants tp
template <class F, class Ctx>
ETG_HD void control_step16_core(const Ctx& c, const KCfg& K, TickPar<F>& tp, State16<F>& L, StepCtl16<F>& S, float* ring,
                                const float* etgp, F action, F donef, float* obs, F& reward, F& done, float* info,
                                const F* hyb = nullptr,     // hyb: (kp, qd_des, kd, tau_ff) of this lane's motor, HYBRID mode
                                bool want_obs = true,       // false (inner steps of the open-loop rollout): the observation row is not
                                                            // read by anybody -- skip the delayed reading and the row (info needs it)
                                float* rec_q = nullptr,     // action-tape rollouts: rows [N,12] / [N,6] receiving info["joint_angle"] and
                                float* rec_imu = nullptr,   // info["obs-IMU"] of this step (Dynamic_parallel_model.py:63-64)
                                bool skip_dead = false,     // fused rollouts under KCfg.stop_at_done: see below
                                float* obs_end = nullptr) { // skip_dead: a second row buffer that receives the LAST row of a robot whose episode
                                                            // ends in this step (tape rollouts: `obs` is the tape's row of the step then)
  const F mj = c.jointf();
  // An ended episode is not simulated any more (skip_dead; the reference's loops leave at `done`: pretrain.py:137-153,
  // train.py:226-247).  SIMD lanes cannot sit a step out, so a finished robot (S.alive == 0) is taken out of everything that
  // costs its wave time or touches memory: no contact / joint-stop rows in its ticks (physics_tick16: live), every store of
  // the step gated off, its control variables put back afterwards.  The caller stored its state when it finished
  // (rollout_dead_store16) and does not store it again; a wave whose robots have all finished leaves the step loop.
  const F live = skip_dead ? S.alive : F(1.0f);
  const auto is_live = live > F(0.5f);
  if (skip_dead) c.set_gate(is_live);
  // EtgConfig.enable_etg = 0 (Dynamic_parallel_model.py:49 `ETG=0`): no generator, the command is pose_ori + action
  F etg = (Ctx::kPlain || K.etg_on) ? etg_action16<F>(c, K, etgp, (float)(S.step_count + 1) * K.etg_dt) : F(0.0f);
  const bool torque_cmd = !Ctx::kPlain && K.motor_mode == 1;
  const bool hybrid_cmd = !Ctx::kPlain && K.motor_mode == 2 && hyb != nullptr;
  F qdes = (torque_cmd || hybrid_cmd) ? mj * action : mj * (c.par_joint(PR_POSE) + etg + action);
  if (!Ctx::kPlain && K.enable_filter) {
    F y = F(K.fb[0]) * qdes + F(K.fb[1]) * S.fx0 + F(K.fb[2]) * S.fx1 - F(K.fa[1]) * S.fy0 - F(K.fa[2]) * S.fy1;
    if (skip_dead) {   // (a finished robot's filter history stays)
      S.fx1 = sel_(is_live, S.fx0, S.fx1); S.fx0 = sel_(is_live, qdes, S.fx0);
      S.fy1 = sel_(is_live, S.fy0, S.fy1); S.fy0 = sel_(is_live, y, S.fy0);
    } else {
      S.fx1 = S.fx0; S.fx0 = qdes;
      S.fy1 = S.fy0; S.fy0 = y;
    }
    qdes = mj * y;
  }
  const F last = S.last, lbx = S.lbx, lby = S.lby, lbz = S.lbz, last_fwx = S.last_fwx;
  L.energy = F(0.0f);
  L.sweeps = 0;
  const bool interp = !Ctx::kPlain && K.enable_interp && S.has_last;
  // The observation at the end of the step reads ring slots tick_end - n and tick_end - n - 1 only, so
  // only the ticks that land there are pushed: i == ia or i == ib (one modulo per step, not two per tick).
  const int n_lat = c.uniform_int(c.par(PR_LAT_N));
  const int R_ = K.action_repeat;
  const float inv_repeat = 1.0f / (float)K.action_repeat;
  const int mlat = n_lat < 0 ? 0 : n_lat % R_;
  const int ia = R_ - 1 - mlat;
  const int ib = n_lat < 0 ? ia : (ia == 0 ? R_ - 1 : ia - 1);
  if (hybrid_cmd) { tp.kp = hyb[0]; tp.qd_des = mj * hyb[1]; tp.kd = hyb[2]; tp.tau_ff = mj * hyb[3]; }
  const bool pdl = !Ctx::kPlain && K.pd_n >= 0;
  const bool cl = !Ctx::kPlain && K.clip_cmd > 0.0f && !torque_cmd;   // the command clip reads the delayed angle of every tick
  int tick = S.tick;
  for (int i = 0; i < K.action_repeat; i++) {
    float lerp = (float)(i + 1) * inv_repeat;
    F proc = interp ? last + F(lerp) * (qdes - last) : qdes;
    F pd[3] = {L.q, L.qd, L.q};   // EtgConfig.pd_latency: the PD law reads a delayed joint state, so every tick's reading enters the ring
    if (pdl) pd_reading16(c, K, ring, tick, false, pd);
    if (cl) pd[2] = clip_reading16(c, K, ring, tick, false, L.q);
#ifdef ETG_TRACE_TICKS
    c.trace_index(i);
#endif
    physics_tick16(c, K, tp, L, proc, torque_cmd, Ctx::kPlain ? (const F*)nullptr : pd, live);   // (ONE inlined copy of the tick)
    tick++;
    if (pdl || cl || i == ia || i == ib) ring_push16(c, ring, tick & (RING - 1), L);
  }
  // (skip_dead: the control variables of a finished robot stay what they were when it finished -- without holding copies of
  // them through the ticks: counters advance by `live`, the floats fall back on values this function keeps anyway)
  const int li = skip_dead ? c.sel_i(is_live, 1, 0) : 1;
  S.tick = tick - (1 - li) * K.action_repeat;
  S.last = (skip_dead && !Ctx::kPlain) ? sel_(is_live, qdes, last) : qdes;   // (read by the action interpolation only: not in the PLAIN layer)
  S.step_count += li;
  S.has_last |= li;

  const float cdt = K.dt * (float)K.action_repeat;
  Rows<F> Rw = quat_rows(L.qx, L.qy, L.qz, L.qw);
  V3<F> rpy = quat_rpy(L.qx, L.qy, L.qz, L.qw);
  FootKin16<F> fk = foot_kin16(c, K, L);
  const F legw = sel_(c.sub_is(0), F(1.0f), F(0.0f));   // count each leg once in robot-level sums
  F vx = (L.p.x - lbx) * F(1.0f / cdt);
  F torso = fminf_(vx, F(K.vel_d));
  F up = (F(1.0f) - c_prec(rpy.x, F(0.0f), F(0.5f))) * (F(1.0f) - c_prec(rpy.y, F(0.0f), F(0.5f)));
  F feet = c.sum16(legw * (fk.fwx - last_fwx) * F(0.25f)) * F(1.0f / cdt);
  feet = fminf_(feet, F(K.vel_d));
  F energy = c.sum16(L.energy);
  F lost = c.sum16(legw * (F(1.0f) - L.contact));
  F bad = c.sum16(legw * sel_(fk.knee_h < F(0.03f), F(1.0f), F(0.0f)));
  F footcontact = -fmaxf_(lost - F(2.0f), F(0.0f));
  F fz_mean = c.sum16(legw * fk.fbz) * F(0.25f);
  F fz_max = c.max16(fk.fbz);
  auto fin = isfinite_(L.p.x) && isfinite_(L.p.z) && isfinite_(c.rbcast(L.q, 0));
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

  // The observation row: wanted by the caller, needed by info -- or the LAST one of a robot whose episode ends in this step
  // (skip_dead: nobody writes its row again, and the inner steps of an open-loop rollout write none).
  const bool ending = skip_dead && c.any(is_live && (done > F(0.5f)));
  F imu[6] = {F(0.0f), F(0.0f), F(0.0f), F(0.0f), F(0.0f), F(0.0f)};
  if (want_obs || info || ending) {
    const Delayed16<F> D = ring_read16<F>(c, K, ring, tick);
    write_obs16(c, K, L, D, S.r0, S.r1, S.r2, etg, lbx, lby, lbz, obs, imu);
    if (ending && obs_end && obs_end != obs) {
      c.set_gate(is_live && (done > F(0.5f)));
      write_obs16(c, K, L, D, S.r0, S.r1, S.r2, etg, lbx, lby, lbz, obs_end, imu);
      c.set_gate(is_live);
    }
  }
  if (rec_q) c.st_row_joint(rec_q, ETG_ACT_DIM, 0, L.q);
  if (rec_imu)
    for (int k = 0; k < 6; k++) c.st_row_env(rec_imu, 6, k, imu[k]);
  if (info) {
    for (int k = 0; k < 8; k++) c.st_row_env(info, ETG_INFO_DIM, k, terms[k]);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_VELX, vx);
    c.st_row_joint(info, ETG_INFO_DIM, ETG_INFO_ETG_ACT, etg);
    c.st_row_joint(info, ETG_INFO_DIM, ETG_INFO_JOINT_ANGLE, L.q);
    c.st_row_joint(info, ETG_INFO_DIM, ETG_INFO_REAL_ACTION, qdes);
    for (int k = 0; k < 6; k++) c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_OBS_IMU + k, imu[k]);
    c.st_row_leg(info, ETG_INFO_DIM, ETG_INFO_FOOT_CONTACT, L.contact);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_BASE + 0, L.p.x);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_BASE + 1, L.p.y);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_BASE + 2, L.p.z);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_RPY + 0, rpy.x);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_RPY + 1, rpy.y);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_RPY + 2, rpy.z);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_ENERGY, energy);
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_STEPS, F((float)S.step_count));
    c.st_row_env(info, ETG_INFO_DIM, ETG_INFO_SWEEPS, F((float)L.sweeps));
  }
  S.lbx = L.p.x; S.lby = L.p.y; S.lbz = L.p.z;
  S.last_fwx = fk.fwx;
  S.ret = S.ret + S.alive * reward;
  S.len = S.len + S.alive;
  S.alive = sel_(done > F(0.5f), F(0.0f), S.alive);
  if (skip_dead) {
    c.open_gate();
    S.lbx = sel_(is_live, S.lbx, lbx); S.lby = sel_(is_live, S.lby, lby); S.lbz = sel_(is_live, S.lbz, lbz);
    S.last_fwx = sel_(is_live, S.last_fwx, last_fwx);
  }
}

// env.step for one robot row: load the control state, one step, store it
template <class F, class Ctx>
ETG_HD void control_step16(const Ctx& c, const KCfg& K, State16<F>& L, float* ring, float* ctl, int* ictl, float* legctl,
                           const float* etgp, F action, F donef, float* obs, F& reward, F& done, float* info,
                           const F* hyb = nullptr) {
  StepCtl16<F> S = load_ctl16<F>(c, K, ctl, ictl, legctl);
  TickPar<F> tp = load_tick_par<F>(c);
  if (!Ctx::kPlain && K.ext_force) tp.fext = load_fext16<F>(c, ctl);
  control_step16_core(c, K, tp, L, S, ring, etgp, action, donef, obs, reward, done, info, hyb);
  store_ctl16(c, K, S, ctl, ictl, legctl);
}

// ---- fused rollouts under KCfg.stop_at_done: the bookkeeping of a step loop around control_step16_core(..., skip_dead = true)
// after a step: a robot whose episode ended IN this step (it was alive before, the step said done) stores its state -- the
// terminal state -- now; nothing of it is stored again (rollout_store16).  obs_call: stream position of the observation row this
// step wrote (KCfg.noise_call + step index); robots whose row still waits for its sensor noise from the launch's epilogue
// (k_add_noise_rows) -- the last step's rows, and the last row of a robot that ends mid-launch when `noise_ending` -- note it.
template <class F, class Ctx>
ETG_HD void rollout_dead_store16(const Ctx& c, const KCfg& K, const State16<F>& L, F was_alive, F done, bool last_step, bool noise_ending,
                                 int obs_call, float* base, float* leg, int* ictl) {
  const auto was = was_alive > F(0.5f);
  const auto ended = was && (done > F(0.5f));
  if (c.any(ended)) {
    c.set_gate(ended);
    store_state16(c, base, leg, L);
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
// at the end of the launch: the state of the robots that are still running (all of them without stop_at_done)
template <class F, class Ctx>
ETG_HD void rollout_store16(const Ctx& c, const KCfg& K, const State16<F>& L, F alive, float* base, float* leg) {
  if (K.stop_at_done) c.set_gate(alive > F(0.5f));
  store_state16(c, base, leg, L);
  c.open_gate();
}

// open-loop rollout (pretrain.py:129-154): n_steps env.steps with zero residual action in ONE kernel -- the robot's
// state, control variables and tick constants stay in registers between the steps, and the per-launch cost
// (parameter staging, state load/store, launch ramp) is paid once
template <class F, class Ctx>
ETG_HD void rollout_steps16(Ctx& c, const KCfg& K, State16<F>& L, float* base, float* leg, float* ring, float* ctl, int* ictl,
                            float* legctl, const float* etgp, int n_steps, float* obs) {
  StepCtl16<F> S = load_ctl16<F>(c, K, ctl, ictl, legctl);
  TickPar<F> tp = load_tick_par<F>(c);
  V3<F> fext = {F(0.0f), F(0.0f), F(0.0f)};
  if (!Ctx::kPlain && K.ext_force) fext = load_fext16<F>(c, ctl);
  tp.fext = fext;
  F reward, done;
  const bool skip = K.stop_at_done != 0;
  // only the LAST observation of an open-loop rollout is ever read (etg_rollout_openloop hands it to the caller): the inner
  // steps skip the delayed reading and the 49-float row; the ring itself is pushed every step, so the last reading is exact
  // (stop_at_done: a robot's last row is the one of the step that ended its episode)
  for (int s = 0; s < n_steps; s++) {
    if (skip && !c.any(S.alive > F(0.5f))) break;     // every robot of the wave has finished
    if (Ctx::kStepLocal) {
      // Contexts whose tick constants sit in LDS (GpuCtx16W) form everything that is constant over the rollout anew at the top
      // of every control step -- the lane's coordinates go through an empty asm, the tick constants are read back from LDS --
      // instead of keeping ~60 hoisted registers alive from the kernel's head through every tick: the tick's register peak
      // falls, and with it the accumulator-register moves the allocator puts around it.
      c.launder_lane();
      tp = load_tick_par<F>(c);
      tp.fext = fext;
    }
    const F was_alive = S.alive;
    control_step16_core(c, K, tp, L, S, ring, etgp, F(0.0f), F(0.0f), obs, reward, done, (float*)nullptr, (const F*)nullptr,
                        s == n_steps - 1, (float*)nullptr, (float*)nullptr, skip);
    if (skip) rollout_dead_store16(c, K, L, was_alive, done, s == n_steps - 1, true, (int)K.noise_call + s, base, leg, ictl);
  }
  store_ctl16(c, K, S, ctl, ictl, legctl);
  rollout_store16(c, K, L, S.alive, base, leg);
}

// ------------------------------------------------------------------ reset
template <class F, class Ctx>
ETG_HD void reset_settle16(const Ctx& c, const KCfg& K, State16<F>& L, float* ring, F offx = F(0.0f), F offy = F(0.0f)) {
  const F mj = c.jointf();
  L.p = {F(K.init_pos[0]) + offx, F(K.init_pos[1]) + offy, F(K.init_pos[2])};   // (offx, offy): etg_set_reset_offsets
  L.qx = F(0.0f); L.qy = F(0.0f); L.qz = F(0.0f); L.qw = F(1.0f);
  L.wb = {F(0.0f), F(0.0f), F(0.0f)};
  L.vb = {F(0.0f), F(0.0f), F(0.0f)};
  const F pose = mj * c.par_joint(PR_POSE);
  L.q = pose; L.qd = F(0.0f); L.lam = F(0.0f);
  L.contact = F(0.0f);
  L.energy = F(0.0f);
  L.sweeps = 0;
  for (int sl = 0; sl < RING; sl++) ring_push16(c, ring, sl, L);
  int tick = 0;
  const TickPar<F> tp = load_tick_par<F>(c);
  for (int i = 0; i < K.settle_ticks; i++) {
    F pd[3] = {L.q, L.qd, L.q};
    if (!Ctx::kPlain && K.pd_n >= 0) pd_reading16(c, K, ring, tick, true, pd);
    if (!Ctx::kPlain && K.clip_cmd > 0.0f) pd[2] = clip_reading16(c, K, ring, tick, true, L.q);
    physics_tick16(c, K, tp, L, pose, false, Ctx::kPlain ? (const F*)nullptr : pd);
    tick++;
    ring_push16(c, ring, tick & (RING - 1), L);
  }
  L.energy = F(0.0f);
}
// everything of a reset that comes after the settle: control-loop state, episode accumulators, first observation.
// L and the ring are the settled ones -- freshly computed, or restored from the per-robot settle cache.
template <class F, class Ctx>
ETG_HD void reset_finish16(const Ctx& c, const KCfg& K, State16<F>& L, float* ring, float* ctl, int* ictl, float* legctl,
                           const float* etgp, float* obs) {
  const F pose = c.jointf() * c.par_joint(PR_POSE);
  const int tick = K.settle_ticks;
  L.energy = F(0.0f);
  c.st_env_i(ictl, IC_STEP, 0);
  c.st_env_i(ictl, IC_TICK, tick);
  c.st_env_i(ictl, IC_HAS_LAST, 0);
  c.st_env(ctl, CT_RET, F(0.0f)); c.st_env(ctl, CT_LEN, F(0.0f)); c.st_env(ctl, CT_ALIVE, F(1.0f));
  c.st_env(ctl, CT_LAST_BASE + 0, L.p.x); c.st_env(ctl, CT_LAST_BASE + 1, L.p.y); c.st_env(ctl, CT_LAST_BASE + 2, L.p.z);
  c.st_joint(legctl, LC_LAST_QDES, pose);
  c.st_joint(legctl, LC_FX0, pose); c.st_joint(legctl, LC_FX1, pose);
  c.st_joint(legctl, LC_FY0, pose); c.st_joint(legctl, LC_FY1, pose);
  FootKin16<F> fk = foot_kin16(c, K, L);
  c.st_legf(legctl, LC_LAST_FOOT_X, fk.fwx);
  F imu[6];
  F etg = (Ctx::kPlain || K.etg_on) ? etg_action16<F>(c, K, etgp, 0.0f) : F(0.0f);
  // the first reading after reset defines the rpy reference (EnvWrapper.py:79-84)
  const Delayed16<F> D0 = ring_read16<F>(c, K, ring, tick);
  const V3<F> rpy0 = quat_rpy(D0.qx, D0.qy, D0.qz, D0.qw);
  c.st_env(ctl, CT_FIRST_RPY + 0, rpy0.x); c.st_env(ctl, CT_FIRST_RPY + 1, rpy0.y); c.st_env(ctl, CT_FIRST_RPY + 2, rpy0.z);
  write_obs16(c, K, L, D0, rpy0.x, rpy0.y, rpy0.z, etg, L.p.x, L.p.y, L.p.z, obs, imu);
}
template <class F, class Ctx>
ETG_HD void reset_row16(const Ctx& c, const KCfg& K, State16<F>& L, float* ring, float* ctl, int* ictl, float* legctl,
                        const float* etgp, float* obs, F offx = F(0.0f), F offy = F(0.0f)) {
  reset_settle16(c, K, L, ring, offx, offy);
  reset_finish16(c, K, L, ring, ctl, ictl, legctl, etgp, obs);
}

}  // namespace etg
