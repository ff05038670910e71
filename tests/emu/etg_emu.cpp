// etg_emu.cpp -- TEST-ONLY host emulation of the HIP kernels' per-quad math.
//
// Compiles paddlerobotics_amd/csrc/etg_core.h with the lane scalar F bound to a 4-wide
// struct (one robot's quad executed in lock-step on the host), so the exact source the
// GPU runs can be checked against the oracle in the CPU test suite.  Never part of the
// product: paddlerobotics_amd/ does not build, load or call this file.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "emu_lanes.h"

#include "../../paddlerobotics_amd/csrc/etg_core.h"
#include "../../paddlerobotics_amd/csrc/etg_core16.h"

namespace etg {
// The emulation runs one robot at a time, the GPU several per wave: a converged robot then sits through the sweeps its
// wave neighbours still need, FROZEN.  g_extra_sweeps > 0 makes the emulated robot sit through that many sweeps after
// its own convergence (every tick), so a CPU test can check that they change nothing, bit for bit.
static int g_extra_sweeps = 0;
// g_force_body: the emulated robot behaves as if a wave neighbour had (1) a body sphere inside the margin on every tick (the
// tick finishes on the body path) and (2) a loaded body normal from the first sweep on (the second row set is built at once
// and its friction phase runs every sweep) -- neither may change the robot's own result, bit for bit.
static int g_force_body = 0;
struct WaveAny {
  mutable int forced = 0;
  bool more(bool live) const {
    if (live) { forced = 0; return true; }
    if (forced < g_extra_sweeps) { forced++; return true; }
    forced = 0;
    return false;
  }
};
struct EmuCtxBase {
  WaveAny wa;
  // store gate of the fused rollouts (one robot per emulated wave: the mask is uniform)
  mutable bool gate = true;
  void set_gate(B4 m) const { gate = any(m); }
  void open_gate() const { gate = true; }
  int sel_i(B4 m, int a, int b) const { return any(m) ? a : b; }
  bool wave_any(B4 b) const { return wa.more(any(b)); }
  B4 robot_any(B4 b) const { const bool a = any(b); B4 r; for (int l = 0; l < 4; l++) r.v[l] = a; return r; }
  int env, N;
  const float* parp;
  F4 par(int k) const { return ld_lane(parp, k); }
  F4 tpar(int k) const { return par(k); }
  int NL() const { return 4 * N; }
  F4 ld_lane(const float* p, int f) const { F4 r; for (int l = 0; l < 4; l++) r.v[l] = p[(size_t)f * NL() + 4 * env + l]; return r; }
  void st_lane(float* p, int f, F4 v) const { if (!gate) return; for (int l = 0; l < 4; l++) p[(size_t)f * NL() + 4 * env + l] = v.v[l]; }
  F4 ld_env(const float* p, int f) const { return F4(p[(size_t)f * N + env]); }
  void st_env(float* p, int f, F4 v) const { if (!gate) return; p[(size_t)f * N + env] = v.v[0]; }
  int ld_env_i(const int* p, int f) const { return p[(size_t)f * N + env]; }
  void st_env_i(int* p, int f, int v) const { if (!gate) return; p[(size_t)f * N + env] = v; }
  void st_ring(float* r, int slot, int k, F4 v) const { if (!gate) return; for (int l = 0; l < 4; l++) r[((size_t)slot * 8 + k) * NL() + 4 * env + l] = v.v[l]; }
  F4 ld_ring(const float* r, int slot, int k) const { F4 o; for (int l = 0; l < 4; l++) o.v[l] = r[((size_t)slot * 8 + k) * NL() + 4 * env + l]; return o; }
  void ring_fence() const {}
  void phase(int) const {}
  void st_row_env(float* p, int rowlen, int col, F4 v) const { if (!gate) return; p[(size_t)env * rowlen + col] = v.v[0]; }
  void st_row_lane(float* p, int rowlen, int col0, int stride, F4 v) const { if (!gate) return; for (int l = 0; l < 4; l++) p[(size_t)env * rowlen + col0 + stride * l] = v.v[l]; }
  F4 ld_row_env(const float* p, int rowlen, int col) const { return F4(p[(size_t)env * rowlen + col]); }
  F4 ld_row_lane(const float* p, int rowlen, int col0, int stride) const { F4 o; for (int l = 0; l < 4; l++) o.v[l] = p[(size_t)env * rowlen + col0 + stride * l]; return o; }
  F4 qsum(F4 a) const { return F4((a.v[0] + a.v[1]) + (a.v[2] + a.v[3])); }
  F4 qmax(F4 a) const { return F4(fmaxf(fmaxf(a.v[0], a.v[1]), fmaxf(a.v[2], a.v[3]))); }
  F4 qbcast(F4 a, int j) const { return F4(a.v[j]); }
  // acc[i] (at lane l) += a@lane_i * b@lane_l   (v_mfma_f32_4x4x1_16b_f32 semantics, one block per quad)
  void quad_outer(F4 a, F4 b, F4* acc) const {
    for (int i = 0; i < 4; i++)
      for (int l = 0; l < 4; l++) acc[i].v[l] = fmaf(a.v[i], b.v[l], acc[i].v[l]);
  }
  B4 lane_is(int j) const { B4 r; for (int l = 0; l < 4; l++) r.v[l] = (l == j); return r; }
  bool any(B4 b) const { return b.v[0] || b.v[1] || b.v[2] || b.v[3]; }
  unsigned uniform_bits(unsigned v) const { return v; }
  int uniform_int(F4 a) const { return (int)a.v[0]; }
  void terrain(const KCfg& K, F4 x, F4 y, F4& h, F4& nx, F4& ny, F4& nz) const {
    for (int l = 0; l < 4; l++) {
      if (K.terrain == 0 || K.hf == nullptr) { h.v[l] = 0; nx.v[l] = 0; ny.v[l] = 0; nz.v[l] = 1; }
      else heightfield_query(K, env, x.v[l], y.v[l], h.v[l], nx.v[l], ny.v[l], nz.v[l]);
    }
  }
  // the two halves of the query (the kernels issue the corner loads early and finish phases later)
  void terrain_fetch(const KCfg& K, F4 x, F4 y, F4* tap) const {
    for (int l = 0; l < 4; l++) {
      float t[6] = {0, 0, 0, 0, 0, 0};
      if (!(K.terrain == 0 || K.hf == nullptr)) heightfield_fetch(K, env, x.v[l], y.v[l], t);
      for (int k = 0; k < 6; k++) tap[k].v[l] = t[k];
    }
  }
  void terrain_finish(const KCfg& K, const F4* tap, F4& h, F4& nx, F4& ny, F4& nz) const {
    for (int l = 0; l < 4; l++) {
      float t[6];
      for (int k = 0; k < 6; k++) t[k] = tap[k].v[l];
      if (K.terrain == 0 || K.hf == nullptr) { h.v[l] = 0; nx.v[l] = 0; ny.v[l] = 0; nz.v[l] = 1; }
      else heightfield_finish(K, t, h.v[l], nx.v[l], ny.v[l], nz.v[l]);
    }
  }
};

template <bool FLAT, bool PLAIN = false, int BODY = 0> struct EmuCtxT : EmuCtxBase {
  static constexpr bool kFlat = FLAT;
  static constexpr bool kPlain = PLAIN;
  static constexpr int kBody = BODY;
  EmuCtxT(int e, int n, const float* p) { env = e; N = n; parp = p; }
};
typedef EmuCtxT<false> EmuCtx;   // generic-terrain instantiation; the flat fast path is EmuCtxT<true>

// ---- 16 lanes per robot (etg_core16.h): lane r = 4*leg + sub
struct EmuCtx16Base {
  WaveAny wa;
#ifdef ETG_TRACE_TICKS
  mutable int trace_i = 0;
  void trace_index(int i) const { trace_i = i; }
  void trace_tick(const KCfg& K, const F16* v) const {
    if (!K.trace || trace_i >= 16) return;
    for (int r = 0; r < 16; r++) {
      float* row = K.trace + (((size_t)env * 16 + trace_i) * 16 + r) * 10;
      for (int k = 0; k < 10; k++) row[k] = v[k].v[r];
    }
  }
#endif
  mutable bool gate = true;   // store gate: see EmuCtxBase
  void set_gate(B16 m) const { gate = any(m); }
  void open_gate() const { gate = true; }
  int sel_i(B16 m, int a, int b) const { return any(m) ? a : b; }
  bool wave_any(B16 b) const { return wa.more(any(b)); }
  B16 robot_any(B16 b) const { const bool a = any(b); B16 r; for (int l = 0; l < 16; l++) r.v[l] = a; return r; }
  B16 vote(B16 b) const { return b; }
  B16 vote_or(B16 a, B16 b) const { return a || b; }
  bool vote_wave(B16 b) const { return wave_any(b); }
  B16 vote_robot(B16 b) const { return robot_any(b); }
  void fence() const {}
  int env, N;
  const float* parp;
  int NL() const { return 4 * N; }
  static int leg(int r) { return r >> 2; }
  static int sub(int r) { return r & 3; }
  static int sc(int r) { return (r & 3) < 2 ? (r & 3) : 2; }
  size_t col(int r) const { return (size_t)4 * env + leg(r); }
  F16 jointf() const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = sub(r) < 3 ? 1.f : 0.f; return o; }
  B16 sub_is(int j) const { B16 o; for (int r = 0; r < 16; r++) o.v[r] = sub(r) == j; return o; }
  B16 leg_is(int j) const { B16 o; for (int r = 0; r < 16; r++) o.v[r] = leg(r) == j; return o; }
  bool any(B16 b) const { for (int r = 0; r < 16; r++) if (b.v[r]) return true; return false; }
  bool any_body(B16 b) const { return g_force_body != 0 || any(b); }   // the two wave-uniform tests of the body paths
  unsigned long long body_mask(B16 b) const {
    if (g_force_body) return ~0ull;
    unsigned long long r = 0;
    for (int l = 0; l < 16; l++) if (b.v[l]) r |= 1ull << l;
    return r;
  }
  bool mask_any(unsigned long long m) const { return m != 0ull; }
  bool mask_leg(unsigned long long m, int lp) const { return (m & (0x000F000F000F000Full << (4 * lp))) != 0ull; }

  unsigned uniform_bits(unsigned v) const { return v; }
  int uniform_int(F16 a) const { return (int)a.v[0]; }
  F16 par(int k) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = parp[(size_t)k * NL() + col(r)]; return o; }
  F16 par_link(int k) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = sub(r) < 3 ? parp[(size_t)(PR_LINK + 10 * sub(r) + k) * NL() + col(r)] : 0.f; return o; }
  F16 tpar(int k) const { return par(k); }
  F16 tpar_joint(int base) const { return par_joint(base); }
  F16 tpar_link(int k) const { return par_link(k); }
  F16 par_joint(int base) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = parp[(size_t)(base + sc(r)) * NL() + col(r)]; return o; }
  // quad (leg) exchanges
  F16 qb(F16 x, int j) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = x.v[(r & ~3) + j]; return o; }
  F16 qperm(F16 x, int a, int b, int c_, int d) const { int p[4] = {a, b, c_, d}; F16 o; for (int r = 0; r < 16; r++) o.v[r] = x.v[(r & ~3) + p[r & 3]]; return o; }
  F16 qup1(F16 x) const { return qperm(x, 1, 2, 3, 3); }
  F16 qup2(F16 x) const { return qperm(x, 2, 3, 3, 3); }
  F16 qdn1(F16 x) const { return qperm(x, 3, 0, 1, 2); }
  F16 qdn2(F16 x) const { return qperm(x, 3, 3, 0, 1); }
  F16 qswap12(F16 x) const { return qperm(x, 0, 2, 1, 3); }
  F16 qsum(F16 x) const { F16 t = x + qperm(x, 1, 0, 3, 2); return t + qperm(t, 2, 3, 0, 1); }
  // robot (row) exchanges: same butterfly order as the DPP sequence of the GPU context
  F16 half_mirror(F16 x) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = x.v[(r & 8) + 7 - (r & 7)]; return o; }
  F16 mirror(F16 x) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = x.v[15 - r]; return o; }
  F16 sum16(F16 x) const { F16 t = qsum(x); t = t + half_mirror(t); return t + mirror(t); }
  F16 max16(F16 x) const {
    F16 t = fmaxf_(x, qperm(x, 1, 0, 3, 2)); t = fmaxf_(t, qperm(t, 2, 3, 0, 1)); t = fmaxf_(t, half_mirror(t)); return fmaxf_(t, mirror(t));
  }
  F16 rbcast(F16 x, int r0) const { return F16(x.v[r0]); }
  void fmac_rbcast(F16& acc, F16 x, F16 y, int r0) const { for (int r = 0; r < 16; r++) acc.v[r] = std::fmaf(x.v[r0], y.v[r], acc.v[r]); }
  void fmac_qb(F16& acc, F16 x, F16 y, int j) const { for (int r = 0; r < 16; r++) acc.v[r] = std::fmaf(x.v[(r & ~3) + j], y.v[r], acc.v[r]); }
  void dpp_ready(F16*, int) const {}
  mutable F16 slotb_[32];
  void slotb_st(int k, F16 v) const { slotb_[k] = v; }
  F16 slotb_ld(int k) const { return slotb_[k]; }
  void opaque(F16&) const {}
  void opaque3(F16*) const {}
  void dpp_ready10(F16*, F16*, F16*) const {}
  void sum16x6(F16* v) const { for (int k = 0; k < 6; k++) v[k] = sum16(v[k]); }
  void sum16xn(F16* v, int n) const { for (int k = 0; k < n; k++) v[k] = sum16(v[k]); }
  void fmac_rbcast12(F16& acc, F16 x, const F16* a) const { for (int i = 0; i < 12; i++) fmac_rbcast(acc, x, a[i], 4 * (i / 3) + i % 3); }
  F16 row2_velocity(F16 acc, F16 x, const F16* a, F16 y, const F16* b) const {   // same four partial sums as the device's asm block
    F16 p[4] = {acc, F16(0.0f), F16(0.0f), F16(0.0f)};
    for (int i = 0; i < 16; i++) fmac_rbcast(p[i & 3], x, a[i], i);
    for (int i = 0; i < 8; i++) fmac_rbcast(p[i & 3], y, b[i], 4 * (i / 2) + 1 + i % 2);
    return (p[0] + p[1]) + (p[2] + p[3]);
  }
  F16 legrot(F16 x, int kk) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = x.v[(r + 4 * kk) & 15]; return o; }
  void quad_outer(F16 a, F16 b, F16* acc) const {
    for (int i = 0; i < 4; i++)
      for (int r = 0; r < 16; r++) acc[i].v[r] = fmaf(a.v[(r & ~3) + i], b.v[r], acc[i].v[r]);
  }
  // memory
  F16 ld_joint(const float* p, int f0) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = p[(size_t)(f0 + sc(r)) * NL() + col(r)]; return o; }
  void st_joint(float* p, int f0, F16 v) const { if (!gate) return; for (int r = 0; r < 16; r++) if (sub(r) < 3) p[(size_t)(f0 + sub(r)) * NL() + col(r)] = v.v[r]; }
  F16 ld_quad(const float* p, int f0) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = p[(size_t)(f0 + sub(r)) * NL() + col(r)]; return o; }
  void st_quad(float* p, int f0, F16 v) const { if (!gate) return; for (int r = 0; r < 16; r++) p[(size_t)(f0 + sub(r)) * NL() + col(r)] = v.v[r]; }
  F16 ld_legf(const float* p, int f) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = p[(size_t)f * NL() + col(r)]; return o; }
  void st_legf(float* p, int f, F16 v) const { if (!gate) return; for (int r = 0; r < 16; r += 4) p[(size_t)f * NL() + col(r)] = v.v[r]; }
  F16 ld_env(const float* p, int f) const { return F16(p[(size_t)f * N + env]); }
  F16 ld_env_sub(const float* p, int f0, int stride) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = p[(size_t)(f0 + stride * (r & 3)) * N + env]; return o; }
  void st_env(float* p, int f, F16 v) const { if (!gate) return; p[(size_t)f * N + env] = v.v[0]; }
  int ld_env_i(const int* p, int f) const { return p[(size_t)f * N + env]; }
  void st_env_i(int* p, int f, int v) const { if (!gate) return; p[(size_t)f * N + env] = v; }
  void st_ring_joint(float* rg, int slot, int k0, F16 v) const { if (!gate) return; for (int r = 0; r < 16; r++) if (sub(r) < 3) rg[((size_t)slot * 8 + k0 + sub(r)) * NL() + col(r)] = v.v[r]; }
  void st_ring_aux(float* rg, int slot, int k, F16 v) const { if (!gate) return; for (int r = 3; r < 16; r += 4) rg[((size_t)slot * 8 + k) * NL() + col(r)] = v.v[r]; }
  F16 ld_ring_joint(const float* rg, int slot, int k0) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = rg[((size_t)slot * 8 + k0 + sc(r)) * NL() + col(r)]; return o; }
  F16 ld_ring_k(const float* rg, int slot, int k) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = rg[((size_t)slot * 8 + k) * NL() + col(r)]; return o; }
  F16 ld_row_motor(const float* p, int rowlen, int stride, int k) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = sub(r) < 3 ? p[(size_t)env * rowlen + stride * (3 * leg(r) + sub(r)) + k] : 0.f; return o; }
  F16 ld_row_joint(const float* p, int rowlen, int col0) const { F16 o; for (int r = 0; r < 16; r++) o.v[r] = sub(r) < 3 ? p[(size_t)env * rowlen + col0 + 3 * leg(r) + sub(r)] : 0.f; return o; }
  void st_row_joint(float* p, int rowlen, int col0, F16 v) const { if (!gate) return; for (int r = 0; r < 16; r++) if (sub(r) < 3) p[(size_t)env * rowlen + col0 + 3 * leg(r) + sub(r)] = v.v[r]; }
  void st_row_leg(float* p, int rowlen, int col0, F16 v) const { if (!gate) return; for (int r = 0; r < 16; r += 4) p[(size_t)env * rowlen + col0 + leg(r)] = v.v[r]; }
  void st_row_env(float* p, int rowlen, int col_, F16 v) const { if (!gate) return; p[(size_t)env * rowlen + col_] = v.v[0]; }
  F16 ld_row_env(const float* p, int rowlen, int col_) const { return F16(p[(size_t)env * rowlen + col_]); }
  void phase(int) const {}
  void phase_p(int) const {}
  void terrain(const KCfg& K, F16 x, F16 y, F16& h, F16& nx, F16& ny, F16& nz) const {
    for (int r = 0; r < 16; r++) {
      if (K.terrain == 0 || K.hf == nullptr) { h.v[r] = 0; nx.v[r] = 0; ny.v[r] = 0; nz.v[r] = 1; }
      else heightfield_query(K, env, x.v[r], y.v[r], h.v[r], nx.v[r], ny.v[r], nz.v[r]);
    }
  }
  void terrain_fetch(const KCfg& K, F16 x, F16 y, F16* tap) const {
    for (int r = 0; r < 16; r++) {
      float t[6] = {0, 0, 0, 0, 0, 0};
      if (!(K.terrain == 0 || K.hf == nullptr)) heightfield_fetch(K, env, x.v[r], y.v[r], t);
      for (int k = 0; k < 6; k++) tap[k].v[r] = t[k];
    }
  }
  void terrain_finish(const KCfg& K, const F16* tap, F16& h, F16& nx, F16& ny, F16& nz) const {
    for (int r = 0; r < 16; r++) {
      float t[6];
      for (int k = 0; k < 6; k++) t[k] = tap[k].v[r];
      if (K.terrain == 0 || K.hf == nullptr) { h.v[r] = 0; nx.v[r] = 0; ny.v[r] = 0; nz.v[r] = 1; }
      else heightfield_finish(K, t, h.v[r], nx.v[r], ny.v[r], nz.v[r]);
    }
  }
};
template <bool FLAT, bool KNEE = false, bool PLAIN = false> struct EmuCtx16T : EmuCtx16Base {
  static constexpr bool kFlat = FLAT;
  static constexpr bool kKnee = KNEE;
  static constexpr bool kPlain = PLAIN;
  static constexpr bool kAsmSweep = false;   // the C++ statement of the sweep (the device build hand-schedules it)
  static constexpr bool kStepLocal = false;
  void launder_lane() {}
#ifdef ETG_EMU_SLOTB_LDS
  static constexpr bool kSlotBLds = true;    // (build variant: the parked form of the body friction rows' Delassus columns)
#else
  static constexpr bool kSlotBLds = false;
#endif
  template <class A> void pgs_normals(F16&, F16&, F16, F16, const A&, const F16*) const {}
  template <class A> void pgs_tangents_disc(F16&, F16&, F16, F16, const A&, const F16*) const {}
  void pgs_pair_body(F16&, F16&, F16&, F16, F16, F16, F16, F16, F16, F16, int) const {}
  template <class A> void pgs_normals_body(F16&, F16&, F16, F16, const A&, const F16*, const F16*, const F16*) const {}
  template <class A> void pgs_normals_body2(F16&, F16&, F16&, F16, F16, const A&, const F16*, const F16*, const F16*, const F16*, F16&, F16&) const {}
  template <class A> void pgs_tangents_disc2(F16&, F16&, F16&, F16, F16, const A&, const F16*, const F16*) const {}
  EmuCtx16T(int e, int n, const float* p) { env = e; N = n; parp = p; }
};

struct Emu {
  int lanes = 4;
  KCfg K;
  ModelF M;
  int N;
  std::vector<float> base, leg, ctl, legctl, etgp, par, ring, hf, reset_off;
  std::vector<int> ictl;
  unsigned obs_calls = 0;
};
}  // namespace etg

using namespace etg;

extern "C" void* emu_create(const EtgConfig* cfg, const EtgRobotModel* model) {
  Emu* e = new Emu();
  e->K = make_kcfg(*cfg, *model);
  e->M = make_modelf(*model);
  e->N = cfg->num_envs;
  size_t N = e->N, NL = 4 * N;
  e->base.assign(BS_N * N, 0.f); e->leg.assign(LG_N * NL, 0.f); e->ctl.assign(CT_N * N, 0.f);
  e->ictl.assign(IC_N * N, 0); e->legctl.assign(LC_N * NL, 0.f); e->etgp.assign(EP_N * N, 0.f);
  e->par.assign(PR_N * NL, 0.f); e->ring.assign((size_t)RING * 8 * NL, 0.f);
  for (size_t k = PR_STR; k < PR_N; k++)
    for (size_t c = 0; c < NL; c++) e->par[k * NL + c] = 1.0f;   // motor strength ratios: 1 unless set
  return e;
}
extern "C" void emu_destroy(void* h) { delete (Emu*)h; }
extern "C" void emu_set_lanes(void* h, int lanes) { ((Emu*)h)->lanes = lanes; }
extern "C" void emu_set_extra_sweeps(int n) { g_extra_sweeps = n; }
extern "C" void emu_set_force_body(int on) { g_force_body = on; }
extern "C" void emu_set_params(void* h, const float* dyn, const float* w, const float* b, int per_env, const uint8_t* mask) {
  Emu* e = (Emu*)h;
  int N = e->N;
  for (int i = 0; i < N; i++) {
    if (mask && !mask[i]) continue;
    if (dyn)
      for (int l = 0; l < 4; l++) {
        float out[PR_DERIVED];
        derive_lane_params(e->M, dyn + (size_t)i * ETG_DYN_DIM, l, e->K.dt, out);
        for (int k = 0; k < PR_DERIVED; k++) e->par[(size_t)k * 4 * N + 4 * i + l] = out[k];
      }
    if (w)
      for (int k = 0; k < 60; k++) e->etgp[(size_t)(EP_W + k) * N + i] = w[(per_env ? (size_t)i * 60 : 0) + k];
    if (b)
      for (int k = 0; k < 3; k++) e->etgp[(size_t)(EP_B + k) * N + i] = b[(per_env ? (size_t)i * 3 : 0) + k];
  }
}
extern "C" void emu_set_heightfield(void* h, const float* hts) {
  Emu* e = (Emu*)h;
  e->hf.assign(hts, hts + (size_t)e->K.hf_nx * e->K.hf_ny * e->K.hf_bands);
  e->K.hf = e->hf.data();
}
extern "C" void emu_set_external_force(void* h, const float* force) {
  Emu* e = (Emu*)h;
  for (int i = 0; i < e->N; i++)
    for (int k = 0; k < 3; k++) e->ctl[(size_t)(CT_FEXT + k) * e->N + i] = force ? force[(size_t)i * 3 + k] : 0.0f;
  e->K.ext_force = force ? 1 : 0;
}
extern "C" void emu_set_motor_strength(void* h, const float* ratios) {
  Emu* e = (Emu*)h;
  const size_t N = e->N;
  for (size_t i = 0; i < N; i++)
    for (int l = 0; l < 4; l++)
      for (int j = 0; j < 3; j++) e->par[(size_t)(PR_STR + j) * 4 * N + 4 * i + l] = ratios ? ratios[i * 12 + 3 * l + j] : 1.0f;
  e->K.strength_on = ratios ? 1 : 0;
}
extern "C" void emu_set_reset_offsets(void* h, const float* xy) {
  Emu* e = (Emu*)h;
  e->reset_off.assign((size_t)2 * e->N, 0.0f);
  if (xy) std::copy(xy, xy + (size_t)2 * e->N, e->reset_off.begin());
}
template <class Ctx> static void emu_reset4(Emu* e, int i, float* obs, float ox, float oy) {
  Ctx c(i, e->N, e->par.data());
  LaneState<F4> L;
  reset_quad(c, e->K, L, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), obs, F4(ox), F4(oy));
  store_state(c, e->base.data(), e->leg.data(), L);
}
template <class Ctx> static void emu_step4(Emu* e, int i, LaneState<F4>& L, const F4* act, F4 dn, float* obs, F4& r, F4& d, float* info, const F4* hyb) {
  Ctx c(i, e->N, e->par.data());
  control_step(c, e->K, L, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), act, dn, obs, r, d, info, hyb);
}
template <class Ctx> static void emu_reset16(Emu* e, int i, float* obs, float ox, float oy) {
  Ctx c(i, e->N, e->par.data());
  State16<F16> S;
  reset_row16(c, e->K, S, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), obs, F16(ox), F16(oy));
  store_state16(c, e->base.data(), e->leg.data(), S);
}
template <class Ctx> static void emu_step16(Emu* e, int i, const float* action, F16 dn, float* obs, F16& r16, F16& d16, float* info) {
  Ctx c(i, e->N, e->par.data());
  State16<F16> S = load_state16<F16>(c, e->base.data(), e->leg.data());
  const bool hybrid = e->K.motor_mode == 2;
  F16 hyb[4];
  for (int k = 0; k < 4; k++) hyb[k] = hybrid ? c.ld_row_motor(action, ETG_HYBRID_DIM, 5, 1 + k) : F16(0.0f);
  control_step16(c, e->K, S, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(),
                 hybrid ? c.ld_row_motor(action, ETG_HYBRID_DIM, 5, 0) : c.ld_row_joint(action, 12, 0), dn, obs, r16, d16,
                 info, hybrid ? hyb : nullptr);
  store_state16(c, e->base.data(), e->leg.data(), S);
}
// what k_add_noise does after the step / reset kernels
static void emu_obs_noise(Emu* e, const uint8_t* mask, float* obs) {
  if (!e->K.noise_on) return;
  for (int i = 0; i < e->N; i++) {
    if (mask && !mask[i]) continue;
    for (unsigned slot = 0; slot < 16; slot++) add_sensor_noise(e->K, i, e->K.noise_call, slot, obs + (size_t)i * ETG_OBS_DIM);
  }
}
extern "C" void emu_set_sensor_noise(void* h, const float* stdev, uint64_t seed) {
  Emu* e = (Emu*)h;
  e->K.noise_on = 0;
  for (int k = 0; k < 5; k++) { e->K.noise_std[k] = stdev ? stdev[k] : 0.0f; if (e->K.noise_std[k] > 0.0f) e->K.noise_on = 1; }
  e->K.noise_seed = seed;
}
extern "C" void emu_reset(void* h, const uint8_t* mask, float* obs) {
  Emu* e = (Emu*)h;
  e->K.noise_call = e->obs_calls++;
  for (int i = 0; i < e->N; i++) {
    if (mask && !mask[i]) continue;
    const float ox = e->reset_off.empty() ? 0.0f : e->reset_off[2 * i], oy = e->reset_off.empty() ? 0.0f : e->reset_off[2 * i + 1];
    if (e->lanes == 16) {
      const bool pl = plain_config(e->K);                       // same instantiation choice as LAUNCH16 in etg_kernels.hip
      const bool kn = e->K.knee != 0, fl = e->K.terrain == 0;   // (DISPATCH16)
      if (fl && pl && kn) emu_reset16<EmuCtx16T<true, true, true>>(e, i, obs, ox, oy);
      else if (fl && pl) emu_reset16<EmuCtx16T<true, false, true>>(e, i, obs, ox, oy);
      else if (fl) emu_reset16<EmuCtx16T<true, true, false>>(e, i, obs, ox, oy);
      else if (pl && kn) emu_reset16<EmuCtx16T<false, true, true>>(e, i, obs, ox, oy);
      else if (pl) emu_reset16<EmuCtx16T<false, false, true>>(e, i, obs, ox, oy);
      else emu_reset16<EmuCtx16T<false, true, false>>(e, i, obs, ox, oy);
      continue;
    }
    const bool pl4 = plain_config(e->K);                        // same choice as LAUNCH4 in etg_kernels.hip
    if (e->K.knee == 3 && e->K.terrain == 0) emu_reset4<EmuCtxT<true, false, 3>>(e, i, obs, ox, oy);
    else if (e->K.knee == 3) emu_reset4<EmuCtxT<false, false, 3>>(e, i, obs, ox, oy);
    else if (e->K.knee && e->K.terrain == 0) emu_reset4<EmuCtxT<true, false, 1>>(e, i, obs, ox, oy);
    else if (e->K.knee) emu_reset4<EmuCtxT<false, false, 1>>(e, i, obs, ox, oy);
    else if (e->K.terrain == 0 && pl4) emu_reset4<EmuCtxT<true, true>>(e, i, obs, ox, oy);
    else if (e->K.terrain == 0) emu_reset4<EmuCtxT<true>>(e, i, obs, ox, oy);
    else if (pl4) emu_reset4<EmuCtxT<false, true>>(e, i, obs, ox, oy);
    else emu_reset4<EmuCtxT<false>>(e, i, obs, ox, oy);
  }
  emu_obs_noise(e, mask, obs);
}
extern "C" void emu_step(void* h, const float* action, const uint8_t* donef, float* obs, float* reward, uint8_t* done, float* info) {
  Emu* e = (Emu*)h;
  e->K.noise_call = e->obs_calls++;
  for (int i = 0; i < e->N; i++) {
    if (e->lanes == 16) {
      F16 r16, d16;
      F16 dn(donef ? (float)donef[i] : 0.f);
      const bool pl = plain_config(e->K);
      const bool kn = e->K.knee != 0, fl = e->K.terrain == 0;   // (DISPATCH16)
      if (fl && pl && kn) emu_step16<EmuCtx16T<true, true, true>>(e, i, action, dn, obs, r16, d16, info);
      else if (fl && pl) emu_step16<EmuCtx16T<true, false, true>>(e, i, action, dn, obs, r16, d16, info);
      else if (fl) emu_step16<EmuCtx16T<true, true, false>>(e, i, action, dn, obs, r16, d16, info);
      else if (pl && kn) emu_step16<EmuCtx16T<false, true, true>>(e, i, action, dn, obs, r16, d16, info);
      else if (pl) emu_step16<EmuCtx16T<false, false, true>>(e, i, action, dn, obs, r16, d16, info);
      else emu_step16<EmuCtx16T<false, true, false>>(e, i, action, dn, obs, r16, d16, info);
      reward[i] = r16.v[0];
      done[i] = d16.v[0] > 0.5f;
      continue;
    }
    EmuCtx c0(i, e->N, e->par.data());
    LaneState<F4> L = load_state<F4>(c0, e->base.data(), e->leg.data());
    F4 act[3], hyb[12];
    const bool hybrid = e->K.motor_mode == 2;
    for (int j = 0; j < 3; j++) {
      act[j] = hybrid ? c0.ld_row_lane(action, ETG_HYBRID_DIM, 5 * j, 15) : c0.ld_row_lane(action, 12, j, 3);
      for (int k = 0; k < 4; k++) hyb[4 * j + k] = hybrid ? c0.ld_row_lane(action, ETG_HYBRID_DIM, 5 * j + 1 + k, 15) : F4(0.0f);
    }
    F4 r, d;
    const F4 dn4(donef ? (float)donef[i] : 0.f);
    const bool pl4 = plain_config(e->K);
    if (e->K.knee == 3 && e->K.terrain == 0) emu_step4<EmuCtxT<true, false, 3>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    else if (e->K.knee == 3) emu_step4<EmuCtxT<false, false, 3>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    else if (e->K.knee && e->K.terrain == 0) emu_step4<EmuCtxT<true, false, 1>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    else if (e->K.knee) emu_step4<EmuCtxT<false, false, 1>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    else if (e->K.terrain == 0 && pl4) emu_step4<EmuCtxT<true, true>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    else if (e->K.terrain == 0) emu_step4<EmuCtxT<true>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    else if (pl4) emu_step4<EmuCtxT<false, true>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    else emu_step4<EmuCtxT<false>>(e, i, L, act, dn4, obs, r, d, info, hybrid ? hyb : nullptr);
    store_state(c0, e->base.data(), e->leg.data(), L);
    reward[i] = r.v[0];
    done[i] = d.v[0] > 0.5f;
  }
  emu_obs_noise(e, nullptr, obs);
}
// ---- fused open-loop rollout (rollout_steps16 / rollout_steps: what k_rollout16 / k_rollout run), one robot per emulated wave.
// stop_at_done = KCfg.stop_at_done of the launch: a finished robot is not simulated any more, its last observation row is the one
// of the step that ended its episode, its state the terminal state.
template <class Ctx> static void emu_rollout16(Emu* e, int i, int n_steps, float* obs) {
  Ctx c(i, e->N, e->par.data());
  State16<F16> S = load_state16<F16>(c, e->base.data(), e->leg.data());
  rollout_steps16(c, e->K, S, e->base.data(), e->leg.data(), e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), n_steps, obs);
}
template <class Ctx> static void emu_rollout4(Emu* e, int i, int n_steps, float* obs) {
  Ctx c(i, e->N, e->par.data());
  LaneState<F4> L = load_state<F4>(c, e->base.data(), e->leg.data());
  rollout_steps(c, e->K, L, e->base.data(), e->leg.data(), e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), n_steps, obs);
}
extern "C" void emu_rollout_openloop(void* h, int n_steps, int stop_at_done, float* obs, float* ret, int* len) {
  Emu* e = (Emu*)h;
  e->K.stop_at_done = stop_at_done;
  e->K.noise_call = e->obs_calls;
  e->obs_calls += (unsigned)n_steps;
  for (int i = 0; i < e->N; i++) {
    if (e->lanes == 16) {
      const bool pl = plain_config(e->K);
      const bool kn = e->K.knee != 0, fl = e->K.terrain == 0;   // (DISPATCH16)
      if (fl && pl && kn) emu_rollout16<EmuCtx16T<true, true, true>>(e, i, n_steps, obs);
      else if (fl && pl) emu_rollout16<EmuCtx16T<true, false, true>>(e, i, n_steps, obs);
      else if (fl) emu_rollout16<EmuCtx16T<true, true, false>>(e, i, n_steps, obs);
      else if (pl && kn) emu_rollout16<EmuCtx16T<false, true, true>>(e, i, n_steps, obs);
      else if (pl) emu_rollout16<EmuCtx16T<false, false, true>>(e, i, n_steps, obs);
      else emu_rollout16<EmuCtx16T<false, true, false>>(e, i, n_steps, obs);
    } else {
      const bool pl4 = plain_config(e->K);                        // (LAUNCH4)
      if (e->K.knee == 3 && e->K.terrain == 0) emu_rollout4<EmuCtxT<true, false, 3>>(e, i, n_steps, obs);
      else if (e->K.knee == 3) emu_rollout4<EmuCtxT<false, false, 3>>(e, i, n_steps, obs);
      else if (e->K.knee && e->K.terrain == 0) emu_rollout4<EmuCtxT<true, false, 1>>(e, i, n_steps, obs);
      else if (e->K.knee) emu_rollout4<EmuCtxT<false, false, 1>>(e, i, n_steps, obs);
      else if (e->K.terrain == 0 && pl4) emu_rollout4<EmuCtxT<true, true>>(e, i, n_steps, obs);
      else if (e->K.terrain == 0) emu_rollout4<EmuCtxT<true>>(e, i, n_steps, obs);
      else if (pl4) emu_rollout4<EmuCtxT<false, true>>(e, i, n_steps, obs);
      else emu_rollout4<EmuCtxT<false>>(e, i, n_steps, obs);
    }
    if (ret) ret[i] = e->ctl[(size_t)CT_RET * e->N + i];
    if (len) len[i] = (int)e->ctl[(size_t)CT_LEN * e->N + i];
  }
  e->K.stop_at_done = 1;
}
#ifdef ETG_TRACE_TICKS
extern "C" void emu_debug_set_trace(void* h, float* buf) { ((Emu*)h)->K.trace = buf; }
#endif
extern "C" void emu_set_contact_impulses(void* h, const float* lam) {   // [N,16] per leg (n, t1, t2, body normal): the warm start (after emu_set_state)
  Emu* e = (Emu*)h;
  const size_t N = e->N, NL = 4 * N;
  for (size_t i = 0; i < N; i++)
    for (int l = 0; l < 4; l++)
      for (int k = 0; k < 4; k++) e->leg[(size_t)(LG_LAM + k) * NL + 4 * i + l] = lam[i * 16 + 4 * l + k];
}
extern "C" void emu_get_contact_impulses(void* h, float* lam) {
  Emu* e = (Emu*)h;
  const size_t N = e->N, NL = 4 * N;
  for (size_t i = 0; i < N; i++)
    for (int l = 0; l < 4; l++)
      for (int k = 0; k < 4; k++) lam[i * 16 + 4 * l + k] = e->leg[(size_t)(LG_LAM + k) * NL + 4 * i + l];
}
extern "C" void emu_get_state(void* h, float* st) {
  Emu* e = (Emu*)h;
  for (int i = 0; i < e->N; i++) {
    EmuCtx c(i, e->N, e->par.data());
    LaneState<F4> L = load_state<F4>(c, e->base.data(), e->leg.data());
    get_state_quad(c, L, st);
  }
}
extern "C" void emu_set_state(void* h, const float* st) {
  Emu* e = (Emu*)h;
  for (int i = 0; i < e->N; i++) {
    EmuCtx c(i, e->N, e->par.data());
    LaneState<F4> L;
    set_state_quad(c, st, L, e->ring.data(), e->ctl.data(), e->ictl.data());
    store_state(c, e->base.data(), e->leg.data(), L);
  }
}
// checks that the replicated base state is bit-identical across a quad's lanes after a tick
extern "C" int emu_tick_replication_check(void* h, int env, int nticks) {
  Emu* e = (Emu*)h;
  EmuCtxT<true> c(env, e->N, e->par.data());
  LaneState<F4> L = load_state<F4>(c, e->base.data(), e->leg.data());
  F4 qdes[3] = {c.par(PR_POSE), c.par(PR_POSE + 1), c.par(PR_POSE + 2)};
  int bad = 0;
  for (int t = 0; t < nticks; t++) {
    physics_tick(c, e->K, load_tick_par4<F4>(c), L, qdes, V3<F4>{F4(0.0f), F4(0.0f), F4(0.0f)});
    const F4* f[] = {&L.p.x, &L.p.y, &L.p.z, &L.qx, &L.qy, &L.qz, &L.qw, &L.wb.x, &L.wb.y, &L.wb.z, &L.vb.x, &L.vb.y, &L.vb.z};
    for (auto* x : f)
      for (int l = 1; l < 4; l++) bad += std::memcmp(&x->v[0], &x->v[l], 4) != 0;
  }
  return bad;
}

// 16-lane variant: replicated base state must be bit-identical across the 16 lanes of the row
extern "C" int emu16_tick_replication_check(void* h, int env, int nticks) {
  Emu* e = (Emu*)h;
  EmuCtx16T<true> c(env, e->N, e->par.data());
  State16<F16> L = load_state16<F16>(c, e->base.data(), e->leg.data());
  F16 qdes = c.jointf() * c.par_joint(PR_POSE);
  int bad = 0;
  const TickPar<F16> tp = load_tick_par<F16>(c);
  for (int t = 0; t < nticks; t++) {
    physics_tick16(c, e->K, tp, L, qdes);
    const F16* f[] = {&L.p.x, &L.p.y, &L.p.z, &L.qx, &L.qy, &L.qz, &L.qw, &L.wb.x, &L.wb.y, &L.wb.z, &L.vb.x, &L.vb.y, &L.vb.z};
    for (auto* x : f)
      for (int l = 1; l < 16; l++) bad += std::memcmp(&x->v[0], &x->v[l], 4) != 0;
  }
  return bad;
}
