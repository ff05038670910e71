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

#include "../../paddlerobotics_amd/csrc/etg_layout.h"

namespace etg {
struct B4 { bool v[4]; };
struct F4 {
  float v[4];
  F4() {}
  explicit F4(float s) { v[0] = v[1] = v[2] = v[3] = s; }
};
#define OP2(op) inline F4 operator op(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
OP2(+) OP2(-) OP2(*) OP2(/)
#undef OP2
inline F4 operator-(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = -a.v[i]; return r; }
#define CMP(op) inline B4 operator op(F4 a, F4 b) { B4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i]; return r; }
CMP(<) CMP(>) CMP(<=) CMP(>=)
#undef CMP
inline B4 operator&&(B4 a, B4 b) { B4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
inline B4 operator||(B4 a, B4 b) { B4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
inline B4 operator!(B4 a) { B4 r; for (int i = 0; i < 4; i++) r.v[i] = !a.v[i]; return r; }
inline F4 sel_(B4 c, F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = c.v[i] ? a.v[i] : b.v[i]; return r; }
#define FN1(name) inline F4 name(F4 a) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = name(a.v[i]); return r; }
FN1(fabsf_) FN1(sqrt_) FN1(rsqrt_) FN1(rcp_) FN1(sin_) FN1(cos_) FN1(exp_) FN1(tanh_) FN1(acos_) FN1(asin_) FN1(wrap_pi_)
#undef FN1
inline F4 fminf_(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = fminf(a.v[i], b.v[i]); return r; }
inline F4 fmaxf_(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = fmaxf(a.v[i], b.v[i]); return r; }
inline F4 atan2_(F4 a, F4 b) { F4 r; for (int i = 0; i < 4; i++) r.v[i] = atan2f(a.v[i], b.v[i]); return r; }
inline B4 isfinite_(F4 a) { B4 r; for (int i = 0; i < 4; i++) r.v[i] = std::isfinite(a.v[i]); return r; }
inline void sincos_(F4 a, F4& s, F4& c) { for (int i = 0; i < 4; i++) { s.v[i] = sinf(a.v[i]); c.v[i] = cosf(a.v[i]); } }
}  // namespace etg

#include "../../paddlerobotics_amd/csrc/etg_core.h"

namespace etg {
struct EmuCtxBase {
  int env, N;
  const float* parp;
  F4 par(int k) const { return ld_lane(parp, k); }
  int NL() const { return 4 * N; }
  F4 ld_lane(const float* p, int f) const { F4 r; for (int l = 0; l < 4; l++) r.v[l] = p[(size_t)f * NL() + 4 * env + l]; return r; }
  void st_lane(float* p, int f, F4 v) const { for (int l = 0; l < 4; l++) p[(size_t)f * NL() + 4 * env + l] = v.v[l]; }
  F4 ld_env(const float* p, int f) const { return F4(p[(size_t)f * N + env]); }
  void st_env(float* p, int f, F4 v) const { p[(size_t)f * N + env] = v.v[0]; }
  int ld_env_i(const int* p, int f) const { return p[(size_t)f * N + env]; }
  void st_env_i(int* p, int f, int v) const { p[(size_t)f * N + env] = v; }
  void st_ring(float* r, int slot, int k, F4 v) const { for (int l = 0; l < 4; l++) r[((size_t)slot * 8 + k) * NL() + 4 * env + l] = v.v[l]; }
  F4 ld_ring(const float* r, int slot, int k) const { F4 o; for (int l = 0; l < 4; l++) o.v[l] = r[((size_t)slot * 8 + k) * NL() + 4 * env + l]; return o; }
  void ring_fence() const {}
  void phase(int) const {}
  void st_row_env(float* p, int rowlen, int col, F4 v) const { p[(size_t)env * rowlen + col] = v.v[0]; }
  void st_row_lane(float* p, int rowlen, int col0, int stride, F4 v) const { for (int l = 0; l < 4; l++) p[(size_t)env * rowlen + col0 + stride * l] = v.v[l]; }
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
  int uniform_int(F4 a) const { return (int)a.v[0]; }
  void terrain(const KCfg& K, F4 x, F4 y, F4& h, F4& nx, F4& ny, F4& nz) const {
    for (int l = 0; l < 4; l++) {
      if (K.terrain == 0 || K.hf == nullptr) { h.v[l] = 0; nx.v[l] = 0; ny.v[l] = 0; nz.v[l] = 1; }
      else heightfield_query(K, x.v[l], y.v[l], h.v[l], nx.v[l], ny.v[l], nz.v[l]);
    }
  }
};

template <bool FLAT> struct EmuCtxT : EmuCtxBase {
  static constexpr bool kFlat = FLAT;
  EmuCtxT(int e, int n, const float* p) { env = e; N = n; parp = p; }
};
typedef EmuCtxT<false> EmuCtx;   // generic-terrain instantiation; the flat fast path is EmuCtxT<true>

struct Emu {
  KCfg K;
  ModelF M;
  int N;
  std::vector<float> base, leg, ctl, legctl, etgp, par, ring, hf;
  std::vector<int> ictl;
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
  return e;
}
extern "C" void emu_destroy(void* h) { delete (Emu*)h; }
extern "C" void emu_set_params(void* h, const float* dyn, const float* w, const float* b, int per_env, const uint8_t* mask) {
  Emu* e = (Emu*)h;
  int N = e->N;
  for (int i = 0; i < N; i++) {
    if (mask && !mask[i]) continue;
    if (dyn)
      for (int l = 0; l < 4; l++) {
        float out[PR_N];
        derive_lane_params(e->M, dyn + (size_t)i * ETG_DYN_DIM, l, e->K.dt, out);
        for (int k = 0; k < PR_N; k++) e->par[(size_t)k * 4 * N + 4 * i + l] = out[k];
      }
    if (w)
      for (int k = 0; k < 60; k++) e->etgp[(size_t)(EP_W + k) * N + i] = w[(per_env ? (size_t)i * 60 : 0) + k];
    if (b)
      for (int k = 0; k < 3; k++) e->etgp[(size_t)(EP_B + k) * N + i] = b[(per_env ? (size_t)i * 3 : 0) + k];
  }
}
extern "C" void emu_set_heightfield(void* h, const float* hts) {
  Emu* e = (Emu*)h;
  e->hf.assign(hts, hts + (size_t)e->K.hf_nx * e->K.hf_ny);
  e->K.hf = e->hf.data();
}
extern "C" void emu_reset(void* h, const uint8_t* mask, float* obs) {
  Emu* e = (Emu*)h;
  for (int i = 0; i < e->N; i++) {
    if (mask && !mask[i]) continue;
    LaneState<F4> L;
    if (e->K.terrain == 0) {
      EmuCtxT<true> c(i, e->N, e->par.data());
      reset_quad(c, e->K, L, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), obs);
      store_state(c, e->base.data(), e->leg.data(), L);
    } else {
      EmuCtxT<false> c(i, e->N, e->par.data());
      reset_quad(c, e->K, L, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), obs);
      store_state(c, e->base.data(), e->leg.data(), L);
    }
  }
}
extern "C" void emu_step(void* h, const float* action, const uint8_t* donef, float* obs, float* reward, uint8_t* done, float* info) {
  Emu* e = (Emu*)h;
  for (int i = 0; i < e->N; i++) {
    EmuCtx c0(i, e->N, e->par.data());
    LaneState<F4> L = load_state<F4>(c0, e->base.data(), e->leg.data());
    F4 act[3];
    for (int j = 0; j < 3; j++) act[j] = c0.ld_row_lane(action, 12, j, 3);
    F4 r, d;
    if (e->K.terrain == 0) {
      EmuCtxT<true> c(i, e->N, e->par.data());
      control_step(c, e->K, L, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), act,
                   F4(donef ? (float)donef[i] : 0.f), obs, r, d, info);
    } else {
      control_step(c0, e->K, L, e->ring.data(), e->ctl.data(), e->ictl.data(), e->legctl.data(), e->etgp.data(), act,
                   F4(donef ? (float)donef[i] : 0.f), obs, r, d, info);
    }
    store_state(c0, e->base.data(), e->leg.data(), L);
    reward[i] = r.v[0];
    done[i] = d.v[0] > 0.5f;
  }
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
    physics_tick(c, e->K, L, qdes);
    const F4* f[] = {&L.p.x, &L.p.y, &L.p.z, &L.qx, &L.qy, &L.qz, &L.qw, &L.wb.x, &L.wb.y, &L.wb.z, &L.vb.x, &L.vb.y, &L.vb.z};
    for (auto* x : f)
      for (int l = 1; l < 4; l++) bad += std::memcmp(&x->v[0], &x->v[l], 4) != 0;
  }
  return bad;
}
