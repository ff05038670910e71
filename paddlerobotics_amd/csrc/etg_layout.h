// etg_layout.h -- HBM data layout of the simulator state, the kernel-uniform config
// block, and the float lane-math primitives etg_core.h is written against.
//
// Layout (DESIGN.md "data layout in HBM"): everything is structure-of-arrays so that a
// wave's 64 lanes touch 64 consecutive floats:
//   base  [BS_N ][N]      floating-base state, one column per robot (read by its 4 lanes)
//   leg   [LG_N ][4N]     per-leg state, one column per lane  (lane id = 4*env + leg)
//   ctl   [CT_N ][N]      per-robot control-loop floats ; ictl [IC_N][N] ints
//   legctl[LC_N ][4N]     per-leg control-loop floats
//   etgp  [EP_N ][N]      ETG weights w[3][20] + b[3] per robot
//   par   [PR_N ][4N]     derived physical parameters per lane
//   ring  [RING][8][4N]   latency ring (minitaur.py:1142-1193), 8 floats per lane per tick
#pragma once

#include <math.h>
#include <stdint.h>

#include "../../include/etgsim.h"

#if defined(__HIPCC__)
#define ETG_HD __device__ __forceinline__
#else
#define ETG_HD inline
#endif

namespace etg {

constexpr int RING = 64;  // ticks of history; latency <= 80 ms at dt = 2 ms needs 42

enum { BS_PX, BS_PY, BS_PZ, BS_QX, BS_QY, BS_QZ, BS_QW, BS_WX, BS_WY, BS_WZ, BS_VX, BS_VY, BS_VZ, BS_N };
enum { LG_Q = 0, LG_QD = 3, LG_LAM = 6, LG_LAMB = 9, LG_CONTACT = 10, LG_N = 11 };   // LG_LAMB = LG_LAM + 3: the body contact's normal impulse (the aux lane's fourth "joint" slot)
// CT_RET/LEN/ALIVE: per-robot episode accumulators (return, length, alive mask) updated by every step
// CT_FEXT: external force on the trunk COM (world frame, N), etg_set_external_force(); CT_PUSH: the random push of
// etg_random_pushes() -- separate columns, the kernels apply their sum (a set force survives pushes and their clearing)
enum { CT_FIRST_RPY = 0, CT_LAST_BASE = 3, CT_RET = 6, CT_LEN = 7, CT_ALIVE = 8, CT_FEXT = 9, CT_PUSH = 12, CT_N = 15 };
// IC_PUSH_LEFT: control steps the current random push still lasts (etg_random_pushes)
// IC_OBS_CALL: sensor-noise stream position of the robot's observation row that still waits for its noise from the epilogue of a
// fused rollout launch under stop_at_done (rows are then written at different steps: a robot's last one when its episode ends)
enum { IC_STEP = 0, IC_TICK = 1, IC_HAS_LAST = 2, IC_PUSH_LEFT = 3, IC_OBS_CALL = 4, IC_N = 5 };
enum { LC_LAST_QDES = 0, LC_FX0 = 3, LC_FX1 = 6, LC_FY0 = 9, LC_FY1 = 12, LC_LAST_FOOT_X = 15, LC_N = 16 };
enum { EP_W = 0, EP_B = 60, EP_N = 63 };
// per-lane derived parameters, in the order derive_lane_params() writes them
enum { PR_LINK = 0 /* 3 x (m, com3, Ic6) */, PR_O1 = 30, PR_SY = 33, PR_KP = 34, PR_KD = 37, PR_MU = 40, PR_M0 = 41,
       PR_I0 = 42, PR_G = 48, PR_LAT_N = 51, PR_LAT_A = 52, PR_BASE_FOOT = 53, PR_POSE = 56, PR_EMEAN = 59,
       PR_ESTD = 62, PR_HIPSIGN = 65, PR_DERIVED = 66 /* derive_lane_params() writes [0, PR_DERIVED) */,
       PR_STR = 66 /* 3: motor strength ratios of the leg's joints (etg_set_motor_strength), 1 unless set */, PR_N = 69 };

struct KCfg {
  int n_env;
  int action_repeat, settle_ticks, iters, enable_interp, enable_filter, obs_normal, terrain;
  float dt, erp, margin, warmstart, torque_limit;
  float upper_len, lower_len, foot_radius;
  float init_pos[3];
  float etg_T, etg_T2, etg_amp, etg_sigma_sq, etg_phase0, etg_phase1, etg_omega, etg_dt;
  float etg_u[ETG_RBF_H][2];
  float rw[8], reward_p, vel_d;
  float fb[3], fa[3];
  int hf_nx, hf_ny;      // hf_ny = rows of ONE band
  int hf_bands;          // bands stacked along y in the heights array; robot e uses band e % hf_bands
  float hf_cell, hf_x0, hf_y0, hf_inv_cell;
  const float* hf;
  // the settle cache's copy of the latency ring (DevState.cache_ring), or null (test emulation).  Ring slots of ticks up to
  // the reset tick (settle_ticks) are READ FROM THE CACHE: a robot's history before its reset is its settle, which the cache
  // holds, so a reset -- etg_reset from the cache, the auto-reset after a step -- never copies 8 KB of ring per robot.
  const float* cring;
  int ext_force;         // 1 while a set force (ctl[CT_FEXT..]) or random pushes (ctl[CT_PUSH..]) are installed
  int etg_on;            // EtgConfig.enable_etg: 0 = no trajectory generator (command = pose_ori + action)
  int jlim;              // EtgConfig.joint_limits
  float jlo[3], jhi[3];  // joint-limit stops (hip, thigh, calf)
  int motor_mode;        // 0 POSITION (PD on a joint-angle command), 1 TORQUE (the command is the torque)
  float clip_cmd;        // > 0: clip the position command to q +- clip_cmd every tick (a1.py:439-457)
  int knee;              // EtgConfig.body_contacts: 1 knee spheres collide; 2 deepest of knee / shin / trunk corner (16-lane kernels)
  float knee_radius;
  float body_mu;         // EtgConfig.body_friction: friction coefficient of the body contacts (body_contacts 1 / 2)
  float blend_inv;       // 1 / EtgConfig.body_blend (0 = the deepest sphere is the contact): softness of the choice among the leg's three spheres
  float trunk_half[3];   // knee == 2: half extents of the trunk box (its corners collide too)
  // Gaussian sensor noise (minitaur.py:1206-1211): stdev of motor angle, motor velocity, motor torque (not part of
  // the 49-float observation), base rpy, base rpy rate -- the order of SENSOR_NOISE_STDDEV (minitaur.py:102)
  int noise_on;
  float noise_std[5];
  unsigned long long noise_seed;
  unsigned noise_call;   // stream position of the first observation this launch writes
  float res_thr;         // EtgConfig.solver_residual: > 0 = sweep until the robot's squared row residual is below it (iters = cap)
  float res_sqrt;        // sqrt(res_thr): the kernels compare |d lambda| with res_sqrt / A_rr
  int fric_pyramid;      // EtgConfig.friction_model == 1: per-direction clamp instead of the disc projection
  int pd_n;              // EtgConfig.pd_latency: n_steps_ago of the PD law's reading (minitaur.py:1185), -1 = off (true state)
  float pd_a;            // its blend_alpha (minitaur.py:1188)
  float warmstart_b;     // warm-start factor of a leg's body contact's normal row: warmstart when the contact is one persistent point
                         // (body_contacts 1, or 2 with body_blend > 0), else 0 (oracle: "Warm start" in tick())
  float warmstart_t;     // EtgConfig.warmstart_friction: warm-start factor of the friction rows (warmstart: the normal rows)
  float slop;            // EtgConfig.contact_slop: added to a contact's distance before the velocity target is formed
  float restitution;     // EtgConfig.foot_restitution (combined coefficient; 0 = off)
  int strength_on;       // motor strength ratios other than 1 are installed (etg_set_motor_strength)
  int block0;            // first workgroup of a sub-batch launch (etg_step_range): the launch's own block index is added to it; 0 otherwise
  int stop_at_done;      // fused rollouts: a robot whose episode has ended is not simulated any more (etg_set_rollout_mode; default 1)
#ifdef ETG_TRACE_TICKS   // debugging build only: [N][16 ticks][16 lanes][10] floats written by physics_tick16 (etg_debug_set_trace)
  float* trace;
#endif
};

// the default robot layer (what train.py / pretrain.py run): the PLAIN kernel instantiations compile the options out
inline bool plain_config(const KCfg& K) {
  return K.motor_mode == 0 && !K.enable_filter && !K.enable_interp && !(K.torque_limit > 0.0f) && !(K.clip_cmd > 0.0f) &&
         !K.ext_force && K.knee != 3 && K.etg_on && !K.fric_pyramid && K.pd_n < 0 && !(K.restitution > 0.0f) && !K.strength_on;   // (joint limits: in every instantiation)
}

// counter-based standard normal pair for (seed, robot, observation index, channel): splitmix64 finaliser twice, then
// Box-Muller.  The same few lines are restated in the oracle (oracle/etgsim_oracle.cpp: gauss_pair).
ETG_HD unsigned long long mix64_(unsigned long long z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
ETG_HD void gauss_pair(unsigned long long seed, unsigned env, unsigned call, unsigned ch, float& a, float& b) {
  unsigned long long z = mix64_(seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)env * 32ull + ch + 1ull));
  z = mix64_(z + call);
  const float u1 = (float)((z >> 40) + 1ull) * (1.0f / 16777216.0f);           // (0, 1]
  const float u2 = (float)((z >> 8) & 0xFFFFFFull) * (1.0f / 16777216.0f);    // [0, 1)
  const float r = sqrtf(-2.0f * logf(u1)), th = 6.283185307179586f * u2;
  a = r * cosf(th);
  b = r * sinf(th);
}
// _AddSensorNoise (minitaur.py:1206-1211) on one written observation row, one call per channel slot 0..15:
// slots 0..11 = joint j (angle column 13+j, velocity column 25+j), slots 12..14 = rpy k / rpy rate k (columns 7+k,
// 10+k; RNG channels 16..18), slot 15 unused.  Scales follow the row's normalisation (write_obs).
ETG_HD void add_sensor_noise(const KCfg& K, unsigned env, unsigned call, unsigned slot, float* row) {
  if (slot >= 15) return;
  const bool nrm = K.obs_normal != 0;
  float n0, n1;
  if (slot < 12) {
    gauss_pair(K.noise_seed, env, call, slot, n0, n1);
    row[13 + slot] += K.noise_std[0] * n0 * (nrm ? 10.0f : 1.0f);
    row[25 + slot] += K.noise_std[1] * n1;
  } else {
    const unsigned k = slot - 12;
    gauss_pair(K.noise_seed, env, call, 16 + k, n0, n1);
    row[7 + k] += K.noise_std[3] * n0 * (nrm ? 10.0f : 1.0f);
    row[10 + k] += K.noise_std[4] * n1 * (nrm ? 2.0f : 1.0f);
  }
}

struct DevState {
  float *base, *leg, *ctl, *legctl, *etgp, *par, *ring;
  float* dyn;   // [N,48] the dynamic_param rows as etg_set_params received them (dynamic_vec sensor)
  int* ictl;
  // settle cache: the state (base, leg) and latency ring right after the 500-tick reset settle of each robot.
  // The settle only depends on the robot's dynamic parameters and terrain, so a later reset of the same robot
  // copies it back instead of re-simulating (cache_ok[env]; cleared when its dyn row or the heightfield changes).
  float *cache_base, *cache_leg, *cache_ring;
  unsigned char* cache_ok;
  // start offsets [2][N] (x, y; etg_set_reset_offsets) of the next reset, and the ones each cached settle ran at
  float *reset_off, *cache_off;
};

// ---- float lane math ------------------------------------------------------------
ETG_HD float sel_(bool c, float a, float b) { return c ? a : b; }
ETG_HD float fminf_(float a, float b) { return fminf(a, b); }
ETG_HD float fmaxf_(float a, float b) { return fmaxf(a, b); }
ETG_HD float fabsf_(float a) { return fabsf(a); }
ETG_HD float acos_(float a) { return acosf(a); }
ETG_HD float asin_(float a) { return asinf(a); }
ETG_HD float atan2_(float a, float b) { return atan2f(a, b); }
// bit test, so that -ffinite-math-only (used to fold the x*0 / x*1 terms of the structured frames)
// cannot optimise the NaN guards away
ETG_HD bool isfinite_(float a) {
  unsigned u;
  __builtin_memcpy(&u, &a, 4);
  return (u & 0x7f800000u) != 0x7f800000u;
}
#if defined(__HIPCC__)
// Hardware 1-ulp reciprocal / rsqrt / sqrt: the IEEE-exact expansions cost 10-15 dependent
// instructions each and sit on the serial critical path of the LDL^T and of every PGS turn.
#if defined(ETG_IEEE_ALL)   // A/B build variant (tools/hf_tracking_probe.py): IEEE division / square root everywhere
ETG_HD float rcp_(float a) { return 1.0f / a; }
ETG_HD float rsqrt_(float a) { return 1.0f / sqrtf(a); }
ETG_HD float sqrt_(float a) { return sqrtf(a); }
#else
ETG_HD float rcp_(float a) { return __builtin_amdgcn_rcpf(a); }
ETG_HD float rsqrt_(float a) { return __builtin_amdgcn_rsqf(a); }
ETG_HD float sqrt_(float a) { return __builtin_amdgcn_sqrtf(a); }
#endif
// sin/cos for joint angles: Cody-Waite reduction by pi/2 + cephes minimax polynomials
// (|error| < 2e-7 for |x| < 1e3; joint angles are within +-4.2 rad).  libm's sincosf carries a
// Payne-Hanek slow path whose branches alone cost more than this whole routine.
ETG_HD void sincos_(float x, float& s, float& c) {
  const float k = rintf(x * 0.636619772f);
  float r = fmaf(k, -1.57079637f, x);
  r = fmaf(k, 4.37113900e-08f, r);
  const float z = r * r;
  const float ps = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  const float pc = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z,
                        fmaf(-0.5f, z, 1.0f));
  const int q = (int)k;
  const float sv = (q & 1) ? pc : ps, cv = (q & 1) ? ps : pc;
  s = (q & 2) ? -sv : sv;
  c = ((q + 1) & 2) ? -cv : cv;
}
// sin/cos of a joint angle inside the physics tick (13 times per control step and lane): the hardware's v_sin_f32 /
// v_cos_f32 take the argument in revolutions; over the joint range (|x| < 4.2 rad) their error is 3.9e-7 / 3.5e-7 against
// double precision and |s^2 + c^2 - 1| < 2.5e-7 (tools/ubench/hw_sincos.hip) -- the polynomial routine's accuracy for
// 3 instructions instead of 27 (-1.5 % per control step).  NOT for the ETG phase: 160 revolutions leave 1e-4 rad.
// Domain of the instructions: |x| < 256 revolutions = 1608 rad (beyond: sin -> 0, cos -> 1, still finite) -- only a
// limit-less joint spun up by a constant torque for seconds gets there, on a robot that has long terminated.
ETG_HD void sincos_tick_(float x, float& s, float& c) {
#ifdef ETG_POLY_SINCOS_IN_TICK
  sincos_(x, s, c);
#else
  const float rev = x * 0.15915494309189535f;
  s = __builtin_amdgcn_sinf(rev);
  c = __builtin_amdgcn_cosf(rev);
#endif
}
// ETG phase sine (|argument| < 1e3: phase + 2 pi t / T over an episode) and the RBF / reward exponentials: the
// bounded-range routine above and the hardware exp2 (1 ulp) instead of libm's full-range versions
ETG_HD float sin_(float a) { float s, c; sincos_(a, s, c); return s; }
ETG_HD float cos_(float a) { float s, c; sincos_(a, s, c); return c; }   // IK angles, within +-pi
ETG_HD float exp_(float a) { return __builtin_amdgcn_exp2f(a * 1.44269504088896341f); }
// tanh(a) = 1 - 2 / (e^{2a} + 1); the reward terms call it with a >= 0 (c_prec: a = x^2); |error| < 2e-7
ETG_HD float tanh_(float a) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(fminf(a, 40.0f) * 2.88539008177792681f) + 1.0f); }
#else
ETG_HD float sin_(float a) { return sinf(a); }
ETG_HD float cos_(float a) { return cosf(a); }
ETG_HD float exp_(float a) { return expf(a); }
ETG_HD float tanh_(float a) { return tanhf(a); }
ETG_HD float rcp_(float a) { return 1.0f / a; }
ETG_HD float rsqrt_(float a) { return 1.0f / sqrtf(a); }
ETG_HD float sqrt_(float a) { return sqrtf(a); }
ETG_HD void sincos_(float a, float& s, float& c) { s = sinf(a); c = cosf(a); }
ETG_HD void sincos_tick_(float a, float& s, float& c) { s = sinf(a); c = cosf(a); }
#endif
// 1 / sqrt in the terrain normal and the contact frame: the hardware's v_rsq_f32, or (A/B build variant ETG_IEEE_HF) IEEE ops
ETG_HD float rsqrt_hf_(float a) {
#if defined(ETG_IEEE_HF) || defined(ETG_IEEE_ALL)
  return 1.0f / sqrtf(a);
#else
  return rsqrt_(a);
#endif
}
// MapToMinusPiToPi (minitaur.py:67-83)
ETG_HD float wrap_pi_(float a) {
  const float two_pi = 6.283185307179586f, pi = 3.141592653589793f;
  a = fmodf(a, two_pi);
  if (a >= pi) a -= two_pi;
  else if (a < -pi) a += two_pi;
  return a;
}

// ---- derived per-lane parameters (shared by the set_params kernel and the test emulator)
// dyn: one 48-float dynamic_param row (train.py:112-126 layout), leg in 0..3.
// I' = S I S with S = diag(sqrt(ratio)) for the per-axis inertia ratios.
struct ModelF {  // float copy of EtgRobotModel
  float trunk_m, trunk_I[6];
  float link_m[4][4], link_com[4][4][3], link_I[4][4][6];  // [leg][hip,thigh,calf,foot]
  float hip_origin[4][3], thigh_y[4];
  float lower_len;
  float pose[12], base_foot[12], emean[12], estd[12];
};

ETG_HD void scale_inertia(const float* I, const float* r, float* o) {
  float s0 = sqrtf(r[0]), s1 = sqrtf(r[1]), s2 = sqrtf(r[2]);
  o[0] = I[0] * s0 * s0; o[1] = I[1] * s1 * s1; o[2] = I[2] * s2 * s2;
  o[3] = I[3] * s0 * s1; o[4] = I[4] * s0 * s2; o[5] = I[5] * s1 * s2;
}

ETG_HD void derive_lane_params(const ModelF& M, const float* dyn, int leg, float sim_dt, float* out /*PR_DERIVED*/) {
  int k = 0;
  // hip, thigh
  for (int i = 0; i < 2; i++) {
    float I[6];
    scale_inertia(M.link_I[leg][i], dyn + 9 + 3 * i, I);
    out[k++] = M.link_m[leg][i] * dyn[6 + i];
    for (int a = 0; a < 3; a++) out[k++] = M.link_com[leg][i][a];
    for (int a = 0; a < 6; a++) out[k++] = I[a];
  }
  {  // calf + rigidly attached foot at (0,0,-lower_len)  (fixed toe joint, a1.py:99)
    float Ic[6], If[6];
    scale_inertia(M.link_I[leg][2], dyn + 15, Ic);
    scale_inertia(M.link_I[leg][3], dyn + 18, If);
    float mc = M.link_m[leg][2] * dyn[8], mf = M.link_m[leg][3];
    const float* cc = M.link_com[leg][2];
    float cf[3] = {M.link_com[leg][3][0], M.link_com[leg][3][1], M.link_com[leg][3][2] - M.lower_len};
    float m = mc + mf;
    float c[3] = {(mc * cc[0] + mf * cf[0]) / m, (mc * cc[1] + mf * cf[1]) / m, (mc * cc[2] + mf * cf[2]) / m};
    float I[6] = {0, 0, 0, 0, 0, 0};
    const float* srcI[2] = {Ic, If};
    const float* srcc[2] = {cc, cf};
    float srcm[2] = {mc, mf};
    for (int b = 0; b < 2; b++) {
      float d[3] = {srcc[b][0] - c[0], srcc[b][1] - c[1], srcc[b][2] - c[2]};
      I[0] += srcI[b][0] + srcm[b] * (d[1] * d[1] + d[2] * d[2]);
      I[1] += srcI[b][1] + srcm[b] * (d[0] * d[0] + d[2] * d[2]);
      I[2] += srcI[b][2] + srcm[b] * (d[0] * d[0] + d[1] * d[1]);
      I[3] += srcI[b][3] - srcm[b] * d[0] * d[1];
      I[4] += srcI[b][4] - srcm[b] * d[0] * d[2];
      I[5] += srcI[b][5] - srcm[b] * d[1] * d[2];
    }
    out[k++] = m;
    for (int a = 0; a < 3; a++) out[k++] = c[a];
    for (int a = 0; a < 6; a++) out[k++] = I[a];
  }
  for (int a = 0; a < 3; a++) out[k++] = M.hip_origin[leg][a];
  out[k++] = M.thigh_y[leg];
  for (int j = 0; j < 3; j++) out[k++] = dyn[21 + 3 * leg + j];  // kp
  for (int j = 0; j < 3; j++) out[k++] = dyn[33 + 3 * leg + j];  // kd
  out[k++] = dyn[1];                                            // foot friction
  out[k++] = M.trunk_m * dyn[2];
  {
    float I[6];
    scale_inertia(M.trunk_I, dyn + 3, I);
    for (int a = 0; a < 6; a++) out[k++] = I[a];
  }
  for (int a = 0; a < 3; a++) out[k++] = dyn[45 + a];  // gravity
  {  // control latency (ms) -> n_steps_ago, blend_alpha (minitaur.py:1185-1188)
    float lat = dyn[0] * 0.001f;
    if (lat <= 0.0f) {
      out[k++] = -1.0f; out[k++] = 0.0f;
    } else {
      int n = (int)(lat / sim_dt);
      if (n > RING - 2) n = RING - 2;
      out[k++] = (float)n;
      out[k++] = (lat - (float)n * sim_dt) / sim_dt;
    }
  }
  for (int a = 0; a < 3; a++) out[k++] = M.base_foot[3 * leg + a];
  for (int a = 0; a < 3; a++) out[k++] = M.pose[3 * leg + a];
  for (int a = 0; a < 3; a++) out[k++] = M.emean[3 * leg + a];
  for (int a = 0; a < 3; a++) out[k++] = M.estd[3 * leg + a];
  out[k++] = (leg & 1) ? 1.0f : -1.0f;  // (-1)**(leg+1), a1.py:485
}

// host helpers: EtgConfig/EtgRobotModel (double) -> kernel blocks (float)
inline KCfg make_kcfg(const EtgConfig& c, const EtgRobotModel& m) {
  KCfg K;
  K.n_env = c.num_envs;
  K.action_repeat = c.action_repeat; K.settle_ticks = c.settle_ticks; K.iters = c.solver_iters;
  K.enable_interp = c.enable_action_interp; K.enable_filter = c.enable_action_filter;
  K.obs_normal = c.obs_normal; K.terrain = c.terrain;
  K.dt = (float)c.sim_dt; K.erp = (float)c.erp; K.margin = (float)c.contact_margin;
  K.warmstart = (float)c.warmstart; K.torque_limit = (float)c.torque_limit;
  K.upper_len = (float)m.upper_len; K.lower_len = (float)m.lower_len; K.foot_radius = (float)m.foot_radius;
  for (int k = 0; k < 3; k++) K.init_pos[k] = (float)m.init_pos[k];
  K.etg_T = (float)c.etg_T; K.etg_T2 = (float)c.etg_T2; K.etg_amp = (float)c.etg_amp;
  K.etg_sigma_sq = (float)c.etg_sigma_sq; K.etg_phase0 = (float)c.etg_phase[0]; K.etg_phase1 = (float)c.etg_phase[1];
  const double omega = 2.0 * M_PI / c.etg_T;
  K.etg_omega = (float)omega; K.etg_dt = (float)c.etg_dt;
  for (int h = 0; h < ETG_RBF_H; h++) {
    double t = h * c.etg_T / (ETG_RBF_H - 0.9);
    K.etg_u[h][0] = (float)(c.etg_amp * sin(c.etg_phase[0] + t * omega));
    K.etg_u[h][1] = (float)(c.etg_amp * sin(c.etg_phase[1] + t * omega));
  }
  for (int k = 0; k < 8; k++) K.rw[k] = (float)c.reward_w[k];
  K.reward_p = (float)c.reward_p; K.vel_d = (float)c.vel_d;
  for (int k = 0; k < 3; k++) { K.fb[k] = (float)c.filter_b[k]; K.fa[k] = (float)c.filter_a[k]; }
  K.hf_bands = c.hf_bands > 1 ? c.hf_bands : 1;
  K.hf_nx = c.hf_nx; K.hf_ny = c.hf_ny / K.hf_bands;
  K.ext_force = 0;
  K.noise_on = 0; K.noise_seed = 0; K.noise_call = 0;
  for (int k = 0; k < 5; k++) K.noise_std[k] = 0.0f;
  K.motor_mode = c.motor_mode;
  K.clip_cmd = (float)c.clip_motor_commands;
  K.knee = c.body_contacts; K.knee_radius = (float)c.knee_radius; K.body_mu = (float)c.body_friction;
  K.blend_inv = c.body_blend > 0 ? (float)(1.0 / c.body_blend) : 0.0f;
  for (int k = 0; k < 3; k++) K.trunk_half[k] = (float)c.trunk_half[k];
  K.etg_on = c.enable_etg != 0;
  K.res_thr = (float)c.solver_residual;
  K.res_sqrt = (float)sqrt(c.solver_residual > 0 ? c.solver_residual : 0.0);
  K.fric_pyramid = c.friction_model == 1;
  K.pd_n = -1; K.pd_a = 0.0f;
  if (c.pd_latency > 0) {   // minitaur.py:1185-1188 with the ring depth as the history length
    int n = (int)(c.pd_latency / c.sim_dt);
    if (n > RING - 2) n = RING - 2;
    K.pd_n = n;
    K.pd_a = (float)((c.pd_latency - n * c.sim_dt) / c.sim_dt);
  }
  K.warmstart_b = (c.body_contacts == 1 || (c.body_contacts == 2 && c.body_blend > 0)) ? (float)c.warmstart : 0.0f;
  K.warmstart_t = (float)c.warmstart_friction; K.slop = (float)c.contact_slop; K.restitution = (float)c.foot_restitution;
  K.strength_on = 0;
  K.stop_at_done = 1;
  K.block0 = 0;
#ifdef ETG_TRACE_TICKS
  K.trace = nullptr;
#endif
  K.jlim = c.joint_limits != 0;
  for (int k = 0; k < 3; k++) { K.jlo[k] = (float)c.joint_lower[k]; K.jhi[k] = (float)c.joint_upper[k]; }
  K.hf_cell = (float)c.hf_cell; K.hf_x0 = (float)c.hf_x0; K.hf_y0 = (float)c.hf_y0;
  K.hf_inv_cell = c.hf_cell > 0 ? (float)(1.0 / c.hf_cell) : 0.0f;
  K.hf = nullptr;
  K.cring = nullptr;
  return K;
}

inline ModelF make_modelf(const EtgRobotModel& m) {
  ModelF M;
  M.trunk_m = (float)m.trunk.mass;
  for (int a = 0; a < 6; a++) M.trunk_I[a] = (float)m.trunk.inertia[a];
  for (int l = 0; l < 4; l++) {
    const EtgLink* links[4] = {&m.hip[l], &m.thigh[l], &m.calf[l], &m.foot[l]};
    for (int i = 0; i < 4; i++) {
      M.link_m[l][i] = (float)links[i]->mass;
      for (int a = 0; a < 3; a++) M.link_com[l][i][a] = (float)links[i]->com[a];
      for (int a = 0; a < 6; a++) M.link_I[l][i][a] = (float)links[i]->inertia[a];
    }
    for (int a = 0; a < 3; a++) M.hip_origin[l][a] = (float)m.hip_origin[l][a];
    M.thigh_y[l] = (float)m.thigh_y[l];
  }
  M.lower_len = (float)m.lower_len;
  for (int j = 0; j < 12; j++) {
    M.pose[j] = (float)m.pose_ori[j]; M.base_foot[j] = (float)m.base_foot[j];
    M.emean[j] = (float)m.etg_mean[j]; M.estd[j] = (float)m.etg_std[j];
  }
  return M;
}

// bilinear heightfield query shared by both builds (clamped at the border), in two halves: the four corner loads
// (`tap` = h00, h10, h01, h11, tx, ty) can be issued as soon as the query point is known -- contact detection works on the
// start-of-tick pose -- and consumed phases later, so a lone wave does not sit out their latency (heightfield_finish)
ETG_HD void heightfield_fetch(const KCfg& K, int env, float x, float y, float* tap) {
  const float* hf = K.hf + (size_t)(env % K.hf_bands) * K.hf_ny * K.hf_nx;
#if defined(ETG_IEEE_HF) || defined(ETG_IEEE_ALL)
  float fx = (x - K.hf_x0) / K.hf_cell, fy = (y - K.hf_y0) / K.hf_cell;
#else
  float fx = (x - K.hf_x0) * K.hf_inv_cell, fy = (y - K.hf_y0) * K.hf_inv_cell;   // (a true division is ~10 instructions)
#endif
  fx = fminf(fmaxf(fx, 0.0f), (float)(K.hf_nx - 1));
  fy = fminf(fmaxf(fy, 0.0f), (float)(K.hf_ny - 1));
  int ix = (int)fx, iy = (int)fy;
  if (ix > K.hf_nx - 2) ix = K.hf_nx - 2;
  if (iy > K.hf_ny - 2) iy = K.hf_ny - 2;
  tap[4] = fx - (float)ix; tap[5] = fy - (float)iy;
  tap[0] = hf[iy * K.hf_nx + ix]; tap[1] = hf[iy * K.hf_nx + ix + 1];
  tap[2] = hf[(iy + 1) * K.hf_nx + ix]; tap[3] = hf[(iy + 1) * K.hf_nx + ix + 1];
}
ETG_HD void heightfield_finish(const KCfg& K, const float* tap, float& h, float& nx, float& ny, float& nz) {
  const float h00 = tap[0], h10 = tap[1], h01 = tap[2], h11 = tap[3], tx = tap[4], ty = tap[5];
  h = (1 - tx) * (1 - ty) * h00 + tx * (1 - ty) * h10 + (1 - tx) * ty * h01 + tx * ty * h11;
#if defined(ETG_IEEE_HF) || defined(ETG_IEEE_ALL)
  float dhdx = ((1 - ty) * (h10 - h00) + ty * (h11 - h01)) / K.hf_cell;
  float dhdy = ((1 - tx) * (h01 - h00) + tx * (h11 - h10)) / K.hf_cell;
#else
  float dhdx = ((1 - ty) * (h10 - h00) + ty * (h11 - h01)) * K.hf_inv_cell;
  float dhdy = ((1 - tx) * (h01 - h00) + tx * (h11 - h10)) * K.hf_inv_cell;
#endif
  float inv = rsqrt_hf_(dhdx * dhdx + dhdy * dhdy + 1.0f);
  nx = -dhdx * inv; ny = -dhdy * inv; nz = inv;
}
ETG_HD void heightfield_query(const KCfg& K, int env, float x, float y, float& h, float& nx, float& ny, float& nz) {
  float tap[6];
  heightfield_fetch(K, env, x, y, tap);
  heightfield_finish(K, tap, h, nx, ny, nz);
}

}  // namespace etg
