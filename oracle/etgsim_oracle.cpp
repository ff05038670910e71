/*
 * etgsim_oracle.cpp -- CPU restatement of the ETGRL env.step()/reset() hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle (and the timed
 * "port" CPU baseline of bench.py).  Nothing under paddlerobotics_amd/ may
 * import, link or call it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * What it restates, and from where (paths relative to
 * /root/reference/QuadrupedalRobots/ETGRL):
 *   - ETG RBF basis / trot fan-out        rlschool ETG_layer (ABSENT from the tree);
 *                                         form pinned by gait_action_list_ETG_exp.npy
 *                                         and train.py:81-110,296-297 (SURVEY 8a a1,a2)
 *   - leg IK / FK / Jacobian              deployment/robots/a1.py:97-173,464-497
 *   - PD motor model                      deployment/robots/laikago_motor.py:103-175
 *   - sub-step loop, action interpolation deployment/robots/minitaur.py:242-260,1384-1401
 *   - latency-delayed observation         deployment/robots/minitaur.py:1142-1204
 *   - reset / settle                      deployment/robots/minitaur.py:403-445, a1.py:289-349
 *   - observation layout                  deployment/envs/EnvWrapper.py:28-121
 *   - Butterworth action filter           deployment/robots/action_filter.py:46-216
 *   - policy forward                      model/mujoco_model.py:44-60, alg/sac.py:60-63
 *   - stepSimulation()                    Bullet (pybullet), ABSENT and unpinned: this
 *                                         file DEFINES the rigid-body model (floating-base
 *                                         articulated dynamics by CRBA + RNEA + dense
 *                                         Cholesky, foot-sphere/ground contact with
 *                                         Coulomb friction by projected Gauss-Seidel,
 *                                         semi-implicit Euler).  PARITY UNPINNED for
 *                                         dynamics/contact/reward/termination: neither
 *                                         pybullet nor rlschool exists in the reference
 *                                         tree or in this image.
 *
 * The implementation is deliberately GENERIC (13-body kinematic tree, 6x6
 * spatial algebra, dense 18x18 factorisation) so that it is an independent
 * check of the hand-specialised HIP kernels (16 and 4 lanes per robot).
 *
 * Build: see oracle/Makefile (g++ -O2 -shared).  Exports a C ABI (etgo_*),
 * each entry point in a double ("64") and a float ("32") instantiation.
 */
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../include/etgsim.h"

namespace {

constexpr int NB = 13;       // trunk + 4 x (hip, thigh, calf[+foot])
constexpr int NV = 18;       // 6 base + 12 joints
constexpr int RING = 64;     // latency ring depth (ticks)
constexpr int HIST = 31;     // q12 qd12 quat4 wlocal3  (minitaur.py:1142-1149 minus torques)

// ---------------------------------------------------------------- small math
template <class T> inline void cross(const T* a, const T* b, T* c) {
  T x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
template <class T> inline T dot3(const T* a, const T* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <class T> inline void mat3_mul_vec(const T R[3][3], const T* v, T* o) {
  T x = R[0][0] * v[0] + R[0][1] * v[1] + R[0][2] * v[2];
  T y = R[1][0] * v[0] + R[1][1] * v[1] + R[1][2] * v[2];
  T z = R[2][0] * v[0] + R[2][1] * v[1] + R[2][2] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
template <class T> inline void mat3T_mul_vec(const T R[3][3], const T* v, T* o) {
  T x = R[0][0] * v[0] + R[1][0] * v[1] + R[2][0] * v[2];
  T y = R[0][1] * v[0] + R[1][1] * v[1] + R[2][1] * v[2];
  T z = R[0][2] * v[0] + R[1][2] * v[1] + R[2][2] * v[2];
  o[0] = x; o[1] = y; o[2] = z;
}
template <class T> inline void mat3_mul(const T A[3][3], const T B[3][3], T C[3][3]) {
  T t[3][3];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) t[i][j] = A[i][0] * B[0][j] + A[i][1] * B[1][j] + A[i][2] * B[2][j];
  std::memcpy(C, t, sizeof(t));
}
// quaternion xyzw (pybullet order) -> rotation matrix (body -> world)
template <class T> inline void quat_to_mat(const T* q, T R[3][3]) {
  T n = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  T s = T(2) / n;
  T x = q[0], y = q[1], z = q[2], w = q[3];
  R[0][0] = 1 - s * (y * y + z * z); R[0][1] = s * (x * y - z * w); R[0][2] = s * (x * z + y * w);
  R[1][0] = s * (x * y + z * w); R[1][1] = 1 - s * (x * x + z * z); R[1][2] = s * (y * z - x * w);
  R[2][0] = s * (x * z - y * w); R[2][1] = s * (y * z + x * w); R[2][2] = 1 - s * (x * x + y * y);
}
template <class T> inline void quat_mul(const T* a, const T* b, T* o) {  // xyzw
  T x = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  T y = a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0];
  T z = a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3];
  T w = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}
// roll-pitch-yaw (ZYX) of a quaternion, the getEulerFromQuaternion convention
// (minitaur.py:620,633)
template <class T> inline void quat_to_rpy(const T* q, T* rpy) {
  T R[3][3];
  quat_to_mat(q, R);
  T sp = -R[2][0];
  if (sp > T(1)) sp = T(1);
  if (sp < T(-1)) sp = T(-1);
  rpy[0] = std::atan2(R[2][1], R[2][2]);
  rpy[1] = std::asin(sp);
  rpy[2] = std::atan2(R[1][0], R[0][0]);
}

// ---------------------------------------------------------------- spatial algebra
// motion vector [w; v], force vector [n; f]; 6x6 inertia as dense matrix.
template <class T> struct Xform {  // Plucker transform parent(A) -> child(B): E = rot A->B coords, r = B origin in A coords
  T E[3][3];
  T r[3];
};
template <class T> inline void x_apply_motion(const Xform<T>& X, const T* v, T* o) {
  T t[3], rw[3];
  cross(X.r, v, rw);  // r x w
  for (int i = 0; i < 3; i++) t[i] = v[3 + i] - rw[i];
  T a[3], b[3];
  mat3_mul_vec(X.E, v, a);
  mat3_mul_vec(X.E, t, b);
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i]; }
}
// X^T f : child coords force -> parent coords
template <class T> inline void x_applyT_force(const Xform<T>& X, const T* f, T* o) {
  T n[3], l[3], rl[3];
  mat3T_mul_vec(X.E, f, n);
  mat3T_mul_vec(X.E, f + 3, l);
  cross(X.r, l, rl);
  for (int i = 0; i < 3; i++) { o[i] = n[i] + rl[i]; o[3 + i] = l[i]; }
}
template <class T> inline void x_to_mat(const Xform<T>& X, T M[6][6]) {
  T rx[3][3] = {{0, -X.r[2], X.r[1]}, {X.r[2], 0, -X.r[0]}, {-X.r[1], X.r[0], 0}};
  T Erx[3][3];
  mat3_mul(X.E, rx, Erx);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      M[i][j] = X.E[i][j]; M[i][3 + j] = 0;
      M[3 + i][j] = -Erx[i][j]; M[3 + i][3 + j] = X.E[i][j];
    }
}
// I_parent += X^T I_child X
template <class T> inline void inertia_accumulate(const Xform<T>& X, const T Ic[6][6], T Ip[6][6]) {
  T M[6][6], t[6][6];
  x_to_mat(X, M);
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      T s = 0;
      for (int k = 0; k < 6; k++) s += Ic[i][k] * M[k][j];
      t[i][j] = s;
    }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      T s = 0;
      for (int k = 0; k < 6; k++) s += M[k][i] * t[k][j];
      Ip[i][j] += s;
    }
}
template <class T> inline void mat6_mul_vec(const T M[6][6], const T* v, T* o) {
  T t[6];
  for (int i = 0; i < 6; i++) {
    T s = 0;
    for (int k = 0; k < 6; k++) s += M[i][k] * v[k];
    t[i] = s;
  }
  std::memcpy(o, t, sizeof(t));
}
// spatial cross products
template <class T> inline void crm(const T* v, const T* m, T* o) {  // v x m (motion)
  T a[3], b[3], c[3];
  cross(v, m, a);
  cross(v, m + 3, b);
  cross(v + 3, m, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i]; o[3 + i] = b[i] + c[i]; }
}
template <class T> inline void crf(const T* v, const T* f, T* o) {  // v x* f (force)
  T a[3], b[3], c[3];
  cross(v, f, a);
  cross(v + 3, f + 3, b);
  cross(v, f + 3, c);
  for (int i = 0; i < 3; i++) { o[i] = a[i] + b[i]; o[3 + i] = c[i]; }
}
// spatial inertia (about the link-frame origin) from mass, com, inertia about com
template <class T> inline void make_spatial_inertia(T m, const T* c, const T Ic[3][3], T I[6][6]) {
  T cx[3][3] = {{0, -c[2], c[1]}, {c[2], 0, -c[0]}, {-c[1], c[0], 0}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      T cc = 0;
      for (int k = 0; k < 3; k++) cc += cx[i][k] * cx[j][k];  // cx cx^T
      I[i][j] = Ic[i][j] + m * cc;
      I[i][3 + j] = m * cx[i][j];
      I[3 + i][j] = -m * cx[i][j];  // m cx^T
      I[3 + i][3 + j] = (i == j) ? m : T(0);
    }
}

// ---------------------------------------------------------------- kinematics from the reference
// a1.py:97-110
template <class T> inline void leg_ik(const T* foot, T l_hip_sign, T* ang) {
  const T l_up = T(0.2), l_low = T(0.2);
  T l_hip = T(0.08505) * l_hip_sign;
  T x = foot[0], y = foot[1], z = foot[2];
  T theta_knee = -std::acos((x * x + y * y + z * z - l_hip * l_hip - l_low * l_low - l_up * l_up) /
                            (2 * l_low * l_up));
  T l = std::sqrt(l_up * l_up + l_low * l_low + 2 * l_up * l_low * std::cos(theta_knee));
  T theta_hip = std::asin(-x / l) - theta_knee / 2;
  T c1 = l_hip * y - l * std::cos(theta_hip + theta_knee / 2) * z;
  T s1 = l * std::cos(theta_hip + theta_knee / 2) * y + l_hip * z;
  ang[0] = std::atan2(s1, c1);
  ang[1] = theta_hip;
  ang[2] = theta_knee;
}
// a1.py:113-129
template <class T> inline void leg_fk(const T* ang, T l_hip_sign, T* p) {
  const T l_up = T(0.2), l_low = T(0.2);
  T l_hip = T(0.08505) * l_hip_sign;
  T leg_distance = std::sqrt(l_up * l_up + l_low * l_low + 2 * l_up * l_low * std::cos(ang[2]));
  T eff_swing = ang[1] + ang[2] / 2;
  T off_x_hip = -leg_distance * std::sin(eff_swing);
  T off_z_hip = -leg_distance * std::cos(eff_swing);
  T off_y_hip = l_hip;
  p[0] = off_x_hip;
  p[1] = std::cos(ang[0]) * off_y_hip - std::sin(ang[0]) * off_z_hip;
  p[2] = std::sin(ang[0]) * off_y_hip + std::cos(ang[0]) * off_z_hip;
}
// a1.py:132-160
template <class T> inline void leg_jacobian(const T* a, int leg_id, T J[3][3]) {
  const T l_up = T(0.2), l_low = T(0.2);
  T l_hip = T(0.08505) * ((leg_id + 1) % 2 == 0 ? T(1) : T(-1));
  T t1 = a[0], t2 = a[1], t3 = a[2];
  T l_eff = std::sqrt(l_up * l_up + l_low * l_low + 2 * l_up * l_low * std::cos(t3));
  T t_eff = t2 + t3 / 2;
  J[0][0] = 0;
  J[0][1] = -l_eff * std::cos(t_eff);
  J[0][2] = l_low * l_up * std::sin(t3) * std::sin(t_eff) / l_eff - l_eff * std::cos(t_eff) / 2;
  J[1][0] = -l_hip * std::sin(t1) + l_eff * std::cos(t1) * std::cos(t_eff);
  J[1][1] = -l_eff * std::sin(t1) * std::sin(t_eff);
  J[1][2] = -l_low * l_up * std::sin(t1) * std::sin(t3) * std::cos(t_eff) / l_eff -
            l_eff * std::sin(t1) * std::sin(t_eff) / 2;
  J[2][0] = l_hip * std::cos(t1) + l_eff * std::sin(t1) * std::cos(t_eff);
  J[2][1] = l_eff * std::sin(t_eff) * std::cos(t1);
  J[2][2] = l_low * l_up * std::sin(t3) * std::cos(t1) * std::cos(t_eff) / l_eff +
            l_eff * std::sin(t_eff) * std::cos(t1) / 2;
}
inline double hip_sign(int leg) { return ((leg + 1) % 2 == 0) ? 1.0 : -1.0; }  // (-1)**(leg+1), a1.py:485

// ---------------------------------------------------------------- ETG
// RBF basis of the (absent) rlschool ETG_layer; validated against the two
// gait_action_list_*.npy fixtures to 4e-15 (tests/test_golden_etg.py).
template <class T> struct EtgBasis {
  T u[ETG_RBF_H][2];
  T omega, amp, sigma_sq, phase[2], Tperiod, T2;
  void init(const EtgConfig& c) {
    Tperiod = T(c.etg_T); T2 = T(c.etg_T2); amp = T(c.etg_amp); sigma_sq = T(c.etg_sigma_sq);
    phase[0] = T(c.etg_phase[0]); phase[1] = T(c.etg_phase[1]);
    omega = T(2.0 * M_PI / c.etg_T);
    for (int h = 0; h < ETG_RBF_H; h++) {
      T t = T(h) * Tperiod / T(ETG_RBF_H - 0.9);
      u[h][0] = amp * std::sin(phase[0] + t * omega);
      u[h][1] = amp * std::sin(phase[1] + t * omega);
    }
  }
  void rbf(T t, T* r) const {
    T x0 = amp * std::sin(phase[0] + t * omega), x1 = amp * std::sin(phase[1] + t * omega);
    for (int h = 0; h < ETG_RBF_H; h++) {
      T d0 = x0 - u[h][0], d1 = x1 - u[h][1];
      r[h] = std::exp(-(d0 * d0 + d1 * d1) / sigma_sq);
    }
  }
};

// ---------------------------------------------------------------- per-env data
template <class T> struct Env {
  // rigid-body state
  T pos[3], quat[4], wb[3], vb[3];  // base twist in base coordinates
  T q[12], qd[12];
  T lam[12];  // warm-start impulses, per foot (n, t1, t2)
  T lamb[4];  // the normal impulse of each leg's body contact in the last tick (0: the row was outside the margin): its warm start
  // latency ring
  T hist[RING][HIST];
  int64_t tick;  // ticks since reset (hist[(tick) % RING] is the newest)
  // control
  int step_count;
  T last_qdes[12];
  int has_last;
  T first_rpy[3];
  int first_rpy_set;
  T last_base[3];
  T last_foot_w[12];
  T fx[2][12], fy[2][12];  // action filter history
  // parameters
  T etg_w[3][ETG_RBF_H], etg_b[3];
  T dyn[ETG_DYN_DIM];
  // derived model
  T I[NB][6][6];  // link spatial inertias (link frame)
  T kp[12], kd[12], mu, latency, grav[3];
  T fext[3];  // external force on the trunk COM, world frame (etg_set_external_force)
  T strength[12];  // motor strength ratios (laikago_motor.py:67-76,138,167; etg_set_motor_strength), 1 unless set
  T reset_off[2];  // start offset (x, y) of the next reset (etg_set_reset_offsets)
  int band;   // terrain band of this robot (env index % hf_bands)
  // outputs of the last tick
  T tau[12];
  int contact[4];
  T energy;
  long long sweep_hist[64];   // ticks by the number of PGS sweeps they ran since etgo_create (etgo_sweep_hist)
  long long sweeps_total;     // all sweeps so far; step_env reports the step's share in info[ETG_INFO_SWEEPS]
  long long body_ticks[3];    // ticks with a body row inside the margin / with a loaded body row (normal impulse > 0) / all ticks (etgo_body_stats)
  // per-tick trace (etgo_set_trace; tests/divergence.py: where two evaluations of one control step part): TRACE_W doubles per tick
  // into the caller's buffer -- [0] rows inside the margin / joint rows at a stop, as a bit mask over the 36 rows; [1] rows with a
  // positive impulse after the solve; [2] body candidate picked per leg (2 bits each); [3] sweeps; [4..7] foot distances phi;
  // [8..11] distance of the picked body sphere; [12..47] impulses of the 36 rows; [48..59] joint angles after the tick;
  // [60..62] base position after the tick; [63] the tick's index since reset
  double* trace = nullptr;
  int trace_cap = 0, trace_n = 0;
};
constexpr int TRACE_W = 64;

template <class T> struct Sim {
  EtgConfig cfg;
  EtgRobotModel model;
  EtgBasis<T> basis;
  int N;
  std::vector<Env<T>> env;
  std::vector<float> heights;
  // Gaussian sensor noise (minitaur.py:1206-1211), etgo_set_sensor_noise; stream position of this call's observation
  int noise_on = 0;
  float noise_std[5] = {0, 0, 0, 0, 0};
  uint64_t noise_seed = 0;
  unsigned obs_calls = 0, noise_call = 0;
  // TEST KNOB (etgo_set_solve_noise; tests/parity_util.OracleEnsemble): every impulse the contact solve returns is multiplied by
  // 1 + solve_noise * U(-1, 1) (counter-based: robot, tick, row, seed) before it is applied -- a model of the rounding noise of an
  // fp32 solve (incremental row velocities over tens of sweeps: ~5e-6 relative on the impulses of a hard landing, measured on the
  // kernel source), so that an ensemble of oracles shows how far that noise moves a robot's trajectory.  0 (always, outside tests).
  double solve_noise = 0;
  uint64_t solve_noise_seed = 0;
  mutable T* dbgM = nullptr;  // optional taps (tests): 18x18 mass matrix, 18 bias
  mutable T* dbgC = nullptr;
  // tree description
  int parent[NB];
  int axis[NB];     // 0 = x, 1 = y
  T jorigin[NB][3]; // joint origin in the parent frame
};

template <class T> void build_tree(Sim<T>& s) {
  const EtgRobotModel& m = s.model;
  s.parent[0] = -1; s.axis[0] = -1;
  for (int l = 0; l < 4; l++) {
    int h = 1 + 3 * l, t = h + 1, c = h + 2;
    s.parent[h] = 0; s.axis[h] = 0;
    for (int k = 0; k < 3; k++) s.jorigin[h][k] = T(m.hip_origin[l][k]);
    s.parent[t] = h; s.axis[t] = 1;
    s.jorigin[t][0] = 0; s.jorigin[t][1] = T(m.thigh_y[l]); s.jorigin[t][2] = 0;
    s.parent[c] = t; s.axis[c] = 1;
    s.jorigin[c][0] = 0; s.jorigin[c][1] = 0; s.jorigin[c][2] = T(-m.upper_len);
  }
}

// scaled link inertia: mass ratio rm, per-axis inertia ratios ri: I' = S I S, S = diag(sqrt(ri))
template <class T> void link_spatial(const EtgLink& L, double rm, const double* ri, const double* shift, T I[6][6]) {
  T Ic[3][3];
  double s[3] = {std::sqrt(ri[0]), std::sqrt(ri[1]), std::sqrt(ri[2])};
  double M[3][3] = {{L.inertia[0], L.inertia[3], L.inertia[4]},
                    {L.inertia[3], L.inertia[1], L.inertia[5]},
                    {L.inertia[4], L.inertia[5], L.inertia[2]}};
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Ic[i][j] = T(s[i] * M[i][j] * s[j]);
  T c[3] = {T(L.com[0] + shift[0]), T(L.com[1] + shift[1]), T(L.com[2] + shift[2])};
  make_spatial_inertia(T(L.mass * rm), c, Ic, I);
}

template <class T> void derive_params(Sim<T>& s, Env<T>& e) {
  const EtgRobotModel& m = s.model;
  const T* d = e.dyn;
  e.latency = d[0] * T(0.001);  // ms -> s (train.py:116)
  e.mu = d[1];
  double zero[3] = {0, 0, 0};
  double bi[3] = {(double)d[3], (double)d[4], (double)d[5]};
  link_spatial<T>(m.trunk, (double)d[2], bi, zero, e.I[0]);
  for (int l = 0; l < 4; l++) {
    int h = 1 + 3 * l;
    double r0[3] = {(double)d[9], (double)d[10], (double)d[11]};
    double r1[3] = {(double)d[12], (double)d[13], (double)d[14]};
    double r2[3] = {(double)d[15], (double)d[16], (double)d[17]};
    double r3[3] = {(double)d[18], (double)d[19], (double)d[20]};
    link_spatial<T>(m.hip[l], (double)d[6], r0, zero, e.I[h]);
    link_spatial<T>(m.thigh[l], (double)d[7], r1, zero, e.I[h + 1]);
    // calf + rigidly attached foot (fixed joint at (0,0,-lower_len), a1.py:99)
    T Ic[6][6], If[6][6];
    link_spatial<T>(m.calf[l], (double)d[8], r2, zero, Ic);
    double fs[3] = {0, 0, -m.lower_len};
    link_spatial<T>(m.foot[l], 1.0, r3, fs, If);
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) e.I[h + 2][i][j] = Ic[i][j] + If[i][j];
  }
  for (int j = 0; j < 12; j++) { e.kp[j] = d[21 + j]; e.kd[j] = d[33 + j]; }
  for (int k = 0; k < 3; k++) e.grav[k] = d[45 + k];
}

// ---------------------------------------------------------------- terrain
template <class T> inline void terrain_query(const Sim<T>& s, int band, T x, T y, T* h, T* n) {
  if (s.cfg.terrain == 0 || s.heights.empty()) {
    *h = 0; n[0] = 0; n[1] = 0; n[2] = 1;
    return;
  }
  // bilinear heightfield, clamped at the border
  // hf_bands terrain variants stacked along y in the heights array (etgsim.h), clamped inside the band
  const int bands = s.cfg.hf_bands > 1 ? s.cfg.hf_bands : 1;
  const int nx = s.cfg.hf_nx, ny = s.cfg.hf_ny / bands;
  const float* hts = s.heights.data() + (size_t)band * ny * nx;
  T fx = (x - T(s.cfg.hf_x0)) / T(s.cfg.hf_cell), fy = (y - T(s.cfg.hf_y0)) / T(s.cfg.hf_cell);
  if (!(fx > 0)) fx = 0;   // (also a NaN coordinate of a robot that blew up: max(NaN, 0) = 0, what the kernels' fmaxf does)
  if (!(fy > 0)) fy = 0;
  if (fx > T(nx - 1)) fx = T(nx - 1);
  if (fy > T(ny - 1)) fy = T(ny - 1);
  int ix = (int)fx, iy = (int)fy;
  if (ix > nx - 2) ix = nx - 2;
  if (iy > ny - 2) iy = ny - 2;
  T tx = fx - T(ix), ty = fy - T(iy);
  T h00 = T(hts[iy * nx + ix]), h10 = T(hts[iy * nx + ix + 1]);
  T h01 = T(hts[(iy + 1) * nx + ix]), h11 = T(hts[(iy + 1) * nx + ix + 1]);
  *h = (1 - tx) * (1 - ty) * h00 + tx * (1 - ty) * h10 + (1 - tx) * ty * h01 + tx * ty * h11;
  T dhdx = ((1 - ty) * (h10 - h00) + ty * (h11 - h01)) / T(s.cfg.hf_cell);
  T dhdy = ((1 - tx) * (h01 - h00) + tx * (h11 - h10)) / T(s.cfg.hf_cell);
  T inv = T(1) / std::sqrt(dhdx * dhdx + dhdy * dhdy + 1);
  n[0] = -dhdx * inv; n[1] = -dhdy * inv; n[2] = inv;
}

// ---------------------------------------------------------------- one physics tick
// stepSimulation() of minitaur.py:244, as defined by this repo (DESIGN.md "physics model"):
//   M(q) [a_b; qdd] + C(q, v) = [0; tau] + J^T f
// 1. CRBA -> M (18x18), RNEA(qdd = 0, a_b = -g) -> C
// 2. v* = v + dt M^-1 ([0;tau] - C)
// 3. contacts: foot spheres vs ground; rows (n, t1, t2) per active foot; Delassus
//    A = J M^-1 J^T; projected Gauss-Seidel in fixed order with cone projection
// 4. v+ = v* + M^-1 J^T lambda ; semi-implicit Euler on positions
template <class T> void physics_tick(const Sim<T>& s, Env<T>& e, const T* tau) {
  const T dt = T(s.cfg.sim_dt);
  T R[3][3];
  quat_to_mat(e.quat, R);

  // ---- kinematics
  Xform<T> Xup[NB];
  T S[NB][6];
  T v[NB][6], avp[NB][6], fvp[NB][6];
  T IC[NB][6][6];
  for (int k = 0; k < 3; k++) { v[0][k] = e.wb[k]; v[0][3 + k] = e.vb[k]; }
  T gb[3];
  mat3T_mul_vec(R, e.grav, gb);
  for (int k = 0; k < 3; k++) { avp[0][k] = 0; avp[0][3 + k] = -gb[k]; }
  {
    T Iv[6], Ia[6], c[6];
    mat6_mul_vec(e.I[0], v[0], Iv);
    mat6_mul_vec(e.I[0], avp[0], Ia);
    crf(v[0], Iv, c);
    for (int k = 0; k < 6; k++) fvp[0][k] = Ia[k] + c[k];
  }
  std::memcpy(IC, e.I, sizeof(IC));
  for (int i = 1; i < NB; i++) {
    T ang = e.q[i - 1];
    T cs = std::cos(ang), sn = std::sin(ang);
    // E = rot(axis, ang)^T (parent -> child coordinates)
    T E[3][3];
    if (s.axis[i] == 0) {
      T t[3][3] = {{1, 0, 0}, {0, cs, sn}, {0, -sn, cs}};
      std::memcpy(E, t, sizeof(t));
    } else {
      T t[3][3] = {{cs, 0, -sn}, {0, 1, 0}, {sn, 0, cs}};
      std::memcpy(E, t, sizeof(t));
    }
    std::memcpy(Xup[i].E, E, sizeof(E));
    for (int k = 0; k < 3; k++) Xup[i].r[k] = s.jorigin[i][k];
    for (int k = 0; k < 6; k++) S[i][k] = 0;
    S[i][s.axis[i]] = 1;
    int p = s.parent[i];
    T vj[6];
    for (int k = 0; k < 6; k++) vj[k] = S[i][k] * e.qd[i - 1];
    x_apply_motion(Xup[i], v[p], v[i]);
    for (int k = 0; k < 6; k++) v[i][k] += vj[k];
    T t6[6];
    x_apply_motion(Xup[i], avp[p], avp[i]);
    crm(v[i], vj, t6);
    for (int k = 0; k < 6; k++) avp[i][k] += t6[k];
    T Iv[6], Ia[6], c[6];
    mat6_mul_vec(e.I[i], v[i], Iv);
    mat6_mul_vec(e.I[i], avp[i], Ia);
    crf(v[i], Iv, c);
    for (int k = 0; k < 6; k++) fvp[i][k] = Ia[k] + c[k];
  }
  // ---- backward pass: composite inertias and bias forces
  for (int i = NB - 1; i >= 1; i--) {
    int p = s.parent[i];
    inertia_accumulate(Xup[i], IC[i], IC[p]);
    T f[6];
    x_applyT_force(Xup[i], fvp[i], f);
    for (int k = 0; k < 6; k++) fvp[p][k] += f[k];
  }
  T M[NV][NV];
  T C[NV];
  std::memset(M, 0, sizeof(M));
  for (int a = 0; a < 6; a++) {
    C[a] = fvp[0][a];
    for (int b = 0; b < 6; b++) M[a][b] = IC[0][a][b];
  }
  for (int i = 1; i < NB; i++) {
    C[5 + i] = dot3(S[i], fvp[i]) + dot3(S[i] + 3, fvp[i] + 3);
    T fh[6];
    mat6_mul_vec(IC[i], S[i], fh);
    M[5 + i][5 + i] = dot3(S[i], fh) + dot3(S[i] + 3, fh + 3);
    int j = i;
    while (s.parent[j] > 0) {
      T t[6];
      x_applyT_force(Xup[j], fh, t);
      std::memcpy(fh, t, sizeof(t));
      j = s.parent[j];
      T h = dot3(S[j], fh) + dot3(S[j] + 3, fh + 3);
      M[5 + i][5 + j] = h; M[5 + j][5 + i] = h;
    }
    T t[6];
    x_applyT_force(Xup[j], fh, t);
    for (int a = 0; a < 6; a++) { M[a][5 + i] = t[a]; M[5 + i][a] = t[a]; }
  }
  if (s.dbgM) std::memcpy(s.dbgM, M, sizeof(M));
  if (s.dbgC) std::memcpy(s.dbgC, C, sizeof(C));
  // ---- Cholesky M = L L^T
  T L[NV][NV];
  std::memset(L, 0, sizeof(L));
  for (int i = 0; i < NV; i++)
    for (int j = 0; j <= i; j++) {
      T sum = M[i][j];
      for (int k = 0; k < j; k++) sum -= L[i][k] * L[j][k];
      if (i == j) L[i][i] = std::sqrt(sum);
      else L[i][j] = sum / L[j][j];
    }
  auto chol_solve = [&](const T* b, T* x) {
    T y[NV];
    for (int i = 0; i < NV; i++) {
      T sum = b[i];
      for (int k = 0; k < i; k++) sum -= L[i][k] * y[k];
      y[i] = sum / L[i][i];
    }
    for (int i = NV - 1; i >= 0; i--) {
      T sum = y[i];
      for (int k = i + 1; k < NV; k++) sum -= L[k][i] * x[k];
      x[i] = sum / L[i][i];
    }
  };
  T rhs[NV], acc[NV], vel[NV];
  for (int a = 0; a < 6; a++) rhs[a] = -C[a];
  {  // external force on the trunk COM: world -> base coordinates
    T fb[3];
    mat3T_mul_vec(R, e.fext, fb);
    for (int k = 0; k < 3; k++) rhs[3 + k] += fb[k];
  }
  for (int j = 0; j < 12; j++) rhs[6 + j] = tau[j] - C[6 + j];
  chol_solve(rhs, acc);
  for (int k = 0; k < 3; k++) { vel[k] = e.wb[k]; vel[3 + k] = e.vb[k]; }
  for (int j = 0; j < 12; j++) vel[6 + j] = e.qd[j];
  for (int i = 0; i < NV; i++) vel[i] += dt * acc[i];

  // ---- contacts
  // world poses of the leg links: R_w[i], p_w[i] (link origin in world)
  T Rw[NB][3][3], pw[NB][3];
  std::memcpy(Rw[0], R, sizeof(R));
  for (int k = 0; k < 3; k++) pw[0][k] = e.pos[k];
  for (int i = 1; i < NB; i++) {
    int p = s.parent[i];
    T Et[3][3];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) Et[a][b] = Xup[i].E[b][a];
    mat3_mul(Rw[p], Et, Rw[i]);
    T o[3];
    mat3_mul_vec(Rw[p], Xup[i].r, o);
    for (int k = 0; k < 3; k++) pw[i][k] = pw[p][k] + o[k];
  }
  // rows 0..11: feet (n, t1, t2 per leg); rows 12 + 3 l + b: the body rows of leg l (cfg.body_contacts) -- for body_contacts 1 / 2
  // ONE contact per leg with its normal and two friction rows (b = n, t1, t2), for body_contacts 3 three frictionless normal
  // rows (b = knee, shin midpoint, trunk corner); then 12 joint-limit rows (cfg.joint_limits), one per joint, row NRC + j
  constexpr int NRMAX = 36;
  const bool all_bodies = s.cfg.body_contacts == 3;
  const bool body_fric = s.cfg.body_contacts == 1 || s.cfg.body_contacts == 2;   // the leg's body contact has friction rows
  const int NRC = 24;                                      // contact rows
  const int NR = NRC + 12;
  const int NBR = all_bodies ? 3 : 1;                      // body contact points per leg
  const bool body_warm = s.cfg.body_contacts == 1 || (s.cfg.body_contacts == 2 && s.cfg.body_blend > 0);   // (see "Warm start" below)
  auto body_row = [&](int l, int b) { return 12 + 3 * l + b; };
  T J[NRMAX][NV];
  T target[NRMAX];
  double tr_phi[8] = {0, 0, 0, 0, 0, 0, 0, 0};              // trace only: foot distances, picked body sphere distances
  int tr_pick = 0;
  int active[4], kactive[NRMAX];                            // kactive / klam: indexed by row
  T klam[NRMAX];                                            // body impulses: no warm start
  for (int r = 0; r < NRMAX; r++) { kactive[r] = 0; klam[r] = 0; target[r] = 0; }
  std::memset(J, 0, sizeof(J));
  const T rad = T(s.model.foot_radius);
  for (int l = 0; l < 4; l++) {
    int c = 3 + 3 * l;  // calf body
    T fl[3] = {0, 0, T(-s.model.lower_len)}, fw[3];
    mat3_mul_vec(Rw[c], fl, fw);
    for (int k = 0; k < 3; k++) fw[k] += pw[c][k];  // foot centre, world
    T h, n[3];
    terrain_query(s, e.band, fw[0], fw[1], &h, n);
    T phi = (fw[2] - h) * n[2] - rad;  // distance along the normal to the tangent plane
    tr_phi[l] = (double)phi;
    active[l] = phi < T(s.cfg.contact_margin);
    e.contact[l] = 0;
    if (!active[l]) {
      for (int k = 0; k < 3; k++) { e.lam[3 * l + k] = 0; target[3 * l + k] = 0; }
      continue;
    }
    // contact frame: n, t1 = normalised projection of world x, t2 = n x t1
    T t1[3] = {1 - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
    T inv = T(1) / std::sqrt(dot3(t1, t1));
    for (int k = 0; k < 3; k++) t1[k] *= inv;
    T t2[3];
    cross(n, t1, t2);
    T cp[3];  // contact point (sphere bottom along the normal), world
    for (int k = 0; k < 3; k++) cp[k] = fw[k] - rad * n[k];
    const T* dirs[3] = {n, t1, t2};
    for (int r = 0; r < 3; r++) {
      const T* d = dirs[r];
      T* row = J[3 * l + r];
      T rel[3], rxd[3], tmp[3];
      for (int k = 0; k < 3; k++) rel[k] = cp[k] - e.pos[k];
      cross(rel, d, rxd);
      mat3T_mul_vec(R, rxd, tmp);  // base angular part in base coords
      for (int k = 0; k < 3; k++) row[k] = tmp[k];
      mat3T_mul_vec(R, d, tmp);
      for (int k = 0; k < 3; k++) row[3 + k] = tmp[k];
      for (int b = c; b > 0; b = s.parent[b]) {  // joints of this leg
        T axw[3] = {Rw[b][0][s.axis[b]], Rw[b][1][s.axis[b]], Rw[b][2][s.axis[b]]};
        T rj[3], cr[3];
        for (int k = 0; k < 3; k++) rj[k] = cp[k] - pw[b][k];
        cross(axw, rj, cr);
        row[5 + b] = dot3(d, cr);
      }
    }
    // normal target velocity: speculative for a gap, Baumgarte for penetration
    // (Bullet: penetration = distance + m_linearSlop; pybullet's server sets the slop to 1e-5 m: cfg.contact_slop)
    const T pen = phi + T(s.cfg.contact_slop);
    target[3 * l] = (pen > 0) ? -pen / dt : -T(s.cfg.erp) * pen / dt;
    target[3 * l + 1] = 0; target[3 * l + 2] = 0;
    // restitution (cfg.foot_restitution = the COMBINED coefficient of foot and ground, Bullet multiplies the two bodies'):
    // an approach faster than Bullet's restitutionVelocityThreshold (0.2 m/s) at the start of the tick adds e * (-u_n)
    if (s.cfg.foot_restitution > 0) {
      T un0 = 0;
      const T* row = J[3 * l];
      for (int k = 0; k < 3; k++) un0 += row[k] * e.wb[k] + row[3 + k] * e.vb[k];
      for (int j = 0; j < 12; j++) un0 += row[6 + j] * e.qd[j];
      if (un0 < -T(0.2)) target[3 * l] += T(s.cfg.foot_restitution) * (-un0);
    }
    // warm start: the previous tick's normal impulse x cfg.warmstart, the friction impulses x cfg.warmstart_friction
    // (Bullet's multibody solver restarts friction rows from zero: setupMultiBodyContactConstraint, isFriction ? 0 : ...)
    e.lam[3 * l] *= T(s.cfg.warmstart);
    for (int k = 1; k < 3; k++) e.lam[3 * l + k] *= T(s.cfg.warmstart_friction);
  }
  // body contacts (cfg.body_contacts): spheres of knee_radius standing in for the link shapes Bullet collides (a1.py:276-287 loads
  // the URDF with every link's collision shape; no self-collision).  1: one contact per leg on a sphere at the knee (the calf
  // joint origin, carried by the thigh: the calf joint does not move it).  2: one contact per leg on the DEEPEST of three
  // spheres -- knee, shin midpoint (carried by the calf), trunk corner next to the leg's hip (carried by the base: no joint
  // moves it).  A contact of modes 1 / 2 has a normal row and two friction rows like a foot's (Bullet gives every contact point
  // friction), with the coefficient cfg.body_friction (the product of the link's and the ground's lateralFriction: only the
  // FEET's is ever changed by the reference, minitaur.py:1100-1110).  3: all three spheres of every leg collide at once, a
  // frictionless normal row each (solved in that order).
  // Warm start (Bullet's persistent manifold keeps a contact point's applied impulse from step to step and
  // setupMultiBodyContactConstraint restarts the NORMAL row from it x warmstartingFactor, friction rows from zero): a leg's
  // body contact is one persistent point when the single knee sphere carries it (mode 1) or when it is the blended contact
  // (mode 2 with body_blend > 0: its position is continuous in the state) -- its normal row then starts from cfg.warmstart x
  // the impulse of the tick before (0 if the row was outside the margin then), like a foot's.  Under the hard deepest-of-three
  // choice (mode 2, body_blend = 0) the point changes identity from tick to tick, and in mode 3 the rows are the legacy
  // frictionless set: no warm start there.
  bool any_margin = false;
  for (int l = 0; l < 4; l++) {
    if (!s.cfg.body_contacts) continue;
    const int c = 3 + 3 * l;
    const T krad = T(s.cfg.knee_radius);
    struct Cand { T p[3]; int first_body; };   // first_body: the chain of joints moving the point starts at this body (0 = none)
    Cand cand[3];
    int ncand = 1;
    for (int k = 0; k < 3; k++) cand[0].p[k] = pw[c][k];
    cand[0].first_body = s.parent[c];
    if (s.cfg.body_contacts >= 2) {
      T sl[3] = {0, 0, T(-0.5 * s.model.lower_len)}, o[3];
      mat3_mul_vec(Rw[c], sl, o);
      for (int k = 0; k < 3; k++) cand[1].p[k] = pw[c][k] + o[k];
      cand[1].first_body = c;
      T corner[3] = {T(s.model.hip_origin[l][0] > 0 ? s.cfg.trunk_half[0] : -s.cfg.trunk_half[0]),
                     T(s.model.hip_origin[l][1] > 0 ? s.cfg.trunk_half[1] : -s.cfg.trunk_half[1]), T(-s.cfg.trunk_half[2])};
      mat3_mul_vec(R, corner, o);
      for (int k = 0; k < 3; k++) cand[2].p[k] = e.pos[k] + o[k];
      cand[2].first_body = 0;
      ncand = 3;
    }
    T cphi[3], cn[3][3];
    for (int q = 0; q < ncand; q++) {
      T h;
      terrain_query(s, e.band, cand[q].p[0], cand[q].p[1], &h, cn[q]);
      cphi[q] = (cand[q].p[2] - h) * cn[q][2] - krad;
    }
    for (int b = 0; b < NBR; b++) {
      // which spheres carry the contact, with what weights.  body_contacts 3: slot b is candidate b.  1 / 2: the deepest candidate
      // (ties to the earlier one) -- or, with cfg.body_blend > 0, ALL candidates with weights exp(-(d_i - d_min) / body_blend):
      // the contact's impulse is distributed over the spheres (etgsim.h: body_blend).
      T wq[3] = {0, 0, 0};
      int pick = b;
      if (all_bodies) {
        wq[b] = 1;
      } else {
        pick = 0;
        for (int q = 1; q < ncand; q++)
          if (cphi[q] < cphi[pick]) pick = q;
        if (s.cfg.body_blend > 0 && ncand > 1) {
          T sum = 0;
          for (int q = 0; q < ncand; q++) { wq[q] = std::exp(-(cphi[q] - cphi[pick]) / T(s.cfg.body_blend)); sum += wq[q]; }
          for (int q = 0; q < ncand; q++) wq[q] /= sum;
        } else {
          wq[pick] = 1;
        }
      }
      const int rk = body_row(l, b);
      // the contact is a sphere of knee_radius centred at the weighted mean of the spheres' centres; the ground's height and
      // normal are taken under THAT point (one-hot weights: the picked sphere's own)
      T pc[3] = {0, 0, 0};
      for (int q = 0; q < ncand; q++)
        for (int k = 0; k < 3; k++) pc[k] += wq[q] * cand[q].p[k];
      T hc, n[3];
      terrain_query(s, e.band, pc[0], pc[1], &hc, n);
      const T phi = (pc[2] - hc) * n[2] - krad;
      if (b == 0) { tr_phi[4 + l] = (double)phi; tr_pick |= pick << (2 * l); }
      const bool on = phi < T(s.cfg.contact_margin);
      kactive[rk] = on;
      if (body_fric) kactive[rk + 1] = kactive[rk + 2] = on;
      if (!on) continue;
      any_margin = true;
      T cp[3] = {0, 0, 0}, rel[3];                     // the contact point: weighted mean of the spheres' points
      for (int q = 0; q < ncand; q++)
        for (int k = 0; k < 3; k++) cp[k] += wq[q] * (cand[q].p[k] - krad * n[k]);
      for (int k = 0; k < 3; k++) rel[k] = cp[k] - e.pos[k];
      // contact frame as for a foot: n, t1 = normalised projection of world x, t2 = n x t1
      T t1[3] = {1 - n[0] * n[0], -n[0] * n[1], -n[0] * n[2]};
      T inv = T(1) / std::sqrt(dot3(t1, t1));
      for (int k = 0; k < 3; k++) t1[k] *= inv;
      T t2[3];
      cross(n, t1, t2);
      const T* dirs[3] = {n, t1, t2};
      for (int r = 0; r < (body_fric ? 3 : 1); r++) {
        const T* d = dirs[r];
        T* row = J[rk + r];
        T rxd[3], tmp[3];
        cross(rel, d, rxd);
        mat3T_mul_vec(R, rxd, tmp);
        for (int k = 0; k < 3; k++) row[k] = tmp[k];
        mat3T_mul_vec(R, d, tmp);
        for (int k = 0; k < 3; k++) row[3 + k] = tmp[k];
        for (int q = 0; q < ncand; q++) {
          if (wq[q] == 0) continue;
          T cq[3];
          for (int k = 0; k < 3; k++) cq[k] = cand[q].p[k] - krad * n[k];   // sphere q's own point
          for (int bd = cand[q].first_body; bd > 0; bd = s.parent[bd]) {  // the joints that move sphere q, weighted
            T axw[3] = {Rw[bd][0][s.axis[bd]], Rw[bd][1][s.axis[bd]], Rw[bd][2][s.axis[bd]]};
            T rj[3], cr[3];
            for (int k = 0; k < 3; k++) rj[k] = cq[k] - pw[bd][k];
            cross(axw, rj, cr);
            row[5 + bd] += wq[q] * dot3(d, cr);
          }
        }
      }
      const T pen = phi + T(s.cfg.contact_slop);
      target[rk] = (pen > 0) ? -pen / dt : -T(s.cfg.erp) * pen / dt;
      if (body_warm) klam[rk] = T(s.cfg.warmstart) * e.lamb[l];
    }
  }
  // joint-limit rows (cfg.joint_limits; bounds of a1.py:186-195 = the URDF limits Bullet turns into btMultiBodyJointLimitConstraint
  // rows): a joint at or beyond a bound gets a unilateral row along the joint coordinate, pushing back into the range, with the
  // velocity target erp * violation / dt; no warm start.  Solved INSIDE the sweeps, before the contact rows (Bullet's
  // solveSingleIteration: non-contact constraints, then normal contacts, then friction).
  for (int j = 0; j < 12; j++) {
    if (!s.cfg.joint_limits) break;
    const T lo = T(s.cfg.joint_lower[j % 3]), hi = T(s.cfg.joint_upper[j % 3]);
    T sgn = 0, viol = 0;
    if (e.q[j] >= hi) { sgn = -1; viol = e.q[j] - hi; }
    else if (e.q[j] <= lo) { sgn = 1; viol = lo - e.q[j]; }
    if (sgn == 0) continue;
    const int rk = NRC + j;
    kactive[rk] = 1;
    J[rk][6 + j] = sgn;
    target[rk] = T(s.cfg.erp) * viol / dt;
  }
  auto row_active = [&](int r) { return r < 12 ? active[r / 3] : kactive[r]; };
  auto lam_of = [&](int r) -> T& { return r < 12 ? e.lam[r] : klam[r]; };
  // Delassus operator
  T MiJt[NRMAX][NV];
  T A[NRMAX][NRMAX];
  for (int r = 0; r < NR; r++) {
    if (!row_active(r)) { std::memset(MiJt[r], 0, sizeof(MiJt[r])); continue; }
    chol_solve(J[r], MiJt[r]);
  }
  for (int r = 0; r < NR; r++)
    for (int c = 0; c < NR; c++) {
      T sum = 0;
      for (int k = 0; k < NV; k++) sum += J[r][k] * MiJt[c][k];
      A[r][c] = sum;
    }
  T u[NRMAX];
  for (int r = 0; r < NR; r++) {
    T sum = 0;
    for (int k = 0; k < NV; k++) sum += J[r][k] * vel[k];
    for (int c = 0; c < NR; c++) sum += A[r][c] * lam_of(c);
    u[r] = sum;
  }
  auto apply = [&](int row, T d) {
    for (int r = 0; r < NR; r++) u[r] += A[r][row] * d;
  };
  // Stopping rule (EtgConfig.solver_residual; etgsim.h states it): after every sweep the velocity-level change each row's
  // own impulse change produced, d_r = (lam_r - lam_r at the start of the sweep) * A_rr, is squared and the largest value
  // compared with the threshold -- the least-squares-residual test of Bullet's sequential-impulse loop, which pybullet runs
  // with numSolverIterations = 50 and solverResidualThreshold = 1e-7.  solver_residual = 0: exactly solver_iters sweeps.
  const T res_thr = T(s.cfg.solver_residual);
  const bool pyramid = s.cfg.friction_model == 1;
  int sweeps = 0;
  for (int it = 0; it < s.cfg.solver_iters; it++) {
    T lam_start[NRMAX];
    for (int r = 0; r < NR; r++) lam_start[r] = lam_of(r);
    // One sweep in the order of Bullet's btMultiBodyConstraintSolver::solveSingleIteration (what stepSimulation() runs,
    // minitaur.py:244): (1) the non-contact constraints = joint-limit rows, (2) every NORMAL contact row -- feet FR, FL, RR, RL,
    // then the body rows -- (3) the friction rows of every foot, their limits taken from the normal impulse of this sweep.
    auto unilateral = [&](int rk) {
      T lk = klam[rk] - (u[rk] - target[rk]) / A[rk][rk];
      if (lk < 0) lk = 0;
      apply(rk, lk - klam[rk]);
      klam[rk] = lk;
    };
    for (int j = 0; j < 12; j++)
      if (kactive[NRC + j]) unilateral(NRC + j);
    for (int l = 0; l < 4; l++) {
      if (!active[l]) continue;
      const int r0 = 3 * l;
      T ln = e.lam[r0] - (u[r0] - target[r0]) / A[r0][r0];
      if (ln < 0) ln = 0;
      apply(r0, ln - e.lam[r0]);
      e.lam[r0] = ln;
    }
    for (int l = 0; l < 4; l++)
      for (int b = 0; b < NBR; b++)
        if (kactive[body_row(l, b)]) unilateral(body_row(l, b));
    // both candidates from the SAME velocities, the pair projected on the friction disc `lim`, then both changes applied:
    // the implicit cone friction of pybullet's default (resolveConeFrictionConstraintRows; enableConeFriction = 1).
    // friction_model 1: each direction clamped on its own to +-lim (the pyramid of enableConeFriction = 0).
    auto friction_pair = [&](int r0, T lim, T* lamv) {
      T cand[3];
      for (int k = 1; k < 3; k++) cand[k] = lamv[r0 + k] - u[r0 + k] / A[r0 + k][r0 + k];
      if (pyramid) {
        for (int k = 1; k < 3; k++) {
          if (cand[k] > lim) cand[k] = lim;
          if (cand[k] < -lim) cand[k] = -lim;
        }
      } else {
        T nt = std::sqrt(cand[1] * cand[1] + cand[2] * cand[2]);
        if (nt > lim) {
          T sc = (nt > 0) ? lim / nt : T(0);
          cand[1] *= sc; cand[2] *= sc;
        }
      }
      for (int k = 1; k < 3; k++) {
        apply(r0 + k, cand[k] - lamv[r0 + k]);
        lamv[r0 + k] = cand[k];
      }
    };
    for (int l = 0; l < 4; l++) {
      if (!active[l]) continue;
      const int r0 = 3 * l;
      const T ln = e.lam[r0];
      // Bullet solves a contact's friction rows only while its normal impulse is positive (`if (totalImpulse > 0)`): a foot
      // whose normal impulse is zero keeps the friction impulses it has
      if (!(ln > 0)) continue;
      friction_pair(r0, e.mu * ln, e.lam);
    }
    // the friction pairs of the body contacts (body_contacts 1 / 2), after the feet's, same rule with cfg.body_friction
    for (int l = 0; l < 4 && body_fric; l++) {
      const int r0 = body_row(l, 0);
      if (!kactive[r0]) continue;
      const T ln = klam[r0];
      if (!(ln > 0)) continue;
      friction_pair(r0, T(s.cfg.body_friction) * ln, klam);
    }
    sweeps++;
    if (res_thr > 0) {
      T res = 0;
      for (int r = 0; r < NR; r++) {
        if (!row_active(r)) continue;
        const T d = (lam_of(r) - lam_start[r]) * A[r][r];
        if (d * d > res) res = d * d;
      }
      if (res <= res_thr) break;
    }
  }
  e.sweep_hist[sweeps < 63 ? sweeps : 63]++;
  e.sweeps_total += sweeps;
  if (s.solve_noise > 0) {   // test knob: see Sim::solve_noise
    const uint64_t envi = (uint64_t)(&e - &s.env[0]);
    for (int r = 0; r < NR; r++) {
      if (!row_active(r)) continue;
      uint64_t z = s.solve_noise_seed * 0x9E3779B97F4A7C15ull + (envi << 40) + ((uint64_t)e.tick << 8) + (uint64_t)r;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      const double u01 = (double)(z >> 11) * (1.0 / 9007199254740992.0);
      lam_of(r) = lam_of(r) * T(1.0 + s.solve_noise * (2.0 * u01 - 1.0));
    }
  }
  {
    bool loaded = false;
    for (int l = 0; l < 4; l++)
      for (int b = 0; b < NBR; b++) loaded = loaded || (kactive[body_row(l, b)] && klam[body_row(l, b)] > 0);
    e.body_ticks[0] += any_margin; e.body_ticks[1] += loaded; e.body_ticks[2] += 1;
  }
  for (int r = 0; r < NR; r++) {
    if (!row_active(r)) continue;
    for (int k = 0; k < NV; k++) vel[k] += MiJt[r][k] * lam_of(r);
  }
  for (int l = 0; l < 4; l++) e.contact[l] = active[l] && e.lam[3 * l] > 0;
  for (int l = 0; l < 4; l++) e.lamb[l] = (body_warm && kactive[body_row(l, 0)]) ? klam[body_row(l, 0)] : T(0);
  double* tr = (e.trace && e.trace_n < e.trace_cap) ? e.trace + (size_t)TRACE_W * e.trace_n++ : nullptr;
  if (tr) {
    unsigned long long am = 0, lm = 0;
    for (int r = 0; r < NR; r++) {
      if (row_active(r)) am |= 1ull << r;
      if (row_active(r) && lam_of(r) > 0) lm |= 1ull << r;
      tr[12 + r] = row_active(r) ? (double)lam_of(r) : 0.0;
    }
    tr[0] = (double)am; tr[1] = (double)lm; tr[2] = (double)tr_pick; tr[3] = (double)sweeps;
    for (int k = 0; k < 8; k++) tr[4 + k] = tr_phi[k];
  }

  // ---- integrate (semi-implicit Euler)
  for (int k = 0; k < 3; k++) { e.wb[k] = vel[k]; e.vb[k] = vel[3 + k]; }
  for (int j = 0; j < 12; j++) { e.qd[j] = vel[6 + j]; e.q[j] += dt * e.qd[j]; }
  T vw[3];
  mat3_mul_vec(R, e.vb, vw);
  for (int k = 0; k < 3; k++) e.pos[k] += dt * vw[k];
  T th[3] = {e.wb[0] * dt, e.wb[1] * dt, e.wb[2] * dt};
  T ang = std::sqrt(dot3(th, th));
  T dq[4];
  if (ang > T(1e-12)) {
    T sh = std::sin(ang / 2) / ang;
    dq[0] = th[0] * sh; dq[1] = th[1] * sh; dq[2] = th[2] * sh; dq[3] = std::cos(ang / 2);
  } else {
    dq[0] = th[0] / 2; dq[1] = th[1] / 2; dq[2] = th[2] / 2; dq[3] = 1;
  }
  T qn[4];
  quat_mul(e.quat, dq, qn);
  T nn = T(1) / std::sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
  for (int k = 0; k < 4; k++) e.quat[k] = qn[k] * nn;
  if (tr) {
    for (int j = 0; j < 12; j++) tr[48 + j] = (double)e.q[j];
    for (int k = 0; k < 3; k++) tr[60 + k] = (double)e.pos[k];
    tr[63] = (double)e.tick;
  }
}

// ---------------------------------------------------------------- robot layer
// GetTrueObservation (minitaur.py:1142-1149) without the torques
template <class T> void push_history(Env<T>& e) {
  e.tick++;
  T* h = e.hist[e.tick % RING];
  for (int j = 0; j < 12; j++) { h[j] = e.q[j]; h[12 + j] = e.qd[j]; }
  for (int k = 0; k < 4; k++) h[24 + k] = e.quat[k];
  // body-frame angular velocity (TransformAngularVelocityToLocalFrame, minitaur.py:849-870)
  for (int k = 0; k < 3; k++) h[28 + k] = e.wb[k];
}
// _GetDelayedObservation (minitaur.py:1172-1193); the ring is always full after the settle
template <class T> void delayed_obs(const Sim<T>& s, const Env<T>& e, T* o) {
  const T dt = T(s.cfg.sim_dt);
  if (e.latency <= 0) {
    std::memcpy(o, e.hist[e.tick % RING], sizeof(T) * HIST);
    return;
  }
  int n = (int)(e.latency / dt);
  if (n > RING - 2) n = RING - 2;
  T rem = e.latency - T(n) * dt;
  T alpha = rem / dt;
  const T* a = e.hist[(e.tick - n + 4 * RING) % RING];
  const T* b = e.hist[(e.tick - n - 1 + 4 * RING) % RING];
  for (int k = 0; k < HIST; k++) o[k] = (T(1) - alpha) * a[k] + alpha * b[k];
}

// LaikagoMotorModel.convert_to_torque (laikago_motor.py:103-175): POSITION / HYBRID  tau = -kp (q - q_des) - kd (qd - qd_des) + tau_ff,
// TORQUE the command itself (passed as tau_ff); then x strength_ratio (:138,:167), then the optional clip to +-limit (:168-173;
// the TORQUE branch returns before the clip, :137-139)
template <class T> inline T motor_torque(T q, T qd, T q_des, T kp, T kd, T qd_des, T tau_ff, T strength, T limit, bool torque_cmd) {
  if (torque_cmd) return strength * tau_ff;
  T t = ((-(kp * (q - q_des))) - kd * (qd - qd_des)) + tau_ff;
  t = strength * t;
  if (limit > 0) {
    if (t > limit) t = limit;
    if (t < -limit) t = -limit;
  }
  return t;
}
// MapToMinusPiToPi (minitaur.py:67-83)
template <class T> inline T wrap_to_pi(T a) {
  const T two_pi = T(6.283185307179586), pi = T(3.141592653589793);
  a = std::fmod(a, two_pi);
  if (a >= pi) a -= two_pi;
  else if (a < -pi) a += two_pi;
  return a;
}

// one sub-step: ApplyAction (PD, laikago_motor.py:165-173; pd_latency = 0) -> tick -> history
// hyb (HYBRID mode, laikago_motor.py:152-167): 12 x (kp, qd_des, kd, tau_ff) replacing the model's gains
template <class T> void sub_step(const Sim<T>& s, Env<T>& e, const T* qdes, bool torque_cmd = false, const T* hyb = nullptr) {
  T tau[12];
  // _GetPDObservation (minitaur.py:1195-1199): with pd_latency the PD law reads the joint state of that long ago, blended
  // from the two bracketing history entries like the control observation (minitaur.py:1172-1193)
  T qm[12], qdm[12];
  for (int j = 0; j < 12; j++) { qm[j] = e.q[j]; qdm[j] = e.qd[j]; }
  if (s.cfg.pd_latency > 0) {
    const T dt = T(s.cfg.sim_dt), lat = T(s.cfg.pd_latency);
    int n = (int)(s.cfg.pd_latency / s.cfg.sim_dt);
    if (n > RING - 2) n = RING - 2;
    const T alpha = (lat - T(n) * dt) / dt;
    const int64_t ta = e.tick - n < 0 ? 0 : e.tick - n, tb = e.tick - n - 1 < 0 ? 0 : e.tick - n - 1;   // before the first tick: the initial reading
    const T* a = e.hist[ta % RING];
    const T* b = e.hist[tb % RING];
    for (int j = 0; j < 12; j++) { qm[j] = (T(1) - alpha) * a[j] + alpha * b[j]; qdm[j] = (T(1) - alpha) * a[12 + j] + alpha * b[12 + j]; }
  }
  // A1._ClipMotorCommands (a1.py:439-457) clips against GetMotorAngles() = the control-latency-delayed reading of the motor
  // angles, wrapped to [-pi, pi] (minitaur.py:753-764, 67-83) -- without the sensor noise the reference adds to every reading
  T qclip[12];
  if (s.cfg.clip_motor_commands > 0 && !torque_cmd) {
    T dl[HIST];
    delayed_obs(s, e, dl);
    for (int j = 0; j < 12; j++) qclip[j] = wrap_to_pi(dl[j]);
  }
  for (int j = 0; j < 12; j++) {
    // POSITION: laikago_motor.py:165-173; TORQUE: the command is the torque (laikago_motor.py:140-143)
    T cmd = qdes[j];
    if (s.cfg.clip_motor_commands > 0 && !torque_cmd) {
      const T lim = T(s.cfg.clip_motor_commands);
      if (cmd > qclip[j] + lim) cmd = qclip[j] + lim;
      if (cmd < qclip[j] - lim) cmd = qclip[j] - lim;
    }
    const T t = torque_cmd ? motor_torque<T>(0, 0, 0, 0, 0, 0, cmd, e.strength[j], T(s.cfg.torque_limit), true)
                : hyb ? motor_torque<T>(qm[j], qdm[j], cmd, hyb[4 * j], hyb[4 * j + 2], hyb[4 * j + 1], hyb[4 * j + 3], e.strength[j], T(s.cfg.torque_limit), false)
                      : motor_torque<T>(qm[j], qdm[j], cmd, e.kp[j], e.kd[j], T(0), T(0), e.strength[j], T(s.cfg.torque_limit), false);
    tau[j] = t;
    e.tau[j] = t;
  }
  physics_tick(s, e, tau);
  for (int j = 0; j < 12; j++) e.energy += std::fabs(tau[j] * e.qd[j]) * T(s.cfg.sim_dt);
  push_history(e);
}

template <class T> void foot_world(const Sim<T>& s, const Env<T>& e, T* fw /*12*/, T* fb /*12 base frame*/) {
  T R[3][3];
  quat_to_mat(e.quat, R);
  for (int l = 0; l < 4; l++) {
    T p[3];
    leg_fk(e.q + 3 * l, T(hip_sign(l)), p);
    for (int k = 0; k < 3; k++) p[k] += T(s.model.hip_origin[l][k]);
    T w[3];
    mat3_mul_vec(R, p, w);
    for (int k = 0; k < 3; k++) { fb[3 * l + k] = p[k]; fw[3 * l + k] = w[k] + e.pos[k]; }
  }
}

// ETG joint-space action at time t (SURVEY 8a a1-a3,a5): two phases, trot fan-out,
// IK with the 0.95 shrink guard, minus pose_ori
template <class T> void etg_action(const Sim<T>& s, const Env<T>& e, T t, T* act) {
  T r1[ETG_RBF_H], r2[ETG_RBF_H];
  s.basis.rbf(t, r1);
  s.basis.rbf(t + s.basis.T2 * s.basis.Tperiod, r2);
  T a1[3], a2[3];
  for (int k = 0; k < 3; k++) {
    T s1 = 0, s2 = 0;
    for (int h = 0; h < ETG_RBF_H; h++) { s1 += e.etg_w[k][h] * r1[h]; s2 += e.etg_w[k][h] * r2[h]; }
    a1[k] = s1 + e.etg_b[k]; a2[k] = s2 + e.etg_b[k];
  }
  for (int l = 0; l < 4; l++) {
    const T* d = (l == 0 || l == 3) ? a1 : a2;
    T scale = 1;
    T ang[3];
    for (int it = 0; it < 200; it++) {
      T foot[3];
      for (int k = 0; k < 3; k++)
        foot[k] = T(s.model.base_foot[3 * l + k]) + d[k] * scale - T(s.model.hip_origin[l][k]);
      leg_ik(foot, T(hip_sign(l)), ang);
      if (std::isfinite(ang[0]) && std::isfinite(ang[1]) && std::isfinite(ang[2])) break;
      scale *= T(0.95);
      if (it == 199) { ang[0] = T(s.model.pose_ori[0]); ang[1] = T(s.model.pose_ori[1]); ang[2] = T(s.model.pose_ori[2]); }
    }
    for (int k = 0; k < 3; k++) act[3 * l + k] = ang[k] - T(s.model.pose_ori[3 * l + k]);
  }
}

// counter-based standard normal pair (seed, robot, observation index, channel): splitmix64 finaliser twice, Box-Muller.
// Restates the generator of the product (csrc/etg_layout.h: gauss_pair) so that both draw the same noise.
static inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
template <class T> void gauss_pair(uint64_t seed, unsigned env, unsigned call, unsigned ch, T& a, T& b) {
  uint64_t z = mix64(seed + 0x9E3779B97F4A7C15ull * ((uint64_t)env * 32ull + ch + 1ull));
  z = mix64(z + call);
  const T u1 = T((z >> 40) + 1ull) / T(16777216.0);
  const T u2 = T((z >> 8) & 0xFFFFFFull) / T(16777216.0);
  const T r = std::sqrt(T(-2) * std::log(u1)), th = T(2 * M_PI) * u2;
  a = r * std::cos(th);
  b = r * std::sin(th);
}

// observation assembly (EnvWrapper.py:60-109): sorted keys
// BaseDisplacement(3) FootContactSensor(4) IMU(6) MotorAngleAcc(24) + ETG(12) = 49
template <class T> void build_obs(const Sim<T>& s, Env<T>& e, const T* etg_act, T* obs, T* imu_raw) {
  T d[HIST];
  delayed_obs(s, e, d);
  T rpy[3];
  quat_to_rpy(d + 24, rpy);
  if (!e.first_rpy_set) {
    for (int k = 0; k < 3; k++) e.first_rpy[k] = rpy[k];
    e.first_rpy_set = 1;
  }
  if (s.noise_on) {   // _AddSensorNoise: motor angles [0], velocities [1], rpy [3], rpy rate [4] (torques [2] are not observed)
    const unsigned ei = (unsigned)(&e - s.env.data());
    T n0, n1;
    for (int j = 0; j < 12; j++) {
      gauss_pair<T>(s.noise_seed, ei, s.noise_call, j, n0, n1);
      d[j] += T(s.noise_std[0]) * n0; d[12 + j] += T(s.noise_std[1]) * n1;
    }
    for (int k = 0; k < 3; k++) {
      gauss_pair<T>(s.noise_seed, ei, s.noise_call, 16 + k, n0, n1);
      rpy[k] += T(s.noise_std[3]) * n0; d[28 + k] += T(s.noise_std[4]) * n1;
    }
  }
  const bool nrm = s.cfg.obs_normal != 0;
  const T ctrl_dt = T(s.cfg.sim_dt) * T(s.cfg.action_repeat);
  int o = 0;
  for (int k = 0; k < 3; k++) {
    T disp = e.pos[k] - e.last_base[k];
    obs[o++] = nrm ? disp / ctrl_dt : disp;
  }
  for (int l = 0; l < 4; l++) obs[o++] = e.contact[l] ? T(1) : T(0);
  for (int k = 0; k < 3; k++) {
    T r = rpy[k] - e.first_rpy[k];
    imu_raw[k] = r;
    obs[o++] = nrm ? r / T(0.1) : r;
  }
  for (int k = 0; k < 3; k++) {
    imu_raw[3 + k] = d[28 + k];
    obs[o++] = nrm ? d[28 + k] / T(0.5) : d[28 + k];
  }
  for (int j = 0; j < 12; j++) {
    T a = d[j];
    // MapToMinusPiToPi (minitaur.py:67-83)
    a = std::fmod(a, T(2 * M_PI));
    if (a >= T(M_PI)) a -= T(2 * M_PI);
    else if (a < T(-M_PI)) a += T(2 * M_PI);
    obs[o++] = nrm ? (a - T(s.model.pose_ori[j])) / T(0.1) : a;
  }
  for (int j = 0; j < 12; j++) obs[o++] = d[12 + j];
  for (int j = 0; j < 12; j++)
    obs[o++] = nrm ? (etg_act[j] - T(s.model.etg_mean[j])) / T(s.model.etg_std[j]) : etg_act[j];
}

template <class T> void reset_env(Sim<T>& s, Env<T>& e, T* obs) {
  const EtgRobotModel& m = s.model;
  for (int k = 0; k < 3; k++) { e.pos[k] = T(m.init_pos[k]); e.wb[k] = 0; e.vb[k] = 0; }
  e.pos[0] += e.reset_off[0]; e.pos[1] += e.reset_off[1];
  e.quat[0] = e.quat[1] = e.quat[2] = 0; e.quat[3] = 1;
  for (int j = 0; j < 12; j++) { e.q[j] = T(m.pose_ori[j]); e.qd[j] = 0; e.lam[j] = 0; e.tau[j] = 0; }
  for (int l = 0; l < 4; l++) { e.contact[l] = 0; e.lamb[l] = 0; }
  e.tick = 0; e.energy = 0;
  std::memset(e.hist, 0, sizeof(e.hist));
  // ReceiveObservation before settling (a1.py:290): fill the whole ring with the
  // initial reading so that the delayed observation is defined from tick 0
  e.tick = -1;
  push_history(e);
  for (int r = 1; r < RING; r++) std::memcpy(e.hist[r], e.hist[0], sizeof(T) * HIST);
  T qdes[12];
  for (int j = 0; j < 12; j++) qdes[j] = T(m.pose_ori[j]);
  T fsave[3];   // the settle runs without the external force
  for (int k = 0; k < 3; k++) { fsave[k] = e.fext[k]; e.fext[k] = 0; }
  for (int i = 0; i < s.cfg.settle_ticks; i++) sub_step(s, e, qdes);  // a1.py:294-297
  for (int k = 0; k < 3; k++) e.fext[k] = fsave[k];
  e.step_count = 0; e.has_last = 0; e.first_rpy_set = 0; e.energy = 0;
  for (int j = 0; j < 12; j++) {
    e.last_qdes[j] = qdes[j];
    for (int k = 0; k < 2; k++) { e.fx[k][j] = qdes[j]; e.fy[k][j] = qdes[j]; }  // init_history, action_filter.py:122-126
  }
  for (int k = 0; k < 3; k++) e.last_base[k] = e.pos[k];
  T fb[12];
  foot_world(s, e, e.last_foot_w, fb);
  T act[12], imu[6];
  if (s.cfg.enable_etg) etg_action(s, e, T(0), act);
  else for (int j = 0; j < 12; j++) act[j] = 0;
  build_obs(s, e, act, obs, imu);
}

// reward pieces (own definitions, DESIGN.md "reward"): c_prec saturating kernel
template <class T> inline T c_prec(T v, T t, T m) {
  T w = std::atanh(std::sqrt(T(0.95))) / m;
  T x = (v - t) * w;
  return std::tanh(x * x);
}

template <class T>
void step_env(Sim<T>& s, Env<T>& e, const T* action, int donef, T* obs, T* reward, uint8_t* done, T* info) {
  const EtgRobotModel& m = s.model;
  const int R_ = s.cfg.action_repeat;
  const T ctrl_dt = T(s.cfg.sim_dt) * T(R_);
  const long long sweeps_before = e.sweeps_total;
  // ETG at t = (k+1) dt  (fixture convention of gait_action_list_ETG_exp.npy)
  T t = T(e.step_count + 1) * T(s.cfg.etg_dt);
  T etg[12], qdes[12];
  if (s.cfg.enable_etg) etg_action(s, e, t, etg);
  else for (int j = 0; j < 12; j++) etg[j] = 0;     // `ETG=0` (Dynamic_parallel_model.py:49): command = pose_ori + action
  const bool torque_cmd = s.cfg.motor_mode == 1;
  const bool hybrid_cmd = s.cfg.motor_mode == 2;   // action row = 12 x (q_des, kp, qd_des, kd, tau_ff)
  T hyb[48];
  if (hybrid_cmd)
    for (int j = 0; j < 12; j++)
      for (int k = 0; k < 4; k++) hyb[4 * j + k] = action[5 * j + 1 + k];
  for (int j = 0; j < 12; j++)
    qdes[j] = hybrid_cmd ? action[5 * j] : torque_cmd ? action[j] : T(m.pose_ori[j]) + etg[j] + action[j];
  if (s.cfg.enable_action_filter) {  // action_filter.py:111-120 (order 2)
    for (int j = 0; j < 12; j++) {
      T y = T(s.cfg.filter_b[0]) * qdes[j] + T(s.cfg.filter_b[1]) * e.fx[0][j] + T(s.cfg.filter_b[2]) * e.fx[1][j] -
            T(s.cfg.filter_a[1]) * e.fy[0][j] - T(s.cfg.filter_a[2]) * e.fy[1][j];
      e.fx[1][j] = e.fx[0][j]; e.fx[0][j] = qdes[j];
      e.fy[1][j] = e.fy[0][j]; e.fy[0][j] = y;
      qdes[j] = y;
    }
  }
  e.energy = 0;
  for (int i = 0; i < R_; i++) {  // minitaur.py:254-258
    T proc[12];
    if (s.cfg.enable_action_interp && e.has_last) {
      T lerp = T(i + 1) / T(R_);
      for (int j = 0; j < 12; j++) proc[j] = e.last_qdes[j] + lerp * (qdes[j] - e.last_qdes[j]);
    } else {
      for (int j = 0; j < 12; j++) proc[j] = qdes[j];
    }
    sub_step(s, e, proc, torque_cmd, hybrid_cmd ? hyb : (const T*)nullptr);
  }
  for (int j = 0; j < 12; j++) e.last_qdes[j] = qdes[j];
  e.has_last = 1;
  e.step_count++;

  T imu[6];
  build_obs(s, e, etg, obs, imu);

  // ---- reward / termination (this repo's definitions; rlschool's are absent)
  T Rm[3][3];
  quat_to_mat(e.quat, Rm);
  T rpy[3];
  quat_to_rpy(e.quat, rpy);
  T fw[12], fb[12];
  foot_world(s, e, fw, fb);
  T dx = e.pos[0] - e.last_base[0];
  T vx = dx / ctrl_dt;
  T torso = vx < T(s.cfg.vel_d) ? vx : T(s.cfg.vel_d);
  T up = (T(1) - c_prec(rpy[0], T(0), T(0.5))) * (T(1) - c_prec(rpy[1], T(0), T(0.5)));
  T feet = 0;
  for (int l = 0; l < 4; l++) feet += (fw[3 * l] - e.last_foot_w[3 * l]) * T(0.25);
  feet = feet / ctrl_dt;
  feet = feet < T(s.cfg.vel_d) ? feet : T(s.cfg.vel_d);
  T tau_r = -e.energy;
  int lost = 0, bad = 0;
  for (int l = 0; l < 4; l++) {
    lost += e.contact[l] ? 0 : 1;
    // knee height: thigh-calf joint in world
    T ang[3] = {e.q[3 * l], e.q[3 * l + 1], 0};
    T cs = std::cos(ang[0]), sn = std::sin(ang[0]);
    T ly = T(m.thigh_y[l]);
    T kx = -T(m.upper_len) * std::sin(ang[1]);
    T kzh = -T(m.upper_len) * std::cos(ang[1]);
    T kb[3] = {T(m.hip_origin[l][0]) + kx, T(m.hip_origin[l][1]) + cs * ly - sn * kzh,
               T(m.hip_origin[l][2]) + sn * ly + cs * kzh};
    T kw[3];
    mat3_mul_vec(Rm, kb, kw);
    T hgt, nrm[3];
    terrain_query(s, e.band, kw[0] + e.pos[0], kw[1] + e.pos[1], &hgt, nrm);
    if (kw[2] + e.pos[2] - hgt < T(0.03)) bad++;
  }
  T badfoot = -T(bad);
  T footcontact = -T(lost - 2 > 0 ? lost - 2 : 0);
  T footz_mean = (fb[2] + fb[5] + fb[8] + fb[11]) * T(0.25);
  T footz_max = fb[2];
  for (int l = 1; l < 4; l++) footz_max = fb[3 * l + 2] > footz_max ? fb[3 * l + 2] : footz_max;
  bool finite = std::isfinite(e.pos[0]) && std::isfinite(e.pos[2]) && std::isfinite(e.q[0]);
  bool term = Rm[2][2] < T(0.5) || footz_mean > T(-0.1) || footz_max > 0 || std::fabs(rpy[2]) > T(0.6) || !finite;
  const double* w = s.cfg.reward_w;
  T terms[8] = {T(w[0]) * torso, T(w[1]) * feet, T(w[2]) * up, T(w[3]) * tau_r, T(0),
                T(w[5]) * badfoot, T(w[6]) * footcontact, T(w[7]) * (term ? T(-1) : T(0))};
  T sum = 0;
  for (int k = 0; k < 8; k++) sum += terms[k];
  *reward = T(s.cfg.reward_p) * sum;
  *done = (term || donef) ? 1 : 0;
  if (info) {
    for (int k = 0; k < ETG_INFO_DIM; k++) info[k] = 0;
    for (int k = 0; k < 8; k++) info[k] = terms[k];
    info[ETG_INFO_VELX] = vx;
    for (int j = 0; j < 12; j++) {
      info[ETG_INFO_ETG_ACT + j] = etg[j];
      info[ETG_INFO_JOINT_ANGLE + j] = e.q[j];
      info[ETG_INFO_REAL_ACTION + j] = qdes[j];
    }
    for (int k = 0; k < 6; k++) info[ETG_INFO_OBS_IMU + k] = imu[k];
    for (int l = 0; l < 4; l++) info[ETG_INFO_FOOT_CONTACT + l] = e.contact[l] ? T(1) : T(0);
    for (int k = 0; k < 3; k++) { info[ETG_INFO_BASE + k] = e.pos[k]; info[ETG_INFO_RPY + k] = rpy[k]; }
    info[ETG_INFO_ENERGY] = e.energy;
    info[ETG_INFO_STEPS] = T(e.step_count);
    info[ETG_INFO_SWEEPS] = T(e.sweeps_total - sweeps_before);   // this robot's own sweeps (the kernels report the wave's)
  }
  for (int k = 0; k < 3; k++) e.last_base[k] = e.pos[k];
  for (int k = 0; k < 12; k++) e.last_foot_w[k] = fw[k];
}

template <class T> void get_state(const Env<T>& e, T* st) {
  T R[3][3];
  quat_to_mat(e.quat, R);
  T vw[3], ww[3];
  mat3_mul_vec(R, e.vb, vw);
  mat3_mul_vec(R, e.wb, ww);
  for (int k = 0; k < 3; k++) { st[k] = e.pos[k]; st[7 + k] = vw[k]; st[10 + k] = ww[k]; }
  for (int k = 0; k < 4; k++) st[3 + k] = e.quat[k];
  for (int j = 0; j < 12; j++) { st[13 + j] = e.q[j]; st[25 + j] = e.qd[j]; }
}
template <class T> void set_state(Env<T>& e, const T* st) {
  for (int k = 0; k < 3; k++) e.pos[k] = st[k];
  T n = 0;
  for (int k = 0; k < 4; k++) n += st[3 + k] * st[3 + k];
  n = T(1) / std::sqrt(n);
  for (int k = 0; k < 4; k++) e.quat[k] = st[3 + k] * n;
  T R[3][3];
  quat_to_mat(e.quat, R);
  mat3T_mul_vec(R, st + 7, e.vb);
  mat3T_mul_vec(R, st + 10, e.wb);
  for (int j = 0; j < 12; j++) { e.q[j] = st[13 + j]; e.qd[j] = st[25 + j]; e.lam[j] = 0; }
  for (int l = 0; l < 4; l++) e.lamb[l] = 0;
  // re-seed the latency ring with the new reading
  e.tick = -1;
  push_history(e);
  for (int r = 1; r < RING; r++) std::memcpy(e.hist[r], e.hist[0], sizeof(T) * HIST);
  for (int k = 0; k < 3; k++) e.last_base[k] = e.pos[k];
}

template <class F> void par_for(int n, int threads, F f) {
  if (threads <= 1 || n < 2) {
    for (int i = 0; i < n; i++) f(i);
    return;
  }
  std::vector<std::thread> th;
  int per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; t++) {
    int a = t * per, b = a + per < n ? a + per : n;
    if (a >= b) break;
    th.emplace_back([=]() { for (int i = a; i < b; i++) f(i); });
  }
  for (auto& x : th) x.join();
}
}  // namespace

// ---------------------------------------------------------------- C ABI
#define DEFINE_API(SFX, T)                                                                         \
  extern "C" void* etgo_create##SFX(const EtgConfig* cfg, const EtgRobotModel* model) {            \
    auto* s = new Sim<T>();                                                                         \
    s->cfg = *cfg; s->model = *model; s->N = cfg->num_envs;                                         \
    s->basis.init(*cfg);                                                                            \
    build_tree(*s);                                                                                 \
    s->env.resize(s->N);                                                                            \
    for (int i = 0; i < s->N; i++) {                                                                \
      auto& e = s->env[i];                                                                          \
      std::memset((void*)&e, 0, sizeof(e));                                                         \
      e.quat[3] = 1;                                                                                \
      for (int j = 0; j < 12; j++) e.strength[j] = 1;                                               \
      e.band = i % (cfg->hf_bands > 1 ? cfg->hf_bands : 1);                                         \
    }                                                                                               \
    return s;                                                                                       \
  }                                                                                                 \
  extern "C" void etgo_destroy##SFX(void* h) { delete (Sim<T>*)h; }                                 \
  extern "C" void etgo_set_params##SFX(void* h, const T* dyn, const T* w, const T* b, int per_env,  \
                                        const uint8_t* mask) {                                      \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++) {                                                                \
      if (mask && !mask[i]) continue;                                                               \
      Env<T>& e = s->env[i];                                                                        \
      if (dyn) { std::memcpy(e.dyn, dyn + (size_t)i * ETG_DYN_DIM, sizeof(T) * ETG_DYN_DIM); derive_params(*s, e); } \
      if (w) std::memcpy(e.etg_w, w + (per_env ? (size_t)i * 3 * ETG_RBF_H : 0), sizeof(T) * 3 * ETG_RBF_H); \
      if (b) std::memcpy(e.etg_b, b + (per_env ? (size_t)i * 3 : 0), sizeof(T) * 3);               \
    }                                                                                               \
  }                                                                                                 \
  extern "C" void etgo_set_external_force##SFX(void* h, const T* force) {                           \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++)                                                                  \
      for (int k = 0; k < 3; k++) s->env[i].fext[k] = force ? force[(size_t)i * 3 + k] : T(0);      \
  }                                                                                                 \
  extern "C" void etgo_set_motor_strength##SFX(void* h, const T* ratios, const uint8_t* mask) {     \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++) {                                                                \
      if (mask && !mask[i]) continue;                                                               \
      for (int j = 0; j < 12; j++) s->env[i].strength[j] = ratios ? ratios[(size_t)i * 12 + j] : T(1); \
    }                                                                                               \
  }                                                                                                 \
  /* the motor model alone (golden tests): POSITION law with strength ratios and a torque limit */ \
  extern "C" void etgo_motor_torque##SFX(const T* qdes, const T* q, const T* qd, const T* kp, const T* kd,  \
                                          const T* strength, T limit, int torque_mode, int n, T* tau) {      \
    for (int j = 0; j < n; j++)                                                                     \
      tau[j] = torque_mode ? motor_torque<T>(0, 0, 0, 0, 0, 0, qdes[j], strength[j], limit, true)   \
                           : motor_torque<T>(q[j], qd[j], qdes[j], kp[j], kd[j], T(0), T(0), strength[j], limit, false); \
  }                                                                                                 \
  extern "C" void etgo_set_reset_offsets##SFX(void* h, const T* xy, const uint8_t* mask) {           \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++) {                                                                \
      if (mask && !mask[i]) continue;                                                               \
      for (int k = 0; k < 2; k++) s->env[i].reset_off[k] = xy ? xy[(size_t)i * 2 + k] : T(0);       \
    }                                                                                               \
  }                                                                                                 \
  extern "C" void etgo_set_heightfield##SFX(void* h, const float* hts) {                            \
    auto* s = (Sim<T>*)h;                                                                           \
    s->heights.assign(hts, hts + (size_t)s->cfg.hf_nx * s->cfg.hf_ny);                              \
  }                                                                                                 \
  extern "C" void etgo_reset##SFX(void* h, const uint8_t* mask, T* obs, int threads) {              \
    auto* s = (Sim<T>*)h;                                                                           \
    s->noise_call = s->obs_calls++;                                                                 \
    par_for(s->N, threads, [=](int i) {                                                             \
      if (mask && !mask[i]) return;                                                                 \
      reset_env(*s, s->env[i], obs + (size_t)i * ETG_OBS_DIM);                                      \
    });                                                                                             \
  }                                                                                                 \
  extern "C" void etgo_set_sensor_noise##SFX(void* h, const float* stdev, uint64_t seed) {         \
    auto* s = (Sim<T>*)h;                                                                           \
    s->noise_on = 0;                                                                                \
    for (int k = 0; k < 5; k++) {                                                                   \
      s->noise_std[k] = stdev ? stdev[k] : 0.0f;                                                    \
      if (s->noise_std[k] > 0.0f) s->noise_on = 1;                                                  \
    }                                                                                               \
    s->noise_seed = seed;                                                                           \
  }                                                                                                 \
  extern "C" void etgo_step##SFX(void* h, const T* action, const uint8_t* donef, T* obs, T* reward, \
                                  uint8_t* done, T* info, int threads) {                            \
    auto* s = (Sim<T>*)h;                                                                           \
    s->noise_call = s->obs_calls++;                                                                 \
    par_for(s->N, threads, [=](int i) {                                                             \
      step_env(*s, s->env[i], action + (size_t)i * (s->cfg.motor_mode == 2 ? ETG_HYBRID_DIM : 12), donef ? donef[i] : 0,                        \
               obs + (size_t)i * ETG_OBS_DIM, reward + i, done + i,                                 \
               info ? info + (size_t)i * ETG_INFO_DIM : (T*)nullptr);                               \
    });                                                                                             \
  }                                                                                                 \
  /* the control step of robots [env0, env0 + count) only (arrays are the whole batch's): include/etgsim.h etg_step_range; \
   * the sensor-noise stream moves on with the range that starts at robot 0                                             */ \
  extern "C" void etgo_step_range##SFX(void* h, int env0, int count, const T* action, const uint8_t* donef, T* obs,   \
                                        T* reward, uint8_t* done, T* info, int threads) {           \
    auto* s = (Sim<T>*)h;                                                                           \
    if (env0 == 0) s->noise_call = s->obs_calls++;                                                  \
    par_for(count, threads, [=](int k) {                                                            \
      const int i = env0 + k;                                                                       \
      step_env(*s, s->env[i], action + (size_t)i * (s->cfg.motor_mode == 2 ? ETG_HYBRID_DIM : 12), donef ? donef[i] : 0, \
               obs + (size_t)i * ETG_OBS_DIM, reward + i, done + i,                                 \
               info ? info + (size_t)i * ETG_INFO_DIM : (T*)nullptr);                               \
    });                                                                                             \
  }                                                                                                 \
  /* nsteps control steps with a constant action row set (action [N,12] or NULL = zeros), every thread running ITS   \
   * slice of the robots through all the steps (no per-step thread spawn / join): the all-core CPU baseline of       \
   * bench.py.  ret / len [N]: episode return and length with alive masking (frozen after the first done).          \
   * stop_at_done: the loop of a robot ends with its episode, as the reference's do (pretrain.py:137-153,             \
   * train.py:226-247) -- its state stays the terminal state; 0: finished robots are stepped on (accumulators masked).\
   * A robot that was finished before the call (alive_in[i] == 0, NULL = all alive) is not stepped under stop_at_done; \
   * alive_in receives the flags after the call.                                                                      \
   * Sensor noise is not drawn here (the counter-based stream is per call).                                          */ \
  extern "C" void etgo_run_steps##SFX(void* h, const T* action, int nsteps, int threads, T* ret, int32_t* len,       \
                                       int stop_at_done, uint8_t* alive_in /* in / out */, T* obs_out) {              \
    auto* s = (Sim<T>*)h;                                                                           \
    const int adim = s->cfg.motor_mode == 2 ? ETG_HYBRID_DIM : 12;                                  \
    par_for(s->N, threads, [=](int i) {                                                             \
      std::vector<T> zero(adim, T(0));                                                              \
      T obs[ETG_OBS_DIM], r, acc = 0;                                                               \
      uint8_t d;                                                                                    \
      int alive = alive_in ? (int)alive_in[i] : 1, n = 0;                                           \
      for (int k = 0; k < nsteps; k++) {                                                            \
        if (stop_at_done && !alive) break;                                                          \
        step_env(*s, s->env[i], action ? action + (size_t)i * adim : zero.data(), 0, obs, &r, &d, (T*)nullptr); \
        if (obs_out) std::memcpy(obs_out + (size_t)i * ETG_OBS_DIM, obs, sizeof(obs));              \
        if (alive) { acc += r; n++; }                                                               \
        if (d) alive = 0;                                                                           \
      }                                                                                             \
      if (ret) ret[i] = acc;                                                                        \
      if (len) len[i] = n;                                                                          \
      if (alive_in) alive_in[i] = (uint8_t)alive;                                                   \
    });                                                                                             \
  }                                                                                                 \
  extern "C" void etgo_get_state##SFX(void* h, T* st) {                                             \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++) get_state(s->env[i], st + (size_t)i * ETG_STATE_DIM);            \
  }                                                                                                 \
  extern "C" void etgo_set_state##SFX(void* h, const T* st) {                                       \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++) set_state(s->env[i], st + (size_t)i * ETG_STATE_DIM);            \
  }                                                                                                 \
  /* raw physics ticks with given torques (for invariants tests) */                                 \
  extern "C" void etgo_tick##SFX(void* h, const T* tau, int nticks) {                               \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++)                                                                  \
      for (int k = 0; k < nticks; k++) physics_tick(*s, s->env[i], tau + (size_t)i * 12);           \
  }                                                                                                 \
  /* mass matrix / bias taps of env 0 at its current state (one throw-away tick on a copy) */       \
  extern "C" void etgo_dynamics_terms##SFX(void* h, int env, T* M, T* C) {                           \
    auto* s = (Sim<T>*)h;                                                                           \
    Env<T> copy = s->env[env];                                                                      \
    T tau[12] = {0};                                                                                \
    s->dbgM = M; s->dbgC = C;                                                                       \
    physics_tick(*s, copy, tau);                                                                    \
    s->dbgM = nullptr; s->dbgC = nullptr;                                                           \
  }                                                                                                 \
  /* out[64]: ticks by PGS sweep count, summed over the robots (mask NULL = all); clear != 0 zeroes the counters */   \
  extern "C" void etgo_sweep_hist##SFX(void* h, long long* out, const uint8_t* mask, int clear) {    \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int k = 0; k < 64; k++) out[k] = 0;                                                        \
    for (int i = 0; i < s->N; i++) {                                                                \
      if (mask && !mask[i]) continue;                                                               \
      for (int k = 0; k < 64; k++) { out[k] += s->env[i].sweep_hist[k]; if (clear) s->env[i].sweep_hist[k] = 0; } \
    }                                                                                               \
  }                                                                                                 \
  /* out[N][3]: per robot, ticks with a body row inside the margin / with a loaded body row / all ticks; clear != 0 zeroes them */ \
  extern "C" void etgo_body_stats##SFX(void* h, long long* out, int clear) {                        \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++)                                                                  \
      for (int k = 0; k < 3; k++) { out[3 * i + k] = s->env[i].body_ticks[k]; if (clear) s->env[i].body_ticks[k] = 0; } \
  }                                                                                                 \
  /* per-tick trace of ONE robot into the caller's buffer [cap_ticks][TRACE_W] doubles (NULL switches it off); returns    \
   * the number of ticks recorded so far (reset to 0 by every call that installs a buffer)                               */ \
  extern "C" int etgo_set_trace##SFX(void* h, int env, double* buf, int cap_ticks) {                \
    auto* s = (Sim<T>*)h;                                                                           \
    if (env < 0 || env >= s->N) return -1;                                                          \
    const int n = s->env[env].trace_n;                                                              \
    if (cap_ticks >= 0) { s->env[env].trace = buf; s->env[env].trace_cap = buf ? cap_ticks : 0; s->env[env].trace_n = 0; } \
    return n;                                                                                       \
  }                                                                                                 \
  extern "C" void etgo_set_solve_noise##SFX(void* h, double rel, uint64_t seed) {                    \
    auto* s = (Sim<T>*)h;                                                                           \
    s->solve_noise = rel; s->solve_noise_seed = seed;                                               \
  }                                                                                                 \
  /* the solver's warm start [N,4,4]: per leg the foot's (n, t1, t2) and the body contact's normal impulse of the last tick */ \
  extern "C" void etgo_get_lambda##SFX(void* h, T* lam) {                                           \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++)                                                                  \
      for (int l = 0; l < 4; l++) {                                                                 \
        for (int k = 0; k < 3; k++) lam[(size_t)i * 16 + 4 * l + k] = s->env[i].lam[3 * l + k];     \
        lam[(size_t)i * 16 + 4 * l + 3] = s->env[i].lamb[l];                                        \
      }                                                                                             \
  }                                                                                                 \
  extern "C" void etgo_set_lambda##SFX(void* h, const T* lam) {                                     \
    auto* s = (Sim<T>*)h;                                                                           \
    for (int i = 0; i < s->N; i++)                                                                  \
      for (int l = 0; l < 4; l++) {                                                                 \
        for (int k = 0; k < 3; k++) s->env[i].lam[3 * l + k] = lam[(size_t)i * 16 + 4 * l + k];     \
        s->env[i].lamb[l] = lam[(size_t)i * 16 + 4 * l + 3];                                        \
      }                                                                                             \
  }                                                                                                 \
  extern "C" void etgo_etg_rbf##SFX(void* h, T t, T* r) { ((Sim<T>*)h)->basis.rbf(t, r); }         \
  extern "C" void etgo_etg_action##SFX(void* h, int env, T t, T* act) {                             \
    auto* s = (Sim<T>*)h;                                                                           \
    etg_action(*s, s->env[env], t, act);                                                            \
  }                                                                                                 \
  extern "C" void etgo_leg_ik##SFX(const T* foot, T sign, T* ang) { leg_ik(foot, sign, ang); }      \
  extern "C" void etgo_leg_fk##SFX(const T* ang, T sign, T* p) { leg_fk(ang, sign, p); }            \
  extern "C" void etgo_leg_jacobian##SFX(const T* ang, int leg, T* J) {                             \
    T Jm[3][3];                                                                                     \
    leg_jacobian(ang, leg, Jm);                                                                     \
    std::memcpy(J, Jm, sizeof(Jm));                                                                 \
  }                                                                                                 \
  extern "C" void etgo_pd_torque##SFX(const T* qdes, const T* q, const T* qd, const T* kp,          \
                                       const T* kd, int n, T* tau) {                                \
    for (int j = 0; j < n; j++) tau[j] = -(kp[j] * (q[j] - qdes[j])) - kd[j] * qd[j];               \
  }                                                                                                 \
  /* HYBRID command (laikago_motor.py:152-167): cmd = 12 x (q_des, kp, qd_des, kd, tau_ff) */      \
  extern "C" void etgo_pd_torque_hybrid##SFX(const T* cmd, const T* q, const T* qd, int n, T* tau) { \
    for (int j = 0; j < n; j++)                                                                     \
      tau[j] = (-(cmd[5 * j + 1] * (q[j] - cmd[5 * j])) - cmd[5 * j + 3] * (qd[j] - cmd[5 * j + 2])) + cmd[5 * j + 4]; \
  }                                                                                                 \
  /* policy forward: model/mujoco_model.py:53-57 + alg/sac.py:60-63 */                              \
  extern "C" void etgo_mlp_forward##SFX(const T* obs, int n, int in_dim, int hid, int out_dim,      \
                                         const T* w1, const T* b1, const T* w2, const T* b2,        \
                                         const T* w3, const T* b3, T scale, T* act) {               \
    std::vector<T> h1(hid), h2(hid);                                                                \
    for (int i = 0; i < n; i++) {                                                                   \
      const T* o = obs + (size_t)i * in_dim;                                                        \
      for (int a = 0; a < hid; a++) {                                                               \
        T sum = b1[a];                                                                              \
        for (int k = 0; k < in_dim; k++) sum += w1[(size_t)a * in_dim + k] * o[k];                  \
        h1[a] = sum > 0 ? sum : 0;                                                                  \
      }                                                                                             \
      for (int a = 0; a < hid; a++) {                                                               \
        T sum = b2[a];                                                                              \
        for (int k = 0; k < hid; k++) sum += w2[(size_t)a * hid + k] * h1[k];                       \
        h2[a] = sum > 0 ? sum : 0;                                                                  \
      }                                                                                             \
      for (int a = 0; a < out_dim; a++) {                                                           \
        T sum = b3[a];                                                                              \
        for (int k = 0; k < hid; k++) sum += w3[(size_t)a * hid + k] * h2[k];                       \
        act[(size_t)i * out_dim + a] = std::tanh(sum) * scale;                                      \
      }                                                                                             \
    }                                                                                               \
  }

DEFINE_API(64, double)
DEFINE_API(32, float)

extern "C" int etgo_version(void) { return 1; }
