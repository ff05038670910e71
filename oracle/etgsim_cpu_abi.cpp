/*
 * etgsim_cpu_abi.cpp -- the C-ABI of include/etgsim.h served by the CPU oracle: the "CPU build of the same ABI
 * (device = -1)" that SURVEY.md 8(b) / BASELINE.md 2.1 name as the plumbing path of BASELINE configs[0]
 * ("Single A1, flat terrain, ETG open-loop gait, CPU env.step()") and as the CPU baseline.
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: it is built into oracle/libetgsim_cpu.so, which only
 * tests/ and bench.py's cpu_baseline leg load.  The product library (paddlerobotics_amd/csrc/libetgsim.so) has no
 * CPU path and refuses device < 0.
 *
 * Same entry points, same argument meaning and error codes as the HIP library, with HOST pointers (float32, the
 * header's types) instead of device pointers and the stream argument ignored; the arithmetic inside is the fp64
 * oracle (etgsim_oracle.cpp), converted at the boundary.  etg_create() accepts device = -1 only.  Entry points that
 * only exist as device kernels (random pushes, fused policy rollout, ETG fit, extra sensors, leg kinematics) return
 * ETG_ERR_STATE here.
 */
#include "etgsim_oracle.cpp"

#include <string>

namespace {
thread_local std::string g_cpu_err;
int cfail(int code, const char* msg) { g_cpu_err = msg; return code; }

struct CpuHandle {
  Sim<double>* sim;
  int N;
  bool was_reset;
  std::vector<double> ret, alive;
  std::vector<int32_t> len;
  std::vector<double> dbuf;   // conversion scratch
  // etg_prepare_next_dynamics, sequential restatement: rows waiting for each robot's next episode (installed by its next reset;
  // the settle is simulated then -- on the device it ran ahead of time on scratch state, with the same result)
  std::vector<float> next_dyn;
  std::vector<uint8_t> next_ok;
  std::vector<int> age_ticks;   // physics ticks since each robot's reset (etg_prepare_next_dynamics leaves young robots out)
  bool simulate_finished = false;   // etg_set_rollout_mode
};
CpuHandle* H(EtgHandle* h) { return reinterpret_cast<CpuHandle*>(h); }

std::vector<double> to_d(const float* p, size_t n) { return std::vector<double>(p, p + n); }
}  // namespace

extern "C" {

const char* etg_last_error(void) { return g_cpu_err.c_str(); }
int etg_version(void) { return 2; }
int etg_config_size(void) { return (int)sizeof(EtgConfig); }
int etg_model_size(void) { return (int)sizeof(EtgRobotModel); }
int etg_lanes_per_robot(const EtgHandle* h) { return h ? 0 : ETG_ERR_BAD_ARG; }   /* no lane mapping on the CPU */

int etg_create(const EtgConfig* cfg, const EtgRobotModel* model, int device, EtgHandle** out) {
  if (!cfg || !model || !out) return cfail(ETG_ERR_BAD_ARG, "etg_create: null argument");
  if (device != -1) return cfail(ETG_ERR_BAD_ARG, "etg_create: the CPU build of the ABI serves device = -1 only");
  if (cfg->num_envs <= 0 || cfg->action_repeat <= 0 || cfg->sim_dt <= 0) return cfail(ETG_ERR_BAD_ARG, "etg_create: bad config");
  if (!(cfg->pd_latency >= 0) || cfg->pd_latency >= 62 * cfg->sim_dt)   /* ring depth 64: the same bound as the device library */
    return cfail(ETG_ERR_BAD_ARG, "etg_create: pd_latency must be in [0, 62 ticks)");
  auto* h = new CpuHandle();
  h->sim = (Sim<double>*)etgo_create64(cfg, model);
  h->N = cfg->num_envs;
  h->was_reset = false;
  h->ret.assign(h->N, 0.0); h->alive.assign(h->N, 1.0); h->len.assign(h->N, 0);
  h->age_ticks.assign(h->N, 0);
  /* default physical parameters = param2dynamic_dict(zeros(48)) (train.py:112-126), like the HIP library */
  std::vector<double> dyn((size_t)h->N * ETG_DYN_DIM, 1.0);
  for (int i = 0; i < h->N; i++) {
    double* r = dyn.data() + (size_t)i * ETG_DYN_DIM;
    r[0] = 40.0; r[1] = 0.2; r[2] = 1.5;
    for (int j = 0; j < 12; j++) { r[21 + j] = 80.0; r[33 + j] = (j % 3 == 0) ? 1.0 : 2.0; }
    r[45] = 0.0; r[46] = 0.0; r[47] = -10.0;
  }
  etgo_set_params64(h->sim, dyn.data(), nullptr, nullptr, 0, nullptr);
  *out = reinterpret_cast<EtgHandle*>(h);
  return ETG_OK;
}

void etg_destroy(EtgHandle* h) {
  if (!h) return;
  etgo_destroy64(H(h)->sim);
  delete H(h);
}

int etg_set_params(EtgHandle* h, const float* dyn, const float* etg_w, const float* etg_b, int per_env, const uint8_t* mask, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if ((etg_w == nullptr) != (etg_b == nullptr)) return cfail(ETG_ERR_BAD_ARG, "etg_set_params: pass both etg_w and etg_b or neither");
  const size_t N = H(h)->N;
  std::vector<double> d, w, b;
  if (dyn) d = to_d(dyn, N * ETG_DYN_DIM);
  if (etg_w) { w = to_d(etg_w, (per_env ? N : 1) * 3 * ETG_RBF_H); b = to_d(etg_b, (per_env ? N : 1) * 3); }
  etgo_set_params64(H(h)->sim, dyn ? d.data() : nullptr, etg_w ? w.data() : nullptr, etg_w ? b.data() : nullptr, per_env, mask);
  if (dyn && !H(h)->next_ok.empty())     // new rows for a robot drop the ones waiting for its next episode
    for (size_t i = 0; i < N; i++)
      if (!mask || mask[i]) H(h)->next_ok[i] = 0;
  return ETG_OK;
}

int etg_set_heightfield(EtgHandle* h, const float* heights, void*) {
  if (!h || !heights) return cfail(ETG_ERR_BAD_ARG, "etg_set_heightfield: null argument");
  if (H(h)->sim->cfg.terrain != 1) return cfail(ETG_ERR_STATE, "etg_set_heightfield: config has no heightfield");
  etgo_set_heightfield64(H(h)->sim, heights);
  return ETG_OK;
}

int etg_set_external_force(EtgHandle* h, const float* force, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (!force) { etgo_set_external_force64(H(h)->sim, nullptr); return ETG_OK; }
  std::vector<double> f = to_d(force, (size_t)H(h)->N * 3);
  etgo_set_external_force64(H(h)->sim, f.data());
  return ETG_OK;
}

int etg_set_motor_strength(EtgHandle* h, const float* ratios, const uint8_t* mask, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (!ratios) { etgo_set_motor_strength64(H(h)->sim, nullptr, mask); return ETG_OK; }
  std::vector<double> r = to_d(ratios, (size_t)H(h)->N * ETG_NUM_MOTORS);
  etgo_set_motor_strength64(H(h)->sim, r.data(), mask);
  return ETG_OK;
}

int etg_set_sensor_noise(EtgHandle* h, const float* stdev, uint64_t seed) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  etgo_set_sensor_noise64(H(h)->sim, stdev, seed);
  return ETG_OK;
}

int etg_set_reset_offsets(EtgHandle* h, const float* xy, const uint8_t* mask, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (!xy) { etgo_set_reset_offsets64(H(h)->sim, nullptr, mask); return ETG_OK; }
  std::vector<double> o = to_d(xy, (size_t)H(h)->N * 2);
  etgo_set_reset_offsets64(H(h)->sim, o.data(), mask);
  return ETG_OK;
}

int etg_reset(EtgHandle* h, const uint8_t* mask, float* obs, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (!obs) return cfail(ETG_ERR_BAD_ARG, "etg_reset: obs is null");
  CpuHandle* c = H(h);
  if (!c->next_ok.empty()) {   // rows prepared for the next episode of the robots being reset: this reset starts it
    std::vector<uint8_t> take(c->N, 0);
    bool any = false;
    for (int i = 0; i < c->N; i++)
      if ((!mask || mask[i]) && c->next_ok[i]) { take[i] = 1; c->next_ok[i] = 0; any = true; }
    if (any) {
      std::vector<double> d = to_d(c->next_dyn.data(), (size_t)c->N * ETG_DYN_DIM);
      etgo_set_params64(c->sim, d.data(), nullptr, nullptr, 0, take.data());
    }
  }
  std::vector<double> o((size_t)c->N * ETG_OBS_DIM, 0.0);
  etgo_reset64(c->sim, mask, o.data(), 1);
  for (int i = 0; i < c->N; i++) {
    if (mask && !mask[i]) continue;
    for (int k = 0; k < ETG_OBS_DIM; k++) obs[(size_t)i * ETG_OBS_DIM + k] = (float)o[(size_t)i * ETG_OBS_DIM + k];
    c->ret[i] = 0.0; c->alive[i] = 1.0; c->len[i] = 0;
    c->age_ticks[i] = 0;
  }
  c->was_reset = true;
  return ETG_OK;
}

int etg_step(EtgHandle* h, const float* action, const uint8_t* donef, float* obs, float* reward, uint8_t* done, float* info, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (!obs || !reward || !done) return cfail(ETG_ERR_BAD_ARG, "etg_step: obs/reward/done must be non-null");
  CpuHandle* c = H(h);
  if (!c->was_reset) return cfail(ETG_ERR_STATE, "etg_step: call etg_reset first");
  const int adim = c->sim->cfg.motor_mode == 2 ? ETG_HYBRID_DIM : ETG_ACT_DIM;
  if (c->sim->cfg.motor_mode == 2 && !action) return cfail(ETG_ERR_BAD_ARG, "etg_step: the HYBRID motor mode needs a [N,60] command");
  const size_t N = c->N;
  std::vector<double> a(N * adim, 0.0), o(N * ETG_OBS_DIM), r(N), inf(info ? N * ETG_INFO_DIM : 0);
  if (action) for (size_t k = 0; k < N * adim; k++) a[k] = action[k];
  etgo_step64(c->sim, a.data(), donef, o.data(), r.data(), done, info ? inf.data() : nullptr, 1);
  for (size_t k = 0; k < N * ETG_OBS_DIM; k++) obs[k] = (float)o[k];
  for (size_t i = 0; i < N; i++) {
    reward[i] = (float)r[i];
    c->ret[i] += c->alive[i] * r[i];
    c->len[i] += (int32_t)c->alive[i];
    if (done[i]) c->alive[i] = 0.0;
    c->age_ticks[i] += c->sim->cfg.action_repeat;
  }
  if (info) for (size_t k = 0; k < N * ETG_INFO_DIM; k++) info[k] = (float)inf[k];
  return ETG_OK;
}

int etg_step_range(EtgHandle* h, int env0, int count, const float* action, const uint8_t* donef, float* obs, float* reward,
                   uint8_t* done, float* info, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (!obs || !reward || !done) return cfail(ETG_ERR_BAD_ARG, "etg_step: obs/reward/done must be non-null");
  CpuHandle* c = H(h);
  if (!c->was_reset) return cfail(ETG_ERR_STATE, "etg_step: call etg_reset first");
  const int adim = c->sim->cfg.motor_mode == 2 ? ETG_HYBRID_DIM : ETG_ACT_DIM;
  if (c->sim->cfg.motor_mode == 2 && !action) return cfail(ETG_ERR_BAD_ARG, "etg_step: the HYBRID motor mode needs a [N,60] command");
  const size_t N = c->N;
  if (env0 < 0 || count <= 0 || (size_t)(env0 + count) > N || env0 % 16 != 0 || (count % 16 != 0 && (size_t)(env0 + count) != N))
    return cfail(ETG_ERR_BAD_ARG, "etg_step_range: env0 and count must be multiples of 16 robots (the last range may end at N)");
  const size_t a0 = env0, a1 = env0 + count;
  std::vector<double> a(N * adim, 0.0), o(N * ETG_OBS_DIM), r(N), inf(info ? N * ETG_INFO_DIM : 0);
  if (action) for (size_t k = a0 * adim; k < a1 * adim; k++) a[k] = action[k];
  etgo_step_range64(c->sim, env0, count, a.data(), donef, o.data(), r.data(), done, info ? inf.data() : nullptr, 1);
  for (size_t k = a0 * ETG_OBS_DIM; k < a1 * ETG_OBS_DIM; k++) obs[k] = (float)o[k];
  for (size_t i = a0; i < a1; i++) {
    reward[i] = (float)r[i];
    c->ret[i] += c->alive[i] * r[i];
    c->len[i] += (int32_t)c->alive[i];
    if (done[i]) c->alive[i] = 0.0;
    c->age_ticks[i] += c->sim->cfg.action_repeat;
  }
  if (info) for (size_t k = a0 * ETG_INFO_DIM; k < a1 * ETG_INFO_DIM; k++) info[k] = (float)inf[k];
  return ETG_OK;
}

int etg_step_autoreset(EtgHandle* h, const float* action, const uint8_t* donef, float* obs, float* reward, uint8_t* done, float* info, void* s) {
  int rc = etg_step(h, action, donef, obs, reward, done, info, s);
  if (rc != ETG_OK) return rc;
  return etg_reset(h, done, obs, s);
}

int etg_episode_stats(EtgHandle* h, float* ret, int32_t* len, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  for (int i = 0; i < H(h)->N; i++) {
    if (ret) ret[i] = (float)H(h)->ret[i];
    if (len) len[i] = H(h)->len[i];
  }
  return ETG_OK;
}

int etg_rollout_wave_cycles(EtgHandle* h, int64_t*, int, int* launches, int* waves, void*) {
  if (!h || !launches || !waves) return cfail(ETG_ERR_BAD_ARG, "etg_rollout_wave_cycles: null");
  *launches = 0; *waves = 0;        // no wavefronts on the host
  return ETG_OK;
}

int etg_set_rollout_mode(EtgHandle* h, int simulate_finished) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  H(h)->simulate_finished = simulate_finished != 0;
  return ETG_OK;
}

// the fused open-loop rollout, sequentially: every robot's own loop of zero-action steps, left when its episode ends (the
// default; etg_set_rollout_mode(h, 1): finished robots are stepped on, accumulators masked).  obs: every robot's LAST row.
int etg_rollout_openloop(EtgHandle* h, int n_steps, float* obs, float* ret, int32_t* len, void* s) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (n_steps <= 0 || !ret || !len) return cfail(ETG_ERR_BAD_ARG, "etg_rollout_openloop: bad arguments");
  CpuHandle* c = H(h);
  if (!c->was_reset) return cfail(ETG_ERR_STATE, "etg_rollout_openloop: call etg_reset first");
  if (c->sim->cfg.motor_mode == 2) return cfail(ETG_ERR_BAD_ARG, "etg_rollout_openloop: the HYBRID motor mode needs commands");
  const size_t N = c->N;
  std::vector<uint8_t> alive(N);
  for (size_t i = 0; i < N; i++) alive[i] = c->alive[i] > 0.5 ? 1 : 0;
  std::vector<double> r(N), o(N * ETG_OBS_DIM);
  std::vector<int32_t> n(N);
  if (obs) for (size_t k = 0; k < o.size(); k++) o[k] = obs[k];     // rows of robots that were finished before stay
  etgo_run_steps64(c->sim, nullptr, n_steps, 1, r.data(), n.data(), c->simulate_finished ? 0 : 1, alive.data(), obs ? o.data() : nullptr);
  for (size_t i = 0; i < N; i++) {
    c->ret[i] += r[i];
    c->len[i] += n[i];
    const int stepped = c->simulate_finished ? n_steps : n[i];
    c->age_ticks[i] += stepped * c->sim->cfg.action_repeat;
    c->alive[i] = alive[i] ? 1.0 : 0.0;                              // (in / out: cleared where the episode ended inside the call)
  }
  if (obs) for (size_t k = 0; k < o.size(); k++) obs[k] = (float)o[k];
  return etg_episode_stats(h, ret, len, s);
}

int etg_get_state(EtgHandle* h, float* state, void*) {
  if (!h || !state) return cfail(ETG_ERR_BAD_ARG, "etg_get_state: null");
  std::vector<double> st((size_t)H(h)->N * ETG_STATE_DIM);
  etgo_get_state64(H(h)->sim, st.data());
  for (size_t k = 0; k < st.size(); k++) state[k] = (float)st[k];
  return ETG_OK;
}

int etg_set_state(EtgHandle* h, const float* state, void*) {
  if (!h || !state) return cfail(ETG_ERR_BAD_ARG, "etg_set_state: null");
  std::vector<double> st = to_d(state, (size_t)H(h)->N * ETG_STATE_DIM);
  etgo_set_state64(H(h)->sim, st.data());
  return ETG_OK;
}

int etg_get_contact_impulses(EtgHandle* h, float* lam, void*) {
  if (!h || !lam) return cfail(ETG_ERR_BAD_ARG, "etg_get_contact_impulses: null");
  std::vector<double> l((size_t)H(h)->N * 16);
  etgo_get_lambda64(H(h)->sim, l.data());
  for (size_t k = 0; k < l.size(); k++) lam[k] = (float)l[k];
  return ETG_OK;
}

int etg_set_contact_impulses(EtgHandle* h, const float* lam, void*) {
  if (!h || !lam) return cfail(ETG_ERR_BAD_ARG, "etg_set_contact_impulses: null");
  std::vector<double> l = to_d(lam, (size_t)H(h)->N * 16);
  etgo_set_lambda64(H(h)->sim, l.data());
  return ETG_OK;
}

/* ---- transitions for the off-policy learner (train.py:148-159,240-241), host pointers: the plain sequential statement of
 * what paddlerobotics_amd/csrc/etg_replay.hip does with a prefix sum and scattered rows ------------------------------- */
int etg_replay_begin(const uint8_t* alive, int n, long long max_size, long long* pos_count, int32_t* slot, const float* obs,
                     int obs_dim, const float* act, int act_dim, float* mem_obs, float* mem_act, float act_scale, float* act_scaled,
                     void*) {
  if (n <= 0 || max_size <= 0 || n > max_size || !pos_count || !slot || !obs || !act || !mem_obs || !mem_act || obs_dim <= 0 || act_dim <= 0)
    return cfail(ETG_ERR_BAD_ARG, "etg_replay_begin: bad arguments (a batch must fit the memory)");
  long long pos = pos_count[0];
  for (int i = 0; i < n; i++) {
    const bool a = !alive || alive[i];
    slot[i] = a ? (int32_t)pos : -1;             // the row of a finished robot is not stored
    if (a) {
      std::memcpy(mem_obs + (size_t)pos * obs_dim, obs + (size_t)i * obs_dim, sizeof(float) * obs_dim);
      std::memcpy(mem_act + (size_t)pos * act_dim, act + (size_t)i * act_dim, sizeof(float) * act_dim);
      pos = (pos + 1) % max_size;
      pos_count[1]++;
    }
    if (act_scaled)
      for (int k = 0; k < act_dim; k++) act_scaled[(size_t)i * act_dim + k] = act_scale * act[(size_t)i * act_dim + k];
  }
  pos_count[0] = pos;
  return ETG_OK;
}
int etg_replay_end(const int32_t* slot, int n, const float* reward, const uint8_t* done, const float* next_obs, int obs_dim,
                   float* mem_reward, float* mem_terminal, float* mem_next_obs, const float* info, int info_dim, int n_sum,
                   int velx_col, float* info_sum, uint8_t* alive, void*) {
  if (n <= 0 || !slot || !reward || !done || !next_obs || !mem_reward || !mem_terminal || !mem_next_obs || obs_dim <= 0)
    return cfail(ETG_ERR_BAD_ARG, "etg_replay_end: bad arguments");
  if (info && (info_dim <= 0 || n_sum < 0 || n_sum > info_dim || velx_col >= info_dim)) return cfail(ETG_ERR_BAD_ARG, "etg_replay_end: bad info layout");
  for (int i = 0; i < n; i++) {
    if (slot[i] < 0) continue;                   // a finished robot: nothing stored, nothing summed, alive stays 0
    const size_t s = (size_t)slot[i];
    mem_reward[s] = reward[i];
    mem_terminal[s] = done[i] ? 0.0f : 1.0f;     // the stored flag is the bootstrap mask (train.py:148-149)
    std::memcpy(mem_next_obs + s * obs_dim, next_obs + (size_t)i * obs_dim, sizeof(float) * obs_dim);
    if (!alive) continue;
    if (alive[i] && info && info_sum) {
      for (int k = 0; k < n_sum; k++) info_sum[(size_t)i * (n_sum + 1) + k] += info[(size_t)i * info_dim + k];
      if (velx_col >= 0 && info[(size_t)i * info_dim + velx_col] >= 0.3f) info_sum[(size_t)i * (n_sum + 1) + n_sum] += 1.0f;
    }
    alive[i] = (uint8_t)(alive[i] && !done[i]);
  }
  return ETG_OK;
}

/* ---- device-only entry points ------------------------------------------------------------------------------ */
#define CPU_UNAVAILABLE(name) return cfail(ETG_ERR_STATE, name ": a device kernel of the HIP library, not part of the CPU build")
int etg_random_pushes(EtgHandle*, uint64_t, float, int, float, float, void*) { CPU_UNAVAILABLE("etg_random_pushes"); }
int etg_clear_pushes(EtgHandle*, const uint8_t*, void*) { return ETG_OK; }
int etg_leg_kinematics(EtgHandle*, const float*, int, float*, float*, void*) { CPU_UNAVAILABLE("etg_leg_kinematics"); }
int etg_extra_sensors(EtgHandle*, const float*, float*, void*) { CPU_UNAVAILABLE("etg_extra_sensors"); }
int etg_policy_create(int, int, int, int, EtgPolicy**) { CPU_UNAVAILABLE("etg_policy_create"); }
int etg_policy_load(EtgPolicy*, const float*, const float*, const float*, const float*, const float*, const float*, void*) { CPU_UNAVAILABLE("etg_policy_load"); }
int etg_policy_forward(EtgPolicy*, const float*, int, float, int, float*, void*) { CPU_UNAVAILABLE("etg_policy_forward"); }
int etg_policy_load_std(EtgPolicy*, const float*, const float*, void*) { CPU_UNAVAILABLE("etg_policy_load_std"); }
int etg_policy_sample(EtgPolicy*, const float*, int, const float*, float, int, float*, float*, void*) { CPU_UNAVAILABLE("etg_policy_sample"); }
void etg_policy_destroy(EtgPolicy*) {}
int etg_rollout_policy(EtgHandle*, EtgPolicy*, int, float, int, int, float*, float*, int32_t*, void*) { CPU_UNAVAILABLE("etg_rollout_policy"); }
int etg_rollout_actions(EtgHandle*, const float*, int, float*, float*, float*, float*, float*, uint8_t*, float*, int32_t*, void*) { CPU_UNAVAILABLE("etg_rollout_actions"); }
int etg_prepare_next_dynamics(EtgHandle* h, const float* dyn, const uint8_t* mask, void*) {
  if (!h) return cfail(ETG_ERR_BAD_ARG, "null handle");
  if (!dyn) return cfail(ETG_ERR_BAD_ARG, "etg_prepare_next_dynamics: dyn is null");
  CpuHandle* c = H(h);
  if (!c->was_reset) return cfail(ETG_ERR_STATE, "etg_prepare_next_dynamics: needs a full etg_reset first");
  if (c->next_ok.empty()) { c->next_ok.assign(c->N, 0); c->next_dyn.assign((size_t)c->N * ETG_DYN_DIM, 0.0f); }
  for (int i = 0; i < c->N; i++) {
    if (mask && !mask[i]) continue;
    // the device library leaves out robots in the first RING = 64 ticks of their episode (they still read the settle cache's
    // copy of the latency ring, which the call replaces); their pending flag stays 0 and the caller's next refresh covers them
    if (c->age_ticks[i] < 64) continue;
    for (int k = 0; k < ETG_DYN_DIM; k++) c->next_dyn[(size_t)i * ETG_DYN_DIM + k] = dyn[(size_t)i * ETG_DYN_DIM + k];
    c->next_ok[i] = 1;
  }
  return ETG_OK;
}
int etg_next_dynamics_pending(EtgHandle* h, uint8_t* pending, void*) {
  if (!h || !pending) return cfail(ETG_ERR_BAD_ARG, "etg_next_dynamics_pending: bad arguments");
  CpuHandle* c = H(h);
  for (int i = 0; i < c->N; i++) pending[i] = c->next_ok.empty() ? 0 : c->next_ok[i];
  return ETG_OK;
}
int etg_rollout_policy_record(EtgHandle*, EtgPolicy*, int, float, int, int, float*, const float*, float*, float*, float*, uint8_t*, float*, int32_t*, void*) { CPU_UNAVAILABLE("etg_rollout_policy_record"); }
int etg_fit_etg(const double*, int, const double*, const double*, double, double, double, double, double, int, double*, double*, void*) { CPU_UNAVAILABLE("etg_fit_etg"); }

}  // extern "C"
