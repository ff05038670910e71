// etg_kernels.hip -- gfx950 kernels + C-ABI of the batched A1 simulator (include/etgsim.h).
//
// Two lane mappings of the same algorithm over one HBM state layout (etg_layout.h):
//   k_step16 / k_settle16 / k_finish16 (etg_core16.h): one robot = one 16-lane DPP row, a wave64 = 4 robots;
//       4096 robots = 1024 single-wave workgroups = one wave on every SIMD (the default up to 4096 robots);
//   k_step / k_settle / k_finish (etg_core.h): one robot = one quad (one leg per lane), a wave64 = 16 robots
//       (bigger batches).
// Workgroups are remapped so that each XCD owns a contiguous range of robots (xcd_contiguous_block).  Reductions
// and broadcasts are DPP moves (quad_perm, row butterflies, row_newbcast), fused into the consuming ALU op where
// possible (v_add_f32_dpp by the compiler, v_fmac_f32_dpp by inline asm); the 4-lane contact operator's 4x4
// blocks are v_mfma_f32_4x4x1.  LDS holds each lane's private per-step parameter column (no barriers anywhere);
// the per-tick constants go straight from HBM to registers.  A reset restores the cached 500-tick settle.
// Every step / rollout / reset kernel has compile-time variants picked per launch (LAUNCH16 / LAUNCH4): flat ground
// vs heightfield, body contacts, and PLAIN (the default robot layer with its unused options compiled out).
#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "etg_core16.h"
#include "policy_core.h"

// ---- translation units.  The device code of this file is ~120 instantiations of the physics tick: four minutes of
// single-threaded code generation.  paddlerobotics_amd/build.py therefore compiles the file ETG_TU_PARTS times in parallel:
// part 0 holds the host side (the C-ABI) and the small kernels and declares the tick kernels `extern template`; parts 1..
// hold their explicit instantiations (the table at the end of namespace etg).  Without the macros -- tools/build_variant.sh, a
// plain `hipcc etg_kernels.hip` -- the file is one unit and every kernel is instantiated where it is launched, as before.
#ifndef ETG_TU_PARTS
#define ETG_TU_PARTS 1
#define ETG_TU_PART 0
#endif
#define ETG_TU_HOST (ETG_TU_PART == 0)

namespace etg {

// ---- DPP quad helpers -------------------------------------------------------------
// bound_ctrl = true: every lane of every group is always valid here, and with it the compiler needs no
// "old" value -- without it each v_mov_b32_dpp is preceded by a v_mov_b32 that zero-initialises the
// destination (measured: ~150 extra VALU issues per tick)
template <int CTRL> __device__ __forceinline__ float dpp_(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}

struct GpuCtx {
  int gid, env, lane, N, NL;
  const float* lds;  // this lane's parameter column in LDS: lds[k * BLOCK]
  const float* gpar; // D.par, for the tick constants read straight into registers
  // store gate (fused rollouts under KCfg.stop_at_done, etg_core16.h: control_step16_core): every store of this context happens
  // on the lanes inside the gate only.  Never closed by the step / reset kernels: the constant `true` folds away there.
  mutable bool gate = true;
  __device__ __forceinline__ void set_gate(bool on) const { gate = on; }
  __device__ __forceinline__ void open_gate() const { gate = true; }
  __device__ __forceinline__ int sel_i(bool m, int a, int b) const { return m ? a : b; }
  __device__ __forceinline__ float par(int k) const { return lds[k * 64]; }
  __device__ __forceinline__ float tpar(int k) const { return gpar[(size_t)k * NL + gid]; }
  __device__ __forceinline__ float ld_lane(const float* p, int f) const { return p[(size_t)f * NL + gid]; }
  __device__ __forceinline__ void st_lane(float* p, int f, float v) const { if (gate) p[(size_t)f * NL + gid] = v; }
  __device__ __forceinline__ float ld_env(const float* p, int f) const { return p[(size_t)f * N + env]; }
  __device__ __forceinline__ void st_env(float* p, int f, float v) const { if (gate && lane == 0) p[(size_t)f * N + env] = v; }
  __device__ __forceinline__ int ld_env_i(const int* p, int f) const { return p[(size_t)f * N + env]; }
  __device__ __forceinline__ void st_env_i(int* p, int f, int v) const { if (gate && lane == 0) p[(size_t)f * N + env] = v; }
  __device__ __forceinline__ void st_ring(float* r, int slot, int k, float v) const { if (gate) r[((size_t)slot * 8 + k) * NL + gid] = v; }
  __device__ __forceinline__ float ld_ring(const float* r, int slot, int k) const { return r[((size_t)slot * 8 + k) * NL + gid]; }
  // the ring is written and read back by the SAME lane (program order): no fence needed
  __device__ __forceinline__ void ring_fence() const {}
  // phase boundary: keep the machine scheduler from interleaving whole phases of the tick
  // (it otherwise stretches live ranges to >500 registers and spills)
  __device__ __forceinline__ void phase([[maybe_unused]] int id) const {
    __builtin_amdgcn_sched_barrier(0);
#ifdef ETG_PROFILE_PHASES  // tools/phase_profile.py: per-phase s_memtime deltas of one wave
    long long t = clock64();
    prof[id] += t - prof_last;
    prof_last = t;
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
#ifdef ETG_PROFILE_PHASES
  mutable long long prof[16];
  mutable long long prof_last;
#endif
  int row_base = 0;  // first robot of the [*, rowlen] row buffers (observation tile in LDS: the workgroup's first robot)
  __device__ __forceinline__ void st_row_env(float* p, int rowlen, int col, float v) const { if (gate && lane == 0) p[(size_t)(env - row_base) * rowlen + col] = v; }
  __device__ __forceinline__ void st_row_lane(float* p, int rowlen, int col0, int stride, float v) const { if (gate) p[(size_t)(env - row_base) * rowlen + col0 + stride * lane] = v; }
  __device__ __forceinline__ float ld_row_env(const float* p, int rowlen, int col) const { return p[(size_t)(env - row_base) * rowlen + col]; }
  __device__ __forceinline__ float ld_row_lane(const float* p, int rowlen, int col0, int stride) const { return p[(size_t)(env - row_base) * rowlen + col0 + stride * lane]; }
  // order-symmetric quad sum: (x0+x1)+(x2+x3) on every lane, bit-identical across the quad
  __device__ __forceinline__ float qsum(float a) const {
    float t = a + dpp_<0xB1>(a);  // quad_perm [1,0,3,2]
    return t + dpp_<0x4E>(t);     // quad_perm [2,3,0,1]
  }
  __device__ __forceinline__ float qmax(float a) const {
    float t = fmaxf(a, dpp_<0xB1>(a));
    return fmaxf(t, dpp_<0x4E>(t));
  }
  __device__ __forceinline__ float qbcast(float a, int j) const {  // j is a constant after unrolling
    switch (j) {
      case 0: return dpp_<0x00>(a);
      case 1: return dpp_<0x55>(a);
      case 2: return dpp_<0xAA>(a);
      default: return dpp_<0xFF>(a);
    }
  }
  // acc[i] += a@lane_i * b@this_lane for the 4 lanes of the quad: v_mfma_f32_4x4x1_16b_f32 computes
  // 16 independent 4x4 outer products per wave64, block b = lanes 4b..4b+3, D[i][j] = A[lane i] *
  // B[lane j] held at lane j / register i (layout verified on hardware: tools/ubench/mfma4x4.hip)
  __device__ __forceinline__ void quad_outer(float a, float b, float* acc) const {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v = {acc[0], acc[1], acc[2], acc[3]};
    v = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, v, 0, 0, 0);
    acc[0] = v[0]; acc[1] = v[1]; acc[2] = v[2]; acc[3] = v[3];
  }
  __device__ __forceinline__ bool lane_is(int j) const { return lane == j; }
  __device__ __forceinline__ bool any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
  __device__ __forceinline__ unsigned uniform_bits(unsigned v) const { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }   // a wave-uniform value, kept in an SGPR
  __device__ __forceinline__ bool wave_any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0ull; }   // "does any robot of the wave need another sweep?"
  // "does any lane of MY robot (quad) see b?" from the wave mask of the compare: two ANDs with this lane's quad field
  __device__ __forceinline__ bool robot_any(bool b) const {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(b);
    const unsigned sh = (unsigned)(threadIdx.x & 28);           // first lane of the quad inside its 32-lane half
    const unsigned f = 0xFu << sh;
    return ((threadIdx.x & 32) ? ((unsigned)(m >> 32) & f) : ((unsigned)m & f)) != 0u;
  }
  __device__ __forceinline__ int uniform_int(float a) const { return (int)a; }
  __device__ __forceinline__ void terrain(const KCfg& K, float x, float y, float& h, float& nx, float& ny, float& nz) const {
    if (K.terrain == 0) { h = 0.0f; nx = 0.0f; ny = 0.0f; nz = 1.0f; }
    else heightfield_query(K, env, x, y, h, nx, ny, nz);
  }
  __device__ __forceinline__ void terrain_fetch(const KCfg& K, float x, float y, float* tap) const { heightfield_fetch(K, env, x, y, tap); }
  __device__ __forceinline__ void terrain_finish(const KCfg& K, const float* tap, float& h, float& nx, float& ny, float& nz) const {
    heightfield_finish(K, tap, h, nx, ny, nz);
  }
};

// FLAT selects the plane-z=0 fast path at compile time (contact frame = rows of R, no terrain lookup)
// PLAIN: see GpuCtx16T.  BODY: body contact rows per leg compiled in (EtgConfig.body_contacts 1 / 2 -> 1, 3 -> 3)
template <bool FLAT, bool PLAIN = false, int BODY = 0> struct GpuCtxT : GpuCtx {
  static constexpr bool kFlat = FLAT; static constexpr bool kPlain = PLAIN; static constexpr int kBody = BODY;
};

// Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b % 8), each with its own L2.  Robots
// of neighbouring blocks share 128 B lines of the SoA state (a block covers only 16-64 B of a field), so
// give every XCD a CONTIGUOUS range of robots: the blocks that share a line then share an L2 and the
// line is fetched from HBM once instead of once per XCD.  Bijective for any grid size.
__device__ __forceinline__ int xcd_contiguous_block() {
  const int nb = gridDim.x, b = blockIdx.x, x = b & 7, q = nb >> 3, r = nb & 7;
  return x * q + (x < r ? x : r) + (b >> 3);
}

__device__ __forceinline__ bool make_ctx(const KCfg& K, GpuCtx& c) {
  c.gid = (K.block0 + xcd_contiguous_block()) * blockDim.x + threadIdx.x;   // (block0: etg_step_range; 0 in every other launch)
  c.N = K.n_env;
  c.NL = 4 * K.n_env;
  c.env = c.gid >> 2;
  c.lane = c.gid & 3;
  c.lds = nullptr;
  return c.gid < c.NL;
}

constexpr int BLOCK = 64;

// Parameters waiting for a robot's NEXT episode (etg_prepare_next_dynamics): the derived per-leg rows, the dynamic_param row and a
// pending flag per robot.  The settle that belongs to them already sits in the robot's settle cache; the restart inside
// etg_step_autoreset (or the next etg_reset of the robot) installs the rows.  All null when the feature is unused.
struct NextDyn { float* par; float* dyn; unsigned char* ok; };
// install the pending rows of the masked robots (a reset of theirs is about to use the settle cache that belongs to the rows)
#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void __launch_bounds__(256) k_next_take(KCfg K, DevState D, NextDyn NX, const uint8_t* mask) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = K.n_env, NL = 4 * N;
  if (col >= NL) return;
  const int env = col >> 2;
  if ((mask && !mask[env]) || !NX.ok[env]) return;
  for (int k = 0; k < PR_DERIVED; k++) D.par[(size_t)k * NL + col] = NX.par[(size_t)k * NL + col];
  for (int k = col & 3; k < ETG_DYN_DIM; k += 4) D.dyn[(size_t)env * ETG_DYN_DIM + k] = NX.dyn[(size_t)env * ETG_DYN_DIM + k];
}
// etg_prepare_next_dynamics replaces a robot's settle cache -- cached state AND the cache's copy of the latency ring -- while the
// robot keeps running.  A running robot still READS that ring copy for every delayed reading of a tick at or before its reset
// tick (ring_of_tick / KCfg.cring), i.e. during the first RING ticks of its episode.  Such YOUNG robots are left out of the
// call: out[env] = masked && the robot's episode is at least RING ticks old.  Their pending flag stays 0, so the caller's next
// refresh draws rows for them again.
__global__ void __launch_bounds__(256) k_next_old_enough(KCfg K, DevState D, const uint8_t* mask, uint8_t* out) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= K.n_env) return;
  out[env] = ((!mask || mask[env]) && D.ictl[(size_t)IC_TICK * K.n_env + env] - K.settle_ticks >= RING) ? 1 : 0;
}
// ok[env] = value for the masked robots; scratch flags start as "not cached" so that the settle runs for them
__global__ void __launch_bounds__(256) k_next_flags(KCfg K, unsigned char* ok, const uint8_t* mask, unsigned char value) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= K.n_env || (mask && !mask[env])) return;
  ok[env] = value;
}
#endif


// Stage the lane's 66 derived parameters in LDS, [field][lane] (conflict-free: lane i hits
// bank i).  Each lane reads back only its own column, so no barrier is needed -- the LDS
// is a software-managed register file extension here, not a sharing medium.
// only what the once-per-step code reads through c.par(); the tick constants go straight to registers (tpar)
constexpr int kStaged4[] = {PR_O1, PR_O1 + 1, PR_O1 + 2, PR_SY, PR_LAT_N, PR_LAT_A, PR_BASE_FOOT, PR_BASE_FOOT + 1,
                            PR_BASE_FOOT + 2, PR_POSE, PR_POSE + 1, PR_POSE + 2, PR_EMEAN, PR_EMEAN + 1, PR_EMEAN + 2,
                            PR_ESTD, PR_ESTD + 1, PR_ESTD + 2, PR_HIPSIGN};
constexpr int kNStaged4 = sizeof(kStaged4) / sizeof(int);
// issue / commit: a caller may put its own cold loads between the two (step4_body, like step16_body)
__device__ __forceinline__ void stage4_issue(GpuCtx& c, const DevState& D, float* lds_par, float (&v)[kNStaged4]) {
#pragma unroll
  for (int i = 0; i < kNStaged4; i++) v[i] = D.par[(size_t)kStaged4[i] * c.NL + c.gid];
  c.lds = lds_par + threadIdx.x;
  c.gpar = D.par;
}
__device__ __forceinline__ void stage4_commit(const GpuCtx& c, const float (&v)[kNStaged4]) {
  float* mine = const_cast<float*>(c.lds);
#pragma unroll
  for (int i = 0; i < kNStaged4; i++) mine[kStaged4[i] * BLOCK] = v[i];
}
__device__ __forceinline__ void stage_params(GpuCtx& c, const DevState& D, float* lds_par) {
  float v[kNStaged4];
  stage4_issue(c, D, lds_par, v);
  stage4_commit(c, v);
}

// the same staging for a wave of a multi-wave workgroup (its own [PR_N][64] area, lane = lane in the wave)
__device__ __forceinline__ void stage_params_wave(GpuCtx& c, const DevState& D, float* lds_wave, int lane) {
  constexpr int kStaged[] = {PR_O1, PR_O1 + 1, PR_O1 + 2, PR_SY, PR_LAT_N, PR_LAT_A, PR_BASE_FOOT, PR_BASE_FOOT + 1,
                             PR_BASE_FOOT + 2, PR_POSE, PR_POSE + 1, PR_POSE + 2, PR_EMEAN, PR_EMEAN + 1, PR_EMEAN + 2,
                             PR_ESTD, PR_ESTD + 1, PR_ESTD + 2, PR_HIPSIGN};
#pragma unroll
  for (int k : kStaged) lds_wave[k * 64 + lane] = D.par[(size_t)k * c.NL + c.gid];
  c.lds = lds_wave + lane;
  c.gpar = D.par;
}

#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void __launch_bounds__(BLOCK) k_set_params(KCfg K, ModelF M, DevState D, const float* dyn, const float* w,
                                                       const float* b, int per_env, const uint8_t* mask) {
  GpuCtx c;
  if (!make_ctx(K, c)) return;
  if (mask && !mask[c.env]) return;
  if (dyn) {
    if (c.lane == 0) D.cache_ok[c.env] = 0;   // the settle depends on the dynamic parameters
    float row[ETG_DYN_DIM], out[PR_DERIVED];
    for (int k = 0; k < ETG_DYN_DIM; k++) row[k] = dyn[(size_t)c.env * ETG_DYN_DIM + k];
    for (int k = c.lane; k < ETG_DYN_DIM; k += 4) D.dyn[(size_t)c.env * ETG_DYN_DIM + k] = row[k];   // kept for the dynamic_vec sensor
    derive_lane_params(M, row, c.lane, K.dt, out);
    for (int k = 0; k < PR_DERIVED; k++) D.par[(size_t)k * c.NL + c.gid] = out[k];   // (the strength ratios PR_STR.. are not derived: etg_set_motor_strength)
  }
  // the 63 ETG floats of a robot are copied by its 4 lanes (16 each)
  for (int k = c.lane; k < 60; k += 4)
    if (w) D.etgp[(size_t)(EP_W + k) * c.N + c.env] = w[(per_env ? (size_t)c.env * 60 : 0) + k];
  if (b && c.lane < 3) D.etgp[(size_t)(EP_B + c.lane) * c.N + c.env] = b[(per_env ? (size_t)c.env * 3 : 0) + c.lane];
}
#endif

// ---- reset = settle (only for robots without a valid settle cache) -> restore from the cache -> finish
// copy this lane's 8 ring words of all slots and its leg column between the live arrays and the cache
// (the ring is NOT copied back from the cache: readings of ticks up to the reset tick are served from KCfg.cring)
__device__ __forceinline__ void copy_leg_column(const DevState& D, size_t col, size_t NL, bool to_cache) {
  const float* src_leg = to_cache ? D.leg : D.cache_leg;
  float* dst_leg = to_cache ? D.cache_leg : D.leg;
  for (int f = 0; f < LG_N; f++) dst_leg[(size_t)f * NL + col] = src_leg[(size_t)f * NL + col];
}
__device__ __forceinline__ void copy_base_column(const DevState& D, int env, int N, bool to_cache) {
  const float* src = to_cache ? D.base : D.cache_base;
  float* dst = to_cache ? D.cache_base : D.base;
  for (int f = 0; f < BS_N; f++) dst[(size_t)f * N + env] = src[(size_t)f * N + env];
}

// A cached settle is reusable when it ran at the start offset the robot resets to now; on flat ground the settle is
// translation-invariant, so there any cached settle is reusable and k_cache_sync shifts it to the new offset.
template <bool FLAT> __device__ __forceinline__ bool settle_cached(const KCfg& K, const DevState& D, int env) {
  if (!D.cache_ok[env]) return false;
  if (FLAT) return true;
  return D.cache_off[env] == D.reset_off[env] && D.cache_off[K.n_env + env] == D.reset_off[K.n_env + env];
}
// every lane of the robot writes the same values (the robot's lanes share a wave: the reads above came first)
__device__ __forceinline__ void settle_mark_fresh(const KCfg& K, const DevState& D, int env, float ox, float oy) {
  D.cache_ok[env] = 0;   // "just settled": k_cache_sync snapshots the ring, k_cache_mark sets it again
  D.cache_off[env] = ox;
  D.cache_off[K.n_env + env] = oy;
}

template <bool FLAT, bool PLAIN, int BODY = 0>
__global__ void __launch_bounds__(BLOCK) k_settle(KCfg K, DevState D, const uint8_t* mask) {
  GpuCtxT<FLAT, PLAIN, BODY> c;
  if (!make_ctx(K, c)) return;
  if ((mask && !mask[c.env]) || settle_cached<FLAT>(K, D, c.env)) return;   // whole quads drop out together
  __shared__ float lds_par[PR_N * BLOCK];
  stage_params(c, D, lds_par);
  LaneState<float> L;
  const float ox = D.reset_off[c.env], oy = D.reset_off[K.n_env + c.env];
  reset_settle(c, K, L, D.ring, ox, oy);
  store_state(c, D.cache_base, D.cache_leg, L);              // the ring is copied by k_cache_sync (needs the stores done)
  store_state(c, D.base, D.leg, L);
  settle_mark_fresh(K, D, c.env, ox, oy);
}

// one thread per leg column: after a settle, snapshot the ring into the cache and mark the robot cached;
// for every masked robot, bring state + ring back from the cache
#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void __launch_bounds__(256) k_cache_sync(KCfg K, DevState D, const uint8_t* mask) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = K.n_env, NL = 4 * N;
  if (col >= NL) return;
  const int env = col >> 2;
  if (mask && !mask[env]) return;
  if (!D.cache_ok[env]) {   // just settled by k_settle: live arrays -> cache (ring only; state was stored to both)
    for (int w = 0; w < RING * 8; w++) D.cache_ring[(size_t)w * NL + col] = D.ring[(size_t)w * NL + col];
  } else {                  // cached: cache -> live arrays
    copy_leg_column(D, col, NL, false);
    if ((col & 3) == 0) {
      copy_base_column(D, env, N, false);
      // shift to the offset of THIS reset (non-zero only on flat ground, see settle_cached)
      D.base[(size_t)BS_PX * N + env] += D.reset_off[env] - D.cache_off[env];
      D.base[(size_t)BS_PY * N + env] += D.reset_off[N + env] - D.cache_off[N + env];
    }
  }
}
__global__ void __launch_bounds__(256) k_cache_mark(KCfg K, DevState D, const uint8_t* mask) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= K.n_env) return;
  if (mask && !mask[env]) return;
  D.cache_ok[env] = 1;
}
#endif

template <bool FLAT, bool PLAIN, int BODY = 0>
__global__ void __launch_bounds__(BLOCK) k_finish(KCfg K, DevState D, const uint8_t* mask, float* obs) {
  GpuCtxT<FLAT, PLAIN, BODY> c;
  if (!make_ctx(K, c)) return;
  if (mask && !mask[c.env]) return;
  __shared__ float lds_par[PR_N * BLOCK];
  stage_params(c, D, lds_par);
  LaneState<float> L = load_state<float>(c, D.base, D.leg);
  reset_finish(c, K, L, D.ring, D.ctl, D.ictl, D.legctl, D.etgp, obs);
  store_state(c, D.base, D.leg, L);
}

// ---- cached restart (etg_step_autoreset).  A launch lasts as long as its slowest wave, and with auto-reset
// some wave restarts a robot in nearly every launch; recomputing reset_finish16 there (first ETG action, IK, foot kinematics,
// rpy reference, first observation: ~1200 instructions) costs the whole batch ~3 us per step.  What reset_finish16 produces
// depends only on the settled state, the robot's parameters and its ETG weights, so etg_reset keeps it per robot -- in rows
// FIN_* of the block behind D.cache_off (rows 0, 1 are the offsets the cached settle ran at) -- and the in-kernel restart
// copies it back.  Anything that changes an input clears the robot's FIN_OK flag (etg_set_params, etg_set_heightfield,
// etg_set_reset_offsets); the restart then recomputes as before.  Both lane mappings share the SoA state layout, and this cache.
enum { FIN_OBS = 2, FIN_RPY = FIN_OBS + ETG_OBS_DIM, FIN_FWX = FIN_RPY + 3, FIN_OK = FIN_FWX + 4, FIN_ROWS = FIN_OK + 1 };
#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void __launch_bounds__(256) k_fin_store(KCfg K, DevState D, const uint8_t* mask, const float* obs) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = K.n_env;
  if (env >= N || (mask && !mask[env])) return;
  float* fin = D.cache_off;
  for (int k = 0; k < ETG_OBS_DIM; k++) fin[(size_t)(FIN_OBS + k) * N + env] = obs[(size_t)env * ETG_OBS_DIM + k];
  for (int k = 0; k < 3; k++) fin[(size_t)(FIN_RPY + k) * N + env] = D.ctl[(size_t)(CT_FIRST_RPY + k) * N + env];
  for (int leg = 0; leg < 4; leg++) fin[(size_t)(FIN_FWX + leg) * N + env] = D.legctl[(size_t)LC_LAST_FOOT_X * 4 * N + 4 * env + leg];
  fin[(size_t)FIN_OK * N + env] = 1.0f;
}
// motor strength ratios (etg_set_motor_strength): par[PR_STR + j][4 env + leg] = ratios[env][3 leg + j], or 1 when ratios is null
__global__ void __launch_bounds__(256) k_set_strength(KCfg K, DevState D, const float* ratios, const uint8_t* mask) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int NL = 4 * K.n_env;
  if (col >= NL) return;
  const int env = col >> 2, leg = col & 3;
  if (mask && !mask[env]) return;
  for (int j = 0; j < 3; j++) D.par[(size_t)(PR_STR + j) * NL + col] = ratios ? ratios[(size_t)env * ETG_NUM_MOTORS + 3 * leg + j] : 1.0f;
  if (leg == 0) {   // the reset settle runs under the motor model: the robot's cached settle and cached restart are stale
    D.cache_ok[env] = 0;
    D.cache_off[(size_t)FIN_OK * K.n_env + env] = 0.0f;
  }
}

__global__ void __launch_bounds__(256) k_fin_clear(KCfg K, DevState D, const uint8_t* mask) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= K.n_env || (mask && !mask[env])) return;
  D.cache_off[(size_t)FIN_OK * K.n_env + env] = 0.0f;
}
#endif
// reset_finish from the cache (4 lanes per robot: lane = leg): same stores, no arithmetic
template <class Ctx>
__device__ __forceinline__ void restart_from_cache4(const Ctx& c, const KCfg& K, LaneState<float>& L, float* ctl, int* ictl, float* legctl,
                                                    const float* fin, float* obs) {
  const int N = K.n_env;
  L.energy = 0.0f;
  c.st_env_i(ictl, IC_STEP, 0);
  c.st_env_i(ictl, IC_TICK, K.settle_ticks);
  c.st_env_i(ictl, IC_HAS_LAST, 0);
  c.st_env(ctl, CT_RET, 0.0f); c.st_env(ctl, CT_LEN, 0.0f); c.st_env(ctl, CT_ALIVE, 1.0f);
  c.st_env(ctl, CT_LAST_BASE + 0, L.p.x); c.st_env(ctl, CT_LAST_BASE + 1, L.p.y); c.st_env(ctl, CT_LAST_BASE + 2, L.p.z);
#pragma unroll
  for (int j = 0; j < 3; j++) {
    const float pose = c.par(PR_POSE + j);
    c.st_lane(legctl, LC_LAST_QDES + j, pose);
    c.st_lane(legctl, LC_FX0 + j, pose); c.st_lane(legctl, LC_FX1 + j, pose);
    c.st_lane(legctl, LC_FY0 + j, pose); c.st_lane(legctl, LC_FY1 + j, pose);
  }
  c.st_lane(legctl, LC_LAST_FOOT_X, fin[(size_t)(FIN_FWX + c.lane) * N + c.env]);
#pragma unroll
  for (int k = 0; k < 3; k++) c.st_env(ctl, CT_FIRST_RPY + k, fin[(size_t)(FIN_RPY + k) * N + c.env]);
  for (int k = c.lane; k < ETG_OBS_DIM; k += 4) obs[(size_t)c.env * ETG_OBS_DIM + k] = fin[(size_t)(FIN_OBS + k) * N + c.env];
}

// env.step for the 16 robots of a wave (one quad each); AUTO: see step16_body
template <bool FLAT, bool PLAIN, bool AUTO, int BODY = 0>
__device__ __forceinline__ void step4_body(const KCfg& K, const DevState& D, const float* action, const uint8_t* donef, float* obs,
                                           float* reward, uint8_t* done, float* info, float* lds_par, const NextDyn NX = NextDyn{nullptr, nullptr, nullptr}) {
  GpuCtxT<FLAT, PLAIN, BODY> c;
  if (!make_ctx(K, c)) return;
  // every cold load of the launch head is requested before the first wait (see step16_body)
  float stg[kNStaged4];
  stage4_issue(c, D, lds_par, stg);
  LaneState<float> L = load_state<float>(c, D.base, D.leg);
  StepCtl4<float> S = load_ctl4<float>(c, K, D.ctl, D.ictl, D.legctl);
  TickPar4<float> tp = load_tick_par4<float>(c);
  V3<float> fext = {0.0f, 0.0f, 0.0f};
  if (!PLAIN && K.ext_force) fext = {c.ld_env(D.ctl, CT_FEXT + 0) + c.ld_env(D.ctl, CT_PUSH + 0), c.ld_env(D.ctl, CT_FEXT + 1) + c.ld_env(D.ctl, CT_PUSH + 1),
                                     c.ld_env(D.ctl, CT_FEXT + 2) + c.ld_env(D.ctl, CT_PUSH + 2)};   // set force + random push
  float act[3], hyb[12];
  const bool hybrid = K.motor_mode == 2 && action;   // rows of 60: per motor (q_des, kp, qd_des, kd, tau_ff)
#pragma unroll
  for (int j = 0; j < 3; j++) {
    act[j] = !action ? 0.0f : hybrid ? c.ld_row_lane(action, ETG_HYBRID_DIM, 5 * j, 15) : c.ld_row_lane(action, ETG_ACT_DIM, j, 3);
#pragma unroll
    for (int k = 0; k < 4; k++) hyb[4 * j + k] = hybrid ? c.ld_row_lane(action, ETG_HYBRID_DIM, 5 * j + 1 + k, 15) : 0.0f;
  }
  const float dflag = donef ? (float)donef[c.env] : 0.0f;
  stage4_commit(c, stg);
  float r, d;
#ifdef ETG_PROFILE_PHASES
  for (int k = 0; k < 16; k++) c.prof[k] = 0;
  c.prof_last = clock64();
  long long t_begin = c.prof_last;
#endif
  control_step_core(c, K, tp, fext, L, S, D.ring, D.etgp, act, dflag, obs, r, d, info, hybrid ? hyb : nullptr);
  store_ctl4(c, K, S, D.ctl, D.ictl, D.legctl);
  if (AUTO && d > 0.5f) {   // whole quads take this branch together
    const int N = K.n_env;
    if (NX.ok && NX.ok[c.env]) {   // parameters prepared for the next episode: install them (the cached settle below is theirs)
      for (int k = 0; k < PR_DERIVED; k++) D.par[(size_t)k * c.NL + c.gid] = NX.par[(size_t)k * c.NL + c.gid];
      for (int k = c.lane; k < ETG_DYN_DIM; k += 4) D.dyn[(size_t)c.env * ETG_DYN_DIM + k] = NX.dyn[(size_t)c.env * ETG_DYN_DIM + k];
      __builtin_amdgcn_s_waitcnt(0);          // the quad's reads of the flag are done before its lane 0 clears it
      __builtin_amdgcn_wave_barrier();
      if (c.lane == 0) NX.ok[c.env] = 0;
      stage_params(c, D, lds_par);            // the restart below reads the staged leg parameters: the new ones
    }
    L = load_state<float>(c, D.cache_base, D.cache_leg);
    L.p.x += D.reset_off[c.env] - D.cache_off[c.env];
    L.p.y += D.reset_off[N + c.env] - D.cache_off[N + c.env];
    if (D.cache_off[(size_t)FIN_OK * N + c.env] > 0.5f) restart_from_cache4(c, K, L, D.ctl, D.ictl, D.legctl, D.cache_off, obs);
    else reset_finish(c, K, L, D.ring, D.ctl, D.ictl, D.legctl, D.etgp, obs);
    if (c.lane == 0) {
      D.ictl[(size_t)IC_PUSH_LEFT * N + c.env] = 0;
      for (int k = 0; k < 3; k++) D.ctl[(size_t)(CT_PUSH + k) * N + c.env] = 0.0f;
    }
  }
  store_state(c, D.base, D.leg, L);
#ifdef ETG_PROFILE_PHASES
  if (c.gid == 0 && info) {  // overwrite the first info row with the cycle breakdown (debug build only)
    for (int k = 0; k < 16; k++) info[k] = (float)c.prof[k];
    info[16] = (float)(clock64() - t_begin);
  }
#endif
  if (c.lane == 0) {
    reward[c.env] = r;
    done[c.env] = d > 0.5f ? 1 : 0;
  }
}
template <bool FLAT, bool PLAIN, int BODY = 0>
__global__ void __launch_bounds__(BLOCK) k_step(KCfg K, DevState D, const float* action, const uint8_t* donef, float* obs,
                                                 float* reward, uint8_t* done, float* info) {
  __shared__ float lds_par[PR_N * BLOCK];
  step4_body<FLAT, PLAIN, false, BODY>(K, D, action, donef, obs, reward, done, info, lds_par);
}
template <bool FLAT, bool PLAIN, int BODY = 0>
__global__ void __launch_bounds__(BLOCK) k_step_ar(KCfg K, DevState D, const float* action, const uint8_t* donef, float* obs,
                                                    float* reward, uint8_t* done, float* info, NextDyn NX) {
  __shared__ float lds_par[PR_N * BLOCK];
  step4_body<FLAT, PLAIN, true, BODY>(K, D, action, donef, obs, reward, done, info, lds_par, NX);
}

// episode returns / lengths [N] for the caller, or nulls; cyc: one slot per wavefront of the launch receiving the shader-clock
// cycles the wave spent in the kernel (etg_rollout_wave_cycles: how unevenly the sweeps load the wavefronts), or null
struct StatOut { float* ret; int* len; long long* cyc; };
// n_steps open-loop control steps per launch (rollout_steps), the 4-lanes-per-robot counterpart of k_rollout16
template <bool FLAT, bool PLAIN, int BODY = 0>
__global__ void __launch_bounds__(BLOCK) k_rollout(KCfg K, DevState D, int n_steps, float* obs, StatOut so) {
  GpuCtxT<FLAT, PLAIN, BODY> c;
  if (!make_ctx(K, c)) return;
  const long long t0 = so.cyc ? clock64() : 0;
  __shared__ float lds_par[PR_N * BLOCK];
  stage_params(c, D, lds_par);
  LaneState<float> L = load_state<float>(c, D.base, D.leg);
  rollout_steps(c, K, L, D.base, D.leg, D.ring, D.ctl, D.ictl, D.legctl, D.etgp, n_steps, obs);   // (stores the state itself: stop_at_done)
  if (so.cyc && threadIdx.x == 0) so.cyc[blockIdx.x] = clock64() - t0;
  if (so.ret && c.lane == 0) {   // the last launch of a rollout hands the episode statistics to the caller itself (no extra launch)
    so.ret[c.env] = D.ctl[(size_t)CT_RET * K.n_env + c.env];
    so.len[c.env] = (int)D.ctl[(size_t)CT_LEN * K.n_env + c.env];
  }
}

// n_steps control steps over a caller-supplied action tape [n_steps][N][12] in one launch (etg_rollout_actions): the
// dynamics-identification evaluator's 2 x 100 steps of known joint targets (Dynamic_parallel_model.py:53-77) and teacher
// replays.  T: optional per-step outputs ([n_steps][N][...], null = not recorded).
struct TapeOut { float *q, *imu, *obs, *rew; uint8_t* done; };
template <bool FLAT, bool PLAIN, int BODY = 0>
__global__ void __launch_bounds__(BLOCK) k_rollout_actions(KCfg K, DevState D, int n_steps, const float* actions, float* obs, TapeOut T) {
  GpuCtxT<FLAT, PLAIN, BODY> c;
  if (!make_ctx(K, c)) return;
  __shared__ float lds_par[PR_N * BLOCK];
  stage_params(c, D, lds_par);
  LaneState<float> L = load_state<float>(c, D.base, D.leg);
  StepCtl4<float> S = load_ctl4<float>(c, K, D.ctl, D.ictl, D.legctl);
  TickPar4<float> tp = load_tick_par4<float>(c);
  V3<float> fext = {0.0f, 0.0f, 0.0f};
  if (!PLAIN && K.ext_force) fext = {c.ld_env(D.ctl, CT_FEXT + 0) + c.ld_env(D.ctl, CT_PUSH + 0), c.ld_env(D.ctl, CT_FEXT + 1) + c.ld_env(D.ctl, CT_PUSH + 1),
                                     c.ld_env(D.ctl, CT_FEXT + 2) + c.ld_env(D.ctl, CT_PUSH + 2)};
  const size_t N = K.n_env;
  float reward, done;
  const bool skip = K.stop_at_done != 0;
  for (int s = 0; s < n_steps; s++) {
    const float* a = actions + (size_t)s * N * ETG_ACT_DIM;
    const float act[3] = {c.ld_row_lane(a, ETG_ACT_DIM, 0, 3), c.ld_row_lane(a, ETG_ACT_DIM, 1, 3), c.ld_row_lane(a, ETG_ACT_DIM, 2, 3)};
    const bool last = s == n_steps - 1;
    const float was_alive = S.alive;
    // (stop_at_done: the steps after a robot's episode write reward 0 / done 1 to the tape and leave its other rows alone; once
    // every robot of the wave has finished only those two are written)
    if (!skip || c.any(was_alive > 0.5f)) {
      control_step_core(c, K, tp, fext, L, S, D.ring, D.etgp, act, 0.0f, (T.obs && !last) ? T.obs + (size_t)s * N * ETG_OBS_DIM : obs, reward, done,
                        (float*)nullptr, (const float*)nullptr, last || T.obs || T.imu, T.q ? T.q + (size_t)s * N * ETG_ACT_DIM : nullptr,
                        T.imu ? T.imu + (size_t)s * N * 6 : nullptr, skip, obs);
      // (a robot that ends mid-launch leaves its last row in the tape AND in `obs`; recorded rows and sensor noise exclude each
      // other -- etg_rollout_actions refuses the combination -- so with noise on every row in `obs` notes its stream position)
      if (skip) rollout_dead_store(c, K, L, was_alive, done, last, true, (int)K.noise_call + s, D.base, D.leg, D.ictl);
    } else {
      reward = 0.0f; done = 1.0f;
    }
    if (c.lane == 0) {
      if (T.rew) T.rew[(size_t)s * N + c.env] = reward;
      if (T.done) T.done[(size_t)s * N + c.env] = done > 0.5f ? 1 : 0;
    }
  }
  store_ctl4(c, K, S, D.ctl, D.ictl, D.legctl);
  rollout_store(c, K, L, S.alive, D.base, D.leg);
}

// ====================================================================== 16 lanes per robot
// One robot = one 16-lane DPP row (etg_core16.h): lane r = 4*leg + sub.  A workgroup is still one
// wave64 = 4 robots; 4096 robots -> 1024 workgroups = one wave on every SIMD of the chip.
constexpr int LDS16_FIELDS = PR_N;  // one LDS column per lane, indexed by PR_* (only the staged fields are filled)

struct GpuCtx16 {
  int env, r, leg, sub, sc, tid, N, NL;
  size_t col;        // 4*env + leg : column of the leg-level SoA arrays
  const float* lds;  // this lane's LDS column
  const float* gpar; // D.par: the tick constants are read from it directly (tpar*)
  int row_base = 0;  // first robot of the [*, rowlen] row buffers (obs tile in LDS: the workgroup's first robot)
#ifdef ETG_TRACE_TICKS
  mutable int trace_i = 0;
  __device__ __forceinline__ void trace_index(int i) const { trace_i = i; }
  __device__ __forceinline__ void trace_tick(const KCfg& K, const float* v) const {
    if (!K.trace || trace_i >= 16) return;
    float* row = K.trace + (((size_t)env * 16 + trace_i) * 16 + r) * 10;
    for (int k = 0; k < 10; k++) row[k] = v[k];
  }
#endif
  mutable bool gate = true;   // store gate: see GpuCtx
  __device__ __forceinline__ void set_gate(bool on) const { gate = on; }
  __device__ __forceinline__ void open_gate() const { gate = true; }
  __device__ __forceinline__ int sel_i(bool m, int a, int b) const { return m ? a : b; }
  // The lane's coordinates through an empty asm: everything derived from them after this point (the tick's lane masks, per-lane
  // constants selected from the config block: ~60 registers) is computed HERE again instead of being hoisted out of the caller's
  // loop -- the per-wave closed-loop kernel calls it at the top of every control step, so that none of it lives across the
  // policy tile (where the allocator would spill it to scratch).
  __device__ __forceinline__ void launder_lane() {
    asm volatile("" : "+v"(r), "+v"(leg), "+v"(sub), "+v"(sc));
    int lo = (int)col, e = env;             // (the array columns too: addresses are formed where they are used, not kept from the kernel's head)
    asm volatile("" : "+v"(lo), "+v"(e));
    col = (size_t)(unsigned)lo;
    env = e;
  }
  __device__ __forceinline__ float jointf() const { return sub < 3 ? 1.0f : 0.0f; }
  __device__ __forceinline__ bool sub_is(int j) const { return sub == j; }
  __device__ __forceinline__ bool leg_is(int j) const { return leg == j; }
  __device__ __forceinline__ bool any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
  // the wave-uniform tests of the body paths work on the wave mask of a compare: "any lane", "any lane of leg lp (in any robot
  // of the wave)" are one scalar AND against a constant each
#ifdef ETG_FORCE_BODY   // A/B build variant: every tick takes the body-row paths (what the emulation's force_body knob does)
  __device__ __forceinline__ bool any_body(bool) const { return true; }
  __device__ __forceinline__ unsigned body_mask(bool) const { return ~0u; }
#else
  __device__ __forceinline__ bool any_body(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0ull; }
  // the wave mask folded to 32 bits (a leg's four lanes sit at the same bits of both halves): a leg's test is ONE s_and_b32
  // against a 32-bit literal that sets SCC (a 64-bit mask needs two ANDs and a compare)
  __device__ __forceinline__ unsigned body_mask(bool b) const {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(b);
    return (unsigned)m | (unsigned)(m >> 32);
  }
#endif
  __device__ __forceinline__ bool mask_any(unsigned m) const { return m != 0u; }
  __device__ __forceinline__ bool mask_leg(unsigned m, int lp) const { return (m & (0x000F000Fu << (4 * lp))) != 0u; }
  __device__ __forceinline__ unsigned uniform_bits(unsigned v) const { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }   // a wave-uniform value, kept in an SGPR
  // (the builtin on the i1 itself: HIP's __any / __ballot take an int, and the compare's wave mask went through a
  // v_cndmask 0/1 + v_cmp_ne + a wait state on its way in -- three issue slots per sweep)
  __device__ __forceinline__ bool wave_any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0ull; }   // "does any robot of the wave need another sweep?"
  // "does any lane of MY robot (16-lane row) see b?" from the wave mask of the compare
  __device__ __forceinline__ bool vote_robot(unsigned long long m) const {
    const unsigned f = (tid & 16) ? 0xFFFF0000u : 0x0000FFFFu;
    return ((tid & 32) ? ((unsigned)(m >> 32) & f) : ((unsigned)m & f)) != 0u;
  }
  __device__ __forceinline__ bool robot_any(bool b) const { return vote_robot(__builtin_amdgcn_ballot_w64(b)); }
  // the residual rule's votes (etg_core16.h: sweep_and_test): one wave mask per compare, OR-ed as scalars
  __device__ __forceinline__ unsigned long long vote(bool b) const { return __builtin_amdgcn_ballot_w64(b); }
  __device__ __forceinline__ unsigned long long vote_or(unsigned long long a, unsigned long long b) const { return a | b; }
  __device__ __forceinline__ bool vote_wave(unsigned long long m) const { return m != 0ull; }
  __device__ __forceinline__ void fence() const { asm volatile(""); }   // keeps two tests two branches
  __device__ __forceinline__ int uniform_int(float a) const { return (int)a; }
  __device__ __forceinline__ float par(int k) const { return lds[k * 64]; }
  __device__ __forceinline__ float par_joint(int base) const { return lds[(base + sc) * 64]; }
  __device__ __forceinline__ float tpar(int k) const { return gpar[(size_t)k * NL + col]; }
  __device__ __forceinline__ float tpar_joint(int base) const { return gpar[(size_t)(base + sc) * NL + col]; }
  __device__ __forceinline__ float tpar_link(int k) const { return sub < 3 ? gpar[(size_t)(PR_LINK + 10 * sub + k) * NL + col] : 0.0f; }
  // ---- quad (= leg) exchanges
  __device__ __forceinline__ float qb(float a, int j) const {
    switch (j) { case 0: return dpp_<0x00>(a); case 1: return dpp_<0x55>(a); case 2: return dpp_<0xAA>(a); default: return dpp_<0xFF>(a); }
  }
  __device__ __forceinline__ float qup1(float a) const { return dpp_<0xF9>(a); }     // quad_perm [1,2,3,3]
  __device__ __forceinline__ float qup2(float a) const { return dpp_<0xFE>(a); }     // quad_perm [2,3,3,3]
  __device__ __forceinline__ float qdn1(float a) const { return dpp_<0x93>(a); }     // quad_perm [3,0,1,2]
  __device__ __forceinline__ float qdn2(float a) const { return dpp_<0x4F>(a); }     // quad_perm [3,3,0,1]
  __device__ __forceinline__ float qswap12(float a) const { return dpp_<0xD8>(a); }  // quad_perm [0,2,1,3]
  __device__ __forceinline__ float qsum(float a) const { float t = a + dpp_<0xB1>(a); return t + dpp_<0x4E>(t); }
  // ---- row (= robot) exchanges: xor1, xor2, row_half_mirror, row_mirror -- order-symmetric, so the
  // result is bit-identical on the 16 lanes
  // six row sums advanced stage by stage: every DPP reads a value written six instructions earlier
  __device__ __forceinline__ void sum16xn(float* v, int n) const {   // n is a compile-time constant at the call sites
#pragma unroll
    for (int k = 0; k < n; k++) v[k] = v[k] + dpp_<0xB1>(v[k]);
#pragma unroll
    for (int k = 0; k < n; k++) v[k] = v[k] + dpp_<0x4E>(v[k]);
#pragma unroll
    for (int k = 0; k < n; k++) v[k] = v[k] + dpp_<0x141>(v[k]);
#pragma unroll
    for (int k = 0; k < n; k++) v[k] = v[k] + dpp_<0x140>(v[k]);
  }
  __device__ __forceinline__ void sum16x6(float* v) const {
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = v[k] + dpp_<0xB1>(v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = v[k] + dpp_<0x4E>(v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = v[k] + dpp_<0x141>(v[k]);
#pragma unroll
    for (int k = 0; k < 6; k++) v[k] = v[k] + dpp_<0x140>(v[k]);
  }
  __device__ __forceinline__ float sum16(float a) const {
    float t = a + dpp_<0xB1>(a); t = t + dpp_<0x4E>(t); t = t + dpp_<0x141>(t); return t + dpp_<0x140>(t);
  }
  __device__ __forceinline__ float max16(float a) const {
    float t = fmaxf(a, dpp_<0xB1>(a)); t = fmaxf(t, dpp_<0x4E>(t)); t = fmaxf(t, dpp_<0x141>(t)); return fmaxf(t, dpp_<0x140>(t));
  }
  __device__ __forceinline__ float rbcast(float a, int r0) const {  // row_newbcast:r0 (constant after unrolling)
    switch (r0) {
      case 0: return dpp_<0x150>(a); case 1: return dpp_<0x151>(a); case 2: return dpp_<0x152>(a); case 3: return dpp_<0x153>(a);
      case 4: return dpp_<0x154>(a); case 5: return dpp_<0x155>(a); case 6: return dpp_<0x156>(a); case 7: return dpp_<0x157>(a);
      case 8: return dpp_<0x158>(a); case 9: return dpp_<0x159>(a); case 10: return dpp_<0x15A>(a); case 11: return dpp_<0x15B>(a);
      case 12: return dpp_<0x15C>(a); case 13: return dpp_<0x15D>(a); case 14: return dpp_<0x15E>(a); default: return dpp_<0x15F>(a);
    }
  }
  // value of the same sub-lane in leg (leg + kk) mod 4: row_ror:n delivers lane (i - n) mod 16
  __device__ __forceinline__ float legrot(float a, int kk) const {
    switch (kk) { case 0: return a; case 1: return dpp_<0x12C>(a); case 2: return dpp_<0x128>(a); default: return dpp_<0x124>(a); }
  }
  __device__ __forceinline__ void quad_outer(float a, float b, float* acc) const {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 v = {acc[0], acc[1], acc[2], acc[3]};
    v = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, v, 0, 0, 0);
    acc[0] = v[0]; acc[1] = v[1]; acc[2] = v[2]; acc[3] = v[3];
  }
  // ---- fused broadcast-multiply-accumulate: acc += x@lane(r0 of the row / j of the quad) * y in ONE VOP2-DPP
  // instruction.  The compiler's DPP combiner only folds v_mov_dpp into mul/add (FMAs are still VOP3 when it
  // runs), so the FMA form is spelled out.  HAZARD: the hardware wants 2 wait states between the VALU write
  // of x and a DPP read of it, and the compiler does not see into the asm -- callers pass an x that went
  // through dpp_ready() (one s_nop 1 for a whole group of sources).
#define ETG_FMAC_DPP(CTRL) asm("v_fmac_f32_dpp %0, %1, %2 " CTRL " row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(acc) : "v"(x), "v"(y))
  __device__ __forceinline__ void fmac_rbcast(float& acc, float x, float y, int r0) const {
    switch (r0) {
      case 0: ETG_FMAC_DPP("row_newbcast:0"); break;   case 1: ETG_FMAC_DPP("row_newbcast:1"); break;
      case 2: ETG_FMAC_DPP("row_newbcast:2"); break;   case 3: ETG_FMAC_DPP("row_newbcast:3"); break;
      case 4: ETG_FMAC_DPP("row_newbcast:4"); break;   case 5: ETG_FMAC_DPP("row_newbcast:5"); break;
      case 6: ETG_FMAC_DPP("row_newbcast:6"); break;   case 7: ETG_FMAC_DPP("row_newbcast:7"); break;
      case 8: ETG_FMAC_DPP("row_newbcast:8"); break;   case 9: ETG_FMAC_DPP("row_newbcast:9"); break;
      case 10: ETG_FMAC_DPP("row_newbcast:10"); break; case 11: ETG_FMAC_DPP("row_newbcast:11"); break;
      case 12: ETG_FMAC_DPP("row_newbcast:12"); break; case 13: ETG_FMAC_DPP("row_newbcast:13"); break;
      case 14: ETG_FMAC_DPP("row_newbcast:14"); break; default: ETG_FMAC_DPP("row_newbcast:15"); break;
    }
  }
  __device__ __forceinline__ void fmac_qb(float& acc, float x, float y, int j) const {
    switch (j) {
      case 0: ETG_FMAC_DPP("quad_perm:[0,0,0,0]"); break; case 1: ETG_FMAC_DPP("quad_perm:[1,1,1,1]"); break;
      case 2: ETG_FMAC_DPP("quad_perm:[2,2,2,2]"); break; default: ETG_FMAC_DPP("quad_perm:[3,3,3,3]"); break;
    }
  }
  // the 12 warm-start terms in one asm block: consecutive inline-asm statements that accumulate into one register
  // get a compiler-inserted wait state each (it cannot see that the accumulator is not the DPP operand)
  __device__ __forceinline__ void fmac_rbcast12(float& acc, float x, const float* a) const {
#define ETG_L(R, N) "v_fmac_f32_dpp %0, %1, %" #N " row_newbcast:" #R " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
    asm(ETG_L(0, 2) ETG_L(1, 3) ETG_L(2, 4) ETG_L(4, 5) ETG_L(5, 6) ETG_L(6, 7) ETG_L(8, 8) ETG_L(9, 9) ETG_L(10, 10)
        ETG_L(12, 11) ETG_L(13, 12) ETG_L(14, 13)
        : "+&v"(acc)
        : "v"(x), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11]));
#undef ETG_L
  }
  // body friction rows (etg_core16.h: body_friction): acc + sum over the 16 first rows of x@lane * a[lane] + sum over the eight
  // second rows (the t1 / t2 lanes of the four legs) of y@lane * b[2 leg + t].  24 broadcast-FMAs into ONE register would be a
  // dependent chain (a lone wave sits out every link): four partial sums advance side by side and are added at the end.
  __device__ __forceinline__ float row2_velocity(float acc, float x, const float* a, float y, const float* b) const {
    float p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
#define ETG_L(ACC, SRC, R, N) "v_fmac_f32_dpp %" #ACC ", %" #SRC ", %" #N " row_newbcast:" #R " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
    asm(ETG_L(0, 4, 0, 6) ETG_L(1, 4, 1, 7) ETG_L(2, 4, 2, 8) ETG_L(3, 4, 3, 9) ETG_L(0, 4, 4, 10) ETG_L(1, 4, 5, 11) ETG_L(2, 4, 6, 12) ETG_L(3, 4, 7, 13)
        ETG_L(0, 4, 8, 14) ETG_L(1, 4, 9, 15) ETG_L(2, 4, 10, 16) ETG_L(3, 4, 11, 17) ETG_L(0, 4, 12, 18) ETG_L(1, 4, 13, 19) ETG_L(2, 4, 14, 20) ETG_L(3, 4, 15, 21)
        ETG_L(0, 5, 1, 22) ETG_L(1, 5, 2, 23) ETG_L(2, 5, 5, 24) ETG_L(3, 5, 6, 25) ETG_L(0, 5, 9, 26) ETG_L(1, 5, 10, 27) ETG_L(2, 5, 13, 28) ETG_L(3, 5, 14, 29)
        : "+&v"(acc), "+&v"(p1), "+&v"(p2), "+&v"(p3)     // (early-clobber: an input holding the same VALUE as an accumulator's start --
                                                            // y = 0 and p3 = 0 -- must not share its register: it is read after the writes)
        : "v"(x), "v"(y), "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]),
          "v"(a[10]), "v"(a[11]), "v"(a[12]), "v"(a[13]), "v"(a[14]), "v"(a[15]), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]),
          "v"(b[5]), "v"(b[6]), "v"(b[7]));
#undef ETG_L
    return (acc + p1) + (p2 + p3);
  }
#undef ETG_FMAC_DPP
  // ---- the contact solve's sweep, hand-scheduled (etg_core16.h: pgs_sweep; kAsmSweep).  The sweep is one serial chain --
  // candidate -> broadcast -> apply, row after row -- and every broadcast reads a register the instruction before wrote: the
  // hardware wants 2 wait states between a VALU write and a DPP read of it.  The compiler's version spends a v_mov_b32_dpp plus
  // an s_nop 1 per broadcast (17 + 16 of 104 instructions, 120 issue slots per sweep); here the broadcast is the DPP operand
  // of the applying v_fmac itself and the owner's own impulse update sits in one of the two wait states: 5 slots per normal row,
  // 16 per friction pair, 84 per sweep.  Same arithmetic as the C++ statement in pgs_sweep (which the emulation, the body-row
  // and the pyramid instantiations keep).
#define ETG_DPPC " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
  __device__ __forceinline__ void pgs_normals(float& lam, float& u, float iA, float c0, const float (&A)[4][3], const float* mk0) const {
    float t, d;
#define ETG_ROW(LP, AOP, MOP)                                                    \
    "v_fma_f32 %[t], -%[u], %[iA], %[c0]\n"        /* t = c0 - u / A            */ \
    "v_max_f32_e64 %[d], -%[lam], %[t]\n"          /* d = max(-lam, t)          */ \
    "v_fmac_f32_e32 %[lam], %[" MOP "], %[d]\n"    /* the owner commits         */ \
    "s_nop 0\n"                                                                    \
    "v_fmac_f32_dpp %[u], %[d], %[" AOP "] row_newbcast:" #LP ETG_DPPC
    asm(ETG_ROW(0, "a0", "m0") ETG_ROW(4, "a1", "m1") ETG_ROW(8, "a2", "m2") ETG_ROW(12, "a3", "m3")
        : [lam] "+&v"(lam), [u] "+&v"(u), [t] "=&v"(t), [d] "=&v"(d)
        : [iA] "v"(iA), [c0] "v"(c0), [a0] "v"(A[0][0]), [a1] "v"(A[1][0]), [a2] "v"(A[2][0]), [a3] "v"(A[3][0]),
          [m0] "v"(mk0[0]), [m1] "v"(mk0[1]), [m2] "v"(mk0[2]), [m3] "v"(mk0[3]));
  }
  // the same with the four body normal rows (aux lanes) after the feet's: Bullet's order -- every normal row, then friction
  __device__ __forceinline__ void pgs_normals_body(float& lam, float& u, float iA, float c0, const float (&A)[4][3], const float* Ak,
                                                   const float* mk0, const float* mk3) const {
    float t, d;
    asm(ETG_ROW(0, "a0", "m0") ETG_ROW(4, "a1", "m1") ETG_ROW(8, "a2", "m2") ETG_ROW(12, "a3", "m3")
        ETG_ROW(3, "k0", "n0") ETG_ROW(7, "k1", "n1") ETG_ROW(11, "k2", "n2") ETG_ROW(15, "k3", "n3")
        : [lam] "+&v"(lam), [u] "+&v"(u), [t] "=&v"(t), [d] "=&v"(d)
        : [iA] "v"(iA), [c0] "v"(c0), [a0] "v"(A[0][0]), [a1] "v"(A[1][0]), [a2] "v"(A[2][0]), [a3] "v"(A[3][0]),
          [m0] "v"(mk0[0]), [m1] "v"(mk0[1]), [m2] "v"(mk0[2]), [m3] "v"(mk0[3]),
          [k0] "v"(Ak[0]), [k1] "v"(Ak[1]), [k2] "v"(Ak[2]), [k3] "v"(Ak[3]),
          [n0] "v"(mk3[0]), [n1] "v"(mk3[1]), [n2] "v"(mk3[2]), [n3] "v"(mk3[3]));
  }
#undef ETG_ROW
  // The sweeps of a tick with body rows track the velocity `u2` of the SECOND row set (the body contacts' friction rows,
  // etg_core16.h: finish_tick) through every phase: each impulse change is applied to u2 as well, by one more broadcast-FMA
  // that sits in a wait state the row needs anyway (the change of row i is applied to u2 while row i + 1 waits for its own
  // broadcast: the two changes alternate between the registers dA / dB).  8 rows: 41 issue slots, as many as without u2.
  // The block ends with the two quad broadcasts the friction phases start from -- the leg's foot normal impulse `lnq` and body
  // normal impulse `lbn` -- where the last rows' tails are their wait states (after the block the compiler fences each with an
  // s_nop 1: it does not see into the asm).
  __device__ __forceinline__ void pgs_normals_body2(float& lam, float& u, float& u2, float iA, float c0, const float (&A)[4][3],
                                                    const float* Ak, const float* mk0, const float* mk3, const float* bn, float& lnq,
                                                    float& lbn) const {
    float t, dA, dB;
#define ETG_ROW2(D, LP, AOP, MOP, FILL)                                            \
    "v_fma_f32 %[t], -%[u], %[iA], %[c0]\n"                                        \
    "v_max_f32_e64 %[" D "], -%[lam], %[t]\n"                                      \
    "v_fmac_f32_e32 %[lam], %[" MOP "], %[" D "]\n"                                \
    FILL                                                                           \
    "v_fmac_f32_dpp %[u], %[" D "], %[" AOP "] row_newbcast:" #LP ETG_DPPC
#define ETG_U2(D, LP, BOP) "v_fmac_f32_dpp %[u2], %[" D "], %[" BOP "] row_newbcast:" #LP ETG_DPPC
    asm(ETG_ROW2("dA", 0, "a0", "m0", "s_nop 0\n")
        ETG_ROW2("dB", 4, "a1", "m1", ETG_U2("dA", 0, "b0"))
        ETG_ROW2("dA", 8, "a2", "m2", ETG_U2("dB", 4, "b1"))
        ETG_ROW2("dB", 12, "a3", "m3", ETG_U2("dA", 8, "b2"))
        ETG_ROW2("dA", 3, "k0", "n0", ETG_U2("dB", 12, "b3"))
        ETG_ROW2("dB", 7, "k1", "n1", ETG_U2("dA", 3, "b4"))
        ETG_ROW2("dA", 11, "k2", "n2", ETG_U2("dB", 7, "b5"))
        ETG_ROW2("dB", 15, "k3", "n3", ETG_U2("dA", 11, "b6"))
        ETG_U2("dB", 15, "b7")
        "v_mov_b32_dpp %[lnq], %[lam] quad_perm:[0,0,0,0]" ETG_DPPC
        "v_mov_b32_dpp %[lbn], %[lam] quad_perm:[3,3,3,3]" ETG_DPPC
        : [lam] "+&v"(lam), [u] "+&v"(u), [u2] "+&v"(u2), [t] "=&v"(t), [dA] "=&v"(dA), [dB] "=&v"(dB), [lnq] "=&v"(lnq), [lbn] "=&v"(lbn)
        : [iA] "v"(iA), [c0] "v"(c0), [a0] "v"(A[0][0]), [a1] "v"(A[1][0]), [a2] "v"(A[2][0]), [a3] "v"(A[3][0]),
          [m0] "v"(mk0[0]), [m1] "v"(mk0[1]), [m2] "v"(mk0[2]), [m3] "v"(mk0[3]),
          [k0] "v"(Ak[0]), [k1] "v"(Ak[1]), [k2] "v"(Ak[2]), [k3] "v"(Ak[3]),
          [n0] "v"(mk3[0]), [n1] "v"(mk3[1]), [n2] "v"(mk3[2]), [n3] "v"(mk3[3]),
          [b0] "v"(bn[0]), [b1] "v"(bn[1]), [b2] "v"(bn[2]), [b3] "v"(bn[3]), [b4] "v"(bn[4]), [b5] "v"(bn[5]), [b6] "v"(bn[6]), [b7] "v"(bn[7]));
#undef ETG_ROW2
  }
  // friction pairs of the four feet on the disc: iA / lim already carry the "normal impulse > 0" condition (iA = 0 and
  // lim = 1e30 where it does not hold: the candidate is the current impulse, the scale 1, the change an exact zero)
  __device__ __forceinline__ void pgs_tangents_disc(float& lam, float& u, float iA, float lim, const float (&A)[4][3], const float* mt) const {
    float lc, sq, sc, dl;
#define ETG_PAIR(R1, R2, A1, A2, MOP)                                                                  \
    "v_fma_f32 %[lc], -%[u], %[iA], %[lam]\n"                /* candidate of this lane's row        */ \
    "v_fmaak_f32 %[sq], %[lc], %[lc], 0x0da24260\n"          /* lc^2 + 1e-30                       */ \
    "s_nop 1\n"                                                                                        \
    "v_add_f32_dpp %[sq], %[sq], %[sq] quad_perm:[0,2,1,3]" ETG_DPPC /* + the other row's           */ \
    "v_rsq_f32_e32 %[sq], %[sq]\n"                                                                     \
    "s_nop 0\n"                                                                                        \
    "v_mul_f32_e32 %[sc], %[lim], %[sq]\n"                                                             \
    "v_min_f32_e32 %[sc], 1.0, %[sc]\n"                      /* projection on the disc             */ \
    "v_fma_f32 %[dl], %[lc], %[sc], -%[lam]\n"                                                         \
    "v_fmac_f32_e32 %[lam], %[" MOP "], %[dl]\n"             /* the owners commit                  */ \
    "s_nop 0\n"                                                                                        \
    "v_fmac_f32_dpp %[u], %[dl], %[" A1 "] row_newbcast:" #R1 ETG_DPPC                                  \
    "v_fmac_f32_dpp %[u], %[dl], %[" A2 "] row_newbcast:" #R2 ETG_DPPC
    asm(ETG_PAIR(1, 2, "a01", "a02", "m0") ETG_PAIR(5, 6, "a11", "a12", "m1") ETG_PAIR(9, 10, "a21", "a22", "m2") ETG_PAIR(13, 14, "a31", "a32", "m3")
        : [lam] "+&v"(lam), [u] "+&v"(u), [lc] "=&v"(lc), [sq] "=&v"(sq), [sc] "=&v"(sc), [dl] "=&v"(dl)
        : [iA] "v"(iA), [lim] "v"(lim), [a01] "v"(A[0][1]), [a02] "v"(A[0][2]), [a11] "v"(A[1][1]), [a12] "v"(A[1][2]), [a21] "v"(A[2][1]),
          [a22] "v"(A[2][2]), [a31] "v"(A[3][1]), [a32] "v"(A[3][2]), [m0] "v"(mt[0]), [m1] "v"(mt[1]), [m2] "v"(mt[2]), [m3] "v"(mt[3]));
#undef ETG_PAIR
  }
  // the same with `u2` tracked (pgs_normals_body2): the two changes of pair p reach u2 in the two wait states pair p + 1 spends
  // between its squares and their exchange
  __device__ __forceinline__ void pgs_tangents_disc2(float& lam, float& u, float& u2, float iA, float lim, const float (&A)[4][3],
                                                     const float* mt, const float* bt) const {
    float lc, sq, sc, dA, dB;
#define ETG_PAIR2(D, R1, R2, A1, A2, MOP, FILL)                                                        \
    "v_fma_f32 %[lc], -%[u], %[iA], %[lam]\n"                                                          \
    "v_fmaak_f32 %[sq], %[lc], %[lc], 0x0da24260\n"                                                    \
    FILL                                                                                               \
    "v_add_f32_dpp %[sq], %[sq], %[sq] quad_perm:[0,2,1,3]" ETG_DPPC                                   \
    "v_rsq_f32_e32 %[sq], %[sq]\n"                                                                     \
    "s_nop 0\n"                                                                                        \
    "v_mul_f32_e32 %[sc], %[lim], %[sq]\n"                                                             \
    "v_min_f32_e32 %[sc], 1.0, %[sc]\n"                                                                \
    "v_fma_f32 %[" D "], %[lc], %[sc], -%[lam]\n"                                                      \
    "v_fmac_f32_e32 %[lam], %[" MOP "], %[" D "]\n"                                                    \
    "s_nop 0\n"                                                                                        \
    "v_fmac_f32_dpp %[u], %[" D "], %[" A1 "] row_newbcast:" #R1 ETG_DPPC                              \
    "v_fmac_f32_dpp %[u], %[" D "], %[" A2 "] row_newbcast:" #R2 ETG_DPPC
    asm(ETG_PAIR2("dA", 1, 2, "a01", "a02", "m0", "s_nop 1\n")
        ETG_PAIR2("dB", 5, 6, "a11", "a12", "m1", ETG_U2("dA", 1, "b0") ETG_U2("dA", 2, "b1"))
        ETG_PAIR2("dA", 9, 10, "a21", "a22", "m2", ETG_U2("dB", 5, "b2") ETG_U2("dB", 6, "b3"))
        ETG_PAIR2("dB", 13, 14, "a31", "a32", "m3", ETG_U2("dA", 9, "b4") ETG_U2("dA", 10, "b5"))
        ETG_U2("dB", 13, "b6") ETG_U2("dB", 14, "b7")
        : [lam] "+&v"(lam), [u] "+&v"(u), [u2] "+&v"(u2), [lc] "=&v"(lc), [sq] "=&v"(sq), [sc] "=&v"(sc), [dA] "=&v"(dA), [dB] "=&v"(dB)
        : [iA] "v"(iA), [lim] "v"(lim), [a01] "v"(A[0][1]), [a02] "v"(A[0][2]), [a11] "v"(A[1][1]), [a12] "v"(A[1][2]), [a21] "v"(A[2][1]),
          [a22] "v"(A[2][2]), [a31] "v"(A[3][1]), [a32] "v"(A[3][2]), [m0] "v"(mt[0]), [m1] "v"(mt[1]), [m2] "v"(mt[2]), [m3] "v"(mt[3]),
          [b0] "v"(bt[0]), [b1] "v"(bt[1]), [b2] "v"(bt[2]), [b3] "v"(bt[3]), [b4] "v"(bt[4]), [b5] "v"(bt[5]), [b6] "v"(bt[6]), [b7] "v"(bt[7]));
#undef ETG_PAIR2
#undef ETG_U2
  }
  // ONE friction pair of a body contact (second rows of leg lp, on its t1 / t2 lanes): the pair of pgs_tangents_disc on (lam2, u2),
  // its two changes applied to the second rows' velocity u2 AND to the first rows' u (etg_core16.h: body_friction)
  __device__ __forceinline__ void pgs_pair_body(float& lam2, float& u2, float& u, float iA, float lim, float at0, float at1, float bb0,
                                                float bb1, float mk, int lp) const {
    float lc, sq, sc, dl;
#define ETG_PAIRB(R1, R2)                                                                             \
    asm("v_fma_f32 %[lc], -%[u2], %[iA], %[lam]\n"                                                      \
        "v_fmaak_f32 %[sq], %[lc], %[lc], 0x0da24260\n"                                                 \
        "s_nop 1\n"                                                                                     \
        "v_add_f32_dpp %[sq], %[sq], %[sq] quad_perm:[0,2,1,3]" ETG_DPPC                                \
        "v_rsq_f32_e32 %[sq], %[sq]\n"                                                                  \
        "s_nop 0\n"                                                                                     \
        "v_mul_f32_e32 %[sc], %[lim], %[sq]\n"                                                          \
        "v_min_f32_e32 %[sc], 1.0, %[sc]\n"                                                             \
        "v_fma_f32 %[dl], %[lc], %[sc], -%[lam]\n"                                                      \
        "v_fmac_f32_e32 %[lam], %[mk], %[dl]\n"                                                         \
        "s_nop 0\n"                                                                                     \
        "v_fmac_f32_dpp %[u], %[dl], %[at0] row_newbcast:" #R1 ETG_DPPC                                 \
        "v_fmac_f32_dpp %[u2], %[dl], %[bb0] row_newbcast:" #R1 ETG_DPPC                                \
        "v_fmac_f32_dpp %[u], %[dl], %[at1] row_newbcast:" #R2 ETG_DPPC                                 \
        "v_fmac_f32_dpp %[u2], %[dl], %[bb1] row_newbcast:" #R2 ETG_DPPC                                \
        : [lam] "+&v"(lam2), [u2] "+&v"(u2), [u] "+&v"(u), [lc] "=&v"(lc), [sq] "=&v"(sq), [sc] "=&v"(sc), [dl] "=&v"(dl)           \
        : [iA] "v"(iA), [lim] "v"(lim), [at0] "v"(at0), [at1] "v"(at1), [bb0] "v"(bb0), [bb1] "v"(bb1), [mk] "v"(mk))
    switch (lp) { case 0: ETG_PAIRB(1, 2); break; case 1: ETG_PAIRB(5, 6); break; case 2: ETG_PAIRB(9, 10); break; default: ETG_PAIRB(13, 14); break; }
#undef ETG_PAIRB
  }
#undef ETG_DPPC
  // every value that later feeds fmac_rbcast / fmac_qb as the broadcast source passes through here: the
  // asm "modifies" them, so their producers are ordered before it and the DPP reads after it
  __device__ __forceinline__ void dpp_ready10(float* z, float* hj, float* lam) const {
    asm volatile("s_nop 1" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]), "+v"(z[4]), "+v"(z[5]), "+v"(hj[0]), "+v"(hj[1]), "+v"(hj[2]), "+v"(lam[0]));
  }
  // 32 floats of per-lane scratch in the lane's LDS parameter column: the fields that the 16-lane kernels never stage
  // (kStaged16 below: the link block PR_LINK..PR_LINK+29 and the gains go straight to registers through tpar*)
  __device__ __forceinline__ static int slotb_field(int k) { return k < 30 ? PR_LINK + k : PR_KP + (k - 30); }
  __device__ __forceinline__ void slotb_st(int k, float v) const { const_cast<float*>(lds)[slotb_field(k) * 64] = v; }
  __device__ __forceinline__ float slotb_ld(int k) const { return lds[slotb_field(k) * 64]; }
  // an optimisation barrier without an instruction: the value's producer cannot be fused into its consumers (etg_core16.h)
  __device__ __forceinline__ void opaque(float& v) const { asm volatile("" : "+v"(v)); }
  __device__ __forceinline__ void opaque3(float* v) const { asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])); }
  __device__ __forceinline__ void dpp_ready(float* v, int n) const {
    if (n == 6) asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
    else if (n == 3) asm volatile("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
    else asm volatile("s_nop 1" : "+v"(v[0]));
  }
  // ---- memory (same SoA arrays as the 4-lane kernels)
  __device__ __forceinline__ float ld_joint(const float* p, int f0) const { return p[(size_t)(f0 + sc) * NL + col]; }
  __device__ __forceinline__ void st_joint(float* p, int f0, float v) const { if (gate && sub < 3) p[(size_t)(f0 + sub) * NL + col] = v; }
  // four fields f0 .. f0 + 3, one per lane of the leg (LG_LAM .. LG_LAMB: the foot's three impulses and the body contact's normal)
  __device__ __forceinline__ float ld_quad(const float* p, int f0) const { return p[(size_t)(f0 + sub) * NL + col]; }
  __device__ __forceinline__ void st_quad(float* p, int f0, float v) const { if (gate) p[(size_t)(f0 + sub) * NL + col] = v; }
  __device__ __forceinline__ float ld_legf(const float* p, int f) const { return p[(size_t)f * NL + col]; }
  __device__ __forceinline__ void st_legf(float* p, int f, float v) const { if (gate && sub == 0) p[(size_t)f * NL + col] = v; }
  __device__ __forceinline__ float ld_env(const float* p, int f) const { return p[(size_t)f * N + env]; }
  // field f0 + stride * sub of the per-robot array: the 4 sub-lanes of a leg split a table between them
  __device__ __forceinline__ float ld_env_sub(const float* p, int f0, int stride) const { return p[(size_t)(f0 + stride * sub) * N + env]; }
  __device__ __forceinline__ void st_env(float* p, int f, float v) const { if (gate && r == 0) p[(size_t)f * N + env] = v; }
  __device__ __forceinline__ int ld_env_i(const int* p, int f) const { return p[(size_t)f * N + env]; }
  __device__ __forceinline__ void st_env_i(int* p, int f, int v) const { if (gate && r == 0) p[(size_t)f * N + env] = v; }
  __device__ __forceinline__ void st_ring_joint(float* rg, int slot, int k0, float v) const { if (gate && sub < 3) rg[((size_t)slot * 8 + k0 + sub) * NL + col] = v; }
  __device__ __forceinline__ void st_ring_aux(float* rg, int slot, int k, float v) const { if (gate && sub == 3) rg[((size_t)slot * 8 + k) * NL + col] = v; }
  __device__ __forceinline__ float ld_ring_joint(const float* rg, int slot, int k0) const { return rg[((size_t)slot * 8 + k0 + sc) * NL + col]; }
  __device__ __forceinline__ float ld_ring_k(const float* rg, int slot, int k) const { return rg[((size_t)slot * 8 + k) * NL + col]; }
  __device__ __forceinline__ float ld_row_joint(const float* p, int rowlen, int col0) const { return sub < 3 ? p[(size_t)(env - row_base) * rowlen + col0 + 3 * leg + sub] : 0.0f; }
  // element k of this lane's motor in rows of `stride` values per motor (HYBRID commands: stride 5)
  __device__ __forceinline__ float ld_row_motor(const float* p, int rowlen, int stride, int k) const { return sub < 3 ? p[(size_t)(env - row_base) * rowlen + stride * (3 * leg + sub) + k] : 0.0f; }
  __device__ __forceinline__ void st_row_joint(float* p, int rowlen, int col0, float v) const { if (gate && sub < 3) p[(size_t)(env - row_base) * rowlen + col0 + 3 * leg + sub] = v; }
  __device__ __forceinline__ void st_row_leg(float* p, int rowlen, int col0, float v) const { if (gate && sub == 0) p[(size_t)(env - row_base) * rowlen + col0 + leg] = v; }
  __device__ __forceinline__ void st_row_env(float* p, int rowlen, int c_, float v) const { if (gate && r == 0) p[(size_t)(env - row_base) * rowlen + c_] = v; }
  __device__ __forceinline__ float ld_row_env(const float* p, int rowlen, int c_) const { return p[(size_t)(env - row_base) * rowlen + c_]; }
  __device__ __forceinline__ void phase([[maybe_unused]] int id) const {
#ifndef ETG_NO_PHASE_BARRIER16
    __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef ETG_PROFILE_PHASES
    long long t = clock64();
    prof[id] += t - prof_last;
    prof_last = t;
    __builtin_amdgcn_sched_barrier(0);
#endif
  }
  // a marker that exists in the profiling build only (no scheduling barrier in the product build)
  __device__ __forceinline__ void phase_p([[maybe_unused]] int id) const {
#ifdef ETG_PROFILE_PHASES
    phase(id);
#endif
  }
#ifdef ETG_PROFILE_PHASES
  mutable long long prof[16];
  mutable long long prof_last;
#endif
  __device__ __forceinline__ void terrain(const KCfg& K, float x, float y, float& h, float& nx, float& ny, float& nz) const {
    if (K.terrain == 0) { h = 0.0f; nx = 0.0f; ny = 0.0f; nz = 1.0f; }
    else heightfield_query(K, env, x, y, h, nx, ny, nz);
  }
  __device__ __forceinline__ void terrain_fetch(const KCfg& K, float x, float y, float* tap) const { heightfield_fetch(K, env, x, y, tap); }
  __device__ __forceinline__ void terrain_finish(const KCfg& K, const float* tap, float& h, float& nx, float& ny, float& nz) const {
    heightfield_finish(K, tap, h, nx, ny, nz);
  }
};
// KNEE: the body contacts of EtgConfig.body_contacts 1 / 2 (etg_core16.h) are compiled in; SLOTB_LDS: see kSlotBLds.
// PLAIN: the default robot layer -- POSITION control, no action filter / interpolation, no torque limit, no command
// clip, no external force (plain_config) -- with those options compiled out: their never-taken branches and the
// registers they pin cost the step kernels ~3 % (measured A/B), so the common configuration gets its own instantiation.
template <bool FLAT, bool KNEE = false, bool PLAIN = false, bool SLOTB_LDS = false> struct GpuCtx16T : GpuCtx16 {
  static constexpr bool kFlat = FLAT;
  static constexpr bool kKnee = KNEE;
  static constexpr bool kPlain = PLAIN;
  static constexpr bool kSlotBLds = SLOTB_LDS;   // the body friction rows' Delassus columns live in LDS (etg_core16.h: finish_tick)
  static constexpr bool kStepLocal = false;      // (GpuCtx16W: rollout constants are formed anew every control step)
#ifdef ETG_NO_ASM_SWEEP
  static constexpr bool kAsmSweep = false;
#else
  static constexpr bool kAsmSweep = true;     // pgs_normals / pgs_tangents_disc: the hand-scheduled sweep
#endif
};

// The context of the per-wave closed-loop kernels: the tick constants come from the lane's LDS column (staged in full at kernel
// start) instead of HBM, so that control_step can re-read them at the top of every control step -- they need not stay in
// registers across the policy tile (45 registers per lane that the allocator otherwise spills to scratch around the tile).
template <bool FLAT, bool KNEE, bool PLAIN> struct GpuCtx16W : GpuCtx16T<FLAT, KNEE, PLAIN> {
  static constexpr bool kStepLocal = true;
  __device__ __forceinline__ float tpar(int k) const { return this->lds[k * 64]; }
  __device__ __forceinline__ float tpar_joint(int base) const { return this->lds[(base + this->sc) * 64]; }
  __device__ __forceinline__ float tpar_link(int k) const { return this->sub < 3 ? this->lds[(PR_LINK + 10 * this->sub + k) * 64] : 0.0f; }
};
// register parking around the policy tile: the robot state and the control variables of the lane, in LDS (one column per lane)
constexpr int PARK16 = 36;
__device__ __forceinline__ void park16(float* p, const State16<float>& L, const StepCtl16<float>& S) {
  const float v[PARK16] = {L.p.x, L.p.y, L.p.z, L.qx, L.qy, L.qz, L.qw, L.wb.x, L.wb.y, L.wb.z, L.vb.x, L.vb.y, L.vb.z, L.q, L.qd, L.lam, L.contact,
                           __int_as_float(S.step_count), __int_as_float(S.tick), __int_as_float(S.has_last), S.last, S.lbx, S.lby, S.lbz, S.last_fwx,
                           S.ret, S.len, S.r0, S.r1, S.r2, S.fx0, S.fx1, S.fy0, S.fy1, 0.0f, 0.0f};
#pragma unroll
  for (int k = 0; k < PARK16; k++) p[k * 64] = v[k];
}
__device__ __forceinline__ void unpark16(const float* p, State16<float>& L, StepCtl16<float>& S) {
  float v[PARK16];
#pragma unroll
  for (int k = 0; k < PARK16; k++) v[k] = p[k * 64];
  L.p = {v[0], v[1], v[2]}; L.qx = v[3]; L.qy = v[4]; L.qz = v[5]; L.qw = v[6];
  L.wb = {v[7], v[8], v[9]}; L.vb = {v[10], v[11], v[12]}; L.q = v[13]; L.qd = v[14]; L.lam = v[15]; L.contact = v[16];
  S.step_count = __float_as_int(v[17]); S.tick = __float_as_int(v[18]); S.has_last = __float_as_int(v[19]);
  S.last = v[20]; S.lbx = v[21]; S.lby = v[22]; S.lbz = v[23]; S.last_fwx = v[24]; S.ret = v[25]; S.len = v[26];
  S.r0 = v[27]; S.r1 = v[28]; S.r2 = v[29]; S.fx0 = v[30]; S.fx1 = v[31]; S.fy0 = v[32]; S.fy1 = v[33];
}

// robot_block = index of the group of 4 robots this wave carries, lane = lane in the wave, lds_wave = the wave's
// own [LDS16_FIELDS][64] parameter staging area.  Three steps so that a caller can put its own loads between the staging
// loads and the LDS writes that have to wait for them (step16_body: one HBM round trip at the head of the launch, not two).
constexpr int kStaged16[] = {PR_O1, PR_O1 + 1, PR_O1 + 2, PR_SY, PR_LAT_N, PR_LAT_A, PR_BASE_FOOT, PR_BASE_FOOT + 1,
                             PR_BASE_FOOT + 2, PR_POSE, PR_POSE + 1, PR_POSE + 2, PR_EMEAN, PR_EMEAN + 1, PR_EMEAN + 2,
                             PR_ESTD, PR_ESTD + 1, PR_ESTD + 2, PR_HIPSIGN};
constexpr int kNStaged16 = sizeof(kStaged16) / sizeof(int);
__device__ __forceinline__ bool make_ctx16_fields(const KCfg& K, const DevState& D, GpuCtx16& c, float* lds_wave, int robot_block, int lane) {
  c.tid = lane;
  c.env = robot_block * 4 + (lane >> 4);
  c.r = lane & 15;
  c.leg = c.r >> 2;
  c.sub = c.r & 3;
  c.sc = c.sub < 2 ? c.sub : 2;
  c.N = K.n_env;
  c.NL = 4 * K.n_env;
  if (c.env >= c.N) return false;   // whole rows drop out together, so every DPP/MFMA group stays complete
  c.col = (size_t)4 * c.env + c.leg;
  c.gpar = D.par;
  c.lds = lds_wave + lane;
  return true;
}
// stage in LDS only the leg-level parameters that the once-per-step code reads (c.par / c.par_joint); the tick
// constants (gains, link block, gravity, trunk inertia, mu) go straight to registers through tpar*
__device__ __forceinline__ void stage16_issue(const DevState& D, const GpuCtx16& c, float (&v)[kNStaged16]) {
#pragma unroll
  for (int i = 0; i < kNStaged16; i++) v[i] = D.par[(size_t)kStaged16[i] * c.NL + c.col];
}
__device__ __forceinline__ void stage16_commit(const GpuCtx16& c, const float (&v)[kNStaged16]) {
  float* mine = const_cast<float*>(c.lds);
#pragma unroll
  for (int i = 0; i < kNStaged16; i++) mine[kStaged16[i] * 64] = v[i];
}
__device__ __forceinline__ bool make_ctx16_at(const KCfg& K, const DevState& D, GpuCtx16& c, float* lds_wave, int robot_block, int lane) {
  if (!make_ctx16_fields(K, D, c, lds_wave, robot_block, lane)) return false;
  float v[kNStaged16];
  stage16_issue(D, c, v);
  stage16_commit(c, v);
  return true;
}
__device__ __forceinline__ bool make_ctx16(const KCfg& K, const DevState& D, GpuCtx16& c, float* lds_all) {
  return make_ctx16_at(K, D, c, lds_all, xcd_contiguous_block(), threadIdx.x);
}

template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_settle16(KCfg K, DevState D, const uint8_t* mask) {
  __shared__ float lds_par[LDS16_FIELDS * BLOCK];
  GpuCtx16T<FLAT, KNEE, PLAIN> c;
  if (!make_ctx16(K, D, c, lds_par)) return;
  if ((mask && !mask[c.env]) || settle_cached<FLAT>(K, D, c.env)) return;   // whole rows drop out together
  State16<float> L;
  const float ox = D.reset_off[c.env], oy = D.reset_off[K.n_env + c.env];
  reset_settle16(c, K, L, D.ring, ox, oy);
  store_state16(c, D.cache_base, D.cache_leg, L);
  store_state16(c, D.base, D.leg, L);
  settle_mark_fresh(K, D, c.env, ox, oy);
}

// reset_finish16 from the cache: same stores, no arithmetic
template <class Ctx>
__device__ __forceinline__ void restart_from_cache16(const Ctx& c, const KCfg& K, State16<float>& L, float* ctl, int* ictl, float* legctl,
                                                     const float* fin, float* obs) {
  const int N = K.n_env;
  const float pose = c.jointf() * c.par_joint(PR_POSE);
  L.energy = 0.0f;
  c.st_env_i(ictl, IC_STEP, 0);
  c.st_env_i(ictl, IC_TICK, K.settle_ticks);
  c.st_env_i(ictl, IC_HAS_LAST, 0);
  c.st_env(ctl, CT_RET, 0.0f); c.st_env(ctl, CT_LEN, 0.0f); c.st_env(ctl, CT_ALIVE, 1.0f);
  c.st_env(ctl, CT_LAST_BASE + 0, L.p.x); c.st_env(ctl, CT_LAST_BASE + 1, L.p.y); c.st_env(ctl, CT_LAST_BASE + 2, L.p.z);
  c.st_joint(legctl, LC_LAST_QDES, pose);
  c.st_joint(legctl, LC_FX0, pose); c.st_joint(legctl, LC_FX1, pose);
  c.st_joint(legctl, LC_FY0, pose); c.st_joint(legctl, LC_FY1, pose);
  c.st_legf(legctl, LC_LAST_FOOT_X, fin[(size_t)(FIN_FWX + c.leg) * N + c.env]);
#pragma unroll
  for (int k = 0; k < 3; k++) c.st_env(ctl, CT_FIRST_RPY + k, fin[(size_t)(FIN_RPY + k) * N + c.env]);
  for (int k = c.r; k < ETG_OBS_DIM; k += 16) obs[(size_t)(c.env - c.row_base) * ETG_OBS_DIM + k] = fin[(size_t)(FIN_OBS + k) * N + c.env];
}

template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_finish16(KCfg K, DevState D, const uint8_t* mask, float* obs) {
  __shared__ float lds_par[LDS16_FIELDS * BLOCK];
  GpuCtx16T<FLAT, KNEE, PLAIN> c;
  if (!make_ctx16(K, D, c, lds_par)) return;
  if (mask && !mask[c.env]) return;
  State16<float> L = load_state16<float>(c, D.base, D.leg);
  reset_finish16(c, K, L, D.ring, D.ctl, D.ictl, D.legctl, D.etgp, obs);
  store_state16(c, D.base, D.leg, L);
}

// env.step for the 4 robots of a wave.  AUTO (etg_step_autoreset while every robot has a cached settle): a robot whose step
// ended its episode restarts inside the same launch -- cached settle -> registers, reset_finish (control state, episode
// accumulators, first observation over the step's row; the pre-reset ring readings come from KCfg.cring), pending push
// cleared.  reward / done / info stay the finished step's.
template <bool FLAT, bool KNEE, bool PLAIN, bool AUTO>
__device__ __forceinline__ void step16_body(const KCfg& K, const DevState& D, const float* action, const uint8_t* donef, float* obs,
                                            float* reward, uint8_t* done, float* info, float* lds_par, const NextDyn NX = NextDyn{nullptr, nullptr, nullptr}) {
  GpuCtx16T<FLAT, KNEE, PLAIN> c;
  if (!make_ctx16_fields(K, D, c, lds_par, K.block0 + xcd_contiguous_block(), threadIdx.x)) return;   // (block0: etg_step_range)
  // the head of a launch is a chain of cold loads (the L2s are invalidated at kernel boundaries): ALL of them -- staged
  // parameters, state, control state, tick constants, action, done flag -- are requested before the first use, so the launch
  // pays one HBM round trip here instead of one per group (the LDS writes of the staging used to wait in front of the rest)
  float stg[kNStaged16];
  stage16_issue(D, c, stg);
  State16<float> L = load_state16<float>(c, D.base, D.leg);
  StepCtl16<float> S = load_ctl16<float>(c, K, D.ctl, D.ictl, D.legctl);
  TickPar<float> tp = load_tick_par<float>(c);
  if (!PLAIN && K.ext_force) tp.fext = load_fext16<float>(c, D.ctl);
  const bool hybrid = K.motor_mode == 2 && action;   // rows of 60: per motor (q_des, kp, qd_des, kd, tau_ff)
  float act = !action ? 0.0f : hybrid ? c.ld_row_motor(action, ETG_HYBRID_DIM, 5, 0) : c.ld_row_joint(action, ETG_ACT_DIM, 0);
  float hyb[4];
#pragma unroll
  for (int k = 0; k < 4; k++) hyb[k] = hybrid ? c.ld_row_motor(action, ETG_HYBRID_DIM, 5, 1 + k) : 0.0f;
  const float dflag = donef ? (float)donef[c.env] : 0.0f;
  stage16_commit(c, stg);
  float r, d;
#ifdef ETG_PROFILE_PHASES
  for (int k = 0; k < 16; k++) c.prof[k] = 0;
  c.prof_last = clock64();
  long long t_begin = c.prof_last;
#endif
  control_step16_core(c, K, tp, L, S, D.ring, D.etgp, act, dflag, obs, r, d, info, hybrid ? hyb : nullptr);
  store_ctl16(c, K, S, D.ctl, D.ictl, D.legctl);
  if (AUTO && d > 0.5f) {   // whole 16-lane rows take this branch together (d is the robot's)
    const int N = K.n_env;
    if (NX.ok && NX.ok[c.env]) {   // parameters prepared for the next episode: install them (the cached settle below is theirs)
      for (int k = c.sub; k < PR_DERIVED; k += 4) D.par[(size_t)k * c.NL + c.col] = NX.par[(size_t)k * c.NL + c.col];   // a leg's 4 lanes share its column
      for (int k = c.r; k < ETG_DYN_DIM; k += 16) D.dyn[(size_t)c.env * ETG_DYN_DIM + k] = NX.dyn[(size_t)c.env * ETG_DYN_DIM + k];
      __builtin_amdgcn_s_waitcnt(0);          // the row's reads of the flag and its stores are done before lane 0 clears it
      __builtin_amdgcn_wave_barrier();
      if (c.r == 0) NX.ok[c.env] = 0;
      float stg2[kNStaged16];                 // the restart below reads the staged leg parameters: the new ones
      stage16_issue(D, c, stg2);
      stage16_commit(c, stg2);
    }
    L = load_state16<float>(c, D.cache_base, D.cache_leg);
    L.p.x += D.reset_off[c.env] - D.cache_off[c.env];        // non-zero only on flat ground (settle_cached)
    L.p.y += D.reset_off[N + c.env] - D.cache_off[N + c.env];
    if (D.cache_off[(size_t)FIN_OK * N + c.env] > 0.5f) restart_from_cache16(c, K, L, D.ctl, D.ictl, D.legctl, D.cache_off, obs);
    else reset_finish16(c, K, L, D.ring, D.ctl, D.ictl, D.legctl, D.etgp, obs);
    if (c.r == 0) {
      D.ictl[(size_t)IC_PUSH_LEFT * N + c.env] = 0;
      for (int k = 0; k < 3; k++) D.ctl[(size_t)(CT_PUSH + k) * N + c.env] = 0.0f;
    }
  }
  store_state16(c, D.base, D.leg, L);
#ifdef ETG_PROFILE_PHASES
  if ((c.env & 3) == 0 && c.r == 0 && info) {   // every wave reports, in the info row of its first robot
    float* row = info + (size_t)c.env * ETG_INFO_DIM;
    for (int k = 0; k < 16; k++) row[k] = (float)c.prof[k];
    row[16] = (float)(clock64() - t_begin);
  }
#endif
  if (c.r == 0) {
    reward[c.env] = r;
    done[c.env] = d > 0.5f ? 1 : 0;
  }
}
template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_step16(KCfg K, DevState D, const float* action, const uint8_t* donef, float* obs,
                                                   float* reward, uint8_t* done, float* info) {
  __shared__ float lds_par[LDS16_FIELDS * BLOCK];
  step16_body<FLAT, KNEE, PLAIN, false>(K, D, action, donef, obs, reward, done, info, lds_par);
}
template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_step16_ar(KCfg K, DevState D, const float* action, const uint8_t* donef, float* obs,
                                                      float* reward, uint8_t* done, float* info, NextDyn NX) {
  __shared__ float lds_par[LDS16_FIELDS * BLOCK];
  step16_body<FLAT, KNEE, PLAIN, true>(K, D, action, donef, obs, reward, done, info, lds_par, NX);
}

// n_steps open-loop control steps of every robot in one launch (rollout_steps16): state, control variables and
// tick constants stay in registers between the steps
template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_rollout16(KCfg K, DevState D, int n_steps, float* obs, StatOut so) {
  __shared__ float lds_par[LDS16_FIELDS * BLOCK];
#ifndef ETG_ROLLOUT16_STEP_LOCAL   // the default: rollout constants hoisted to the kernel's head, tick constants from HBM.  (A/B build
  // variant -DETG_ROLLOUT16_STEP_LOCAL: formed anew every control step like the tape and closed-loop kernels -- 343 registers
  // instead of 448 and 60 % fewer accumulator-register moves, and 0.4 % SLOWER on the same box: profiles/r06_ab_experiments.txt)
  GpuCtx16T<FLAT, KNEE, PLAIN> c;
  if (!make_ctx16(K, D, c, lds_par)) return;
#else
  GpuCtx16W<FLAT, KNEE, PLAIN> c;
  if (!make_ctx16_fields(K, D, c, lds_par, xcd_contiguous_block(), threadIdx.x)) return;
  for (int k = 0; k < PR_N; k++) lds_par[k * 64 + threadIdx.x] = D.par[(size_t)k * c.NL + c.col];   // the lane's whole parameter column
#endif
  const long long t0 = so.cyc ? clock64() : 0;
  State16<float> L = load_state16<float>(c, D.base, D.leg);
  rollout_steps16(c, K, L, D.base, D.leg, D.ring, D.ctl, D.ictl, D.legctl, D.etgp, n_steps, obs);   // (stores the state itself: stop_at_done)
  if (so.cyc && threadIdx.x == 0) so.cyc[blockIdx.x] = clock64() - t0;
  if (so.ret && c.r == 0) {   // the last launch of a rollout hands the episode statistics to the caller itself (no extra launch):
    so.ret[c.env] = D.ctl[(size_t)CT_RET * K.n_env + c.env];        // lane 0 of the robot's row re-reads what it has just stored
    so.len[c.env] = (int)D.ctl[(size_t)CT_LEN * K.n_env + c.env];
  }
}

// the 16-lane counterpart of k_rollout_actions
template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_rollout_actions16(KCfg K, DevState D, int n_steps, const float* actions, float* obs, TapeOut T) {
  __shared__ float lds_par[LDS16_FIELDS * BLOCK];
  GpuCtx16W<FLAT, KNEE, PLAIN> c;             // (rollout constants formed anew every control step: see rollout_steps16)
  if (!make_ctx16_fields(K, D, c, lds_par, xcd_contiguous_block(), threadIdx.x)) return;
  for (int k = 0; k < PR_N; k++) lds_par[k * 64 + threadIdx.x] = D.par[(size_t)k * c.NL + c.col];
  State16<float> L = load_state16<float>(c, D.base, D.leg);
  StepCtl16<float> S = load_ctl16<float>(c, K, D.ctl, D.ictl, D.legctl);
  V3<float> fext = {0.0f, 0.0f, 0.0f};
  if (!PLAIN && K.ext_force) fext = load_fext16<float>(c, D.ctl);
  const size_t N = K.n_env;
  float reward, done;
  const bool skip = K.stop_at_done != 0;
  for (int s = 0; s < n_steps; s++) {
    c.launder_lane();
    TickPar<float> tp = load_tick_par<float>(c);
    tp.fext = fext;
    const float act = c.ld_row_joint(actions + (size_t)s * N * ETG_ACT_DIM, ETG_ACT_DIM, 0);
    const bool last = s == n_steps - 1;
    const float was_alive = S.alive;
    if (!skip || c.any(was_alive > 0.5f)) {   // (see k_rollout_actions)
      control_step16_core(c, K, tp, L, S, D.ring, D.etgp, act, 0.0f, (T.obs && !last) ? T.obs + (size_t)s * N * ETG_OBS_DIM : obs, reward, done,
                          (float*)nullptr, (const float*)nullptr, last || T.obs || T.imu, T.q ? T.q + (size_t)s * N * ETG_ACT_DIM : nullptr,
                          T.imu ? T.imu + (size_t)s * N * 6 : nullptr, skip, obs);
      if (skip) rollout_dead_store16(c, K, L, was_alive, done, last, true, (int)K.noise_call + s, D.base, D.leg, D.ictl);
    } else {
      reward = 0.0f; done = 1.0f;
    }
    if (c.r == 0) {
      if (T.rew) T.rew[(size_t)s * N + c.env] = reward;
      if (T.done) T.done[(size_t)s * N + c.env] = done > 0.5f ? 1 : 0;
    }
  }
  store_ctl16(c, K, S, D.ctl, D.ictl, D.legctl);
  rollout_store16(c, K, L, S.alive, D.base, D.leg);
}

// Closed-loop rollout in one launch: a workgroup of 4 waves owns 16 robots = one 16-row tile of the policy MLP
// (policy_core.h).  Per control step: observations of the 16 robots (LDS) -> policy on MFMA (the 4 waves share
// the tile like k_policy) -> tanh(mean) * act_scale in LDS -> every wave runs its 4 robots' control step, writing
// the next observation back to LDS.  Nothing but the final observation, the ring and the episode accumulators
// touches HBM between steps.  (run_EStrain_episode / run_evaluate_episodes, train.py:182-249, with a fixed actor.)
struct PolicyW { const float4 *w1, *w2, *w3; const float *b1, *b2, *b3; int in_dim, out_dim, col0; };   // col0: first observation column the actor sees
// (the per-wave kernels: w1 = the k_pack_wave12 stream of both hidden layers, w2 unused, w3 / RecOut::w3s = the k_pack_head copies)
// per-step outputs of the recording variant ([n_steps][N][...]); noise != NULL: the STOCHASTIC actor of SAC.sample
// (alg/sac.py:65-76) on the caller's N(0,1) draws [n_steps][N][12]: x = mean + exp(clamp(log_std, -20, 2)) * noise,
// action = tanh(x); w3s / b3s = the packed log-std head (etg_policy_load_std)
struct RecOut { float *obs, *act, *rew; uint8_t* done; const float* noise; const float4* w3s; const float* b3s; };

// ONE body for the closed-loop kernel and its recording variant (REC: every step's observation, unscaled action, reward and done
// written out for the replay memory of the ES-SAC loop, run_EStrain_episode with es_rpm, train.py:213-249).  Two __global__
// symbols instantiate it: REC is a compile-time constant in each, so the plain kernel contains none of the recording code (a
// runtime flag in ONE kernel shifts its 512-register allocation; round 2 kept a 70-line copy for that reason, which let fixes
// diverge).  ps_mem: the [NWP][TM][16] partial sums of the log-std head (REC only).
template <bool FLAT, bool BF16, bool KNEE, bool PLAIN, bool REC>
__device__ __forceinline__ void rollout_policy16_body(const KCfg& K, const DevState& D, const PolicyW& P, int n_steps, float act_scale,
                                                      float* obs, const RecOut& R, float* bufA, float* bufB, float (*part)[pol::TM][16],
                                                      float (*part_s)[pol::TM][16], float (*act_lds)[16], float* obs_lds, float* lds_par, float* live_lds) {
  using namespace pol;
  constexpr int NWP = 4;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = xcd_contiguous_block();            // 16 robots; the host guarantees N % 16 == 0
  GpuCtx16T<FLAT, KNEE, PLAIN, true> c;
  make_ctx16_at(K, D, c, lds_par + wave * (LDS16_FIELDS * 64), 4 * tile + wave, lane);
  State16<float> L = load_state16<float>(c, D.base, D.leg);
  StepCtl16<float> S = load_ctl16<float>(c, K, D.ctl, D.ictl, D.legctl);
  TickPar<float> tp = load_tick_par<float>(c);
  if (!PLAIN && K.ext_force) tp.fext = load_fext16<float>(c, D.ctl);
  // current observation of the tile -> LDS
  for (int idx = tid; idx < TM * ETG_OBS_DIM; idx += 256) obs_lds[idx] = obs[(size_t)tile * TM * ETG_OBS_DIM + idx];
  float reward, done;
  // KCfg.stop_at_done: a finished robot is not simulated any more (control_step16_core).  Its row of the observation tile stays
  // its last one (the policy keeps evaluating it: a workgroup's 16 rows are one MFMA tile either way); a wave whose four robots
  // have all finished skips its control steps, a workgroup whose sixteen have leaves the loop.
  const bool skip = K.stop_at_done != 0;
  if (lane < 4) live_lds[4 * wave + lane] = 1.0f;
#ifdef ETG_PROFILE_PHASES
  long long pp[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pl = clock64();
#endif
  int s_at = 0;   // the step the loop stands at (== n_steps unless the workgroup left early)
  for (int s = 0; s < n_steps; s++, s_at = s) {
    if (skip) {
      if (!__syncthreads_or(S.alive > 0.5f)) break;   // (the barrier of the loop top, with the vote riding on it)
    } else {
      __syncthreads();
    }
    if (K.noise_on && s > 0) {   // sensor noise on the row the previous step left in LDS (the last one: the launch's epilogue)
      if (live_lds[tid >> 4] > 0.5f)   // ... if the step wrote one: a robot that had finished before it keeps its row as it is
        add_sensor_noise(K, tile * TM + (tid >> 4), K.noise_call + s - 1, tid & 15, &obs_lds[(tid >> 4) * ETG_OBS_DIM]);
      __syncthreads();
    }
    // fp32: every layer's first weight fragments are requested one stage EARLY (layer 1's before the observation tile is
    // staged, layer 2's before layer 1's MFMAs, the head's before layer 2's): a lone wave per SIMD has nothing else to hide
    // the L2 latency of a layer's first fragments behind
    WRing<4, NWP> ring1;
    WRing<HID / 16, NWP> ring2;
    HeadFrag<NWP> head;
    constexpr bool PRE = !BF16;
    if (PRE) ring_prefetch(ring1, P.w1, wave, lane);
    for (int idx = tid; idx < TM * 64; idx += 256) {   // obs tile, zero padded to the 64-wide K of layer 1
      const int r = idx >> 6, col = idx & 63;
      bufA[r * HS + col] = col < P.in_dim ? obs_lds[r * ETG_OBS_DIM + P.col0 + col] : 0.0f;
    }
    if (REC)   // the observation the actor acts on at this step (noise included), rows of the tile's 16 robots: coalesced
      for (int idx = tid; idx < TM * ETG_OBS_DIM; idx += 256) R.obs[((size_t)s * K.n_env + (size_t)tile * TM) * ETG_OBS_DIM + idx] = obs_lds[idx];
    __syncthreads();
#ifdef ETG_PROFILE_PHASES
    { long long t = clock64(); pp[0] += t - pl; pl = t; __builtin_amdgcn_sched_barrier(0); }
#endif
    if (PRE) ring_prefetch(ring2, P.w2, wave, lane);
    hidden_layer<BF16, 4, NWP, PRE>(bufA, P.w1, P.b1, bufB, wave, lane, &ring1);
    __syncthreads();
#ifdef ETG_PROFILE_PHASES
    { long long t = clock64(); pp[1] += t - pl; pl = t; __builtin_amdgcn_sched_barrier(0); }
#endif
    if (PRE) head_prefetch(head, P.w3, wave, lane);
    hidden_layer<BF16, HID / 16, NWP, PRE>(bufB, P.w2, P.b2, bufA, wave, lane, &ring2);
    __syncthreads();
#ifdef ETG_PROFILE_PHASES
    { long long t = clock64(); pp[2] += t - pl; pl = t; __builtin_amdgcn_sched_barrier(0); }
#endif
    output_partial<BF16, NWP, PRE>(bufA, P.w3, wave, lane, part, &head);
    if (REC && R.noise) output_partial<BF16, NWP>(bufA, R.w3s, wave, lane, part_s);
    __syncthreads();
#ifdef ETG_PROFILE_PHASES
    { long long t = clock64(); pp[3] += t - pl; pl = t; __builtin_amdgcn_sched_barrier(0); }
#endif
    {
      const int r = tid >> 4, cidx = tid & 15;          // 256 threads = 16 rows x 16 columns
      float v = ((part[0][r][cidx] + part[1][r][cidx]) + (part[2][r][cidx] + part[3][r][cidx])) + (cidx < P.out_dim ? P.b3[cidx] : 0.0f);
      if (REC && R.noise && cidx < ETG_ACT_DIM) {
        float ls = ((part_s[0][r][cidx] + part_s[1][r][cidx]) + (part_s[2][r][cidx] + part_s[3][r][cidx])) + R.b3s[cidx];
        ls = fminf(fmaxf(ls, -20.0f), 2.0f);
        v = v + expf(ls) * R.noise[((size_t)s * K.n_env + (size_t)tile * TM + r) * ETG_ACT_DIM + cidx];
      }
      const float t = tanhf(v);
      act_lds[r][cidx] = t * act_scale;
      if (REC && cidx < ETG_ACT_DIM) R.act[((size_t)s * K.n_env + (size_t)tile * TM + r) * ETG_ACT_DIM + cidx] = t;   // the UNSCALED action (train.py:159)
    }
    __syncthreads();
    const float action = c.sub < 3 ? act_lds[4 * wave + (lane >> 4)][3 * c.leg + c.sub] : 0.0f;
#ifdef ETG_PROFILE_PHASES
    { long long t = clock64(); pp[4] += t - pl; pl = t; __builtin_amdgcn_sched_barrier(0); }
#endif
    // the step code addresses observation rows by robot index: rows of the LDS tile start at the tile's first robot.
    // Every step writes its observation to the tile (plain ds_write, no generic pointer); the last one is copied out below.
    c.row_base = tile * TM;
    const float was_alive = S.alive;
    if (!skip || c.any(was_alive > 0.5f)) {
      control_step16_core(c, K, tp, L, S, D.ring, D.etgp, action, 0.0f, obs_lds, reward, done, (float*)nullptr, (const float*)nullptr, true,
                          (float*)nullptr, (float*)nullptr, skip);
      // (a row written mid-launch gets its sensor noise at the top of the next step; only the last step's rows wait for the epilogue)
      if (skip) rollout_dead_store16(c, K, L, was_alive, done, s == n_steps - 1, false, (int)K.noise_call + s, D.base, D.leg, D.ictl);
    } else {
      reward = 0.0f; done = 1.0f;
    }
    if (c.r == 0) live_lds[4 * wave + (lane >> 4)] = skip ? was_alive : 1.0f;   // did this step write the robot's row?
#ifdef ETG_PROFILE_PHASES
    { long long t = clock64(); pp[5] += t - pl; pl = t; __builtin_amdgcn_sched_barrier(0); }
#endif
    if (REC && c.r == 0) {
      R.rew[(size_t)s * K.n_env + c.env] = reward;
      R.done[(size_t)s * K.n_env + c.env] = done > 0.5f ? 1 : 0;
    }
  }
  if (REC && c.r == 0)   // the workgroup left the loop early: the remaining steps read reward 0 / done 1 (their obs / action rows stay unwritten)
    for (int s2 = s_at; s2 < n_steps; s2++) {
      R.rew[(size_t)s2 * K.n_env + c.env] = 0.0f;
      R.done[(size_t)s2 * K.n_env + c.env] = 1;
    }
  store_ctl16(c, K, S, D.ctl, D.ictl, D.legctl);
  rollout_store16(c, K, L, S.alive, D.base, D.leg);
  __syncthreads();
  for (int idx = tid; idx < TM * ETG_OBS_DIM; idx += 256) obs[(size_t)tile * TM * ETG_OBS_DIM + idx] = obs_lds[idx];   // coalesced
#ifdef ETG_PROFILE_PHASES
  __syncthreads();
  if (tile == 0 && tid == 0) for (int k = 0; k < 8; k++) obs[k] = (float)pp[k];   // debug build: cycle breakdown over row 0
#endif
}

template <bool FLAT, bool BF16, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(256) k_rollout_policy16(KCfg K, DevState D, PolicyW P, int n_steps, float act_scale, float* obs) {
  using namespace pol;
  __shared__ __attribute__((aligned(16))) float bufA[TM * HS];
  __shared__ __attribute__((aligned(16))) float bufB[TM * HS];
  __shared__ float part[4][TM][16];
  __shared__ float act_lds[TM][16];
  __shared__ float obs_lds[TM * ETG_OBS_DIM];
  __shared__ float lds_par[4 * LDS16_FIELDS * 64];
  __shared__ float live_lds[TM];
  rollout_policy16_body<FLAT, BF16, KNEE, PLAIN, false>(K, D, P, n_steps, act_scale, obs, RecOut{}, bufA, bufB, part, nullptr, act_lds, obs_lds, lds_par, live_lds);
}

// ---- closed loop, precision 0, the 16-lane mapping: ONE WAVE per workgroup, the policy tile per wave (policy_core.h: wave_layer).
// A wave runs the actor for its own 4 robots on v_mfma_f32_4x4x1_16B_f32 and then their control step; nothing is shared with
// another wave, so there is no workgroup barrier anywhere in the loop (the __syncthreads() below are the LDS hand-overs between
// the lanes of this one wave: a wait, not a rendezvous).  REC: the recording variant (RecOut; see rollout_policy16_body).
template <bool FLAT, bool KNEE, bool PLAIN, bool REC>
__device__ __forceinline__ void rollout_policy16w_body(const KCfg& K, const DevState& D, const PolicyW& P, int n_steps, float act_scale,
                                                       float* obs, const RecOut& R, float* lds_par, float* obs4, float* abuf, float* hA,
                                                       float* hB, float* part, float (*act4)[16], float* live4, float* park) {
  using namespace pol;
  const int lane0 = threadIdx.x;
  int lane = lane0;
  const int blk = xcd_contiguous_block();             // 4 robots; the host guarantees N % 4 == 0
  GpuCtx16W<FLAT, KNEE, PLAIN> c;
  make_ctx16_fields(K, D, c, lds_par, blk, lane);
  for (int k = 0; k < PR_N; k++) lds_par[k * 64 + lane] = D.par[(size_t)k * c.NL + c.col];   // the lane's whole parameter column
  State16<float> L = load_state16<float>(c, D.base, D.leg);
  StepCtl16<float> S = load_ctl16<float>(c, K, D.ctl, D.ictl, D.legctl);
  V3<float> fext = {0.0f, 0.0f, 0.0f};
  if (!PLAIN && K.ext_force) fext = load_fext16<float>(c, D.ctl);
  park16(park + lane0, L, S);
  float alive = S.alive;                              // (the one control variable the loop itself looks at)
  const size_t row0 = (size_t)blk * 4;
  for (int idx = lane; idx < 4 * ETG_OBS_DIM; idx += 64) obs4[idx] = obs[row0 * ETG_OBS_DIM + idx];
  if (lane < 4) live4[lane] = 1.0f;
  c.row_base = (int)row0;                             // the step code addresses observation rows by robot index
  const bool skip = K.stop_at_done != 0;
  float reward, done;
  int s_at = 0;
#ifdef ETG_PROFILE_TILE   // debugging build: cycles in the policy tile / in the control step / sweeps, per wave (written over the wave's first obs row)
  long long pt_tile = 0, pt_step = 0, pt_t = clock64();
#endif
  for (int s = 0; s < n_steps; s++, s_at = s) {
    if (skip && !c.any(alive > 0.5f)) break;          // every robot of the wave has finished: the wave is done (nobody waits for it)
    // (the lane index through an empty asm: what the tile derives from it -- LDS offsets, row addresses -- is formed here, per
    // step, instead of being kept in registers from the kernel's head through the ticks)
    lane = lane0;
    asm volatile("" : "+v"(lane));
    const int rob = lane >> 4;                        // this lane's robot in the physics mapping
    float* pk = park + lane;
    __syncthreads();
    if (K.noise_on && s > 0) {   // sensor noise on the rows the previous step wrote (the last step's: the launch's epilogue)
      if (live4[rob] > 0.5f) add_sensor_noise(K, (unsigned)(row0 + rob), K.noise_call + s - 1, lane & 15, &obs4[rob * ETG_OBS_DIM]);
      __syncthreads();
    }
    WaveRing ring;
    wave_ring_start(ring, P.w1, lane);                // the first k-groups of layer 1 arrive while the rows are staged
    // the actor's rows, zero padded to WS columns; REC: the rows it acts on
    for (int idx = lane; idx < 4 * WS; idx += 64) {
      const int r = idx / WS, col = idx - r * WS;
      abuf[idx] = col < P.in_dim ? obs4[r * ETG_OBS_DIM + P.col0 + col] : 0.0f;
    }
    if (REC)
      for (int idx = lane; idx < 4 * ETG_OBS_DIM; idx += 64) R.obs[((size_t)s * K.n_env + row0) * ETG_OBS_DIM + idx] = obs4[idx];
    __syncthreads();
    HeadW hw;
    wave_hidden12(abuf, hA, hB, ring, P.w1, P.b1, P.b2, hw, P.w3, lane);
    f32x4 mean = wave_head(hB, hw, part, lane);
    f32x4 lstd = {0.f, 0.f, 0.f, 0.f};
    if (REC && R.noise) {
      head_fetch(hw, R.w3s, lane);
      lstd = wave_head(hB, hw, part, lane);
    }
    if (lane < 12) {                                  // lane = output neuron (the actor has 12: ETG_ACT_DIM)
      const float b3 = lane < P.out_dim ? P.b3[lane] : 0.0f;
      const float b3s = (REC && R.noise && lane < ETG_ACT_DIM) ? R.b3s[lane] : 0.0f;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float v = mean[i] + b3;
        if (REC && R.noise && lane < ETG_ACT_DIM) {
          const float ls = fminf(fmaxf(lstd[i] + b3s, -20.0f), 2.0f);
          v = v + expf(ls) * R.noise[((size_t)s * K.n_env + row0 + i) * ETG_ACT_DIM + lane];
        }
        const float t = tanhf(v);
        act4[i][lane] = t * act_scale;
        if (REC && lane < ETG_ACT_DIM) R.act[((size_t)s * K.n_env + row0 + i) * ETG_ACT_DIM + lane] = t;   // the UNSCALED action (train.py:159)
      }
    }
    __syncthreads();
    const float action = c.sub < 3 ? act4[rob][3 * c.leg + c.sub] : 0.0f;
#ifdef ETG_PROFILE_TILE
    { const long long t = clock64(); pt_tile += t - pt_t; pt_t = t; }
#endif
    // the robot state, the control variables and the tick constants come back from LDS for the step and go back after it: none
    // of them occupies a register during the tile
    c.launder_lane();
    unpark16(pk, L, S);
    S.alive = alive;
    TickPar<float> tp = load_tick_par<float>(c);
    tp.fext = fext;
    const float was_alive = alive;
    control_step16_core(c, K, tp, L, S, D.ring, D.etgp, action, 0.0f, obs4, reward, done, (float*)nullptr, (const float*)nullptr, true,
                        (float*)nullptr, (float*)nullptr, skip);
    if (skip) rollout_dead_store16(c, K, L, was_alive, done, s == n_steps - 1, false, (int)K.noise_call + s, D.base, D.leg, D.ictl);
    alive = S.alive;
    park16(pk, L, S);
#ifdef ETG_PROFILE_TILE
    { const long long t = clock64(); pt_step += t - pt_t; pt_t = t; }
#endif
    if (c.r == 0) live4[rob] = skip ? was_alive : 1.0f;   // did this step write the robot's row?
    if (REC && c.r == 0) {
      R.rew[(size_t)s * K.n_env + c.env] = reward;
      R.done[(size_t)s * K.n_env + c.env] = done > 0.5f ? 1 : 0;
    }
  }
  if (REC && c.r == 0)   // the wave left the loop early: the remaining steps read reward 0 / done 1 (their obs / action rows stay unwritten)
    for (int s2 = s_at; s2 < n_steps; s2++) {
      R.rew[(size_t)s2 * K.n_env + c.env] = 0.0f;
      R.done[(size_t)s2 * K.n_env + c.env] = 1;
    }
  c.launder_lane();
  unpark16(park + lane0, L, S);
  S.alive = alive;
  store_ctl16(c, K, S, D.ctl, D.ictl, D.legctl);
  rollout_store16(c, K, L, S.alive, D.base, D.leg);
  __syncthreads();
  for (int idx = lane0; idx < 4 * ETG_OBS_DIM; idx += 64) obs[row0 * ETG_OBS_DIM + idx] = obs4[idx];
#ifdef ETG_PROFILE_TILE
  __syncthreads();
  if (lane0 == 0) { obs[row0 * ETG_OBS_DIM + 0] = (float)pt_tile; obs[row0 * ETG_OBS_DIM + 1] = (float)pt_step; obs[row0 * ETG_OBS_DIM + 2] = (float)s_at; }
#endif
}

#define ETG_POLICY16W_LDS                                                                    \
  __shared__ float lds_par[LDS16_FIELDS * BLOCK];                                            \
  __shared__ __attribute__((aligned(16))) float obs4[4 * ETG_OBS_DIM];                       \
  __shared__ __attribute__((aligned(16))) float abuf[4 * pol::WS];                           \
  __shared__ __attribute__((aligned(16))) float hA[4 * pol::HS];                             \
  __shared__ __attribute__((aligned(16))) float hB[4 * pol::HS];                             \
  __shared__ __attribute__((aligned(16))) float part[64 * 4];                                \
  __shared__ float park[PARK16 * BLOCK];                                                     \
  __shared__ float act4[4][16];                                                              \
  __shared__ float live4[4];
template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_rollout_policy16w(KCfg K, DevState D, PolicyW P, int n_steps, float act_scale, float* obs) {
  ETG_POLICY16W_LDS
  rollout_policy16w_body<FLAT, KNEE, PLAIN, false>(K, D, P, n_steps, act_scale, obs, RecOut{}, lds_par, obs4, abuf, hA, hB, part, act4, live4, park);
}
template <bool FLAT, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(BLOCK) k_rollout_policy16w_rec(KCfg K, DevState D, PolicyW P, int n_steps, float act_scale, float* obs, RecOut R) {
  ETG_POLICY16W_LDS
  rollout_policy16w_body<FLAT, KNEE, PLAIN, true>(K, D, P, n_steps, act_scale, obs, R, lds_par, obs4, abuf, hA, hB, part, act4, live4, park);
}
#undef ETG_POLICY16W_LDS

// Closed loop on the 4-lanes-per-robot mapping (the mapping of every batch above 4096 robots): a workgroup of 4 waves owns
// 64 robots (16 per wave, one leg per lane).  The policy runs over TWO stacked 16-row tiles at a time (hidden_layer_rt): every
// weight fragment fetched from L2 feeds two MFMAs, so the weight delivery per robot is half that of k_rollout_policy16, and a
// workgroup does two such passes per control step.  Activations of the 32 rows of a pass, the 64 observation rows and the
// actions stay in LDS; each wave then runs the control step of its 16 robots.  Same arithmetic per row as the 16-lane kernel.
template <bool FLAT, bool BF16, bool PLAIN, int BODY = 0>
__global__ void __launch_bounds__(256) k_rollout_policy4(KCfg K, DevState D, PolicyW P, int n_steps, float act_scale, float* obs) {
  using namespace pol;
  constexpr int NWP = 4, RT = 2, ROWS = 64;
  __shared__ __attribute__((aligned(16))) float bufA[RT * TM * HS];
  __shared__ __attribute__((aligned(16))) float bufB[RT * TM * HS];
  __shared__ float part[NWP][RT * TM][16];
  __shared__ float act_lds[ROWS][16];
  __shared__ float obs_lds[ROWS * ETG_OBS_DIM];
  __shared__ float lds_par[NWP][PR_N * 64];
  __shared__ float live_lds[ROWS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int tile = xcd_contiguous_block();            // 64 robots; the host guarantees N % 64 == 0
  GpuCtxT<FLAT, PLAIN, BODY> c;
  c.gid = tile * 256 + tid;
  c.N = K.n_env;
  c.NL = 4 * K.n_env;
  c.env = c.gid >> 2;
  c.lane = c.gid & 3;
  stage_params_wave(c, D, lds_par[wave], lane);
  LaneState<float> L = load_state<float>(c, D.base, D.leg);
  StepCtl4<float> S = load_ctl4<float>(c, K, D.ctl, D.ictl, D.legctl);
  TickPar4<float> tp = load_tick_par4<float>(c);
  V3<float> fext = {0.0f, 0.0f, 0.0f};
  if (!PLAIN && K.ext_force) fext = {c.ld_env(D.ctl, CT_FEXT + 0) + c.ld_env(D.ctl, CT_PUSH + 0), c.ld_env(D.ctl, CT_FEXT + 1) + c.ld_env(D.ctl, CT_PUSH + 1),
                                     c.ld_env(D.ctl, CT_FEXT + 2) + c.ld_env(D.ctl, CT_PUSH + 2)};
  for (int idx = tid; idx < ROWS * ETG_OBS_DIM; idx += 256) obs_lds[idx] = obs[(size_t)tile * ROWS * ETG_OBS_DIM + idx];
  c.row_base = tile * ROWS;
  float reward, done;
  const bool skip = K.stop_at_done != 0;     // (see rollout_policy16_body)
  if (tid < ROWS) live_lds[tid] = 1.0f;
  for (int s = 0; s < n_steps; s++) {
    if (skip) {
      if (!__syncthreads_or(S.alive > 0.5f)) break;
    } else {
      __syncthreads();
    }
    if (K.noise_on && s > 0) {   // sensor noise on the rows the previous step wrote to LDS (the last ones: the launch's epilogue)
      for (int idx = tid; idx < ROWS * 16; idx += 256)
        if (live_lds[idx >> 4] > 0.5f)
          add_sensor_noise(K, tile * ROWS + (idx >> 4), K.noise_call + s - 1, idx & 15, &obs_lds[(idx >> 4) * ETG_OBS_DIM]);
      __syncthreads();
    }
    for (int pass = 0; pass < ROWS / (RT * TM); pass++) {
      const int r0 = pass * RT * TM;
      for (int idx = tid; idx < RT * TM * 64; idx += 256) {   // observation rows of the pass, zero padded to the 64-wide K of layer 1
        const int r = idx >> 6, col = idx & 63;
        bufA[r * HS + col] = col < P.in_dim ? obs_lds[(r0 + r) * ETG_OBS_DIM + P.col0 + col] : 0.0f;
      }
      __syncthreads();
      hidden_layer_rt<BF16, 4, NWP, RT>(bufA, P.w1, P.b1, bufB, wave, lane);
      __syncthreads();
      hidden_layer_rt<BF16, HID / 16, NWP, RT>(bufB, P.w2, P.b2, bufA, wave, lane);
      __syncthreads();
      output_partial_rt<BF16, NWP, RT>(bufA, P.w3, wave, lane, part);
      __syncthreads();
      for (int idx = tid; idx < RT * TM * 16; idx += 256) {
        const int r = idx >> 4, cidx = idx & 15;
        const float v = ((part[0][r][cidx] + part[1][r][cidx]) + (part[2][r][cidx] + part[3][r][cidx])) + (cidx < P.out_dim ? P.b3[cidx] : 0.0f);
        act_lds[r0 + r][cidx] = tanhf(v) * act_scale;
      }
      __syncthreads();
    }
    const int rl = 16 * wave + (lane >> 2), leg = lane & 3;
    const float act[3] = {act_lds[rl][3 * leg + 0], act_lds[rl][3 * leg + 1], act_lds[rl][3 * leg + 2]};
    const float was_alive = S.alive;
    if (!skip || c.any(was_alive > 0.5f)) {
      control_step_core(c, K, tp, fext, L, S, D.ring, D.etgp, act, 0.0f, obs_lds, reward, done, (float*)nullptr, (const float*)nullptr, true,
                        (float*)nullptr, (float*)nullptr, skip);
      if (skip) rollout_dead_store(c, K, L, was_alive, done, s == n_steps - 1, false, (int)K.noise_call + s, D.base, D.leg, D.ictl);
    }
    if (c.lane == 0) live_lds[rl] = skip ? was_alive : 1.0f;
  }
  store_ctl4(c, K, S, D.ctl, D.ictl, D.legctl);
  rollout_store(c, K, L, S.alive, D.base, D.leg);
  __syncthreads();
  for (int idx = tid; idx < ROWS * ETG_OBS_DIM; idx += 256) obs[(size_t)tile * ROWS * ETG_OBS_DIM + idx] = obs_lds[idx];   // coalesced
}

// the recording variant of k_rollout_policy16 (etg_rollout_policy_record): the same body with REC = true
template <bool FLAT, bool BF16, bool KNEE, bool PLAIN>
__global__ void __launch_bounds__(256) k_rollout_policy16_rec(KCfg K, DevState D, PolicyW P, int n_steps, float act_scale, float* obs, RecOut R) {
  using namespace pol;
  __shared__ __attribute__((aligned(16))) float bufA[TM * HS];
  __shared__ __attribute__((aligned(16))) float bufB[TM * HS];
  __shared__ float part[4][TM][16];
  __shared__ float part_s[4][TM][16];   // log-std head (stochastic actor only)
  __shared__ float act_lds[TM][16];
  __shared__ float obs_lds[TM * ETG_OBS_DIM];
  __shared__ float lds_par[4 * LDS16_FIELDS * 64];
  __shared__ float live_lds[TM];
  rollout_policy16_body<FLAT, BF16, KNEE, PLAIN, true>(K, D, P, n_steps, act_scale, obs, R, bufA, bufB, part, part_s, act_lds, obs_lds, lds_par, live_lds);
}

// Gaussian sensor noise on freshly written observation rows: one thread per (robot, channel).  A separate tiny
// kernel so that the step kernels' code (and register allocation) is the same with and without noise.
#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void k_add_noise(KCfg K, unsigned call, const uint8_t* mask, int invert, float* obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = i >> 4;
  if (env >= K.n_env || (mask && (mask[env] != 0) == (invert != 0))) return;   // mask: rows to touch (or, inverted, to leave)
  add_sensor_noise(K, env, call, i & 15, obs + (size_t)env * ETG_OBS_DIM);
}

__global__ void k_add_noise_range(KCfg K, unsigned call, int env0, int count, float* obs) {   // etg_step_range's rows
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((i >> 4) >= count) return;
  const int env = env0 + (i >> 4);
  add_sensor_noise(K, env, call, i & 15, obs + (size_t)env * ETG_OBS_DIM);
}

// the epilogue of a fused rollout launch under KCfg.stop_at_done: rows were written at different steps (a robot's last one when
// its episode ended), each noted its stream position in ictl[IC_OBS_CALL]; rows noted in [first, first + n) get their noise
__global__ void k_add_noise_rows(KCfg K, DevState D, unsigned first, unsigned n, float* obs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = i >> 4;
  if (env >= K.n_env) return;
  const unsigned call = (unsigned)D.ictl[(size_t)IC_OBS_CALL * K.n_env + env];
  if (call - first >= n) return;
  add_sensor_noise(K, env, call, i & 15, obs + (size_t)env * ETG_OBS_DIM);
}

// external force rows [N,3] -> the three SoA columns ctl[CT_FEXT + k][N]
__global__ void k_set_fext(KCfg K, DevState D, const float* force) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K.n_env) return;
  for (int k = 0; k < 3; k++) D.ctl[(size_t)(CT_FEXT + k) * K.n_env + i] = force[(size_t)i * 3 + k];
}
#endif

// counter-based uniform in [0,1): a 64-bit mix (splitmix64 finaliser) of (seed, robot, call, stream)
__device__ __forceinline__ float push_uniform(unsigned long long seed, unsigned env, unsigned long long call, unsigned k) {
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((unsigned long long)env << 32 | k) + 0xD1B54A32D192ED03ull * (call + 1));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void k_random_pushes(KCfg K, DevState D, unsigned long long seed, unsigned long long call, float prob, int duration,
                                float fmin, float fmax) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = K.n_env;
  if (i >= N) return;
  int left = D.ictl[(size_t)IC_PUSH_LEFT * N + i];
  float fx = D.ctl[(size_t)(CT_PUSH + 0) * N + i], fy = D.ctl[(size_t)(CT_PUSH + 1) * N + i];
  if (left > 0) {
    left--;
    if (left == 0) { fx = 0.0f; fy = 0.0f; }
  } else if (push_uniform(seed, i, call, 0) < prob) {
    const float ang = 6.283185307179586f * push_uniform(seed, i, call, 1);
    const float mag = fmin + (fmax - fmin) * push_uniform(seed, i, call, 2);
    fx = mag * cosf(ang); fy = mag * sinf(ang);
    left = duration;
  }
  D.ictl[(size_t)IC_PUSH_LEFT * N + i] = left;
  D.ctl[(size_t)(CT_PUSH + 0) * N + i] = fx;
  D.ctl[(size_t)(CT_PUSH + 1) * N + i] = fy;
  D.ctl[(size_t)(CT_PUSH + 2) * N + i] = 0.0f;
}
__global__ void k_clear_pushes(KCfg K, DevState D, const uint8_t* mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int N = K.n_env;
  if (i >= N || (mask && !mask[i])) return;
  D.ictl[(size_t)IC_PUSH_LEFT * N + i] = 0;
  for (int k = 0; k < 3; k++) D.ctl[(size_t)(CT_PUSH + k) * N + i] = 0.0f;   // the set force (CT_FEXT) stays
}

// copy out the per-robot episode accumulators (return, length) kept in ctl[]
__global__ void k_episode_stats(KCfg K, DevState D, float* ret, int* len) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K.n_env) return;
  if (ret) ret[i] = D.ctl[(size_t)CT_RET * K.n_env + i];
  if (len) len[i] = (int)D.ctl[(size_t)CT_LEN * K.n_env + i];
}

__global__ void __launch_bounds__(BLOCK) k_get_state(KCfg K, DevState D, float* st) {
  GpuCtxT<true> c;
  if (!make_ctx(K, c)) return;
  LaneState<float> L = load_state<float>(c, D.base, D.leg);
  get_state_quad(c, L, st);
}
__global__ void __launch_bounds__(BLOCK) k_set_state(KCfg K, DevState D, const float* st) {
  GpuCtxT<true> c;
  if (!make_ctx(K, c)) return;
  LaneState<float> L;
  set_state_quad(c, st, L, D.ring, D.ctl, D.ictl, K.settle_ticks + RING);
  store_state(c, D.base, D.leg, L);
}

// leg FK + analytic Jacobian through the tick's own leg_geometry (etg_leg_kinematics): 4 joint-angle rows per
// wave, lane r = 4*leg + sub of a 16-lane row like the step kernels.  Sub-lane s < 3 writes component s of the
// foot position (base frame) and row s of d foot / d (hip, thigh, calf): the columns are the k1, k2, k3 lever
// arms the contact rows of physics_tick16 are built from, evaluated at the foot centre.
__global__ void __launch_bounds__(BLOCK) k_leg_kin(KCfg K, ModelF M, const float* q, int n, float* foot, float* jac) {
  GpuCtx16T<true> c;
  const int lane = threadIdx.x;
  const int row = blockIdx.x * 4 + (lane >> 4);
  const bool live = row < n;
  const int rr = live ? row : n - 1;          // keep every quad complete for the DPP exchanges
  c.r = lane & 15; c.leg = c.r >> 2; c.sub = c.r & 3; c.sc = c.sub < 2 ? c.sub : 2;
  const float qo = c.sub < 3 ? q[(size_t)rr * 12 + 3 * c.leg + c.sub] : 0.0f;
  const V3<float> o1 = {M.hip_origin[c.leg][0], M.hip_origin[c.leg][1], M.hip_origin[c.leg][2]};
  const LegGeo<float> g = leg_geometry(c, K, o1, M.thigh_y[c.leg], qo);
  const V3<float> xax = {1.0f, 0.0f, 0.0f};
  const V3<float> k1 = cross(xax, g.pf - g.o1), k2 = cross(g.yax, g.pf - g.o2), k3 = cross(g.yax, g.pf - g.o3);
  if (!live || c.sub == 3) return;
  const int s = c.sub;
  foot[(size_t)row * 12 + 3 * c.leg + s] = s == 0 ? g.pf.x : s == 1 ? g.pf.y : g.pf.z;
  if (jac) {
    float* J = jac + (((size_t)row * 4 + c.leg) * 3 + s) * 3;
    J[0] = s == 0 ? k1.x : s == 1 ? k1.y : k1.z;
    J[1] = s == 0 ? k2.x : s == 1 ? k2.y : k2.z;
    J[2] = s == 0 ? k3.x : s == 1 ? k3.y : k3.z;
  }
}

// optional sensors (include/etgsim.h ETG_EXTRA_*): one thread per (robot, column)
__global__ void k_extra_sensors(KCfg K, ModelF M, DevState D, const float* obs, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int env = i / ETG_EXTRA_DIM, col = i % ETG_EXTRA_DIM;
  const int N = K.n_env;
  if (env >= N) return;
  float v = 0.0f;
  if (col < ETG_EXTRA_FOOTPOSE) {            // RBF activations at the ETG time of the row's observation
    const int k = D.ictl[(size_t)IC_STEP * N + env];
    const float t = (float)k * K.etg_dt;
    const float x0 = K.etg_amp * sinf(K.etg_phase0 + t * K.etg_omega), x1 = K.etg_amp * sinf(K.etg_phase1 + t * K.etg_omega);
    const float d0 = x0 - K.etg_u[col][0], d1 = x1 - K.etg_u[col][1];
    v = expf(-(d0 * d0 + d1 * d1) / K.etg_sigma_sq);
  } else if (col < ETG_EXTRA_DYNAMIC) {      // foot_positions_in_base_frame (a1.py:113-140) of the OBSERVED motor angles
    const int j = col - ETG_EXTRA_FOOTPOSE, leg = j / 3, k = j % 3;
    float a[3];
    for (int m = 0; m < 3; m++) {
      const float o = obs[(size_t)env * ETG_OBS_DIM + 13 + 3 * leg + m];
      a[m] = K.obs_normal ? o * 0.1f + M.pose[3 * leg + m] : o;
    }
    const float l_hip = M.thigh_y[leg], lu = K.upper_len, ll = K.lower_len;
    const float dist = sqrtf(lu * lu + ll * ll + 2.0f * lu * ll * cosf(a[2]));
    const float sw = a[1] + 0.5f * a[2];
    const float ox = -dist * sinf(sw), oz = -dist * cosf(sw), oy = l_hip;
    const float p[3] = {ox, cosf(a[0]) * oy - sinf(a[0]) * oz, sinf(a[0]) * oy + cosf(a[0]) * oz};
    v = p[k] + M.hip_origin[leg][k];
  } else if (col < ETG_EXTRA_FORCE) {        // inverse of param2dynamic_dict's affine maps (train.py:112-126)
    const int k = col - ETG_EXTRA_DYNAMIC;
    const float d = D.dyn[(size_t)env * ETG_DYN_DIM + k];
    if (k == 0) v = (d - 40.0f) * 0.1f;
    else if (k == 1) v = (d - 0.2f) * 0.1f;
    else if (k == 2) v = d - 1.5f;
    else if (k < 21) v = d - 1.0f;
    else if (k < 33) v = (d - 80.0f) * 0.025f;
    else if (k < 45) { const float kd0 = ((k - 33) % 3 == 0) ? 1.0f : 2.0f; v = (d - kd0) / kd0; }
    else { const float g0[3] = {0.0f, 0.0f, -10.0f}, gs[3] = {2.0f, 2.0f, 10.0f}; v = (d - g0[k - 45]) / gs[k - 45]; }
  } else if (col < ETG_EXTRA_FORCE + 3) {
    const int k = col - ETG_EXTRA_FORCE;
    v = D.ctl[(size_t)(CT_FEXT + k) * N + env] + D.ctl[(size_t)(CT_PUSH + k) * N + env];
  }
  out[(size_t)env * ETG_EXTRA_DIM + col] = v;
}
#endif

// ---- the tick kernels' instantiations, one row per kernel family, one SLOT per instantiation.  A slot belongs to part
// 1 + (slot - 1) % (ETG_TU_PARTS - 1): that part instantiates the kernel, every other part (the host side included) only
// declares it.  The combinations are the ones DISPATCH16 / LAUNCH4 / LAUNCH_POLICY* launch; a launch of a combination that
// is missing here fails at link time (undefined __device_stub__), not at run time.
#define ETG_ARGS_SETTLE KCfg, DevState, const uint8_t*
#define ETG_ARGS_FINISH KCfg, DevState, const uint8_t*, float*
#define ETG_ARGS_STEP KCfg, DevState, const float*, const uint8_t*, float*, float*, uint8_t*, float*
#define ETG_ARGS_STEP_AR KCfg, DevState, const float*, const uint8_t*, float*, float*, uint8_t*, float*, NextDyn
#define ETG_ARGS_ROLLOUT KCfg, DevState, int, float*, StatOut
#define ETG_ARGS_TAPE KCfg, DevState, int, const float*, float*, TapeOut
#define ETG_ARGS_POLICY KCfg, DevState, PolicyW, int, float, float*
#define ETG_ARGS_POLICY_REC KCfg, DevState, PolicyW, int, float, float*, RecOut
#if ETG_TU_PARTS > 1
#define ETG_SLOT_OWNER(S) (1 + ((S) - 1) % (ETG_TU_PARTS - 1))
#if ETG_SLOT_OWNER(1) == ETG_TU_PART
#define ETG_TPL_1 template
#else
#define ETG_TPL_1 extern template
#endif
#if ETG_SLOT_OWNER(2) == ETG_TU_PART
#define ETG_TPL_2 template
#else
#define ETG_TPL_2 extern template
#endif
#if ETG_SLOT_OWNER(3) == ETG_TU_PART
#define ETG_TPL_3 template
#else
#define ETG_TPL_3 extern template
#endif
#if ETG_SLOT_OWNER(4) == ETG_TU_PART
#define ETG_TPL_4 template
#else
#define ETG_TPL_4 extern template
#endif
#if ETG_SLOT_OWNER(5) == ETG_TU_PART
#define ETG_TPL_5 template
#else
#define ETG_TPL_5 extern template
#endif
#if ETG_SLOT_OWNER(6) == ETG_TU_PART
#define ETG_TPL_6 template
#else
#define ETG_TPL_6 extern template
#endif
#if ETG_SLOT_OWNER(7) == ETG_TU_PART
#define ETG_TPL_7 template
#else
#define ETG_TPL_7 extern template
#endif
// 16-lane families: <FLAT, KNEE, PLAIN> in the order of DISPATCH16
#define ETG_INST16(KERN, S0, S1, S2, S3, S4, S5, ...)                          \
  ETG_TPL_##S0 __global__ void KERN<true, true, true>(__VA_ARGS__);            \
  ETG_TPL_##S1 __global__ void KERN<true, false, true>(__VA_ARGS__);           \
  ETG_TPL_##S2 __global__ void KERN<true, true, false>(__VA_ARGS__);           \
  ETG_TPL_##S3 __global__ void KERN<false, true, true>(__VA_ARGS__);           \
  ETG_TPL_##S4 __global__ void KERN<false, false, true>(__VA_ARGS__);          \
  ETG_TPL_##S5 __global__ void KERN<false, true, false>(__VA_ARGS__);
// closed-loop 16-lane families: <FLAT, BF16, KNEE, PLAIN>, one precision per row
#define ETG_INSTP16(KERN, BF, S0, S1, S2, S3, S4, S5, ...)                     \
  ETG_TPL_##S0 __global__ void KERN<true, BF, true, true>(__VA_ARGS__);        \
  ETG_TPL_##S1 __global__ void KERN<true, BF, false, true>(__VA_ARGS__);       \
  ETG_TPL_##S2 __global__ void KERN<true, BF, true, false>(__VA_ARGS__);       \
  ETG_TPL_##S3 __global__ void KERN<false, BF, true, true>(__VA_ARGS__);       \
  ETG_TPL_##S4 __global__ void KERN<false, BF, false, true>(__VA_ARGS__);      \
  ETG_TPL_##S5 __global__ void KERN<false, BF, true, false>(__VA_ARGS__);
// 4-lane families: <FLAT, PLAIN, BODY> in the order of LAUNCH4
#define ETG_INST4(KERN, S0, S1, S2, S3, S4, S5, S6, S7, ...)                   \
  ETG_TPL_##S0 __global__ void KERN<true, false, 3>(__VA_ARGS__);              \
  ETG_TPL_##S1 __global__ void KERN<false, false, 3>(__VA_ARGS__);             \
  ETG_TPL_##S2 __global__ void KERN<true, false, 1>(__VA_ARGS__);              \
  ETG_TPL_##S3 __global__ void KERN<false, false, 1>(__VA_ARGS__);             \
  ETG_TPL_##S4 __global__ void KERN<true, true, 0>(__VA_ARGS__);               \
  ETG_TPL_##S5 __global__ void KERN<true, false, 0>(__VA_ARGS__);              \
  ETG_TPL_##S6 __global__ void KERN<false, true, 0>(__VA_ARGS__);              \
  ETG_TPL_##S7 __global__ void KERN<false, false, 0>(__VA_ARGS__);
// closed-loop 4-lane family: <FLAT, BF16, PLAIN, BODY> in the order of LAUNCH_POLICY4
#define ETG_INSTP4(KERN, BF, S0, S1, S2, S3, S4, S5, ...)                      \
  ETG_TPL_##S0 __global__ void KERN<true, BF, false, 1>(__VA_ARGS__);          \
  ETG_TPL_##S1 __global__ void KERN<false, BF, false, 1>(__VA_ARGS__);         \
  ETG_TPL_##S2 __global__ void KERN<true, BF, true, 0>(__VA_ARGS__);           \
  ETG_TPL_##S3 __global__ void KERN<true, BF, false, 0>(__VA_ARGS__);          \
  ETG_TPL_##S4 __global__ void KERN<false, BF, true, 0>(__VA_ARGS__);          \
  ETG_TPL_##S5 __global__ void KERN<false, BF, false, 0>(__VA_ARGS__);
ETG_INST16(k_settle16, 1, 2, 3, 4, 5, 6, ETG_ARGS_SETTLE)
ETG_INST16(k_finish16, 7, 1, 2, 3, 4, 5, ETG_ARGS_FINISH)
ETG_INST16(k_step16, 6, 7, 1, 2, 3, 4, ETG_ARGS_STEP)
ETG_INST16(k_step16_ar, 5, 6, 7, 1, 2, 3, ETG_ARGS_STEP_AR)
ETG_INST16(k_rollout16, 4, 5, 6, 7, 1, 2, ETG_ARGS_ROLLOUT)
ETG_INST16(k_rollout_actions16, 3, 4, 5, 6, 7, 1, ETG_ARGS_TAPE)
ETG_INST16(k_rollout_policy16w, 2, 4, 6, 1, 3, 5, ETG_ARGS_POLICY)
ETG_INST16(k_rollout_policy16w_rec, 7, 2, 4, 6, 1, 3, ETG_ARGS_POLICY_REC)
ETG_INSTP16(k_rollout_policy16, false, 2, 3, 4, 5, 6, 7, ETG_ARGS_POLICY)
ETG_INSTP16(k_rollout_policy16, true, 1, 2, 3, 4, 5, 6, ETG_ARGS_POLICY)
ETG_INSTP16(k_rollout_policy16_rec, false, 7, 1, 2, 3, 4, 5, ETG_ARGS_POLICY_REC)
ETG_INSTP16(k_rollout_policy16_rec, true, 6, 7, 1, 2, 3, 4, ETG_ARGS_POLICY_REC)
ETG_INST4(k_settle, 5, 6, 7, 1, 2, 3, 4, 5, ETG_ARGS_SETTLE)
ETG_INST4(k_finish, 6, 7, 1, 2, 3, 4, 5, 6, ETG_ARGS_FINISH)
ETG_INST4(k_step, 7, 1, 2, 3, 4, 5, 6, 7, ETG_ARGS_STEP)
ETG_INST4(k_step_ar, 1, 2, 3, 4, 5, 6, 7, 1, ETG_ARGS_STEP_AR)
ETG_INST4(k_rollout, 2, 3, 4, 5, 6, 7, 1, 2, ETG_ARGS_ROLLOUT)
ETG_INST4(k_rollout_actions, 3, 4, 5, 6, 7, 1, 2, 3, ETG_ARGS_TAPE)
ETG_INSTP4(k_rollout_policy4, false, 4, 5, 6, 7, 1, 2, ETG_ARGS_POLICY)
ETG_INSTP4(k_rollout_policy4, true, 3, 4, 5, 6, 7, 1, ETG_ARGS_POLICY)
#endif   // ETG_TU_PARTS > 1

}  // namespace etg

#if ETG_TU_HOST
// ====================================================================== C ABI
using namespace etg;

struct EtgHandle {
  int device;
  int N;
  KCfg K;
  ModelF M;
  DevState D;
  float* hf;
  int lanes;                    // 4 or 16 lanes per robot (EtgConfig.lanes_per_robot)
  unsigned long long push_calls;  // stream position of etg_random_pushes
  unsigned obs_calls;             // stream position of the sensor noise: observations written so far
  bool was_reset;                 // etg_step before the first etg_reset is a caller error (state undefined)
  bool fext_set, push_on;         // a set force / random pushes are installed: K.ext_force = fext_set || push_on
  bool all_cached;                // every robot has a valid cached settle (true after a full etg_reset until parameters,
                                  // terrain or -- on a heightfield -- start offsets change): etg_step_autoreset's fast path
  // etg_rollout_wave_cycles: [launches of the last open-loop rollout][wavefronts] shader-clock cycles, written by the launches
  long long* wave_cycles;
  int wc_cap, wc_launches, wc_waves;
  int rollout_chunk;            // control steps per fused launch (400: one launch for BASELINE's episode; 25 .. 400 measured, profiles/r06_chunk_sweep.txt; ETG_ROLLOUT_CHUNK overrides it: a measurement aid)
  float *tmp_obs, *tmp_reward;  // sinks for etg_rollout_openloop
  uint8_t* tmp_done;
  // etg_prepare_next_dynamics: rows waiting for the robots' next episodes + the scratch state / ring / flags its settle runs on
  NextDyn NX;
  float *nx_base, *nx_leg, *nx_ring;
  unsigned char* nx_cache_ok;
  uint8_t* nx_mask;       // etg_prepare_next_dynamics: the call's mask without the robots whose episode is younger than the ring
  // all_cached again after MASKED resets: every invalidation bumps inval_seq; a masked etg_reset ends with a count of the robots
  // still without a cached settle, written with the sequence number it ran under to pinned host memory (cached_report[0] = count,
  // [1] = seq); the next etg_step_autoreset / etg_prepare_next_dynamics trusts a zero count only under the CURRENT number
  unsigned inval_seq;
  volatile unsigned* cached_report;   // pinned host memory, also mapped on the device (cached_report_dev)
  unsigned* cached_report_dev;
};

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(ETG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

extern "C" const char* etg_last_error(void) { return g_err.c_str(); }
// shared with policy_mlp.hip so that one etg_last_error() serves the whole ABI
extern "C" void etg_set_last_error_(const char* msg) { g_err = msg ? msg : ""; }
extern "C" int etg_lanes_per_robot(const EtgHandle* h) { return h ? h->lanes : ETG_ERR_BAD_ARG; }
int etg_version(void) { return 2; }
int etg_config_size(void) { return (int)sizeof(EtgConfig); }
int etg_model_size(void) { return (int)sizeof(EtgRobotModel); }

static int grid_for(const EtgHandle* h) { return (4 * h->N + BLOCK - 1) / BLOCK; }

extern "C" int etg_create(const EtgConfig* cfg, const EtgRobotModel* model, int device, EtgHandle** out) {
  if (!cfg || !model || !out) return fail(ETG_ERR_BAD_ARG, "etg_create: null argument");
  if (cfg->num_envs <= 0 || cfg->num_envs > (1 << 20)) return fail(ETG_ERR_BAD_ARG, "etg_create: num_envs must be in 1..1048576");
  if (cfg->action_repeat <= 0 || cfg->sim_dt <= 0) return fail(ETG_ERR_BAD_ARG, "etg_create: bad action_repeat/sim_dt");
  // the rest of the configuration is checked before a device is touched, so a bad one reads the same on any box
  if (cfg->solver_iters <= 0 || cfg->solver_iters > 1000) return fail(ETG_ERR_BAD_ARG, "etg_create: solver_iters must be in 1..1000");
  if (!(cfg->solver_residual >= 0)) return fail(ETG_ERR_BAD_ARG, "etg_create: solver_residual must be >= 0 (0 = a fixed number of sweeps)");
  if (cfg->friction_model != 0 && cfg->friction_model != 1) return fail(ETG_ERR_BAD_ARG, "etg_create: friction_model must be 0 (disc) or 1 (pyramid)");
  if (!(cfg->pd_latency >= 0)) return fail(ETG_ERR_BAD_ARG, "etg_create: pd_latency must be >= 0 seconds");
  // (beyond the ring depth the blend of minitaur.py:1185-1188 would extrapolate between aliased slots)
  if (cfg->pd_latency >= (etg::RING - 2) * cfg->sim_dt)
    return fail(ETG_ERR_BAD_ARG, "etg_create: pd_latency must be below (ring depth - 2) * sim_dt = 62 ticks");
  if (!(cfg->warmstart >= 0) || !(cfg->warmstart_friction >= 0)) return fail(ETG_ERR_BAD_ARG, "etg_create: warm-start factors must be >= 0");
  if (!(cfg->contact_slop >= 0)) return fail(ETG_ERR_BAD_ARG, "etg_create: contact_slop must be >= 0");
  if (!(cfg->foot_restitution >= 0) || cfg->foot_restitution > 1) return fail(ETG_ERR_BAD_ARG, "etg_create: foot_restitution must be in [0, 1]");
  if (cfg->settle_ticks < 0) return fail(ETG_ERR_BAD_ARG, "etg_create: settle_ticks must not be negative");
  if (cfg->motor_mode < 0 || cfg->motor_mode > 2) return fail(ETG_ERR_BAD_ARG, "etg_create: motor_mode must be 0 (POSITION), 1 (TORQUE) or 2 (HYBRID)");
  if (cfg->body_contacts < 0 || cfg->body_contacts > 3) return fail(ETG_ERR_BAD_ARG, "etg_create: body_contacts must be 0, 1, 2 or 3");
  if (!(cfg->body_friction >= 0.0)) return fail(ETG_ERR_BAD_ARG, "etg_create: body_friction must be >= 0");
  if (cfg->body_contacts == 3 && cfg->lanes_per_robot == 16)
    return fail(ETG_ERR_BAD_ARG, "etg_create: body_contacts = 3 (three body rows per leg) needs the 4-lanes-per-robot mapping");
  if (cfg->terrain != 0 && cfg->terrain != 1) return fail(ETG_ERR_BAD_ARG, "etg_create: terrain must be 0 (plane) or 1 (heightfield)");
  if (cfg->terrain == 1 && (cfg->hf_nx < 2 || cfg->hf_ny < 2 || !(cfg->hf_cell > 0) || cfg->hf_bands < 0 ||
                            (cfg->hf_bands > 1 && cfg->hf_ny % cfg->hf_bands != 0)))
    return fail(ETG_ERR_BAD_ARG, "etg_create: a heightfield needs hf_nx, hf_ny >= 2 (per band), hf_cell > 0 and hf_ny divisible by hf_bands");
  if (cfg->etg_dt <= 0 || cfg->etg_T <= 0) return fail(ETG_ERR_BAD_ARG, "etg_create: etg_dt and etg_T must be positive");
  if (cfg->joint_limits)
    for (int k = 0; k < 3; k++)
      if (!(cfg->joint_lower[k] < cfg->joint_upper[k])) return fail(ETG_ERR_BAD_ARG, "etg_create: joint_lower must be below joint_upper");
  if (cfg->lanes_per_robot != 0 && cfg->lanes_per_robot != 4 && cfg->lanes_per_robot != 16)
    return fail(ETG_ERR_BAD_ARG, "etg_create: lanes_per_robot must be 0 (auto), 4 or 16");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(ETG_ERR_NO_DEVICE, "etg_create: no HIP device (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(ETG_ERR_BAD_ARG, "etg_create: bad device index");
  HIP_TRY(hipSetDevice(device));
  EtgHandle* h = new EtgHandle();
  h->device = device;
  h->N = cfg->num_envs;
  h->K = make_kcfg(*cfg, *model);
  h->M = make_modelf(*model);
  h->hf = nullptr;
  h->push_calls = 0;
  h->obs_calls = 0;
  h->was_reset = false;
  h->fext_set = h->push_on = false;
  h->all_cached = false;
  h->NX = NextDyn{nullptr, nullptr, nullptr};
  h->inval_seq = 1;
  h->cached_report = nullptr;
  h->cached_report_dev = nullptr;
  h->nx_base = h->nx_leg = h->nx_ring = nullptr;
  h->nx_cache_ok = nullptr;
  h->nx_mask = nullptr;
  h->wave_cycles = nullptr;
  h->wc_cap = h->wc_launches = h->wc_waves = 0;
  h->rollout_chunk = 400;   // (a 400-step launch lasts ~37 ms; fewer launches = fewer chip-wide barriers: 93.3 us per step against 94.0 at 50)
  if (const char* e = getenv("ETG_ROLLOUT_CHUNK")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 100000) h->rollout_chunk = v;
  }
  // 0 = auto.  Both kernels hold one wave per SIMD (register footprint), so the chip runs 1024 waves
  // at a time: 16 lanes/robot fills it with 4096 robots and is faster per robot up to there; beyond
  // that the 4-lanes/robot kernel packs 4x the robots per wave (measured crossover).
  // auto: the 16-lane mapping while it fills the chip once (4096 robots = one wave per SIMD); with body rows (modes 1 / 2) up to
  // 8192 -- two rounds of the 16-lane kernel (45 M env-steps/s) still beat the half-filled 4-lane one, whose wave waits for the
  // slowest of SIXTEEN robots (36 M; profiles/r05_ab_experiments.txt section 8); above that the 4-lane mapping (16384: 72 M vs 41 M)
  const int auto16 = (cfg->body_contacts == 1 || cfg->body_contacts == 2) ? 8192 : 4096;
  h->lanes = cfg->lanes_per_robot != 0 ? cfg->lanes_per_robot : (cfg->num_envs <= auto16 ? 16 : 4);
  if (cfg->body_contacts == 3) h->lanes = 4;   // six rows per leg: the 16-lane mapping has one spare lane per leg
  size_t N = h->N, NL = 4 * N;
  struct { void** p; size_t bytes; } allocs[] = {
      {(void**)&h->D.base, BS_N * N * 4},   {(void**)&h->D.leg, LG_N * NL * 4},   {(void**)&h->D.ctl, CT_N * N * 4},
      {(void**)&h->D.ictl, IC_N * N * 4},   {(void**)&h->D.legctl, LC_N * NL * 4}, {(void**)&h->D.etgp, EP_N * N * 4},
      {(void**)&h->D.par, PR_N * NL * 4},   {(void**)&h->D.ring, (size_t)RING * 8 * NL * 4}, {(void**)&h->D.dyn, ETG_DYN_DIM * N * 4},
      {(void**)&h->D.cache_base, BS_N * N * 4}, {(void**)&h->D.cache_leg, LG_N * NL * 4},
      {(void**)&h->D.cache_ring, (size_t)RING * 8 * NL * 4}, {(void**)&h->D.cache_ok, N},
      {(void**)&h->D.reset_off, 2 * N * 4}, {(void**)&h->D.cache_off, (size_t)FIN_ROWS * N * 4},
      {(void**)&h->tmp_obs, ETG_OBS_DIM * N * 4}, {(void**)&h->tmp_reward, N * 4}, {(void**)&h->tmp_done, N}};
  for (auto& a : allocs) {
    if (hipMalloc(a.p, a.bytes) != hipSuccess) return fail(ETG_ERR_ALLOC, "etg_create: hipMalloc failed");
    HIP_TRY(hipMemset(*a.p, 0, a.bytes));
  }
  h->K.cring = h->D.cache_ring;
  // default physical parameters = param2dynamic_dict(zeros(48)) (train.py:112-126)
  std::vector<float> row(ETG_DYN_DIM, 1.0f);
  row[0] = 40.0f; row[1] = 0.2f; row[2] = 1.5f;
  for (int j = 0; j < 12; j++) { row[21 + j] = 80.0f; row[33 + j] = (j % 3 == 0) ? 1.0f : 2.0f; }
  row[45] = 0.0f; row[46] = 0.0f; row[47] = -10.0f;
  std::vector<float> dyn(N * ETG_DYN_DIM);
  for (size_t i = 0; i < N; i++) std::copy(row.begin(), row.end(), dyn.begin() + i * ETG_DYN_DIM);
  float* ddyn = nullptr;
  if (hipMalloc((void**)&ddyn, dyn.size() * 4) != hipSuccess) return fail(ETG_ERR_ALLOC, "etg_create: hipMalloc failed");
  HIP_TRY(hipMemcpy(ddyn, dyn.data(), dyn.size() * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_set_params, dim3(grid_for(h)), dim3(BLOCK), 0, 0, h->K, h->M, h->D, ddyn, nullptr, nullptr, 0, nullptr);
  hipLaunchKernelGGL(k_set_strength, dim3((4 * h->N + 255) / 256), dim3(256), 0, 0, h->K, h->D, (const float*)nullptr, (const uint8_t*)nullptr);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipFree(ddyn));
  *out = h;
  return ETG_OK;
}

extern "C" void etg_destroy(EtgHandle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  void* ptrs[] = {h->D.base, h->D.leg, h->D.ctl, h->D.ictl, h->D.legctl, h->D.etgp, h->D.par, h->D.ring, h->hf, h->D.dyn,
                  h->D.cache_base, h->D.cache_leg, h->D.cache_ring, h->D.cache_ok, h->D.reset_off, h->D.cache_off,
                  h->tmp_obs, h->tmp_reward, h->tmp_done, h->wave_cycles, h->NX.par, h->NX.dyn, h->NX.ok, h->nx_base, h->nx_leg, h->nx_ring, h->nx_cache_ok, h->nx_mask};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  if (h->cached_report) (void)hipHostFree((void*)h->cached_report);
  delete h;
}

// the next launch writes `n` observations per robot: give them the next n positions of the sensor-noise stream
static inline void advance_obs_stream(EtgHandle* h, int n) {
  h->K.noise_call = h->obs_calls;
  h->obs_calls += (unsigned)n;
}

// noise for the observation rows the launch before wrote last (stream position K.noise_call + n - 1)
static inline void launch_obs_noise(EtgHandle* h, int n, const uint8_t* mask, float* obs, hipStream_t s, int invert = 0, int back = 0) {
  if (!h->K.noise_on || !obs) return;   // back: stream positions before the newest one (the step's row when a reset row follows it)
  hipLaunchKernelGGL(k_add_noise, dim3((16 * h->N + 255) / 256), dim3(256), 0, s, h->K,
                     h->K.noise_call + (unsigned)(n - 1) - (unsigned)back, mask, invert, obs);
}

// noise for the rows a fused rollout launch of n steps left in `obs` (stream positions K.noise_call .. + n - 1)
static inline void launch_rollout_noise(EtgHandle* h, int n, float* obs, hipStream_t s) {
  if (!h->K.noise_on || !obs) return;
  if (!h->K.stop_at_done) { launch_obs_noise(h, n, nullptr, obs, s); return; }   // every row is the last step's
  hipLaunchKernelGGL(k_add_noise_rows, dim3((16 * h->N + 255) / 256), dim3(256), 0, s, h->K, h->D, h->K.noise_call, (unsigned)n, obs);
}

// the instantiations of a 16-lane kernel: {flat ground, heightfield} x {plain robot layer + body rows (the default), plain
// robot layer with toe spheres only, all options + body rows}.  The last one also serves body_contacts = 0 of the all-options
// layer (the rows are switched off at run time: KCfg.knee == 0), so a kernel still has six instantiations.
#define DISPATCH16(X)                                                                                                 \
  do {                                                                                                                \
    const bool pl_ = plain_config(h->K), kn_ = h->K.knee != 0, fl_ = h->K.terrain == 0;                               \
    if (fl_ && pl_ && kn_) X(true, true, true);                                                                       \
    else if (fl_ && pl_) X(true, false, true);                                                                        \
    else if (fl_) X(true, true, false);                                                                               \
    else if (pl_ && kn_) X(false, true, true);                                                                        \
    else if (pl_) X(false, false, true);                                                                              \
    else X(false, true, false);                                                                                       \
  } while (0)
#define LAUNCH16(KERN, grid, stream, ...)                                                                             \
  do {                                                                                                                \
    auto launch_ = [&](auto f_, auto k_, auto p_) {                                                                   \
      hipLaunchKernelGGL((KERN<decltype(f_)::value, decltype(k_)::value, decltype(p_)::value>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__); \
    };                                                                                                                \
    DISPATCH16(LAUNCH16_X_);                                                                                          \
  } while (0)
#define LAUNCH16_X_(F_, K_, P_) launch_(std::integral_constant<bool, F_>{}, std::integral_constant<bool, K_>{}, std::integral_constant<bool, P_>{})

// 4-lane kernels: {flat ground, heightfield} x {plain robot layer, all options, all options + 1 body row per leg, + 3 body rows}.
// A plain robot layer WITH body rows (the default configuration beyond 8192 robots) runs the all-options instantiation: the
// PLAIN specialisation is worth ~3 % where the tick fits the register file, but the 4-lane tick with body rows is at the
// 512-register budget with 528-544 B of scratch either way (profiles/r06_isa_baseline.json: k_rollout<flat, all options, 1>), its
// time is the spills' and the sweeps', and every further instantiation of it costs ~70 s of build for the six 4-lane kernels.
#define LAUNCH4(KERN, grid, stream, ...)                                                                              \
  do {                                                                                                                \
    const bool pl_ = plain_config(h->K);                                                                              \
    if (h->K.knee == 3 && h->K.terrain == 0) hipLaunchKernelGGL((KERN<true, false, 3>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__); \
    else if (h->K.knee == 3) hipLaunchKernelGGL((KERN<false, false, 3>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);  \
    else if (h->K.knee && h->K.terrain == 0) hipLaunchKernelGGL((KERN<true, false, 1>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__); \
    else if (h->K.knee) hipLaunchKernelGGL((KERN<false, false, 1>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);       \
    else if (h->K.terrain == 0 && pl_) hipLaunchKernelGGL((KERN<true, true>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);  \
    else if (h->K.terrain == 0) hipLaunchKernelGGL((KERN<true, false>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);   \
    else if (pl_) hipLaunchKernelGGL((KERN<false, true>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);                 \
    else hipLaunchKernelGGL((KERN<false, false>), grid, dim3(BLOCK), 0, stream, __VA_ARGS__);                         \
  } while (0)

#define CHECK_HANDLE(h)                                         \
  if (!(h)) return fail(ETG_ERR_BAD_ARG, "null handle");       \
  HIP_TRY(hipSetDevice((h)->device));

extern "C" int etg_set_params(EtgHandle* h, const float* dyn, const float* etg_w, const float* etg_b, int per_env,
                              const uint8_t* mask, void* stream) {
  CHECK_HANDLE(h);
  if ((etg_w == nullptr) != (etg_b == nullptr)) return fail(ETG_ERR_BAD_ARG, "etg_set_params: pass both etg_w and etg_b or neither");
  if (dyn) { h->all_cached = false; h->inval_seq++; }   // the settle depends on the dynamic parameters
  if (dyn && h->NX.ok)              // (and rows prepared for the masked robots' next episodes came with a settle of their own)
    hipLaunchKernelGGL(k_next_flags, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->NX.ok, mask, (unsigned char)0);
  hipLaunchKernelGGL(k_fin_clear, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, mask);   // cached restarts: stale
  hipLaunchKernelGGL(k_set_params, dim3(grid_for(h)), dim3(BLOCK), 0, (hipStream_t)stream, h->K, h->M, h->D, dyn, etg_w,
                     etg_b, per_env, mask);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

extern "C" int etg_set_heightfield(EtgHandle* h, const float* heights, void* stream) {
  CHECK_HANDLE(h);
  if (h->K.terrain != 1 || h->K.hf_nx < 2 || h->K.hf_ny < 2) return fail(ETG_ERR_STATE, "etg_set_heightfield: config has no heightfield");
  size_t bytes = (size_t)h->K.hf_nx * h->K.hf_ny * h->K.hf_bands * 4;
  if (!h->hf && hipMalloc((void**)&h->hf, bytes) != hipSuccess) return fail(ETG_ERR_ALLOC, "etg_set_heightfield: hipMalloc failed");
  HIP_TRY(hipMemcpyAsync(h->hf, heights, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  HIP_TRY(hipMemsetAsync(h->D.cache_ok, 0, h->N, (hipStream_t)stream));   // the settle depends on the terrain
  HIP_TRY(hipMemsetAsync(h->D.cache_off + (size_t)FIN_OK * h->N, 0, (size_t)h->N * 4, (hipStream_t)stream));
  h->all_cached = false;
  h->inval_seq++;
  h->K.hf = h->hf;
  return ETG_OK;
}

extern "C" int etg_set_external_force(EtgHandle* h, const float* force, void* stream) {
  CHECK_HANDLE(h);
  float* dst = h->D.ctl + (size_t)CT_FEXT * h->N;      // ctl[CT_FEXT + k][N]
  if (!force) {
    HIP_TRY(hipMemsetAsync(dst, 0, (size_t)3 * h->N * 4, (hipStream_t)stream));
    h->fext_set = false;
    h->K.ext_force = h->push_on;     // pending random pushes keep acting
    return ETG_OK;
  }
  hipLaunchKernelGGL(k_set_fext, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, force);
  HIP_TRY(hipGetLastError());
  h->fext_set = true;
  h->K.ext_force = 1;
  return ETG_OK;
}

extern "C" int etg_set_motor_strength(EtgHandle* h, const float* ratios, const uint8_t* mask, void* stream) {
  CHECK_HANDLE(h);
  hipLaunchKernelGGL(k_set_strength, dim3((4 * h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, ratios, mask);
  HIP_TRY(hipGetLastError());
  // ratios installed for anybody: the kernels with the robot-layer options read them (a later NULL for everybody switches back)
  h->K.strength_on = ratios ? 1 : (mask ? h->K.strength_on : 0);
  // the reset settle runs under the motor model too: cached settles of the touched robots are stale (k_set_strength cleared
  // their flags), and so are the settles prepared for their next episodes
  h->all_cached = false;
  h->inval_seq++;
  if (h->NX.ok)
    hipLaunchKernelGGL(k_next_flags, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->NX.ok, mask, (unsigned char)0);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

extern "C" int etg_random_pushes(EtgHandle* h, uint64_t seed, float prob, int duration_steps, float fmin, float fmax,
                                 void* stream) {
  CHECK_HANDLE(h);
  if (prob < 0.0f || duration_steps < 1 || fmax < fmin) return fail(ETG_ERR_BAD_ARG, "etg_random_pushes: bad arguments");
  hipLaunchKernelGGL(k_random_pushes, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D,
                     (unsigned long long)seed, h->push_calls++, prob, duration_steps, fmin, fmax);
  HIP_TRY(hipGetLastError());
  h->push_on = true;
  h->K.ext_force = 1;
  return ETG_OK;
}

extern "C" int etg_clear_pushes(EtgHandle* h, const uint8_t* mask, void* stream) {
  CHECK_HANDLE(h);
  hipLaunchKernelGGL(k_clear_pushes, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, mask);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

// ---- parameters for the NEXT episode (etg_prepare_next_dynamics)
// robots without a valid cached settle, counted after a masked reset (one workgroup; see EtgHandle::cached_report)
#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void __launch_bounds__(256) k_count_uncached(KCfg K, DevState D, unsigned seq, unsigned* report) {
  __shared__ unsigned part[256];
  unsigned n = 0;
  for (int env = threadIdx.x; env < K.n_env; env += 256) n += D.cache_ok[env] ? 0u : 1u;
  part[threadIdx.x] = n;
  __syncthreads();
  for (unsigned s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    __hip_atomic_store(&report[0], part[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __hip_atomic_store(&report[1], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
#endif
// host side: has a masked reset, run under the current invalidation number, reported that every robot is cached again?
static inline void refresh_all_cached(EtgHandle* h) {
  if (h->all_cached || !h->cached_report) return;
  const unsigned seq = h->cached_report[1];
  __sync_synchronize();
  if (seq == h->inval_seq && h->cached_report[0] == 0u) h->all_cached = true;
}

extern "C" int etg_prepare_next_dynamics(EtgHandle* h, const float* dyn, const uint8_t* mask, void* stream) {
  CHECK_HANDLE(h);
  if (!dyn) return fail(ETG_ERR_BAD_ARG, "etg_prepare_next_dynamics: dyn is null");
  refresh_all_cached(h);
  if (!h->was_reset || !h->all_cached)
    return fail(ETG_ERR_STATE, "etg_prepare_next_dynamics: every robot needs a valid cached settle (a full etg_reset, or masked resets "
                               "that covered every robot whose parameters, terrain or start offset changed)");
  const size_t N = h->N, NL = 4 * N;
  if (!h->NX.par) {
    struct { void** p; size_t bytes; } allocs[] = {
        {(void**)&h->NX.par, PR_N * NL * 4}, {(void**)&h->NX.dyn, ETG_DYN_DIM * N * 4}, {(void**)&h->NX.ok, N},
        {(void**)&h->nx_base, BS_N * N * 4}, {(void**)&h->nx_leg, LG_N * NL * 4}, {(void**)&h->nx_ring, (size_t)RING * 8 * NL * 4},
        {(void**)&h->nx_cache_ok, N}, {(void**)&h->nx_mask, N}};
    for (auto& a : allocs) {
      if (hipMalloc(a.p, a.bytes) != hipSuccess) return fail(ETG_ERR_ALLOC, "etg_prepare_next_dynamics: hipMalloc failed");
      HIP_TRY(hipMemsetAsync(*a.p, 0, a.bytes, (hipStream_t)stream));
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const dim3 ge((h->N + 255) / 256), gc((4 * h->N + 255) / 256), g16((h->N + 3) / 4), g4(grid_for(h));
  // a view of the robot arrays whose parameters are the NEXT rows, whose live state / ring are scratch (the robots keep running
  // on theirs) and whose settle cache is the real one: k_settle* leaves the settled state of the new rows in the cache
  DevState Dn = h->D;
  Dn.par = h->NX.par;
  Dn.dyn = h->NX.dyn;
  Dn.base = h->nx_base; Dn.leg = h->nx_leg; Dn.ring = h->nx_ring;
  Dn.cache_ok = h->nx_cache_ok;
  // the motor strength ratios are not derived from the dynamic_param row: the scratch settle runs under the robots' own
  // (found by the round-4 soak run: zeros there left the robot limp during the prepared settle whenever a non-plain kernel ran it)
  HIP_TRY(hipMemcpyAsync(h->NX.par + (size_t)PR_STR * NL, h->D.par + (size_t)PR_STR * NL, (size_t)(PR_N - PR_STR) * NL * 4, hipMemcpyDeviceToDevice, s));
  hipLaunchKernelGGL(k_next_old_enough, ge, dim3(256), 0, s, h->K, h->D, mask, h->nx_mask);
  mask = h->nx_mask;   // robots in the first RING ticks of their episode still read the cache's ring: not this time
  hipLaunchKernelGGL(k_set_params, dim3(grid_for(h)), dim3(BLOCK), 0, s, h->K, h->M, Dn, dyn, (const float*)nullptr,
                     (const float*)nullptr, 0, mask);   // (derives the rows; clears the scratch "cached" flag of the masked robots)
  if (h->lanes == 16) {
    LAUNCH16(k_settle16, g16, s, h->K, Dn, mask);
  } else {
    LAUNCH4(k_settle, g4, s, h->K, Dn, mask);
  }
  hipLaunchKernelGGL(k_cache_sync, gc, dim3(256), 0, s, h->K, Dn, mask);      // scratch ring -> the cache's ring
  hipLaunchKernelGGL(k_fin_clear, ge, dim3(256), 0, s, h->K, h->D, mask);     // cached first observations: stale
  hipLaunchKernelGGL(k_next_flags, ge, dim3(256), 0, s, h->K, h->NX.ok, mask, (unsigned char)1);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

extern "C" int etg_next_dynamics_pending(EtgHandle* h, uint8_t* pending, void* stream) {
  CHECK_HANDLE(h);
  if (!pending) return fail(ETG_ERR_BAD_ARG, "etg_next_dynamics_pending: pending is null");
  if (!h->NX.ok) { HIP_TRY(hipMemsetAsync(pending, 0, h->N, (hipStream_t)stream)); return ETG_OK; }
  HIP_TRY(hipMemcpyAsync(pending, h->NX.ok, h->N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return ETG_OK;
}

extern "C" int etg_reset(EtgHandle* h, const uint8_t* mask, float* obs, void* stream) {
  CHECK_HANDLE(h);
  if (!obs) return fail(ETG_ERR_BAD_ARG, "etg_reset: obs is null");
  if (h->K.terrain == 1 && !h->K.hf) return fail(ETG_ERR_STATE, "etg_reset: heightfield not set");
  h->was_reset = true;
  advance_obs_stream(h, 1);
  if (h->NX.ok) {   // rows prepared for the next episode of the masked robots: this reset starts it
    hipLaunchKernelGGL(k_next_take, dim3((4 * h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, h->NX, mask);
    hipLaunchKernelGGL(k_next_flags, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->NX.ok, mask, (unsigned char)0);
  }
  // 1. settle the masked robots that have no valid settle cache (kernel exits at once for the others)
  // 2. snapshot their ring / restore state + ring of the cached ones, mark everything masked as cached
  // 3. the part after the settle: control state, episode accumulators, first observation
  const dim3 g16((h->N + 3) / 4), g4(grid_for(h)), gc((4 * h->N + 255) / 256), ge((h->N + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (h->lanes == 16) {
    LAUNCH16(k_settle16, g16, s, h->K, h->D, mask);
  } else {
    LAUNCH4(k_settle, g4, s, h->K, h->D, mask);
  }
  hipLaunchKernelGGL(k_cache_sync, gc, dim3(256), 0, s, h->K, h->D, mask);
  hipLaunchKernelGGL(k_cache_mark, ge, dim3(256), 0, s, h->K, h->D, mask);
  if (h->lanes == 16) {
    LAUNCH16(k_finish16, g16, s, h->K, h->D, mask, obs);
  } else {
    LAUNCH4(k_finish, g4, s, h->K, h->D, mask, obs);
  }
  hipLaunchKernelGGL(k_fin_store, ge, dim3(256), 0, s, h->K, h->D, mask, (const float*)obs);   // the clean row: before the noise
  launch_obs_noise(h, 1, mask, obs, s);
  if (mask && !h->all_cached) {   // did this masked reset settle the last robots without a cache?  (answer read by a later call)
    if (!h->cached_report) {
      void* p = nullptr;
      if (hipHostMalloc(&p, 2 * sizeof(unsigned), hipHostMallocMapped) == hipSuccess) {
        h->cached_report = (volatile unsigned*)p;
        h->cached_report[0] = 1u; h->cached_report[1] = 0u;
        if (hipHostGetDevicePointer((void**)&h->cached_report_dev, p, 0) != hipSuccess) {
          (void)hipHostFree(p);
          h->cached_report = nullptr;
        }
      }
    }
    if (h->cached_report) hipLaunchKernelGGL(k_count_uncached, dim3(1), dim3(256), 0, s, h->K, h->D, h->inval_seq, h->cached_report_dev);
  }
  HIP_TRY(hipGetLastError());
  if (!mask) h->all_cached = true;
  return ETG_OK;
}

#if ETG_TU_HOST   // small kernels: compiled with the host side only (part 0)
__global__ void k_set_reset_offsets(KCfg K, DevState D, const float* xy, const uint8_t* mask) {
  const int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= K.n_env || (mask && !mask[env])) return;
  D.reset_off[env] = xy ? xy[2 * env] : 0.0f;
  D.reset_off[K.n_env + env] = xy ? xy[2 * env + 1] : 0.0f;
}
#endif
extern "C" int etg_set_sensor_noise(EtgHandle* h, const float* stdev, uint64_t seed) {
  CHECK_HANDLE(h);
  h->K.noise_on = 0;
  for (int k = 0; k < 5; k++) {
    const float v = stdev ? stdev[k] : 0.0f;
    if (!(v >= 0.0f)) return fail(ETG_ERR_BAD_ARG, "etg_set_sensor_noise: standard deviations must be >= 0");
    h->K.noise_std[k] = v;
    if (v > 0.0f) h->K.noise_on = 1;
  }
  h->K.noise_seed = seed;
  return ETG_OK;
}
extern "C" int etg_set_reset_offsets(EtgHandle* h, const float* xy, const uint8_t* mask, void* stream) {
  CHECK_HANDLE(h);
  hipLaunchKernelGGL(k_set_reset_offsets, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, xy, mask);
  hipLaunchKernelGGL(k_fin_clear, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, mask);
  HIP_TRY(hipGetLastError());
  if (h->K.terrain != 0) { h->all_cached = false; h->inval_seq++; }   // a heightfield settle is only valid at the offset it ran at
  return ETG_OK;
}

static int step_checks(EtgHandle* h, const float* action, float* obs, float* reward, uint8_t* done) {
  if (!obs || !reward || !done) return fail(ETG_ERR_BAD_ARG, "etg_step: obs/reward/done must be non-null");
  if (!h->was_reset) return fail(ETG_ERR_STATE, "etg_step: call etg_reset first");
  if (h->K.motor_mode == 2 && !action) return fail(ETG_ERR_BAD_ARG, "etg_step: the HYBRID motor mode needs a [N,60] command");
  return ETG_OK;
}

extern "C" int etg_step(EtgHandle* h, const float* action, const uint8_t* donef, float* obs, float* reward,
                        uint8_t* done, float* info, void* stream) {
  CHECK_HANDLE(h);
  if (int rc = step_checks(h, action, obs, reward, done)) return rc;
  advance_obs_stream(h, 1);
  const dim3 g16((h->N + 3) / 4);
  if (h->lanes == 16) {
    LAUNCH16(k_step16, g16, (hipStream_t)stream, h->K, h->D, action, donef, obs, reward, done, info);
  } else {
    LAUNCH4(k_step, dim3(grid_for(h)), (hipStream_t)stream, h->K, h->D, action, donef, obs, reward, done, info);
  }
  launch_obs_noise(h, 1, nullptr, obs, (hipStream_t)stream);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

// One control step of the sub-batch [env0, env0 + count) only -- every array is the WHOLE batch's ([N, ...], indexed by the
// robot's number as in etg_step), the other robots' rows and states are not touched.  Sub-batches are independent (a robot's
// step depends on its own state, parameters and action: tests/test_gpu_parity3.py::test_wave_neighbours_...), so G ranges
// enqueued on G streams let each sub-batch start its next step when ITS slowest wavefront has finished instead of the batch's
// (env.step(groups=G), rollout_policy(fused=False, groups=G): train.py:129-178's loop with one barrier per group).
extern "C" int etg_step_range(EtgHandle* h, int env0, int count, const float* action, const uint8_t* donef, float* obs,
                              float* reward, uint8_t* done, float* info, void* stream) {
  CHECK_HANDLE(h);
  if (int rc = step_checks(h, action, obs, reward, done)) return rc;
  const int per_block = h->lanes == 16 ? BLOCK / 16 : BLOCK / 4;
  if (env0 < 0 || count <= 0 || env0 + count > h->N || env0 % 16 != 0 || (count % 16 != 0 && env0 + count != h->N))
    return fail(ETG_ERR_BAD_ARG, "etg_step_range: env0 and count must be multiples of 16 robots (the last range may end at N)");
  // the sensor-noise stream moves on with the range that starts at robot 0: the ranges of one control step are called in
  // ascending order and share the position, so that every robot draws what it draws in etg_step
  if (env0 == 0) advance_obs_stream(h, 1);
  KCfg K = h->K;
  K.block0 = env0 / per_block;
  const dim3 grid((count + per_block - 1) / per_block);
  if (h->lanes == 16) {
    LAUNCH16(k_step16, grid, (hipStream_t)stream, K, h->D, action, donef, obs, reward, done, info);
  } else {
    LAUNCH4(k_step, grid, (hipStream_t)stream, K, h->D, action, donef, obs, reward, done, info);
  }
  if (h->K.noise_on)
    hipLaunchKernelGGL(k_add_noise_range, dim3((16 * count + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->K.noise_call, env0, count, obs);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

extern "C" int etg_step_autoreset(EtgHandle* h, const float* action, const uint8_t* donef, float* obs, float* reward,
                                  uint8_t* done, float* info, void* stream) {
  CHECK_HANDLE(h);
  refresh_all_cached(h);
  if (!h->all_cached) {   // some robot needs a simulated settle: step, then the general reset path masked by the done bytes
    int rc = etg_step(h, action, donef, obs, reward, done, info, stream);
    if (rc != ETG_OK) return rc;
    if (h->push_on && (rc = etg_clear_pushes(h, done, stream)) != ETG_OK) return rc;
    return etg_reset(h, done, obs, stream);
  }
  // every robot has a cached settle: step and restart in ONE launch (k_step16_ar / k_step_ar)
  if (int rc = step_checks(h, action, obs, reward, done)) return rc;
  advance_obs_stream(h, 2);                     // two rows per robot at most: the step's (position c) and the reset's (c + 1)
  hipStream_t s = (hipStream_t)stream;
  if (h->lanes == 16) {
    LAUNCH16(k_step16_ar, dim3((h->N + 3) / 4), s, h->K, h->D, action, donef, obs, reward, done, info, h->NX);
  } else {
    LAUNCH4(k_step_ar, dim3(grid_for(h)), s, h->K, h->D, action, donef, obs, reward, done, info, h->NX);
  }
  launch_obs_noise(h, 2, done, obs, s, /*invert=*/1, /*back=*/1);   // the step's rows of the robots that go on
  launch_obs_noise(h, 2, done, obs, s);                            // the reset rows
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

#ifdef ETG_TRACE_TICKS
extern "C" int etg_debug_set_trace(EtgHandle* h, float* buf) { if (!h) return ETG_ERR_BAD_ARG; h->K.trace = buf; return ETG_OK; }
#endif
extern "C" int etg_rollout_wave_cycles(EtgHandle* h, int64_t* cycles, int capacity, int* launches, int* waves, void* stream) {
  CHECK_HANDLE(h);
  if (!launches || !waves) return fail(ETG_ERR_BAD_ARG, "etg_rollout_wave_cycles: launches / waves are null");
  *launches = h->wc_launches;
  *waves = h->wc_waves;
  if (!cycles) return ETG_OK;      // the size query
  if (capacity < h->wc_launches * h->wc_waves) return fail(ETG_ERR_BAD_ARG, "etg_rollout_wave_cycles: the buffer is too small");
  if (h->wc_launches > 0)
    HIP_TRY(hipMemcpyAsync(cycles, h->wave_cycles, (size_t)h->wc_launches * h->wc_waves * sizeof(long long), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return ETG_OK;
}

extern "C" int etg_set_rollout_mode(EtgHandle* h, int simulate_finished) {
  CHECK_HANDLE(h);
  h->K.stop_at_done = simulate_finished ? 0 : 1;
  return ETG_OK;
}

extern "C" int etg_episode_stats(EtgHandle* h, float* ret, int32_t* len, void* stream) {
  CHECK_HANDLE(h);
  hipLaunchKernelGGL(k_episode_stats, dim3((h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, ret, len);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

// n_steps x env.step(action = 0) enqueued back-to-back on the stream (pretrain.py:129-154).
// The same source as etg_step in another kernel: the compiler contracts multiply-adds differently in the two contexts, so the
// two agree to rounding noise amplified by the contacts (joints ~1e-6 rad after one control step), not bit for bit.  The
// per-robot return/length since the last etg_reset are accumulated inside the kernels with alive masking.
extern "C" int etg_rollout_openloop(EtgHandle* h, int n_steps, float* obs, float* ret, int32_t* len, void* stream) {
  CHECK_HANDLE(h);
  if (n_steps <= 0 || !ret || !len) return fail(ETG_ERR_BAD_ARG, "etg_rollout_openloop: bad arguments");
  if (!h->was_reset) return fail(ETG_ERR_STATE, "etg_rollout_openloop: call etg_reset first");
  if (h->K.motor_mode != 2) {
    // fused: up to rollout_chunk control steps per launch, everything in registers in between
    const int ROLLOUT_CHUNK = h->rollout_chunk;
    const dim3 g16((h->N + 3) / 4), g4(grid_for(h));
    hipStream_t s = (hipStream_t)stream;
    // per-wave cycle counters of this call's launches (etg_rollout_wave_cycles): one 8-byte store per wave and launch
    const int n_launch = (n_steps + ROLLOUT_CHUNK - 1) / ROLLOUT_CHUNK, n_waves = (int)(h->lanes == 16 ? g16.x : g4.x);
    if (h->wc_cap < n_launch * n_waves) {
      if (h->wave_cycles) (void)hipFree(h->wave_cycles);
      h->wave_cycles = nullptr;
      h->wc_cap = 0;
      if (hipMalloc((void**)&h->wave_cycles, (size_t)n_launch * n_waves * sizeof(long long)) == hipSuccess) h->wc_cap = n_launch * n_waves;
    }
    h->wc_launches = h->wave_cycles ? n_launch : 0;
    h->wc_waves = n_waves;
    int launch = 0;
    for (int done_steps = 0; done_steps < n_steps; done_steps += ROLLOUT_CHUNK, launch++) {
      const int m = n_steps - done_steps < ROLLOUT_CHUNK ? n_steps - done_steps : ROLLOUT_CHUNK;
      const bool last = done_steps + m == n_steps;
      // stop_at_done: a robot's last row is written by the launch its episode ends in -- every launch writes to the caller's rows
      float* o = (obs && (last || h->K.stop_at_done)) ? obs : h->tmp_obs;
      long long* cyc = h->wave_cycles ? h->wave_cycles + (size_t)launch * n_waves : nullptr;
      const StatOut so = last ? StatOut{ret, (int*)len, cyc} : StatOut{nullptr, nullptr, cyc};
      advance_obs_stream(h, m);
      if (h->lanes == 16) {
        LAUNCH16(k_rollout16, g16, s, h->K, h->D, m, o, so);
      } else {
        LAUNCH4(k_rollout, g4, s, h->K, h->D, m, o, so);
      }
      if (o == obs) launch_rollout_noise(h, m, obs, s);
    }
    HIP_TRY(hipGetLastError());
    return ETG_OK;
  } else {
    for (int k = 0; k < n_steps; k++) {
      float* o = (obs && k == n_steps - 1) ? obs : h->tmp_obs;
      int rc = etg_step(h, nullptr, nullptr, o, h->tmp_reward, h->tmp_done, nullptr, stream);
      if (rc != ETG_OK) return rc;
    }
  }
  HIP_TRY(hipGetLastError());
  return etg_episode_stats(h, ret, len, stream);
}

extern "C" int etg_rollout_actions(EtgHandle* h, const float* actions, int n_steps, float* obs, float* rec_joint_angle, float* rec_imu,
                                   float* rec_obs, float* rec_reward, uint8_t* rec_done, float* ret, int32_t* len, void* stream) {
  CHECK_HANDLE(h);
  if (!actions || n_steps <= 0 || !obs) return fail(ETG_ERR_BAD_ARG, "etg_rollout_actions: actions / obs null or n_steps <= 0");
  if (!h->was_reset) return fail(ETG_ERR_STATE, "etg_rollout_actions: call etg_reset first");
  if (h->K.motor_mode == 2) return fail(ETG_ERR_STATE, "etg_rollout_actions: POSITION / TORQUE commands only ([n_steps,N,12] tapes)");
  if (h->K.noise_on && rec_obs) return fail(ETG_ERR_STATE, "etg_rollout_actions: per-step observations are recorded without sensor noise; switch it off");
  const int ROLLOUT_CHUNK = h->rollout_chunk;
  const dim3 g16((h->N + 3) / 4), g4(grid_for(h));
  hipStream_t s = (hipStream_t)stream;
  const size_t N = h->N;
  for (int d0 = 0; d0 < n_steps; d0 += ROLLOUT_CHUNK) {
    const int m = n_steps - d0 < ROLLOUT_CHUNK ? n_steps - d0 : ROLLOUT_CHUNK;
    advance_obs_stream(h, m);
    const TapeOut T = {rec_joint_angle ? rec_joint_angle + (size_t)d0 * N * ETG_ACT_DIM : nullptr, rec_imu ? rec_imu + (size_t)d0 * N * 6 : nullptr,
                       rec_obs ? rec_obs + (size_t)d0 * N * ETG_OBS_DIM : nullptr, rec_reward ? rec_reward + (size_t)d0 * N : nullptr,
                       rec_done ? rec_done + (size_t)d0 * N : nullptr};
    const float* a = actions + (size_t)d0 * N * ETG_ACT_DIM;
    if (h->lanes == 16) {
      LAUNCH16(k_rollout_actions16, g16, s, h->K, h->D, m, a, obs, T);
    } else {
      LAUNCH4(k_rollout_actions, g4, s, h->K, h->D, m, a, obs, T);
    }
    launch_rollout_noise(h, m, obs, s);
    // the last step of EVERY launch writes its row to `obs`, not to the tape: the tape gets its copy per chunk (rows 49, 99,
    // ... and n_steps - 1; recorded observations carry no sensor noise, which the check above enforces)
    if (rec_obs)
      HIP_TRY(hipMemcpyAsync(rec_obs + (size_t)(d0 + m - 1) * N * ETG_OBS_DIM, obs, N * ETG_OBS_DIM * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  HIP_TRY(hipGetLastError());
  if (ret || len) return etg_episode_stats(h, ret, len, stream);
  return ETG_OK;
}

extern "C" int etg_rollout_policy(EtgHandle* h, EtgPolicy* pol, int n_steps, float act_scale, int precision, int obs_col0,
                                  float* obs, float* ret, int32_t* len, void* stream) {
  CHECK_HANDLE(h);
  if (!pol || n_steps <= 0 || !obs) return fail(ETG_ERR_BAD_ARG, "etg_rollout_policy: bad arguments");
  if (!h->was_reset) return fail(ETG_ERR_STATE, "etg_rollout_policy: call etg_reset first");
  if (pol->device != h->device) return fail(ETG_ERR_BAD_ARG, "etg_rollout_policy: policy and simulator live on different devices");
  if (obs_col0 < 0 || obs_col0 + pol->in_dim > ETG_OBS_DIM || pol->out_dim != ETG_ACT_DIM)
    return fail(ETG_ERR_BAD_ARG, "etg_rollout_policy: the policy must map observation columns [col0, col0 + in_dim) to 12 actions");
  if (h->K.motor_mode == 2 || (h->lanes == 16 ? h->N % 16 != 0 : h->N % 64 != 0))
    return fail(ETG_ERR_STATE, "etg_rollout_policy: needs POSITION/TORQUE mode and whole workgroups: num_envs % 16 == 0 on the 16-lane "
                               "mapping (16 robots per workgroup), num_envs % 64 == 0 on the 4-lane one (64 per workgroup)");
  const bool bf = precision != 0;   // the bf16 kernels read the bf16 fragments packed at load time
  PolicyW P = {(const float4*)(bf ? pol->w1h : pol->w1), (const float4*)(bf ? pol->w2h : pol->w2), (const float4*)(bf ? pol->w3h : pol->w3), pol->b1, pol->b2, pol->b3,
               pol->in_dim, pol->out_dim, obs_col0};
  const int ROLLOUT_CHUNK = h->rollout_chunk;
  const dim3 g(h->N / 16), b(256);
  hipStream_t s = (hipStream_t)stream;
  const bool flat = h->K.terrain == 0;
  if (h->lanes == 4) {   // the 4-lane mapping: 64 robots per workgroup, two stacked policy tiles per pass (k_rollout_policy4)
    const dim3 g4(h->N / 64);
    const bool pl = plain_config(h->K);
    for (int done_steps = 0; done_steps < n_steps; done_steps += ROLLOUT_CHUNK) {
      const int m = n_steps - done_steps < ROLLOUT_CHUNK ? n_steps - done_steps : ROLLOUT_CHUNK;
      advance_obs_stream(h, m);
#define LAUNCH_POLICY4(F_, P_, K_)                                                                                    \
  do {                                                                                                                \
    if (precision == 0) hipLaunchKernelGGL((k_rollout_policy4<F_, false, P_, K_>), g4, b, 0, s, h->K, h->D, P, m, act_scale, obs); \
    else hipLaunchKernelGGL((k_rollout_policy4<F_, true, P_, K_>), g4, b, 0, s, h->K, h->D, P, m, act_scale, obs);    \
  } while (0)
      if (h->K.knee == 3)   // (the closed-loop kernel is not instantiated with three body rows per leg: callers step instead)
        return fail(ETG_ERR_STATE, "etg_rollout_policy: body_contacts = 3 is served by etg_step / etg_rollout_openloop / etg_rollout_actions");
      if (h->K.knee && flat) LAUNCH_POLICY4(true, false, 1);
      else if (h->K.knee) LAUNCH_POLICY4(false, false, 1);
      else if (flat && pl) LAUNCH_POLICY4(true, true, 0);
      else if (flat) LAUNCH_POLICY4(true, false, 0);
      else if (pl) LAUNCH_POLICY4(false, true, 0);
      else LAUNCH_POLICY4(false, false, 0);
#undef LAUNCH_POLICY4
      launch_rollout_noise(h, m, obs, s);
    }
    HIP_TRY(hipGetLastError());
    if (ret || len) return etg_episode_stats(h, ret, len, stream);
    return ETG_OK;
  }
  // precision 0: one wave per workgroup, the policy tile per wave (k_rollout_policy16w: no workgroup barrier); precision 1 (bf16
  // operands, an opt-in arithmetic): the 16-robot tile of 4 waves (k_rollout_policy16)
  const bool per_wave = precision == 0 && pol->in_dim <= 4 * pol::KQ1 && pol->out_dim <= 12;
  const PolicyW Pq = {(const float4*)pol->w12q, nullptr, (const float4*)pol->w3q, pol->b1, pol->b2, pol->b3, pol->in_dim, pol->out_dim, obs_col0};
  for (int done_steps = 0; done_steps < n_steps; done_steps += ROLLOUT_CHUNK) {
    const int m = n_steps - done_steps < ROLLOUT_CHUNK ? n_steps - done_steps : ROLLOUT_CHUNK;
    advance_obs_stream(h, m);
#define LAUNCH_POLICY16(F_, K_, P_)                                                                                   \
  do {                                                                                                                \
    if (per_wave) hipLaunchKernelGGL((k_rollout_policy16w<F_, K_, P_>), dim3(h->N / 4), dim3(BLOCK), 0, s, h->K, h->D, Pq, m, act_scale, obs); \
    else if (precision == 0) hipLaunchKernelGGL((k_rollout_policy16<F_, false, K_, P_>), g, b, 0, s, h->K, h->D, P, m, act_scale, obs); \
    else hipLaunchKernelGGL((k_rollout_policy16<F_, true, K_, P_>), g, b, 0, s, h->K, h->D, P, m, act_scale, obs);    \
  } while (0)
    DISPATCH16(LAUNCH_POLICY16);
#undef LAUNCH_POLICY16
    launch_rollout_noise(h, m, obs, s);   // every chunk ends in the caller's rows: the next chunk reads them
  }
  HIP_TRY(hipGetLastError());
  if (ret || len) return etg_episode_stats(h, ret, len, stream);
  return ETG_OK;
}

extern "C" int etg_rollout_policy_record(EtgHandle* h, EtgPolicy* pol, int n_steps, float act_scale, int precision, int obs_col0,
                                         float* obs, const float* noise, float* rec_obs, float* rec_act, float* rec_reward,
                                         uint8_t* rec_done, float* ret, int32_t* len, void* stream) {
  CHECK_HANDLE(h);
  if (!pol || n_steps <= 0 || !obs || !rec_obs || !rec_act || !rec_reward || !rec_done)
    return fail(ETG_ERR_BAD_ARG, "etg_rollout_policy_record: bad arguments");
  if (!h->was_reset) return fail(ETG_ERR_STATE, "etg_rollout_policy_record: call etg_reset first");
  if (noise && !pol->has_std) return fail(ETG_ERR_STATE, "etg_rollout_policy_record: a stochastic rollout needs etg_policy_load_std() first");
  if (pol->device != h->device) return fail(ETG_ERR_BAD_ARG, "etg_rollout_policy_record: policy and simulator live on different devices");
  if (obs_col0 < 0 || obs_col0 + pol->in_dim > ETG_OBS_DIM || pol->out_dim != ETG_ACT_DIM)
    return fail(ETG_ERR_BAD_ARG, "etg_rollout_policy_record: the policy must map observation columns [col0, col0 + in_dim) to 12 actions");
  if (h->lanes != 16 || h->N % 16 != 0 || h->K.motor_mode == 2)
    return fail(ETG_ERR_STATE, "etg_rollout_policy_record: needs the 16-lanes-per-robot mapping, num_envs % 16 == 0, POSITION/TORQUE mode");
  const bool bf = precision != 0;   // the bf16 kernels read the bf16 fragments packed at load time
  PolicyW P = {(const float4*)(bf ? pol->w1h : pol->w1), (const float4*)(bf ? pol->w2h : pol->w2), (const float4*)(bf ? pol->w3h : pol->w3), pol->b1, pol->b2, pol->b3,
               pol->in_dim, pol->out_dim, obs_col0};
  const int ROLLOUT_CHUNK = h->rollout_chunk;
  const dim3 g(h->N / 16), b(256);
  hipStream_t s = (hipStream_t)stream;
  const bool flat = h->K.terrain == 0;
  const size_t N = h->N;
  for (int done_steps = 0; done_steps < n_steps; done_steps += ROLLOUT_CHUNK) {
    const int m = n_steps - done_steps < ROLLOUT_CHUNK ? n_steps - done_steps : ROLLOUT_CHUNK;
    advance_obs_stream(h, m);
    const bool per_wave = precision == 0 && pol->in_dim <= 4 * pol::KQ1 && pol->out_dim <= 12;
    const RecOut R = {rec_obs + (size_t)done_steps * N * ETG_OBS_DIM, rec_act + (size_t)done_steps * N * ETG_ACT_DIM,
                      rec_reward + (size_t)done_steps * N, rec_done + (size_t)done_steps * N,
                      noise ? noise + (size_t)done_steps * N * ETG_ACT_DIM : nullptr,
                      (const float4*)(per_wave ? pol->w3sq : (bf ? pol->w3sh : pol->w3s)), pol->b3s};
    const PolicyW Pq = {(const float4*)pol->w12q, nullptr, (const float4*)pol->w3q, pol->b1, pol->b2, pol->b3, pol->in_dim, pol->out_dim, obs_col0};
#define LAUNCH_POLICY16R(F_, K_, P_)                                                                                  \
  do {                                                                                                                \
    if (per_wave) hipLaunchKernelGGL((k_rollout_policy16w_rec<F_, K_, P_>), dim3(h->N / 4), dim3(BLOCK), 0, s, h->K, h->D, Pq, m, act_scale, obs, R); \
    else if (precision == 0) hipLaunchKernelGGL((k_rollout_policy16_rec<F_, false, K_, P_>), g, b, 0, s, h->K, h->D, P, m, act_scale, obs, R); \
    else hipLaunchKernelGGL((k_rollout_policy16_rec<F_, true, K_, P_>), g, b, 0, s, h->K, h->D, P, m, act_scale, obs, R);    \
  } while (0)
    DISPATCH16(LAUNCH_POLICY16R);
#undef LAUNCH_POLICY16R
    launch_rollout_noise(h, m, obs, s);   // every chunk ends in the caller's rows: the next chunk reads them
  }
  HIP_TRY(hipGetLastError());
  if (ret || len) return etg_episode_stats(h, ret, len, stream);
  return ETG_OK;
}

extern "C" int etg_leg_kinematics(EtgHandle* h, const float* q, int n, float* foot, float* jac, void* stream) {
  CHECK_HANDLE(h);
  if (!q || !foot || n <= 0) return fail(ETG_ERR_BAD_ARG, "etg_leg_kinematics: bad arguments");
  hipLaunchKernelGGL(k_leg_kin, dim3((n + 3) / 4), dim3(BLOCK), 0, (hipStream_t)stream, h->K, h->M, q, n, foot, jac);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

extern "C" int etg_extra_sensors(EtgHandle* h, const float* obs, float* out, void* stream) {
  CHECK_HANDLE(h);
  if (!obs || !out) return fail(ETG_ERR_BAD_ARG, "etg_extra_sensors: null argument");
  const int total = h->N * ETG_EXTRA_DIM;
  hipLaunchKernelGGL(k_extra_sensors, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->M, h->D, obs, out);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

extern "C" int etg_get_state(EtgHandle* h, float* state, void* stream) {
  CHECK_HANDLE(h);
  if (!state) return fail(ETG_ERR_BAD_ARG, "etg_get_state: null");
  hipLaunchKernelGGL(k_get_state, dim3(grid_for(h)), dim3(BLOCK), 0, (hipStream_t)stream, h->K, h->D, state);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

// contact impulses of the feet, rows [N,12] <-> the SoA columns leg[LG_LAM + k][4N] (both lane mappings share the layout)
__global__ void __launch_bounds__(256) k_lam_io(KCfg K, DevState D, float* lam, int write) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  const int NL = 4 * K.n_env;
  if (col >= NL) return;
  const int env = col >> 2, leg = col & 3;
  for (int k = 0; k < 4; k++) {   // LG_LAM .. LG_LAMB: the foot's (n, t1, t2) and the body contact's normal
    float* cell = D.leg + (size_t)(LG_LAM + k) * NL + col;
    if (write) *cell = lam[(size_t)env * 16 + 4 * leg + k];
    else lam[(size_t)env * 16 + 4 * leg + k] = *cell;
  }
}
extern "C" int etg_get_contact_impulses(EtgHandle* h, float* lam, void* stream) {
  CHECK_HANDLE(h);
  if (!lam) return fail(ETG_ERR_BAD_ARG, "etg_get_contact_impulses: null");
  hipLaunchKernelGGL(k_lam_io, dim3((4 * h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, lam, 0);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}
extern "C" int etg_set_contact_impulses(EtgHandle* h, const float* lam, void* stream) {
  CHECK_HANDLE(h);
  if (!lam) return fail(ETG_ERR_BAD_ARG, "etg_set_contact_impulses: null");
  hipLaunchKernelGGL(k_lam_io, dim3((4 * h->N + 255) / 256), dim3(256), 0, (hipStream_t)stream, h->K, h->D, const_cast<float*>(lam), 1);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}

extern "C" int etg_set_state(EtgHandle* h, const float* state, void* stream) {
  CHECK_HANDLE(h);
  if (!state) return fail(ETG_ERR_BAD_ARG, "etg_set_state: null");
  hipLaunchKernelGGL(k_set_state, dim3(grid_for(h)), dim3(BLOCK), 0, (hipStream_t)stream, h->K, h->D, state);
  HIP_TRY(hipGetLastError());
  return ETG_OK;
}
#endif   // ETG_TU_HOST
