// etg_replay.hip -- the transition side of the step boundary on the device.
//
// The reference's episode loops hand every transition to a replay memory (train.py:159, 240-241:
// rpm.append(obs, action, reward, next_obs, terminal), terminal = 1 - done, train.py:148-149) and sum the reward
// terms of `info` (train.py:150-156).  With thousands of robots per env.step() that is a masked, compacting append of
// up to N rows per control step: HBM-bound byte shuffling.  Done with a dozen framework calls per step it costs 5x the
// simulator step itself (host-bound); here it is three launches per control step:
//
//   etg_replay_begin   ring slots of the robots whose episode is still running (prefix sum over the alive bytes, one
//                      workgroup; rows of finished robots get slot -1 and are not stored), then obs / action rows
//                      scattered to their slots -- BEFORE the step overwrites the observation buffer
//   etg_replay_end     reward, next_obs, terminal = 1 - done scattered to the same slots; the alive-masked sums of the
//                      info columns and the success counter (info["velx"] >= 0.3) accumulated; alive &= !done
//
// Rows are coalesced float copies (one wave per row group); nothing is synchronised with the host: the write position
// and the transition count live in device memory.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/etgsim.h"

extern "C" void etg_set_last_error_(const char* msg);

namespace {

// slot[i] = (pos + #alive before i) % max_size for alive robots, -1 (row not stored) otherwise; pos_count += #alive
__global__ void __launch_bounds__(1024) k_replay_slots(const uint8_t* __restrict__ alive, int n, long long max_size,
                                                        long long* __restrict__ pos_count, int* __restrict__ slot) {
  __shared__ int wave_sum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  const long long pos = pos_count[0];
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int a = (i < n) && (alive == nullptr || alive[i] != 0);
    const unsigned long long bal = __ballot(a);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_sum[wave] = __popcll(bal);
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; w++) off += wave_sum[w];
    if (i < n) slot[i] = a ? (int)((pos + off + before) % max_size) : -1;
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; w++) tot += wave_sum[w];
      carry += tot;
    }
    __syncthreads();
  }
  if (tid == 0) {
    pos_count[0] = (pos + carry) % max_size;
    pos_count[1] += carry;
  }
}

// The same slots for big batches (a recorded episode: n_steps x N rows at once) in three small launches, using slot[] itself
// as scratch: (1) every 1024-row chunk leaves its number of live rows in slot[chunk * 1024]; (2) one workgroup turns those
// into ring offsets (pos + exclusive prefix) % max_size and advances pos_count; (3) every chunk adds its in-chunk prefix.
__global__ void __launch_bounds__(1024) k_replay_chunk_counts(const uint8_t* __restrict__ alive, int n, int* __restrict__ slot) {
  __shared__ int wave_sum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long i = (long long)blockIdx.x * 1024 + tid;
  const int a = (i < n) && (alive == nullptr || alive[i] != 0);
  const unsigned long long bal = __ballot(a);
  if (lane == 0) wave_sum[wave] = __popcll(bal);
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < 16; w++) tot += wave_sum[w];
    slot[(size_t)blockIdx.x * 1024] = tot;
  }
}
__global__ void __launch_bounds__(1024) k_replay_chunk_offsets(int n_chunks, long long max_size, long long* __restrict__ pos_count,
                                                              int* __restrict__ slot) {
  __shared__ long long part[1024];
  __shared__ long long carry;
  const int tid = threadIdx.x;
  if (tid == 0) carry = 0;
  __syncthreads();
  const long long pos = pos_count[0];
  for (int base = 0; base < n_chunks; base += 1024) {
    const int b = base + tid;
    const long long c = b < n_chunks ? slot[(size_t)b * 1024] : 0;
    part[tid] = c;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {           // inclusive Hillis-Steele scan of the chunk counts
      const long long v = tid >= off ? part[tid - off] : 0;
      __syncthreads();
      part[tid] += v;
      __syncthreads();
    }
    if (b < n_chunks) slot[(size_t)b * 1024] = (int)((pos + carry + part[tid] - c) % max_size);
    __syncthreads();
    if (tid == 1023) carry += part[1023];
    __syncthreads();
  }
  if (tid == 0) {
    pos_count[0] = (pos + carry) % max_size;
    pos_count[1] += carry;
  }
}
__global__ void __launch_bounds__(1024) k_replay_chunk_slots(const uint8_t* __restrict__ alive, int n, long long max_size, int* __restrict__ slot) {
  __shared__ int wave_sum[16];
  __shared__ int chunk_off;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long i = (long long)blockIdx.x * 1024 + tid;
  if (tid == 0) chunk_off = slot[(size_t)blockIdx.x * 1024];
  const int a = (i < n) && (alive == nullptr || alive[i] != 0);
  const unsigned long long bal = __ballot(a);
  const int before = __popcll(bal & ((1ull << lane) - 1ull));
  if (lane == 0) wave_sum[wave] = __popcll(bal);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; w++) off += wave_sum[w];
  if (i < n) slot[i] = a ? (int)(((long long)chunk_off + off + before) % max_size) : -1;
}

// mem_obs[slot[i], :] = obs[i, :], mem_act[slot[i], :] = act[i, :]; optionally act_scaled[i, :] = scale * act[i, :] (the
// command the step receives is the stored action times act_bound, train.py:147: one launch less per control step)
constexpr int BIG_BATCH = 65536;   // rows: above this (a recorded episode) a bounded grid walks the rows, skipping dead ones 64 at a time
// Row copies are HBM-bound byte shuffling: a wave copies a row at a time (lane = column, od + ad <= 64 for the 49 + 12
// floats of a transition) and skips the rows of finished robots 64 at a time.  ROWWISE = false: the general element-per-thread form.
template <bool ROWWISE>
__global__ void k_replay_begin_rows(const int* __restrict__ slot, int n, const float* __restrict__ obs, int od, float* __restrict__ mem_obs,
                                    const float* __restrict__ act, int ad, float* __restrict__ mem_act, float scale,
                                    float* __restrict__ act_scaled) {
  const int dsum = od + ad;
  // rows of finished robots are not stored (hundreds of thousands of writes to one scratch row would serialise in a
  // recorded episode); the scaled action is still produced for them: the step needs a command for every robot
  auto copy = [&](int i, int c) {
    const int sl = slot[i];
    const size_t s = (size_t)sl;
    if (c < od) {
      if (sl >= 0) mem_obs[s * od + c] = obs[(size_t)i * od + c];
    } else if (sl >= 0 || act_scaled) {
      const float a = act[(size_t)i * ad + (c - od)];
      if (sl >= 0) mem_act[s * ad + (c - od)] = a;
      if (act_scaled) act_scaled[(size_t)i * ad + (c - od)] = scale * a;
    }
  };
  if (ROWWISE) {
    // a bounded grid of waves walks the rows 64 at a time: one coalesced load of their slots, then only the rows that
    // have something to do (stored, or every row when the scaled action is wanted) are copied, lane = column
    const int lane = threadIdx.x & 63, per = blockDim.x >> 6;
    if (n <= BIG_BATCH) {   // a step's worth of rows: a wave per row, all in flight at once
      const int i = blockIdx.x * per + (threadIdx.x >> 6);
      if (i < n && lane < dsum) copy(i, lane);
      return;
    }
    for (int base = (blockIdx.x * per + (threadIdx.x >> 6)) * 64; base < n; base += gridDim.x * per * 64) {
      const int row = base + lane;
      const int sl = row < n ? slot[row] : -1;
      unsigned long long todo = __ballot(row < n && (sl >= 0 || act_scaled != nullptr));
      while (todo) {
        const int r = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        if (lane < dsum) copy(base + r, lane);
      }
    }
  } else {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * dsum) return;
    const int i = (int)(idx / dsum);
    copy(i, (int)(idx - (long long)i * dsum));
  }
}

// mem_next_obs[slot[i], :] = next_obs[i, :]; the row's first thread also stores reward / terminal, adds the info terms and
// updates the alive byte
template <bool ROWWISE>
__global__ void k_replay_end_rows(const int* __restrict__ slot, int n, const float* __restrict__ next_obs, int od,
                                  float* __restrict__ mem_next_obs, const float* __restrict__ reward, const uint8_t* __restrict__ done,
                                  float* __restrict__ mem_reward, float* __restrict__ mem_terminal, const float* __restrict__ info,
                                  int info_dim, int n_sum, int velx_col, float* __restrict__ info_sum, uint8_t* __restrict__ alive) {
  auto copy = [&](int i, int c) {
    const int sl = slot[i];
    if (sl < 0) return;                  // a finished robot: nothing to store, nothing to sum, alive stays 0
    const size_t s = (size_t)sl;
    mem_next_obs[s * od + c] = next_obs[(size_t)i * od + c];
    if (c != 0) return;
    const int d = done[i] != 0;
    mem_reward[s] = reward[i];
    mem_terminal[s] = d ? 0.0f : 1.0f;
    if (alive) {
      const int a = alive[i] != 0;
      if (a && info && info_sum) {
        for (int k = 0; k < n_sum; k++) info_sum[(size_t)i * (n_sum + 1) + k] += info[(size_t)i * info_dim + k];
        if (velx_col >= 0 && info[(size_t)i * info_dim + velx_col] >= 0.3f) info_sum[(size_t)i * (n_sum + 1) + n_sum] += 1.0f;
      }
      alive[i] = (uint8_t)(a && !d);
    }
  };
  if (ROWWISE) {
    const int lane = threadIdx.x & 63, per = blockDim.x >> 6;
    if (n <= BIG_BATCH) {
      const int i = blockIdx.x * per + (threadIdx.x >> 6);
      if (i < n && lane < od) copy(i, lane);
      return;
    }
    for (int base = (blockIdx.x * per + (threadIdx.x >> 6)) * 64; base < n; base += gridDim.x * per * 64) {
      const int row = base + lane;
      unsigned long long todo = __ballot(row < n && slot[row] >= 0);
      while (todo) {
        const int r = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        if (lane < od) copy(base + r, lane);
      }
    }
  } else {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * od) return;
    const int i = (int)(idx / od);
    copy(i, (int)(idx - (long long)i * od));
  }
}

int fail(const char* msg) {
  etg_set_last_error_(msg);
  return ETG_ERR_BAD_ARG;
}
// launch on the device that owns the memory (the caller's current device may be another one)
bool bind_device(const void* p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) == hipSuccess && hipSetDevice(attr.device) == hipSuccess) return true;
  (void)hipGetLastError();
  return false;
}
int hip_fail(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return ETG_OK;
  static thread_local char buf[256];
  snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  etg_set_last_error_(buf);
  return ETG_ERR_HIP;
}

}  // namespace

extern "C" int etg_replay_begin(const uint8_t* alive, int n, long long max_size, long long* pos_count, int32_t* slot,
                                const float* obs, int obs_dim, const float* act, int act_dim, float* mem_obs, float* mem_act,
                                float act_scale, float* act_scaled, void* stream) {
  if (n <= 0 || max_size <= 0 || n > max_size || !pos_count || !slot || !obs || !act || !mem_obs || !mem_act || obs_dim <= 0 || act_dim <= 0)
    return fail("etg_replay_begin: bad arguments (a batch must fit the memory)");
  if (!bind_device(mem_obs)) return fail("etg_replay_begin: mem_obs is not a device pointer");
  hipStream_t s = (hipStream_t)stream;
  if (n <= 8192) {
    hipLaunchKernelGGL(k_replay_slots, dim3(1), dim3(1024), 0, s, alive, n, max_size, pos_count, slot);
  } else {   // a recorded episode: hundreds of thousands of rows
    const int chunks = (n + 1023) / 1024;
    hipLaunchKernelGGL(k_replay_chunk_counts, dim3(chunks), dim3(1024), 0, s, alive, n, slot);
    hipLaunchKernelGGL(k_replay_chunk_offsets, dim3(1), dim3(1024), 0, s, chunks, max_size, pos_count, slot);
    hipLaunchKernelGGL(k_replay_chunk_slots, dim3(chunks), dim3(1024), 0, s, alive, n, max_size, slot);
  }
  if (obs_dim + act_dim <= 64) {
    hipLaunchKernelGGL(k_replay_begin_rows<true>, dim3((unsigned)(n <= BIG_BATCH ? (n + 3) / 4 : 2048)), dim3(256), 0, s, slot, n, obs, obs_dim, mem_obs, act, act_dim,
                       mem_act, act_scale, act_scaled);
  } else {
    const long long tot = (long long)n * (obs_dim + act_dim);
    hipLaunchKernelGGL(k_replay_begin_rows<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, slot, n, obs, obs_dim, mem_obs, act,
                       act_dim, mem_act, act_scale, act_scaled);
  }
  return hip_fail("etg_replay_begin");
}

extern "C" int etg_replay_end(const int32_t* slot, int n, const float* reward, const uint8_t* done, const float* next_obs, int obs_dim,
                              float* mem_reward, float* mem_terminal, float* mem_next_obs, const float* info, int info_dim,
                              int n_sum, int velx_col, float* info_sum, uint8_t* alive, void* stream) {
  if (n <= 0 || !slot || !reward || !done || !next_obs || !mem_reward || !mem_terminal || !mem_next_obs || obs_dim <= 0)
    return fail("etg_replay_end: bad arguments");
  if (info && (info_dim <= 0 || n_sum < 0 || n_sum > info_dim || velx_col >= info_dim)) return fail("etg_replay_end: bad info layout");
  if (!bind_device(mem_next_obs)) return fail("etg_replay_end: mem_next_obs is not a device pointer");
  hipStream_t s = (hipStream_t)stream;
  if (obs_dim <= 64) {
    hipLaunchKernelGGL(k_replay_end_rows<true>, dim3((unsigned)(n <= BIG_BATCH ? (n + 3) / 4 : 2048)), dim3(256), 0, s, slot, n, next_obs, obs_dim, mem_next_obs, reward,
                       done, mem_reward, mem_terminal, info, info_dim, n_sum, velx_col, info_sum, alive);
  } else {
    const long long tot = (long long)n * obs_dim;
    hipLaunchKernelGGL(k_replay_end_rows<false>, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, slot, n, next_obs, obs_dim, mem_next_obs,
                       reward, done, mem_reward, mem_terminal, info, info_dim, n_sum, velx_col, info_sum, alive);
  }
  return hip_fail("etg_replay_end");
}
