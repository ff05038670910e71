// policy_mlp.hip -- the residual-policy forward (the one dense contraction of the path):
//   act = tanh(W3 relu(W2 relu(W1 obs + b1) + b2) + b3) * act_scale
// reference: model/mujoco_model.py:44-60 (Actor.forward, mean head), alg/sac.py:60-63
// (predict = tanh(mean)), train.py:320 (act_bound scaling).  Weights keep torch's
// [out, in] row-major fp32 layout (mujoco_agent.py:61-65 state_dict keys l1/l2/mean_linear).
//
// One launch fuses the three layers.  A workgroup (NW waves, 16 by default) owns 16 consecutive robots;
// its activations (16 x 256 fp32) never leave LDS; the 327 KB of weights are streamed
// from L2 (they are shared by all workgroups and stay L2/MALL-resident).  At 4096 robots
// that is 256 workgroups = one per CU.
//   precision 0: v_mfma_f32_16x16x4_f32  -- exact fp32 (bitwise an fmaf chain)
//   precision 1: v_mfma_f32_16x16x32_bf16 -- operands rounded to bf16 (RNE) in registers
// MFMA operand mapping (16x16 tiles, wave64): lane l holds A[i = l&15][k-slot l>>4] and
// B[k-slot l>>4][j = l&15]; D[row = 4*(l>>4)+r][col = l&15].  The k index a slot covers is
// free as long as A and B agree, so each lane loads a CONTIGUOUS float4 (fp32 path) or
// 8 floats (bf16 path) of its row and feeds one component per MFMA.
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/etgsim.h"
#include "policy_core.h"

namespace {

using namespace pol;
#ifndef ETG_POLICY_WAVES
#define ETG_POLICY_WAVES 16   // measured at 4096 rows: 4 waves 10.6 us, 8 waves 10.1 us, 16 waves 9.5 us (fp32)
#endif
constexpr int NW = ETG_POLICY_WAVES;          // waves per workgroup
constexpr int THREADS = 64 * NW;

// Weights are re-laid out once (etg_policy_load) into MFMA-fragment order so that every wave-level
// weight fetch is ONE fully coalesced 1-KiB global_load_dwordx4:
//   packed[((tile * nkb + kb) * 64 + lane) * 4 + c] = W[tile*16 + (lane & 15)][kb*16 + 4*(lane >> 4) + c]
// (zero outside the matrix).  torch's [out, in] layout stays the API format.
__global__ void k_pack_weights(const float* __restrict__ w, int nout, int kdim, int ntiles, int nkb, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int total = ntiles * nkb * 64 * 4;
  if (idx >= total) return;
  const int c = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
  const int kb = blk % nkb, tile = blk / nkb;
  const int n = tile * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4) + c;
  out[idx] = (n < nout && k < kdim) ? w[(size_t)n * kdim + k] : 0.0f;
}

// the per-wave tile's packing of the two hidden layers as ONE stream of k-groups (policy_core.h: wave_hidden12):
//   out[((g * NCHW + chunk) * 64 + lane) * 4 + c] = W[64 chunk + lane][4 k + c],  W = W1, k = g for g < KQ1;  W = W2, k = g - KQ1 after
// (zero outside the matrices): a wave's fetch of one k-group is four coalesced 1-KiB loads at consecutive addresses
__global__ void k_pack_wave12(const float* __restrict__ w1, int in_dim, const float* __restrict__ w2, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pol::KG * pol::NCHW * 256) return;
  const int c = idx & 3, lane = (idx >> 2) & 63, blk = idx >> 8;
  const int chunk = blk % pol::NCHW, g = blk / pol::NCHW;
  const int n = 64 * chunk + lane;
  float v;
  if (g < pol::KQ1) { const int k = 4 * g + c; v = k < in_dim ? w1[(size_t)n * in_dim + k] : 0.0f; }
  else { const int k = 4 * (g - pol::KQ1) + c; v = w2[(size_t)n * pol::HID + k]; }
  out[idx] = v;
}

// the per-wave head's packing (policy_core.h: wave_head): out[(kq * 64 + lane) * 4 + c] = W[4 nb + j][64 slice + 4 kq + c] with
// lane = 4 (3 slice + nb) + j, nb < 3, slice < 4; zeros on lanes 48..63 and for neurons >= nout
__global__ void k_pack_head(const float* __restrict__ w, int nout, int kdim, float* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= pol::KQH * 256) return;
  const int c = idx & 3, lane = (idx >> 2) & 63, kq = idx >> 8;
  const int blk = lane >> 2, j = lane & 3, slice = blk / 3, nb = blk % 3;
  const int n = 4 * nb + j, k = 64 * slice + 4 * kq + c;
  out[idx] = (slice < 4 && n < nout && k < kdim) ? w[(size_t)n * kdim + k] : 0.0f;
}

// bf16 fragments from the fp32 ones: out[(tile * nkb / 2 + kp) * 64 + lane] = pack_bf16(in[(tile * nkb + 2 kp) * 64 + lane],
// in[(tile * nkb + 2 kp + 1) * 64 + lane]) -- the B operand of v_mfma_f32_16x16x32_bf16 as the kernels used to build it per use
__global__ void k_pack_bf16(const float4* __restrict__ in, int ntiles, int nkb, float4* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int nkp = nkb / 2;
  if (idx >= ntiles * nkp * 64) return;
  const int lane = idx & 63, blk = idx >> 6;
  const int kp = blk % nkp, tile = blk / nkp;
  const bf16x8 v = pack_bf16(in[(size_t)(tile * nkb + 2 * kp) * 64 + lane], in[(size_t)(tile * nkb + 2 * kp + 1) * 64 + lane]);
  out[idx] = __builtin_bit_cast(float4, v);
}

// SAMPLE = false: act = tanh(mean) * scale                                  (SAC.predict, alg/sac.py:60-63)
// SAMPLE = true : x = mean + exp(clamp(log_std, -20, 2)) * eps, act = tanh(x) * scale, and optionally
//                 logp = sum_j [N(x_j; mean_j, std_j) log-density - log(1 - tanh(x_j)^2 + 1e-6)]
//                 (SAC.sample, alg/sac.py:65-76; clamp of model/mujoco_model.py:58-59); eps is the caller's N(0,1)
template <bool BF16, bool SAMPLE>
__global__ void __launch_bounds__(THREADS) k_policy(const float* __restrict__ obs, int n, int in_dim,
                                                    const float4* __restrict__ w1p, const float* __restrict__ b1,
                                                    const float4* __restrict__ w2p, const float* __restrict__ b2,
                                                    const float4* __restrict__ w3p, const float* __restrict__ b3,
                                                    const float4* __restrict__ w3sp, const float* __restrict__ b3s,
                                                    const float* __restrict__ eps, int out_dim, float scale,
                                                    float* __restrict__ act, float* __restrict__ logp) {
  __shared__ __attribute__((aligned(16))) float bufA[TM * HS];
  __shared__ __attribute__((aligned(16))) float bufB[TM * HS];
  __shared__ float part[SAMPLE ? 2 : 1][NW][TM][16];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int row0 = blockIdx.x * TM;
  // obs tile -> LDS, zero padded (rows past n, columns past in_dim up to the 64-wide padded K)
  for (int idx = tid; idx < TM * 64; idx += THREADS) {
    const int r = idx >> 6, c = idx & 63;
    float v = 0.0f;
    if (c < in_dim && row0 + r < n) v = obs[(size_t)(row0 + r) * in_dim + c];
    bufA[r * HS + c] = v;
  }
  __syncthreads();
  hidden_layer<BF16, 4, NW>(bufA, w1p, b1, bufB, wave, lane);    // K padded to 64 = 4 k-blocks
  __syncthreads();
  hidden_layer<BF16, HID / 16, NW>(bufB, w2p, b2, bufA, wave, lane);
  __syncthreads();
  // output layer: one 16x16 tile per head (out_dim <= 16), K split over the waves
  output_partial<BF16, NW>(bufA, w3p, wave, lane, part[0]);
  if constexpr (SAMPLE) output_partial<BF16, NW>(bufA, w3sp, wave, lane, part[1]);
  __syncthreads();
  {
    const int r = tid >> 4, cidx = tid & 15;  // first 256 threads = 16 rows x 16 cols; a row = one 16-lane DPP row
    if (r >= TM) return;
    const bool live = cidx < out_dim && row0 + r < n;
    // fixed summation order over the NW K-slices
    float v = 0.0f;
#pragma unroll
    for (int q = 0; q < NW; q += 2) v += part[0][q][r][cidx] + part[0][q + 1][r][cidx];
    v += live ? b3[cidx] : 0.0f;
    float lp = 0.0f;
    if constexpr (SAMPLE) {
      float ls = 0.0f;
#pragma unroll
      for (int q = 0; q < NW; q += 2) ls += part[1][q][r][cidx] + part[1][q + 1][r][cidx];
      ls += live ? b3s[cidx] : 0.0f;
      ls = fminf(fmaxf(ls, -20.0f), 2.0f);
      const float e = live ? eps[(size_t)(row0 + r) * out_dim + cidx] : 0.0f;
      v = v + expf(ls) * e;
      const float a = tanhf(v);
      lp = live ? (-0.5f * e * e - ls - 0.9189385332046727f) - logf((1.0f - a * a) + 1e-6f) : 0.0f;
      if (live) act[(size_t)(row0 + r) * out_dim + cidx] = a * scale;
      if (logp) {   // sum over the row's 16 lanes (butterfly inside the DPP row)
        lp += __shfl_xor(lp, 1); lp += __shfl_xor(lp, 2); lp += __shfl_xor(lp, 4); lp += __shfl_xor(lp, 8);
        if (cidx == 0 && row0 + r < n) logp[row0 + r] = lp;
      }
    } else {
      if (live) act[(size_t)(row0 + r) * out_dim + cidx] = tanhf(v) * scale;
    }
  }
}

}  // namespace

extern "C" void etg_set_last_error_(const char* msg);

static size_t packed_floats(int ntiles, int nkb) { return (size_t)ntiles * nkb * 64 * 4; }

static int pfail(int code, const char* msg) {
  etg_set_last_error_(msg);
  return code;
}

extern "C" int etg_policy_create(int in_dim, int hidden, int out_dim, int device, EtgPolicy** out) {
  if (!out || in_dim <= 0 || in_dim > 64 || hidden != HID || out_dim <= 0 || out_dim > 16)
    return pfail(ETG_ERR_BAD_ARG, "etg_policy_create: need in_dim<=64, hidden==256, out_dim<=16");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return pfail(ETG_ERR_NO_DEVICE, "etg_policy_create: no HIP device");
  if (device < 0 || device >= ndev) return pfail(ETG_ERR_BAD_ARG, "etg_policy_create: bad device");
  if (hipSetDevice(device) != hipSuccess) return pfail(ETG_ERR_HIP, "hipSetDevice");
  EtgPolicy* p = new EtgPolicy();
  p->device = device; p->in_dim = in_dim; p->hidden = hidden; p->out_dim = out_dim;
  struct { float** q; size_t n; } a[] = {{&p->w1, packed_floats(HID / 16, 4)}, {&p->b1, (size_t)hidden},
                                         {&p->w2, packed_floats(HID / 16, HID / 16)}, {&p->b2, (size_t)hidden},
                                         {&p->w3, packed_floats(1, HID / 16)}, {&p->b3, (size_t)out_dim},
                                         {&p->w3s, packed_floats(1, HID / 16)}, {&p->b3s, (size_t)out_dim},
                                         {&p->w1h, packed_floats(HID / 16, 4) / 2}, {&p->w2h, packed_floats(HID / 16, HID / 16) / 2},
                                         {&p->w3h, packed_floats(1, HID / 16) / 2}, {&p->w3sh, packed_floats(1, HID / 16) / 2},
                                         {&p->w12q, (size_t)KG * NCHW * 256}, {&p->w3q, (size_t)KQH * 256}, {&p->w3sq, (size_t)KQH * 256}};
  p->has_std = 0;
  for (auto& x : a)
    if (hipMalloc((void**)x.q, x.n * 4) != hipSuccess) return pfail(ETG_ERR_ALLOC, "etg_policy_create: hipMalloc failed");
  *out = p;
  return ETG_OK;
}

extern "C" int etg_policy_load(EtgPolicy* p, const float* w1, const float* b1, const float* w2, const float* b2,
                               const float* w3, const float* b3, void* stream) {
  if (!p || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) return pfail(ETG_ERR_BAD_ARG, "etg_policy_load: null");
  if (hipSetDevice(p->device) != hipSuccess) return pfail(ETG_ERR_HIP, "hipSetDevice");
  hipStream_t s = (hipStream_t)stream;
  struct { float* d; const float* src; int nout, kdim, ntiles, nkb; float* dh; } pk[] = {
      {p->w1, w1, p->hidden, p->in_dim, HID / 16, 4, p->w1h},
      {p->w2, w2, p->hidden, p->hidden, HID / 16, HID / 16, p->w2h},
      {p->w3, w3, p->out_dim, p->hidden, 1, HID / 16, p->w3h}};
  for (auto& x : pk) {
    const int total = x.ntiles * x.nkb * 256;
    hipLaunchKernelGGL(k_pack_weights, dim3((total + 255) / 256), dim3(256), 0, s, x.src, x.nout, x.kdim, x.ntiles, x.nkb, x.d);
    const int th = x.ntiles * (x.nkb / 2) * 64;
    hipLaunchKernelGGL(k_pack_bf16, dim3((th + 255) / 256), dim3(256), 0, s, (const float4*)x.d, x.ntiles, x.nkb, (float4*)x.dh);
  }
  // the per-wave tile's packing (closed-loop kernels, precision 0; observation rows of up to pol::WS columns)
  if (p->in_dim <= 4 * KQ1)
    hipLaunchKernelGGL(k_pack_wave12, dim3(KG * NCHW), dim3(256), 0, s, w1, p->in_dim, w2, p->w12q);
  hipLaunchKernelGGL(k_pack_head, dim3(KQH), dim3(256), 0, s, w3, p->out_dim, p->hidden, p->w3q);
  struct { float* d; const float* src; size_t n; } c[] = {{p->b1, b1, (size_t)p->hidden}, {p->b2, b2, (size_t)p->hidden},
                                                         {p->b3, b3, (size_t)p->out_dim}};
  for (auto& x : c)
    if (hipMemcpyAsync(x.d, x.src, x.n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
      return pfail(ETG_ERR_HIP, "etg_policy_load: hipMemcpyAsync failed");
  if (hipGetLastError() != hipSuccess) return pfail(ETG_ERR_HIP, "etg_policy_load: pack launch failed");
  return ETG_OK;
}

extern "C" int etg_policy_load_std(EtgPolicy* p, const float* w_std, const float* b_std, void* stream) {
  if (!p || !w_std || !b_std) return pfail(ETG_ERR_BAD_ARG, "etg_policy_load_std: null");
  if (hipSetDevice(p->device) != hipSuccess) return pfail(ETG_ERR_HIP, "hipSetDevice");
  hipStream_t s = (hipStream_t)stream;
  const int total = (HID / 16) * 256;
  hipLaunchKernelGGL(k_pack_weights, dim3((total + 255) / 256), dim3(256), 0, s, w_std, p->out_dim, p->hidden, 1, HID / 16, p->w3s);
  hipLaunchKernelGGL(k_pack_bf16, dim3(((HID / 32) * 64 + 255) / 256), dim3(256), 0, s, (const float4*)p->w3s, 1, HID / 16, (float4*)p->w3sh);
  hipLaunchKernelGGL(k_pack_head, dim3(KQH), dim3(256), 0, s, w_std, p->out_dim, p->hidden, p->w3sq);
  if (hipMemcpyAsync(p->b3s, b_std, (size_t)p->out_dim * 4, hipMemcpyDeviceToDevice, s) != hipSuccess)
    return pfail(ETG_ERR_HIP, "etg_policy_load_std: hipMemcpyAsync failed");
  if (hipGetLastError() != hipSuccess) return pfail(ETG_ERR_HIP, "etg_policy_load_std: pack launch failed");
  p->has_std = 1;
  return ETG_OK;
}

#define ETG_POLICY_ARGS obs, n, p->in_dim, (const float4*)p->w1, p->b1, (const float4*)p->w2, p->b2, (const float4*)p->w3, p->b3, \
                        (const float4*)p->w3s, p->b3s
#define ETG_POLICY_ARGS_BF16 obs, n, p->in_dim, (const float4*)p->w1h, p->b1, (const float4*)p->w2h, p->b2, (const float4*)p->w3h, p->b3, \
                             (const float4*)p->w3sh, p->b3s
extern "C" int etg_policy_forward(EtgPolicy* p, const float* obs, int n, float act_scale, int precision, float* act,
                                  void* stream) {
  if (!p || !obs || !act || n <= 0) return pfail(ETG_ERR_BAD_ARG, "etg_policy_forward: bad arguments");
  if (hipSetDevice(p->device) != hipSuccess) return pfail(ETG_ERR_HIP, "hipSetDevice");
  dim3 grid((n + TM - 1) / TM), block(THREADS);
  if (precision == 0)
    hipLaunchKernelGGL((k_policy<false, false>), grid, block, 0, (hipStream_t)stream, ETG_POLICY_ARGS, nullptr, p->out_dim,
                       act_scale, act, nullptr);
  else
    hipLaunchKernelGGL((k_policy<true, false>), grid, block, 0, (hipStream_t)stream, ETG_POLICY_ARGS_BF16, nullptr, p->out_dim,
                       act_scale, act, nullptr);
  if (hipGetLastError() != hipSuccess) return pfail(ETG_ERR_HIP, "etg_policy_forward: launch failed");
  return ETG_OK;
}

extern "C" int etg_policy_sample(EtgPolicy* p, const float* obs, int n, const float* noise, float act_scale, int precision,
                                 float* act, float* logp, void* stream) {
  if (!p || !obs || !act || !noise || n <= 0) return pfail(ETG_ERR_BAD_ARG, "etg_policy_sample: bad arguments");
  if (!p->has_std) return pfail(ETG_ERR_STATE, "etg_policy_sample: etg_policy_load_std() first");
  if (hipSetDevice(p->device) != hipSuccess) return pfail(ETG_ERR_HIP, "hipSetDevice");
  dim3 grid((n + TM - 1) / TM), block(THREADS);
  if (precision == 0)
    hipLaunchKernelGGL((k_policy<false, true>), grid, block, 0, (hipStream_t)stream, ETG_POLICY_ARGS, noise, p->out_dim,
                       act_scale, act, logp);
  else
    hipLaunchKernelGGL((k_policy<true, true>), grid, block, 0, (hipStream_t)stream, ETG_POLICY_ARGS_BF16, noise, p->out_dim,
                       act_scale, act, logp);
  if (hipGetLastError() != hipSuccess) return pfail(ETG_ERR_HIP, "etg_policy_sample: launch failed");
  return ETG_OK;
}
#undef ETG_POLICY_ARGS
#undef ETG_POLICY_ARGS_BF16

extern "C" void etg_policy_destroy(EtgPolicy* p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  float* ptrs[] = {p->w1, p->b1, p->w2, p->b2, p->w3, p->b3, p->w3s, p->b3s, p->w1h, p->w2h, p->w3h, p->w3sh, p->w12q, p->w3q, p->w3sq};
  for (float* q : ptrs)
    if (q) (void)hipFree(q);
  delete p;
}
