// policy_core.h -- device code of the fused policy MLP shared by policy_mlp.hip (stand-alone k_policy, 16 waves per
// 16-row tile) and etg_kernels.hip (the closed-loop rollout kernel, where the 4 waves that own a tile's 16 robots
// also run their physics).  See policy_mlp.hip for the MFMA operand mapping and the packed weight layout.
#pragma once
#include <hip/hip_runtime.h>

namespace pol {

constexpr int TM = 16;         // robots per tile / workgroup
constexpr int HID = 256;       // hidden width (Actor: 256)
constexpr int HS = HID + 4;    // LDS row stride in floats: rotates 16-B slots by one per row

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16(float x) {  // round to nearest even
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  unsigned short h = (unsigned short)(u >> 16);
  return __builtin_bit_cast(__bf16, h);
}
__device__ __forceinline__ bf16x8 pack_bf16(float4 a, float4 b) {
  bf16x8 v = {to_bf16(a.x), to_bf16(a.y), to_bf16(a.z), to_bf16(a.w), to_bf16(b.x), to_bf16(b.y), to_bf16(b.z), to_bf16(b.w)};
  return v;
}
// bf16 weights are packed ONCE at load time (k_pack_bf16): one 16-byte fragment = the bf16x8 B operand of a K = 32 MFMA, i.e.
// pack_bf16 of the fp32 fragments of two consecutive k-blocks, stored at [(tile * nkb / 2 + kb / 2) * 64 + lane]
__device__ __forceinline__ bf16x8 as_bf16x8(float4 v) { return __builtin_bit_cast(bf16x8, v); }

// The same layer over RT stacked 16-row tiles (in / out hold 16 * RT rows): every weight fragment is fetched ONCE and feeds RT
// MFMAs -- the weight delivery from L2 per robot falls by RT.  For the 4-lane closed-loop kernel (64 robots per workgroup).
template <bool BF16, int NKB, int NW_, int RT>
__device__ __forceinline__ void hidden_layer_rt(const float* in, const float4* __restrict__ wp, const float* b,
                                                float* out, int wave, int lane) {
  constexpr int nkb = NKB;
  constexpr int TPW = (HID / 16) / NW_;
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc[RT][TPW];
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int t = 0; t < TPW; t++) acc[rt][t] = {0.f, 0.f, 0.f, 0.f};
  const float4* base = wp + (size_t)(TPW * wave) * nkb * 64 + lane;
  const int tstride = nkb * 64;
  if (!BF16) {
    constexpr int PF = NKB < 2 ? NKB : 2;
    float4 w[PF + 1][TPW];
#pragma unroll
    for (int p = 0; p < PF; p++)
#pragma unroll
      for (int t = 0; t < TPW; t++) w[p][t] = base[t * tstride + p * 64];
#pragma unroll
    for (int kb = 0; kb < nkb; kb++) {
      float4 a[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) a[rt] = *reinterpret_cast<const float4*>(&in[(16 * rt + i) * HS + kb * 16 + 4 * g]);
      if (kb + PF < nkb) {
#pragma unroll
        for (int t = 0; t < TPW; t++) w[(kb + PF) % (PF + 1)][t] = base[t * tstride + (kb + PF) * 64];
        __builtin_amdgcn_sched_barrier(0);   // keep the loads here (see hidden_layer)
      }
      const float4* w0 = w[kb % (PF + 1)];
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int t = 0; t < TPW; t++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].x, w0[t].x, acc[rt][t], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int t = 0; t < TPW; t++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].y, w0[t].y, acc[rt][t], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int t = 0; t < TPW; t++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].z, w0[t].z, acc[rt][t], 0, 0, 0);
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int t = 0; t < TPW; t++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rt].w, w0[t].w, acc[rt][t], 0, 0, 0);
    }
  } else {
    // wp = the bf16 fragments (as_bf16x8): half the bytes of the fp32 ones, no conversion of the weights here
    constexpr int NKP = NKB / 2;
    const float4* bh = wp + (size_t)(TPW * wave) * NKP * 64 + lane;
    constexpr int PFH = NKP < 2 ? NKP : 2;
    float4 w[PFH + 1][TPW];
#pragma unroll
    for (int p = 0; p < PFH; p++)
#pragma unroll
      for (int t = 0; t < TPW; t++) w[p][t] = bh[(t * NKP + p) * 64];
#pragma unroll
    for (int kp = 0; kp < NKP; kp++) {
      bf16x8 av[RT];
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
        av[rt] = pack_bf16(*reinterpret_cast<const float4*>(&in[(16 * rt + i) * HS + 2 * kp * 16 + 4 * g]),
                           *reinterpret_cast<const float4*>(&in[(16 * rt + i) * HS + (2 * kp + 1) * 16 + 4 * g]));
      if (kp + PFH < NKP) {
#pragma unroll
        for (int t = 0; t < TPW; t++) w[(kp + PFH) % (PFH + 1)][t] = bh[(t * NKP + kp + PFH) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int t = 0; t < TPW; t++) {
        const bf16x8 bw = as_bf16x8(w[kp % (PFH + 1)][t]);
#pragma unroll
        for (int rt = 0; rt < RT; rt++) acc[rt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av[rt], bw, acc[rt][t], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      const int col = 16 * TPW * wave + 16 * t + i;
      const float bias = b[col];
#pragma unroll
      for (int r = 0; r < 4; r++) out[(16 * rt + 4 * g + r) * HS + col] = fmaxf(acc[rt][t][r] + bias, 0.0f);
    }
}

// output layer over RT stacked tiles: part[wave][16 * RT rows][16]
template <bool BF16, int NW_, int RT>
__device__ __forceinline__ void output_partial_rt(const float* bufA, const float4* __restrict__ whp, int wave, int lane,
                                                  float (*part)[RT * TM][16]) {
  constexpr int KPW = (HID / 16) / NW_;
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc[RT];
#pragma unroll
  for (int rt = 0; rt < RT; rt++) acc[rt] = {0.f, 0.f, 0.f, 0.f};
  const float4* base = whp + lane;
  if (!BF16) {
#pragma unroll
    for (int kk = 0; kk < KPW; kk++) {
      const int kb = KPW * wave + kk;
      const float4 bw = base[kb * 64];
#pragma unroll
      for (int rt = 0; rt < RT; rt++) {
        const float4 a = *reinterpret_cast<const float4*>(&bufA[(16 * rt + i) * HS + kb * 16 + 4 * g]);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw.x, acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw.y, acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw.z, acc[rt], 0, 0, 0);
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw.w, acc[rt], 0, 0, 0);
      }
    }
  } else {
    static_assert(KPW % 2 == 0, "the bf16 head needs an even number of k-blocks per wave");
#pragma unroll
    for (int kk = 0; kk < KPW; kk += 2) {
      const int kb = KPW * wave + kk;
      const bf16x8 bw = as_bf16x8(base[(kb / 2) * 64]);   // (whp = the bf16 fragments of the head)
#pragma unroll
      for (int rt = 0; rt < RT; rt++)
        acc[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
            pack_bf16(*reinterpret_cast<const float4*>(&bufA[(16 * rt + i) * HS + kb * 16 + 4 * g]),
                      *reinterpret_cast<const float4*>(&bufA[(16 * rt + i) * HS + (kb + 1) * 16 + 4 * g])), bw, acc[rt], 0, 0, 0);
    }
  }
#pragma unroll
  for (int rt = 0; rt < RT; rt++)
#pragma unroll
    for (int r = 0; r < 4; r++) part[wave][16 * rt + 4 * g + r][i] = acc[rt][r];
}

// The register ring of a hidden layer's weight fragments (fp32 path): PF k-blocks of the wave's TPW tiles in flight.  A
// caller that knows the NEXT layer can start its ring early (ring_prefetch before the current layer's MFMAs), so the L2
// latency of a layer's first fragments hides behind the previous layer instead of stalling the lone wave at every layer.
template <int NKB, int NW_>
struct WRing {
  static constexpr int PF = NKB < 3 ? NKB : 3;
  static constexpr int TPW = (HID / 16) / NW_;
  float4 w[PF + 1][TPW];
};

template <int NKB, int NW_>
__device__ __forceinline__ const float4* ring_base(const float4* __restrict__ wp, int wave, int lane) {
  return wp + (size_t)(WRing<NKB, NW_>::TPW * wave) * NKB * 64 + lane;
}

template <int NKB, int NW_>
__device__ __forceinline__ void ring_prefetch(WRing<NKB, NW_>& r, const float4* __restrict__ wp, int wave, int lane) {
  constexpr int PF = WRing<NKB, NW_>::PF, TPW = WRing<NKB, NW_>::TPW;
  const float4* base = ring_base<NKB, NW_>(wp, wave, lane);
#pragma unroll
  for (int p = 0; p < PF; p++)
#pragma unroll
    for (int t = 0; t < TPW; t++) r.w[p][t] = base[t * NKB * 64 + p * 64];
}

// out[16][this wave's 16*TPW columns] = relu(in[16][16*nkb] * W^T + b); wp = packed weights of the layer.
// PRE: the caller already started the ring (ring_prefetch)
template <bool BF16, int NKB, int NW_, bool PRE = false>
__device__ __forceinline__ void hidden_layer(const float* in, const float4* __restrict__ wp, const float* b,
                                             float* out, int wave, int lane, WRing<NKB, NW_>* pre = nullptr) {
  constexpr int nkb = NKB;
  constexpr int TPW = (HID / 16) / NW_;   // 16-column output tiles per wave
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; t++) acc[t] = {0.f, 0.f, 0.f, 0.f};
  // tile t of this wave: packed block ((TPW*wave + t) * nkb + kb); software-pipelined: the next k-block's
  // fragments are in flight while the MFMAs of the current one issue
  const float4* base = ring_base<NKB, NW_>(wp, wave, lane);
#ifdef ETG_PROBE_WEIGHTS_FROM_L1   // upper-bound probe (wrong results): every tile re-reads the same fragments -> L1 hits
  const int tstride = 0;
#else
  const int tstride = nkb * 64;
#endif
  if (!BF16) {
    // weight fragments come from L2 (hundreds of ns) while one k-block is only 16 MFMAs (~0.2 us): keep PF
    // k-blocks in flight in a register ring; the loop is fully unrolled so the ring indices are static
    constexpr int PF = WRing<NKB, NW_>::PF;
    WRing<NKB, NW_> own;
    WRing<NKB, NW_>& R = PRE ? *pre : own;
    if (!PRE) {
#pragma unroll
      for (int p = 0; p < PF; p++)
#pragma unroll
        for (int t = 0; t < TPW; t++) R.w[p][t] = base[t * tstride + p * 64];
    }
    float4 av[2];   // the A fragment of the next k-block is read from LDS while this one's MFMAs issue
    av[0] = *reinterpret_cast<const float4*>(&in[i * HS + 4 * g]);
#pragma unroll
    for (int kb = 0; kb < nkb; kb++) {
      const float4 a = av[kb & 1];
      if (kb + 1 < nkb) av[(kb + 1) & 1] = *reinterpret_cast<const float4*>(&in[i * HS + (kb + 1) * 16 + 4 * g]);
      if (kb + PF < nkb) {
#pragma unroll
        for (int t = 0; t < TPW; t++) R.w[(kb + PF) % (PF + 1)][t] = base[t * tstride + (kb + PF) * 64];
        // under the fused kernel's register pressure the scheduler sinks these loads to just before their use (seen in the
        // ISA: s_waitcnt vmcnt(0..2) in front of every k-block), which exposes the L2 latency per k-block: pin them here
        __builtin_amdgcn_sched_barrier(0);
      }
      const float4* w0 = R.w[kb % (PF + 1)];
      // k-component outer, tile inner: 4 independent accumulators back to back, so the 40-cycle
      // dependent-accumulator latency of v_mfma_f32_16x16x4_f32 never stalls the 32-cycle issue
#pragma unroll
      for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, w0[t].x, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, w0[t].y, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, w0[t].z, acc[t], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, w0[t].w, acc[t], 0, 0, 0);
    }
  } else {
    // one bf16 MFMA (K = 32) consumes two consecutive 16-wide k-blocks; the k-slot order is free as
    // long as A and B agree, so lane group g takes k = kb*16 + 4g..4g+3 from each block.  wp = the bf16 fragments packed at
    // load time (as_bf16x8): half the bytes of the fp32 ones and no weight conversion here; the same register ring
    constexpr int NKP = NKB / 2;
    const float4* bh = wp + (size_t)(TPW * wave) * NKP * 64 + lane;
    constexpr int PFH = NKP < 3 ? NKP : 3;
    float4 w[PFH + 1][TPW];
#pragma unroll
    for (int p = 0; p < PFH; p++)
#pragma unroll
      for (int t = 0; t < TPW; t++) w[p][t] = bh[(t * NKP + p) * 64];
#pragma unroll
    for (int kp = 0; kp < NKP; kp++) {
      const float4 a0 = *reinterpret_cast<const float4*>(&in[i * HS + 2 * kp * 16 + 4 * g]);
      const float4 a1 = *reinterpret_cast<const float4*>(&in[i * HS + (2 * kp + 1) * 16 + 4 * g]);
      if (kp + PFH < NKP) {
#pragma unroll
        for (int t = 0; t < TPW; t++) w[(kp + PFH) % (PFH + 1)][t] = bh[(t * NKP + kp + PFH) * 64];
        __builtin_amdgcn_sched_barrier(0);
      }
      const bf16x8 av = pack_bf16(a0, a1);
#pragma unroll
      for (int t = 0; t < TPW; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, as_bf16x8(w[kp % (PFH + 1)][t]), acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < TPW; t++) {
    const int col = 16 * TPW * wave + 16 * t + i;
    const float bias = b[col];
#pragma unroll
    for (int r = 0; r < 4; r++) out[(4 * g + r) * HS + col] = fmaxf(acc[t][r] + bias, 0.0f);
  }
}

// output layer: one 16x16 tile of head weights `whp`, K split over the NW_ waves; every wave leaves its partial
// 16x16 product in part[wave]
template <int NW_>
struct HeadFrag { float4 w[(HID / 16) / NW_]; };   // this wave's k-blocks of the head tile, fetched ahead (head_prefetch)

template <int NW_>
__device__ __forceinline__ void head_prefetch(HeadFrag<NW_>& f, const float4* __restrict__ whp, int wave, int lane) {
  constexpr int KPW = (HID / 16) / NW_;
#pragma unroll
  for (int kk = 0; kk < KPW; kk++) f.w[kk] = whp[lane + (KPW * wave + kk) * 64];
}

template <bool BF16, int NW_, bool PRE = false>
__device__ __forceinline__ void output_partial(const float* bufA, const float4* __restrict__ whp, int wave, int lane,
                                               float (*part)[TM][16], const HeadFrag<NW_>* pre = nullptr) {
  constexpr int KPW = (HID / 16) / NW_;   // k-blocks per wave
  const int i = lane & 15, g = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const float4* base = whp + lane;
  if (!BF16) {
#pragma unroll
    for (int kk = 0; kk < KPW; kk++) {
      const int kb = KPW * wave + kk;
      const float4 a = *reinterpret_cast<const float4*>(&bufA[i * HS + kb * 16 + 4 * g]);
      const float4 bw = PRE ? pre->w[kk] : base[kb * 64];
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw.w, acc, 0, 0, 0);
    }
  } else {
    // one bf16 MFMA spans two 16-wide k-blocks: with KPW = 1 only the even waves work here (odd ones add zeros)
    constexpr int KP = KPW < 2 ? 2 : KPW;
    const bool active = (wave % (KP / KPW)) == 0;
#pragma unroll
    for (int kk = 0; kk < (active ? KP : 0); kk += 2) {
      const int kb = KPW * wave + kk;
      const float4 a0 = *reinterpret_cast<const float4*>(&bufA[i * HS + kb * 16 + 4 * g]);
      const float4 a1 = *reinterpret_cast<const float4*>(&bufA[i * HS + (kb + 1) * 16 + 4 * g]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pack_bf16(a0, a1), as_bf16x8(base[(kb / 2) * 64]), acc, 0, 0, 0);   // whp = bf16 fragments
    }
  }
#pragma unroll
  for (int r = 0; r < 4; r++) part[wave][4 * g + r][i] = acc[r];
}

// ---------------------------------------------------------------------------------------------------------------------------
// The policy tile of ONE wave (closed-loop kernels of the 16-lane mapping, precision 0): the whole MLP for the wave's 4 robots
// on v_mfma_f32_4x4x1_16B_f32.  That instruction computes 16 independent 4 x 4 outer products per wave64 -- block b = lanes
// 4b .. 4b+3, D_b[i][j] += A[lane 4b+i] * B[lane 4b+j], D held at lane 4b+j / register i (layout verified on the hardware:
// tools/ubench/mfma4x4.hip).  Mapping: i = ROBOT of the wave (A at lane l = activation of robot l & 3: the same in every block),
// j = NEURON (B at lane l = weight of neuron 64 * chunk + l): one instruction = 4 robots x 64 neurons x one input, fp32 products
// accumulated in fp32, the matrix rate of v_mfma_f32_16x16x4_f32 with all four "rows" used.  Nothing is shared with another
// wave, so the closed loop has no workgroup barrier: a wave waits for its own 4 robots only (the 16-robot tile of rounds 1-5
// made 16 robots wait for the slowest at 5 barriers per control step: +32 us at 4096 robots).
// Price: every wave streams ALL the weights (325 KB fp32) from L2 per control step instead of a quarter of them; they are packed
// so that a wave's fetch is one coalesced 1-KiB global_load_dwordx4 (k_pack_wave12) and kept WR - 1 k-groups ahead in a register
// ring (wave_hidden12).
constexpr int WS = 68;       // row stride (floats) of the wave's padded observation rows: in_dim <= 64 columns in 16 float4 groups
                             // (68 = 4 mod 32: the four robots' rows start 4 banks apart, a 16-byte read of each is conflict-free)
constexpr int KQ1 = 16;      // k-groups (float4) of layer 1
constexpr int KQ2 = HID / 4; // k-groups of layer 2
constexpr int KQH = HID / 16;// k-groups of the head PER SLICE: the head splits K over 4 slices (wave_head)
constexpr int WR = 16; // ring slots of a hidden layer's weight stream: WR - 1 k-groups (x 4 chunks x 1 KiB) in flight

constexpr int NCHW = HID / 64;          // chunks of 64 neurons in a hidden layer
constexpr int KG = KQ1 + KQ2;           // k-groups of the two hidden layers, streamed as ONE sequence (k_pack_wave12)

// The two hidden layers of the wave tile.  Their weights are ONE stream of KG groups of NCHW x 1 KiB -- the 16 groups of layer
// 1, then the 64 of layer 2 (k_pack_wave12: wq[(g * NCHW + c) * 64 + lane] = W[64 c + lane][4 k .. 4 k + 3]) -- kept WR - 1 groups
// ahead in a register ring that never drains between the layers: the L2 latency of a layer's first groups is hidden behind the
// layer before it (and layer 1's behind the staging of the observation rows: wave_ring_start is called before it).  The k
// loop is ROLLED in blocks of WR groups (static ring indices inside a block); KQ1 is a multiple of WR, so the hand-over between
// the layers -- bias, relu, the activations through LDS into the other operand's layout -- falls between two blocks.
struct WaveRing { float4 w[WR][NCHW]; };
__device__ __forceinline__ void wave_ring_start(WaveRing& r, const float4* __restrict__ wq, int lane) {
#pragma unroll
  for (int p = 0; p < WR - 1; p++)
#pragma unroll
    for (int c = 0; c < NCHW; c++) r.w[p][c] = wq[(p * NCHW + c) * 64 + lane];
}
struct HeadW { float4 w[KQH]; };         // the output head's weights of this lane (wave_head below)
__device__ __forceinline__ void head_fetch(HeadW& h, const float4* __restrict__ wq, int lane) {
#pragma unroll
  for (int kq = 0; kq < KQH; kq++) h.w[kq] = wq[kq * 64 + lane];
}
// abuf: the 4 observation rows (stride WS); hA / hB: the activations of layers 1 / 2 (rows of HS floats); hw / wh: the head's
// weights, fetched while layer 2 runs
__device__ __forceinline__ void wave_hidden12(const float* abuf, float* hA, float* hB, WaveRing& ring, const float4* __restrict__ wq,
                                              const float* __restrict__ b1, const float* __restrict__ b2, HeadW& hw,
                                              const float4* __restrict__ wh, int lane) {
  static_assert(KQ1 % WR == 0 && KQ2 % WR == 0, "the layers' k-groups come in whole blocks of the ring");
  f32x4 acc[NCHW];
#pragma unroll
  for (int c = 0; c < NCHW; c++) acc[c] = {0.f, 0.f, 0.f, 0.f};
  const float* row = abuf + (lane & 3) * WS;
  float4 a_next = *reinterpret_cast<const float4*>(row);
  const float4* base = wq + lane;
  auto epilogue = [&](const float* __restrict__ b, float* out) {
#pragma unroll
    for (int c = 0; c < NCHW; c++) {
      const float bias = b[64 * c + lane];
#pragma unroll
      for (int i = 0; i < 4; i++) out[i * HS + 64 * c + lane] = fmaxf(acc[c][i] + bias, 0.0f);
      acc[c] = {0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();                      // (one wave per workgroup: a wait for its own LDS writes)
  };
#pragma unroll 1
  for (int kb = 0; kb < KG; kb += WR) {
    if (kb == KQ1) {                      // layer 1 is complete: its activations become layer 2's input rows
      epilogue(b1, hA);
      row = hA + (lane & 3) * HS - 4 * KQ1;   // (so that row + 4 g addresses column 4 (g - KQ1))
      a_next = *reinterpret_cast<const float4*>(row + 4 * kb);
    }
    if (kb == KQ1 + KQ2 / 2) head_fetch(hw, wh, lane);
#pragma unroll
    for (int j = 0; j < WR; j++) {
      const int g = kb + j;
      const float4 a = a_next;
      a_next = *reinterpret_cast<const float4*>(row + 4 * (g + 1));       // (one group past a layer's end: inside the row's padding)
      const int gl = g + (WR - 1) < KG ? g + (WR - 1) : KG - 1;           // (past the end: the last group again, unused)
#pragma unroll
      for (int c = 0; c < NCHW; c++) ring.w[(j + WR - 1) % WR][c] = base[(gl * NCHW + c) * 64];
      __builtin_amdgcn_sched_barrier(0);   // keep the fetches WR - 1 groups ahead (the scheduler otherwise sinks them to their use)
      const float4* w = ring.w[j];
      // input outer, chunk inner: NCHW independent accumulators back to back
#pragma unroll
      for (int c = 0; c < NCHW; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, w[c].x, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NCHW; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, w[c].y, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NCHW; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, w[c].z, acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NCHW; c++) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, w[c].w, acc[c], 0, 0, 0);
    }
  }
  epilogue(b2, hB);
}

// The output head (<= 12 neurons): 3 blocks of 4 neurons, K = 256 split over 4 SLICES of 64 -- block b = 3 * slice + nb of the
// instruction works on neurons 4 nb .. 4 nb + 3 and inputs 64 slice .. 64 slice + 63 (blocks 12..15 idle): 64 instructions and
// 16 KiB of weights instead of 256 and 64 KiB for a whole chunk of 64 lanes with 12 live ones.  k_pack_head lays the weights
// out accordingly: wq[kq * 64 + lane] = W[4 nb + j][64 slice + 4 kq ..], lane = 4 (3 slice + nb) + j.  The fetch is started
// by head_fetch (before the layer in front of it: nothing of it is exposed) and consumed by wave_head; the four slices'
// partial sums meet in LDS (part: [64 lanes][4 robots]): lane n < 12 returns neuron n's pre-activations of the 4 robots.
__device__ __forceinline__ f32x4 wave_head(const float* in, const HeadW& h, float* part, int lane) {
  const int blk = lane >> 2, slice = blk / 3;           // (blocks 12..15: slice 4 -- their weights are zeros, their rows clamped)
  const float* row = in + (lane & 3) * HS + 64 * (slice < 4 ? slice : 3);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kq = 0; kq < KQH; kq++) {
    const float4 a = *reinterpret_cast<const float4*>(row + 4 * kq);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, h.w[kq].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, h.w[kq].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, h.w[kq].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, h.w[kq].w, acc, 0, 0, 0);
  }
  *reinterpret_cast<f32x4*>(part + 4 * lane) = acc;
  __syncthreads();                                      // (one wave per workgroup: a wait for its own LDS writes)
  f32x4 out = {0.f, 0.f, 0.f, 0.f};
  if (lane < 12) {
    const f32x4 p0 = *reinterpret_cast<const f32x4*>(part + 4 * lane), p1 = *reinterpret_cast<const f32x4*>(part + 4 * (lane + 12));
    const f32x4 p2 = *reinterpret_cast<const f32x4*>(part + 4 * (lane + 24)), p3 = *reinterpret_cast<const f32x4*>(part + 4 * (lane + 36));
    out = (p0 + p1) + (p2 + p3);
  }
  __syncthreads();
  return out;
}

}  // namespace pol

// the policy handle (C-ABI EtgPolicy of include/etgsim.h)
struct EtgPolicy {
  int device, in_dim, hidden, out_dim;
  float *w1, *b1, *w2, *b2, *w3, *b3;  // w1/w2/w3 hold the PACKED (MFMA-fragment order) copies
  float *w3s, *b3s;                    // log-std head (etg_policy_load_std), packed like w3
  int has_std;
  float *w1h, *w2h, *w3h, *w3sh;       // the bf16 fragments (k_pack_bf16) behind precision = 1: half the size, no per-use conversion
  float *w12q, *w3q, *w3sq;            // the per-wave tile's packing (k_pack_wave12 / k_pack_head; pol::wave_hidden12, wave_head), fp32
};
