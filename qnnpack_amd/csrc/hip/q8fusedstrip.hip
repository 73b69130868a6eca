/*
 * q8fusedstrip.hip -- the inverted-residual block [1x1 expand ->] 3x3 depthwise (pad 1, stride 1 | 2) -> 1x1 project
 * [-> + block input] as ONE launch, round 4's rewrite of q8fused.hip (the reference runs the
 * three or four operators one after another: src/q8gemm, src/q8dwconv, src/q8vadd; bench/convolution.cc:464-536 lists
 * the shape triples).
 *
 * Why a second kernel: q8fused.hip keeps every weight in LDS (so it refuses MobileNetV2's blocks b7-b16), walks small
 * tiles with four waves in lock step, computes the depthwise stage with ~50 VALU instructions per output dword and the
 * kernel-zero-point row sums of both GEMMs on top: 0.62x of not fusing (profiles/r03). Here:
 *   - a workgroup of sixteen waves owns a strip of output rows of one image over the image's full width (the whole image
 *     for the 14x14 / 7x7 blocks at large batches) and walks the hidden channels in CHUNKS of 2 / 4 / 8 blocks of 32, so
 *     nothing has to fit LDS but one chunk of the hidden tensor; weights stream from L2 (every workgroup reads the same);
 *   - all three stages use zero-point-CENTRED int8 images (q8gemm256c.hip's trick: w ^ flip, activations ^ flip, flip =
 *     0x7F for kernel zero point 127 and 0x80 for 128): no row sums anywhere, accumulators start at the folded bias;
 *   - the DEPTHWISE stage runs on the matrix cores as well: for a 32-channel block and a tap, the weight operand is a
 *     32 x 32 matrix with the tap's 32 weights on its diagonal, built in registers (five instructions per tap and
 *     chunk), and the activation operand is a plain ds_read_b128 of the hidden tile at the tap's pixel offset -- nine
 *     MFMAs per 32 pixels x 32 channels, 1/32 of the pipe's arithmetic but none of the VALU's, which the three
 *     requantizations need;
 *   - every intermediate is requantized with the stand-alone operator's own parameters (the sequence and clamp class
 *     are picked per stage by a wave-uniform switch), so the output is bit-identical to running the operators in sequence.
 *
 * LDS (dynamic, planned by plan()): input strip [rows][kb1 * 32 + 16] (^ flip1; also the residual operand), hidden chunk
 * as a zero-point-padded image [(rows + 2)][W + 2][CB * 32 + 16] (^ flip2), depthwise output chunk [rows][CB * 32 + 16]
 * (^ flip3), the depthwise weights [9][hidden_pad] and the three bias vectors.
 * Requirements (plan()): kernel zero points 127 / 128 (the caller builds the images only then), channel counts % 4 == 0,
 * input channels <= 160, project tiles <= 64, LDS fit.
 */
#include <hip/hip_runtime.h>

#include <mutex>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "add_math.hip.h"
#include "igemm_epilogue.hip.h"
#include "igemm_params.h"
#include "per_device.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#ifndef QNNP_STRIP_WAVES
#define QNNP_STRIP_WAVES 16                // A/B at build time (make EXTRA=-DQNNP_STRIP_WAVES=8): sixteen waves of <= 128 registers,
#endif                                     // or eight of <= 256 with two row tiles per round and prefetched weights: the stages
                                           // are latency-bound (12 cycles per instruction and SIMD with two waves), so four waves
                                           // per SIMD win: MobileNetV2 at batch 128, 549.5 against 687.3 us (profiles/r04)
constexpr int kWaves = QNNP_STRIP_WAVES;
constexpr int kThreads = kWaves * 64;
constexpr bool kTwoTiles = kWaves == 8;    // two independent accumulator chains per wave
constexpr bool kPrefetch = kWaves == 8;    // weight fragments fetched a stage ahead (register budget)
constexpr int kTilesPerRound = kTwoTiles ? 2 : 1;
constexpr int kMaxKb1 = 5;                 // input channels <= 160
constexpr int kMaxPairs = kWaves == 8 ? 3 : 2;   // project accumulators per wave (row tiles x output blocks <= 24 / 32 per strip)
constexpr uint32_t kLdsLimit = 160 * 1024;

struct StripParams {
  const uint8_t* input;
  uint8_t* output;
  uint32_t batch, H, W, OH, OW, cin, ch, cout, in_stride, out_stride, stride;
  uint32_t strips, rps;                      // strips per image, output rows per strip
  uint32_t kb1, nb1, nb3, cb, nchunks;       // K blocks of the expand stage, hidden blocks, output blocks, blocks per chunk
  uint32_t in_pitch, hid_pitch, dw_pitch;
  uint32_t in_off, hid_off, dw_off, w2_off, b1_off, b2_off, b3_off, hid_bytes;
  uint32_t wlds, we_off, wp_off;             // 1: a chunk's expand / project fragments are staged in LDS (below), at these offsets
  uint32_t inv_w, inv_ow;                    // magic32(W), magic32(OW): x / d == udiv(x, magic)
  uint32_t in_piece, ppp_magic;              // bytes per input staging piece (16 / 8 / 4), magic32(cin / in_piece)
  uint32_t flip1, flip2, flip3, hid_pad4;
  uint32_t mode1, mode2, mode3;
  uint32_t has_expand, has_res, store_mode, hidden_pad, output_pad;
  const int8_t* w1; const int32_t* b1;
  const int8_t* w2; const int32_t* b2;
  const int8_t* w3; const int32_t* b3;
  RequantDev rq1, rq2, rq3;
  qnnp_hip_add_params add;
  unsigned long long* trace;                 // measurement builds only
};

#ifdef QNNP_ENABLE_ABLATION
#define QNNP_S_STAMP(slot)                                                                            \
  do {                                                                                                 \
    if (p.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 4096)                                   \
      p.trace[(blockIdx.x * 4) * 8 + (slot)] = __builtin_readcyclecounter();   /* slots 0..31 = items 0..3 */                          \
  } while (0)
#else
#define QNNP_S_STAMP(slot) do { } while (0)
#endif

/* rounding sequence x clamp class of a stage: seq * 3 + clamp (requant.hip.h: q31_requantize_pack4_clamp) */
inline uint32_t stage_mode(const RequantDev& rq, bool* offset)
{
  uint32_t seq = 2, clamp = 0;
  requant_dispatch_ofs(rq, [&](auto s, auto full) {
    constexpr int kSeq = decltype(s)::value;
    seq = kSeq == kRqShift0Ofs ? 0u : (kSeq == kRqBoundedOfs ? 1u : 2u);
    clamp = decltype(full)::value ? 0u : (rq.zp_late == 0 ? 1u : 2u);
  });
  *offset = seq != 2;
  return seq * 3 + clamp;
}

/* a wave-uniform pointer, in scalar registers for good (q8gemm256c.hip) */
__device__ __forceinline__ const uint8_t* scalar_ptr(const uint8_t* ptr)
{
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}

/* LDS-DMA of one 1 KiB weight fragment (q8gemm256c.hip): 16 bytes per lane from base + lane * 16 to dst + lane * 16,
 * as an instruction the compiler does not track -- behind the builtin it drains vmcnt(0) before EVERY later LDS access
 * of the wave, and the point of these loads is to land under the stage that runs meanwhile. The waits are explicit. */
__device__ __forceinline__ void dma_fragment(const uint8_t* base, uint32_t lane_offset, uint8_t* dst)
{
  const uint32_t m0v = __builtin_amdgcn_readfirstlane(
      static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*) dst)));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(lane_offset), "s"(scalar_ptr(base)), "s"(m0v) : "memory");
}

/* x / d by multiplication: magic = floor(2^32 / d) + 1 is exact for x < 2^32 / d (pixel and dword indices here are
 * below 2^16); 0 stands for d == 1 */
__device__ __forceinline__ uint32_t udiv(uint32_t x, uint32_t magic) { return magic == 0 ? x : __umulhi(x, magic); }

/* global -> LDS staging of `total` dwords by the whole workgroup, U loads in flight per thread before the first store
 * (a plain load-store loop is one L2 / HBM round trip per iteration: fourteen of them in front of the first MFMA of b0) */
template <int U, typename LoadF, typename StoreF>
__device__ __forceinline__ void stage_dwords(uint32_t total, uint32_t tid, LoadF load, StoreF store)
{
  for (uint32_t base = tid; base < total; base += kThreads * U) {
    uint32_t v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t i = base + static_cast<uint32_t>(u) * kThreads;
      v[u] = i < total ? load(i) : 0u;
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t i = base + static_cast<uint32_t>(u) * kThreads;
      if (i < total) store(i, v[u]);
    }
  }
}

/* 16 accumulators of one lane (4 groups of 4 consecutive channels) -> 4 packed dwords */
__device__ __forceinline__ void requant16(const v16i& a, uint32_t mode, const RequantDev& rq, uint32_t (&pk)[4])
{
#define QNNP_STRIP_RQ(SEQ, CLAMP)                                                                                     \
  _Pragma("unroll") for (int rg = 0; rg < 4; rg++)                                                                    \
    pk[rg] = q31_requantize_pack4_clamp<SEQ, CLAMP>(a[rg * 4 + 0], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3], rq);
  switch (mode) {                               // wave-uniform
    case 0: { QNNP_STRIP_RQ(kRqShift0Ofs, 0) } break;
    case 1: { QNNP_STRIP_RQ(kRqShift0Ofs, 1) } break;
    case 2: { QNNP_STRIP_RQ(kRqShift0Ofs, 2) } break;
    case 3: { QNNP_STRIP_RQ(kRqBoundedOfs, 0) } break;
    case 6: { QNNP_STRIP_RQ(kRqGeneral, 0) } break;
    case 7: { QNNP_STRIP_RQ(kRqGeneral, 1) } break;
    default: { QNNP_STRIP_RQ(kRqGeneral, 2) } break;
  }
#undef QNNP_STRIP_RQ
}

/* the lane's 16 bytes of its row: lane l gets channels 0..15 of the 32-channel tile, lane l + 32 channels 16..31 */
__device__ __forceinline__ uint4 gather16(const uint32_t (&pk)[4])
{
  const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
  const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
  return make_uint4(s02[0], s02[1], s13[0], s13[1]);
}

/* MODE: the three stages' common rounding mode when it is one the kernel is specialised for (0: shift 0, saturating
 * clamp -- the reference bench's; 3: bounded shift >= 1, saturating clamp -- what data-derived scales give), -1: a
 * wave-uniform switch per stage on p.mode1 / 2 / 3. */
template <int MODE>
__device__ __forceinline__ void requant16m(const v16i& a, uint32_t mode, const RequantDev& rq, uint32_t (&pk)[4])
{
  if constexpr (MODE == 0) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) pk[rg] = q31_requantize_pack4_clamp<kRqShift0Ofs, 0>(a[rg * 4 + 0], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3], rq);
  } else if constexpr (MODE == 3) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) pk[rg] = q31_requantize_pack4_clamp<kRqBoundedOfs, 0>(a[rg * 4 + 0], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3], rq);
  } else {
    requant16(a, mode, rq, pk);
  }
}

/* C MFMAs of one project accumulator over the chunk's blocks: straight-line code per block count */
template <int C>
__device__ __forceinline__ void project_blocks(v16i& acc, const v4i (&w)[8], const uint8_t* arow)
{
#pragma unroll
  for (int kbl = 0; kbl < C; kbl++) {
    const v4i a = *reinterpret_cast<const v4i*>(arow + kbl * 32);
    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[kbl], a, acc, 0, 0, 0);
  }
}

/*
 * KB1: 32-deep K blocks of the expand stage (input channels / 32, rounded up) -- compile time, so that the expand
 * loop is straight-line code; HAS_EXPAND: false = the depthwise stage reads the block input (MobileNetV2's first block).
 */
/*
 * WLDS (the 14 x 14 / 7 x 7 blocks, where a stage is one tile per wave and its weight fetch from L2 was most of it -- 2.3 k
 * of a 7.9 k-cycle chunk in stage E, 1.9 k in stage P, plus the barrier waits behind the slowest wave): a chunk's expand
 * and project fragments arrive by LDS-DMA while the previous stages run, and the stages read them with ds_read_b128.
 *   after the barrier behind stage E of chunk c: every wave is done with E(c) and with P(c - 1), so the expand buffer
 *   takes chunk c + 1 and the project buffer chunk c; each wave waits for its own pieces in front of the barrier
 *   behind stage D, which publishes them -- a whole depthwise stage to land in.
 */
template <int KB1, int MODE, bool HAS_EXPAND, bool WLDS = false>
__global__ __launch_bounds__(kThreads)
void q8_fused_strip_kernel(const StripParams p)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  uint8_t* in_lds = lds + p.in_off;        // [nE][in_pitch] block input ^ flip1
  uint8_t* hid = lds + p.hid_off;          // [padded rows][W + 2][hid_pitch] hidden chunk ^ flip2
  uint8_t* dwb = lds + p.dw_off;           // [nD][dw_pitch] depthwise output chunk ^ flip3
  const int8_t* w2_lds = reinterpret_cast<const int8_t*>(lds + p.w2_off);   // [9][hidden_pad]
  const int32_t* b1_lds = reinterpret_cast<const int32_t*>(lds + p.b1_off);
  const int32_t* b2_lds = reinterpret_cast<const int32_t*>(lds + p.b2_off);
  const int32_t* b3_lds = reinterpret_cast<const int32_t*>(lds + p.b3_off);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t col = lane & 31u;
  const uint32_t khalf = lane >> 5;
  const bool residual = p.has_res != 0;

  QNNP_S_STAMP(0);
  // ---- this workgroup's strip ----
  const uint32_t img = blockIdx.x / p.strips;
  const uint32_t sidx = blockIdx.x - img * p.strips;
  const uint32_t oy0 = sidx * p.rps;
  const uint32_t oy1 = min(p.OH, oy0 + p.rps);
  const uint32_t s = p.stride;
  const int32_t hy_lo = static_cast<int32_t>(oy0 * s) - 1;                       // hidden row of padded row 0
  const uint32_t prows = (oy1 - oy0 - 1) * s + 3;                                // padded rows the strip's taps touch
  const uint32_t hy0c = hy_lo < 0 ? 0u : static_cast<uint32_t>(hy_lo);
  const uint32_t hy1c = min(static_cast<uint32_t>(hy_lo + static_cast<int32_t>(prows) - 1), p.H - 1);
  const uint32_t nE = (hy1c - hy0c + 1) * p.W;                                   // hidden pixels the strip needs
  const uint32_t nD = (oy1 - oy0) * p.OW;                                        // output pixels of the strip
  const uint32_t rtE = (nE + 31u) >> 5, rtD = (nD + 31u) >> 5;
  const uint32_t PW = p.W + 2;
  const uint32_t prow0 = hy0c - static_cast<uint32_t>(hy_lo);                    // padded row of hidden row hy0c (0 or 1)

  // ---- weight fragments come straight from L2 into registers, a stage or more ahead of their first use ----
  const uint32_t cbl = wave % p.cb;              // this wave's block inside a chunk (kWaves % cb == 0: the same in every item)
  const uint32_t rt_first = wave / p.cb, rt_step = kWaves / p.cb;
  uint8_t* we_lds = lds + p.we_off;         // WLDS: [cb][KB1] expand fragments of the chunk
  uint8_t* wp_lds = lds + p.wp_off;         // WLDS: [nb3][cb] project fragments of the chunk
  auto load_w1 = [&](uint32_t block, v4i (&w)[KB1]) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < KB1; kb++) {
      if constexpr (WLDS) w[kb] = *reinterpret_cast<const v4i*>(we_lds + ((block % p.cb) * KB1 + kb) * 1024u + lane * 16u);
      else w[kb] = *reinterpret_cast<const v4i*>(p.w1 + ((static_cast<uint64_t>(block) * KB1 + kb) * 64 + lane) * 16);
    }
  };
  // WLDS: the pieces of a chunk, dealt round-robin over the waves (the fragments of a chunk's blocks are contiguous in
  // the expand image; in the project image, per output block, the chunk's K blocks are)
  auto dma_expand = [&](uint32_t chunk) __attribute__((always_inline)) {
    const uint32_t first = chunk * p.cb, n = min(p.cb, p.nb1 - first) * KB1;
    for (uint32_t i = wave; i < n; i += kWaves) {
      dma_fragment(reinterpret_cast<const uint8_t*>(p.w1) + (static_cast<uint64_t>(first) * KB1 + i) * 1024u, lane * 16u, we_lds + i * 1024u);
    }
  };
  auto dma_project = [&](uint32_t chunk) __attribute__((always_inline)) {
    const uint32_t first = chunk * p.cb, cbn = min(p.cb, p.nb1 - first), n = p.nb3 * cbn;
    for (uint32_t i = wave; i < n; i += kWaves) {
      const uint32_t nbo = i / cbn, kbl = i - nbo * cbn;
      dma_fragment(reinterpret_cast<const uint8_t*>(p.w3) + (static_cast<uint64_t>(nbo) * p.nb1 + first + kbl) * 1024u, lane * 16u,
                   wp_lds + (nbo * p.cb + kbl) * 1024u);
    }
  };
  if constexpr (WLDS) {
    if constexpr (HAS_EXPAND) dma_expand(0);
    dma_project(0);
  }
  v4i w1f[KB1];
  if constexpr (HAS_EXPAND && kPrefetch && !WLDS) load_w1(min(cbl, p.nb1 - 1u), w1f);     // chunk 0 (clamped: an idle wave loads something valid)

  // ---- once: biases and depthwise weights -> LDS, hidden tile = padding everywhere, input strip ----
  // ONE memory round trip: every load of the three is issued before the first store (a load-store loop is a round trip
  // per iteration -- fourteen of them in front of the first MFMA of b0 in the first version), 16 bytes per lane where
  // the tensors allow; the hidden tile's fill (LDS stores only) runs under the loads.
  const uint8_t* strip_in = p.input + (static_cast<uint64_t>(img) * p.H + hy0c) * p.W * p.in_stride;
  {
    // {b1 | b2 | b3 | w2} as one space of 16-byte pieces (all four are multiples of 16 bytes, 16-byte aligned on both sides)
    const uint32_t hp4 = p.hidden_pad, op4 = p.output_pad >> 2 << 2;           // (dwords; /4 below)
    const uint32_t e1 = (HAS_EXPAND ? hp4 : 0u) >> 2, e2 = e1 + (hp4 >> 2), e3 = e2 + (op4 >> 2), e4 = e3 + ((9u * p.hidden_pad) >> 4);
    auto param_src = [&](uint32_t i) __attribute__((always_inline)) -> const uint4* {
      return i < e1 ? reinterpret_cast<const uint4*>(p.b1) + i : (i < e2 ? reinterpret_cast<const uint4*>(p.b2) + (i - e1) :
          (i < e3 ? reinterpret_cast<const uint4*>(p.b3) + (i - e2) : reinterpret_cast<const uint4*>(p.w2) + (i - e3)));
    };
    auto param_dst = [&](uint32_t i) __attribute__((always_inline)) -> uint4* {
      return reinterpret_cast<uint4*>(i < e1 ? lds + p.b1_off + i * 16 : (i < e2 ? lds + p.b2_off + (i - e1) * 16 :
          (i < e3 ? lds + p.b3_off + (i - e2) * 16 : lds + p.w2_off + (i - e3) * 16)));
    };
    constexpr int kPU = 1536 / kThreads + (1536 % kThreads != 0);   // 1536 pieces = 24 KiB of parameters per round (plan() bounds them)
    uint4 pv[kPU];
#pragma unroll
    for (int u = 0; u < kPU; u++) {
      const uint32_t i = tid + static_cast<uint32_t>(u) * kThreads;
      pv[u] = i < e4 ? *param_src(i) : make_uint4(0, 0, 0, 0);
    }
    // the input strip: pieces of 16 / 8 / 4 bytes (p.in_piece), up to kIU per thread in flight, further rounds if needed
    constexpr int kIU = 4096 / kThreads;
    const uint32_t ppp = p.cin / p.in_piece;                                     // pieces per pixel
    const uint32_t ipieces = (HAS_EXPAND || residual) ? nE * ppp : (HAS_EXPAND ? 0u : nE * ppp);
    const uint32_t iflip = HAS_EXPAND ? p.flip1 : p.flip2;
    auto in_dst = [&](uint32_t i) __attribute__((always_inline)) -> uint8_t* {
      const uint32_t px = udiv(i, p.ppp_magic);
      const uint32_t k = i - px * ppp;
      if constexpr (HAS_EXPAND) {
        return in_lds + px * p.in_pitch + k * p.in_piece;
      } else {                                   // the hidden tensor IS the block input: straight into the padded tile
        const uint32_t rr = udiv(px, p.inv_w);
        return hid + ((prow0 + rr) * PW + (px - rr * p.W) + 1u) * p.hid_pitch + k * p.in_piece;
      }
    };
    auto in_src = [&](uint32_t i) __attribute__((always_inline)) -> const uint8_t* {
      const uint32_t px = udiv(i, p.ppp_magic);
      return strip_in + static_cast<uint64_t>(px) * p.in_stride + (i - px * ppp) * p.in_piece;
    };
    uint4 iv[kIU];
    auto in_load = [&](uint32_t base) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < kIU; u++) {
        const uint32_t i = base + static_cast<uint32_t>(u) * kThreads;
        iv[u] = make_uint4(0, 0, 0, 0);
        if (i < ipieces) {
          const uint8_t* src = in_src(i);
          if (p.in_piece == 16) iv[u] = *reinterpret_cast<const uint4*>(src);
          else if (p.in_piece == 8) { const uint2 t = *reinterpret_cast<const uint2*>(src); iv[u].x = t.x; iv[u].y = t.y; }
          else iv[u].x = *reinterpret_cast<const uint32_t*>(src);
        }
      }
    };
    auto in_store = [&](uint32_t base) __attribute__((always_inline)) {
#pragma unroll
      for (int u = 0; u < kIU; u++) {
        const uint32_t i = base + static_cast<uint32_t>(u) * kThreads;
        if (i < ipieces) {
          uint8_t* dst = in_dst(i);
          if (p.in_piece == 16) *reinterpret_cast<uint4*>(dst) = make_uint4(iv[u].x ^ iflip, iv[u].y ^ iflip, iv[u].z ^ iflip, iv[u].w ^ iflip);
          else if (p.in_piece == 8) *reinterpret_cast<uint2*>(dst) = make_uint2(iv[u].x ^ iflip, iv[u].y ^ iflip);
          else *reinterpret_cast<uint32_t*>(dst) = iv[u].x ^ iflip;
        }
      }
    };
    in_load(tid);
    QNNP_S_STAMP(8);
    // hidden tile = padding everywhere (16 bytes per store; hid_bytes is a multiple of 256)
    {
      const uint4 pad = make_uint4(p.hid_pad4, p.hid_pad4, p.hid_pad4, p.hid_pad4);
      for (uint32_t i = tid; i < p.hid_bytes >> 4; i += kThreads) reinterpret_cast<uint4*>(hid)[i] = pad;
    }
    if constexpr (HAS_EXPAND) {
      // zeros between the pixel's channels and the K block boundary (they meet zero weights): Cin % 32 != 0 only
      const uint32_t cdw = p.cin >> 2, zdw = KB1 * 8u - cdw;
      for (uint32_t i = tid; i < nE * zdw; i += kThreads) {
        const uint32_t px = i / zdw;
        *reinterpret_cast<uint32_t*>(in_lds + px * p.in_pitch + (cdw + i - px * zdw) * 4) = 0u;
      }
    }
    QNNP_S_STAMP(9);
#pragma unroll
    for (int u = 0; u < kPU; u++) {
      const uint32_t i = tid + static_cast<uint32_t>(u) * kThreads;
      if (i < e4) *param_dst(i) = pv[u];
    }
    if constexpr (!HAS_EXPAND) __syncthreads();  // (the fill, by other threads, before the pixels go into the same tile)
    in_store(tid);
    for (uint32_t base = tid + kIU * kThreads; base < ipieces; base += kIU * kThreads) {   // (rare: strips beyond 64 KiB)
      in_load(base);
      in_store(base);
    }
  }
  QNNP_S_STAMP(10);
  QNNP_S_STAMP(11);

  // ---- the project accumulators this wave owns: pair = (row tile, output block), dealt round-robin ----
  const uint32_t npairs = rtD * p.nb3;
  uint32_t prt[kMaxPairs], pnb[kMaxPairs];
  v16i pacc[kMaxPairs];
#pragma unroll
  for (int j = 0; j < kMaxPairs; j++) {
    const uint32_t pair = wave + static_cast<uint32_t>(j) * kWaves;
    const uint32_t pc = pair < npairs ? pair : 0u;
    prt[j] = pc / p.nb3;
    pnb[j] = pc - prt[j] * p.nb3;
  }
  if constexpr (WLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of chunk 0's fragments
  __syncthreads();                               // b3 resident (and the input / hidden tile complete)
  QNNP_S_STAMP(1);
#pragma unroll
  for (int j = 0; j < kMaxPairs; j++) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const int4 b = *reinterpret_cast<const int4*>(b3_lds + pnb[j] * 32 + rg * 8 + khalf * 4);
      pacc[j][rg * 4 + 0] = b.x; pacc[j][rg * 4 + 1] = b.y; pacc[j][rg * 4 + 2] = b.z; pacc[j][rg * 4 + 3] = b.w;
    }
  }
  // the fragments of pair j for the blocks [first, first + blocks): the index is clamped, not guarded (no branches; the
  // surplus registers repeat the last block and are never multiplied)
  auto load_w3 = [&](int j, uint32_t first_block, uint32_t blocks, v4i (&w)[8]) __attribute__((always_inline)) {
    const int8_t* wf = p.w3 + ((static_cast<uint64_t>(pnb[j]) * p.nb1 + first_block) * 64 + lane) * 16;
    if constexpr (WLDS) wf = reinterpret_cast<const int8_t*>(wp_lds) + (pnb[j] * p.cb * 64u + lane) * 16u;
#pragma unroll
    for (int kbl = 0; kbl < 8; kbl++) {
      w[kbl] = *reinterpret_cast<const v4i*>(wf + min(static_cast<uint32_t>(kbl), blocks - 1u) * 1024u);
    }
  };

  // ---- loop-invariant pieces of the depthwise stage ----
  // diagonal fragment: lane l = (n = l & 31, k = (l >> 5) * 16 + j) is non-zero only where k == n
  uint32_t dsel[4];
#pragma unroll
  for (int d = 0; d < 4; d++) {
    dsel[d] = (khalf == (col >> 4) && static_cast<uint32_t>(d) == ((col & 15u) >> 2)) ? (0xFFu << (8u * (col & 3u))) : 0u;
  }
  uint32_t tapoff[9];
#pragma unroll
  for (int t = 0; t < 9; t++) tapoff[t] = (static_cast<uint32_t>(t / 3) * PW + static_cast<uint32_t>(t % 3)) * p.hid_pitch;

  for (uint32_t chunk = 0; chunk < p.nchunks; chunk++) {
    const uint32_t cb0 = chunk * p.cb;
    const uint32_t cbn = min(p.cb, p.nb1 - cb0);             // blocks of this chunk
    const bool mine = cbl < cbn;
    const uint32_t cbg = cb0 + cbl;                          // this wave's hidden block

    // ---- stage E: expand (pixels x Cin) x (Cin x 32), two row tiles at a time (two independent accumulator chains),
    //      requantize, ^ flip2 -> hidden tile. A second tile beyond the strip recomputes the last row and stores nothing. ----
    if constexpr (HAS_EXPAND) {
      if (mine) {
        // the block's folded bias in accumulator layout, once per chunk: it is the FIRST MFMA's addend (a distinct C
        // operand), not sixteen moves per tile
        v16i bias;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int4 b = *reinterpret_cast<const int4*>(b1_lds + cbg * 32 + rg * 8 + khalf * 4);
          bias[rg * 4 + 0] = b.x; bias[rg * 4 + 1] = b.y; bias[rg * 4 + 2] = b.z; bias[rg * 4 + 3] = b.w;
        }
        if constexpr (!kPrefetch || WLDS) load_w1(cbg, w1f);
        for (uint32_t rt = rt_first; rt < rtE; rt += kTilesPerRound * rt_step) {
          const uint32_t m0 = rt * 32u + col, m1 = (rt + rt_step) * 32u + col;
          const uint8_t* a0p = in_lds + min(m0, nE - 1u) * p.in_pitch + khalf * 16;
          const uint8_t* a1p = in_lds + min(m1, nE - 1u) * p.in_pitch + khalf * 16;
          v16i acc0, acc1;
#pragma unroll
          for (int kb = 0; kb < KB1; kb++) {
            const v4i a0 = *reinterpret_cast<const v4i*>(a0p + kb * 32);
            const v4i a1 = *reinterpret_cast<const v4i*>(a1p + kb * 32);
#ifdef QNNP_STRIP_BIAS_MOVES               // (A/B at build time: the accumulators initialised by moves, as first written)
            if (kb == 0) { acc0 = bias; acc1 = bias; }
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1f[kb], a0, acc0, 0, 0, 0);
            if constexpr (kTwoTiles) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1f[kb], a1, acc1, 0, 0, 0);
#else
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1f[kb], a0, kb == 0 ? bias : acc0, 0, 0, 0);
            if constexpr (kTwoTiles) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(w1f[kb], a1, kb == 0 ? bias : acc1, 0, 0, 0);
#endif
          }
          uint32_t pk0[4], pk1[4];
          requant16m<MODE>(acc0, p.mode1, p.rq1, pk0);
          if constexpr (kTwoTiles) requant16m<MODE>(acc1, p.mode1, p.rq1, pk1);
#pragma unroll
          for (int rg = 0; rg < 4; rg++) { pk0[rg] ^= p.flip2; if constexpr (kTwoTiles) pk1[rg] ^= p.flip2; }
          const uint4 v0 = gather16(pk0);
          if (m0 < nE) {
            const uint32_t rr = udiv(m0, p.inv_w);
            *reinterpret_cast<uint4*>(hid + ((prow0 + rr) * PW + (m0 - rr * p.W) + 1u) * p.hid_pitch + cbl * 32 + khalf * 16) = v0;
          }
          if constexpr (kTwoTiles) {
            const uint4 v1 = gather16(pk1);
            if (m1 < nE) {
              const uint32_t rr = udiv(m1, p.inv_w);
              *reinterpret_cast<uint4*>(hid + ((prow0 + rr) * PW + (m1 - rr * p.W) + 1u) * p.hid_pitch + cbl * 32 + khalf * 16) = v1;
            }
          }
        }
      }
      // the next chunk's expand fragments (clamped like the first)
      if constexpr (kPrefetch && !WLDS) { if (chunk + 1 < p.nchunks) load_w1(min(cb0 + p.cb + cbl, p.nb1 - 1u), w1f); }
    }
    v4i w3f[8];
    if constexpr (kPrefetch && !WLDS) load_w3(0, cb0, cbn, w3f);      // the first owned pair's project fragments of this chunk
    if (chunk == 0) QNNP_S_STAMP(2);
    if (chunk == 1) QNNP_S_STAMP(17);
    __syncthreads();
    if (chunk == 0) QNNP_S_STAMP(3);
    if (chunk == 1) QNNP_S_STAMP(18);
    if constexpr (WLDS) {                          // (kernel comment: both buffers are free from this barrier on)
      if constexpr (HAS_EXPAND) { if (chunk + 1 < p.nchunks) dma_expand(chunk + 1); }
      if (chunk > 0) dma_project(chunk);
    }

    // ---- stage D: depthwise as nine MFMAs per (32 pixels, 32 channels) against diagonal weight fragments ----
    if (mine) {
      v4i dfrag[9];
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const uint32_t wb = static_cast<uint32_t>(static_cast<uint8_t>(w2_lds[t * p.hidden_pad + cbg * 32 + col])) * 0x01010101u;
        dfrag[t].x = static_cast<int>(wb & dsel[0]);
        dfrag[t].y = static_cast<int>(wb & dsel[1]);
        dfrag[t].z = static_cast<int>(wb & dsel[2]);
        dfrag[t].w = static_cast<int>(wb & dsel[3]);
      }
      for (uint32_t rt = rt_first; rt < rtD; rt += kTilesPerRound * rt_step) {
        const uint32_t m0 = rt * 32u + col, m1 = (rt + rt_step) * 32u + col;
        const uint32_t mc0 = min(m0, nD - 1u), mc1 = min(m1, nD - 1u);
        const uint32_t oy0l = udiv(mc0, p.inv_ow), oy1l = udiv(mc1, p.inv_ow);
        const uint8_t* base0 = hid + ((oy0l * s) * PW + (mc0 - oy0l * p.OW) * s) * p.hid_pitch + cbl * 32 + khalf * 16;
        const uint8_t* base1 = hid + ((oy1l * s) * PW + (mc1 - oy1l * p.OW) * s) * p.hid_pitch + cbl * 32 + khalf * 16;
        // (the bias is re-read per tile -- four ds_read_b128 straight into the accumulator registers: held across the
        //  tiles as stage E holds its own it costs sixteen registers, and the kernel then spills 1-9 at its 128)
        v16i acc0, acc1;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int4 b = *reinterpret_cast<const int4*>(b2_lds + cbg * 32 + rg * 8 + khalf * 4);
          acc0[rg * 4 + 0] = b.x; acc0[rg * 4 + 1] = b.y; acc0[rg * 4 + 2] = b.z; acc0[rg * 4 + 3] = b.w;
        }
        acc1 = acc0;
#pragma unroll
        for (int t = 0; t < 9; t++) {
          const v4i a0 = *reinterpret_cast<const v4i*>(base0 + tapoff[t]);
          const v4i a1 = *reinterpret_cast<const v4i*>(base1 + tapoff[t]);
          acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(dfrag[t], a0, acc0, 0, 0, 0);
          if constexpr (kTwoTiles) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(dfrag[t], a1, acc1, 0, 0, 0);
        }
        uint32_t pk0[4], pk1[4];
        requant16m<MODE>(acc0, p.mode2, p.rq2, pk0);
        if constexpr (kTwoTiles) requant16m<MODE>(acc1, p.mode2, p.rq2, pk1);
#pragma unroll
        for (int rg = 0; rg < 4; rg++) { pk0[rg] ^= p.flip3; if constexpr (kTwoTiles) pk1[rg] ^= p.flip3; }
        const uint4 v0 = gather16(pk0);
        if (m0 < nD) *reinterpret_cast<uint4*>(dwb + m0 * p.dw_pitch + cbl * 32 + khalf * 16) = v0;
        if constexpr (kTwoTiles) {
          const uint4 v1 = gather16(pk1);
          if (m1 < nD) *reinterpret_cast<uint4*>(dwb + m1 * p.dw_pitch + cbl * 32 + khalf * 16) = v1;
        }
      }
    }
    if (chunk == 0) QNNP_S_STAMP(4);
    if (chunk == 1) QNNP_S_STAMP(19);
    if constexpr (WLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces; the barrier publishes everybody's
    __syncthreads();
    if (chunk == 0) QNNP_S_STAMP(5);
    if (chunk == 1) QNNP_S_STAMP(20);

    // ---- stage P: project, accumulated over the chunks ----
#pragma unroll
    for (int j = 0; j < kMaxPairs; j++) {
      if (wave + static_cast<uint32_t>(j) * kWaves < npairs) {
        if (j > 0 || !kPrefetch || WLDS) load_w3(j, cb0, cbn, w3f);  // (8 waves: the first pair's fragments were fetched in front of stage D)
        const uint8_t* arow = dwb + (prt[j] * 32u + col) * p.dw_pitch + khalf * 16;
        switch (cbn) {                                       // wave-uniform: straight-line MFMAs per block count
          case 8: project_blocks<8>(pacc[j], w3f, arow); break;
          case 7: project_blocks<7>(pacc[j], w3f, arow); break;
          case 6: project_blocks<6>(pacc[j], w3f, arow); break;
          case 5: project_blocks<5>(pacc[j], w3f, arow); break;
          case 4: project_blocks<4>(pacc[j], w3f, arow); break;
          case 3: project_blocks<3>(pacc[j], w3f, arow); break;
          case 2: project_blocks<2>(pacc[j], w3f, arow); break;
          default: project_blocks<1>(pacc[j], w3f, arow); break;
        }
      }
    }
    // (no barrier here: the next chunk's stage E touches the hidden tile only, and its barrier orders stage D behind
    //  every wave's stage P)
    if (chunk == 0) { QNNP_S_STAMP(6); QNNP_S_STAMP(16); }
    if (chunk == 1) QNNP_S_STAMP(21);
  }
  QNNP_S_STAMP(7);

  // ---- epilogue: requantize [+ residual] -> global ----
  IgemmParams sp{};                         // what igemm_store_pk4 reads
  sp.n = p.cout;
  sp.store_mode = p.store_mode;
#pragma unroll
  for (int j = 0; j < kMaxPairs; j++) {
    if (wave + static_cast<uint32_t>(j) * kWaves < npairs) {
      uint32_t pk[4];
      requant16m<MODE>(pacc[j], p.mode3, p.rq3, pk);
      const uint32_t m = prt[j] * 32u + col;
      const uint32_t mc = m < nD ? m : nD - 1u;
      if (residual) {
        // stride 1: output pixel (oy0 + oyl, ox) is hidden / input pixel of the same coordinates
        const uint32_t oyl = udiv(mc, p.inv_ow);
        const uint32_t ox = mc - oyl * p.OW;
        const uint8_t* res = in_lds + ((oy0 + oyl - hy0c) * p.W + ox) * p.in_pitch;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const uint32_t c = pnb[j] * 32u + rg * 8u + khalf * 4u;
          if (c < p.cout) pk[rg] = add_quantize4(*reinterpret_cast<const uint32_t*>(res + c) ^ p.flip1, pk[rg], p.add);
        }
      }
      uint8_t* out_row = p.output + ((static_cast<uint64_t>(img) * p.OH + oy0) * p.OW + mc) * p.out_stride;
      igemm_store_pk4(pk, out_row, pnb[j] * 32u, khalf, m < nD, sp);
    }
  }
  QNNP_S_STAMP(22);
}

inline uint32_t magic32(uint32_t d) { return d <= 1 ? 0u : static_cast<uint32_t>((1ull << 32) / d) + 1u; }

/* strip height, chunk width and LDS layout; false when the block does not fit the kernel */
bool plan(const qnnp_hip_fused_strip_args& a, StripParams* p, uint32_t* lds_bytes)
{
  if (a.stride != 1 && a.stride != 2) return false;
  if (a.input_channels % 4 != 0 || a.hidden_channels % 4 != 0 || a.output_channels % 4 != 0) return false;
  if (a.input_channels > 32u * kMaxKb1) return false;
  if (!a.has_expand && (a.hidden_channels != a.input_channels || a.has_residual)) return false;
  if (a.has_residual && (a.stride != 1 || a.input_channels != a.output_channels)) return false;
  if (a.input_stride % 4 != 0 || reinterpret_cast<uintptr_t>(a.input) % 4 != 0) return false;
  if (a.input_width == 0 || a.input_height == 0 || a.output_width == 0 || a.output_height == 0) return false;
  if (a.input_width > 4096 || a.input_height > 65535) return false;
  const uint32_t s = a.stride;
  p->kb1 = (a.input_channels + 31u) / 32u;
  p->nb1 = (a.hidden_channels + 31u) / 32u;
  p->nb3 = (a.output_channels + 31u) / 32u;
  p->hidden_pad = p->nb1 * 32u;
  p->output_pad = p->nb3 * 32u;
  p->in_pitch = p->kb1 * 32u + 16u;
  const uint32_t fixed = ((9u * p->hidden_pad + 255u) & ~255u) + (2u * p->hidden_pad + p->output_pad) * 4u;
  if (9u * p->hidden_pad + (2u * p->hidden_pad + p->output_pad) * 4u > 1536u * 16u) return false;   // one staging round
  int cus = qnnp_hip_compute_units();
  if (cus <= 0) cus = 256;
  // candidates: 1, 2, 3, ... strips of EQUAL height per image (the last one may be shorter by less than a strip count).
  // Among those that fit LDS: least (rounds of workgroups over the CUs) x (stage rounds per workgroup + 4 for its
  // staging) -- one workgroup per CU at a time. (A first version walked the height down from the whole image and
  // stopped at 13 + 1 rows for 14 x 14.)
  bool found = false;
  uint64_t best_total = ~0ull;
  const uint32_t forced = a.rows_per_strip != 0 ? (a.rows_per_strip < a.output_height ? a.rows_per_strip : a.output_height) : 0u;
  uint32_t last_rps = 0;
  for (uint32_t want = 1; want <= a.output_height; want++) {
    const uint32_t rps = forced != 0 ? forced : (a.output_height + want - 1) / want;
    if (rps == last_rps) continue;
    last_rps = rps;
    const uint32_t strips = (a.output_height + rps - 1) / rps;
    const uint32_t prows = (rps - 1) * s + 3;
    const uint32_t erows = prows < a.input_height ? prows : a.input_height;
    const uint32_t nE = erows * a.input_width, nD = rps * a.output_width;
    const uint32_t rtE = (nE + 31u) / 32u, rtD = (nD + 31u) / 32u;
    if (rtD * p->nb3 <= static_cast<uint32_t>(kMaxPairs * kWaves) && nE <= 65535u) {
      // chunk width: the one whose stages cost least -- chunks x (two-tile rounds of stage E + of stage D), each wave
      // owning one of the chunk's blocks and every (8 / cb)-th row tile -- among those that fit; ties go to wider chunks
      uint32_t best_cb = 0, best_cost = 0xFFFFFFFFu, best_need = 0, best_hid = 0, best_in = 0, best_wlds = 0;
      for (uint32_t cb = 8; cb >= 1; cb >>= 1) {
        if (!a.has_expand && cb < p->nb1) continue;          // the copied input must be one chunk
        const uint32_t hid_pitch = cb * 32u + 16u;
        const uint32_t in_bytes = a.has_expand ? ((rtE * 32u * p->in_pitch + 255u) & ~255u) : 0u;
        const uint32_t hid_bytes = (prows * (a.input_width + 2u) * hid_pitch + 255u) & ~255u;
        const uint32_t dw_bytes = (rtD * 32u * hid_pitch + 255u) & ~255u;
        uint32_t need = fixed + in_bytes + hid_bytes + dw_bytes;
        if (need > kLdsLimit) continue;
        // the chunk's expand and project fragments beside the tiles, if they fit (kernel comment, WLDS)
        const uint32_t w_bytes = ((a.has_expand ? cb * p->kb1 : 0u) + cb * p->nb3) * 1024u;
        // (measured level to slightly behind fetching from L2 -- profiles/r04 -- hence only on request, and only in
        //  measurement builds: the product library does not carry the flavour)
#ifdef QNNP_ENABLE_ABLATION
        const bool wlds = a.weights_in_lds == 1u && need + w_bytes <= kLdsLimit;
        if (a.weights_in_lds == 1u && !wlds) continue;        // (forced, for tests)
#else
        const bool wlds = false;
        (void) w_bytes;
#endif
        if (wlds) need += w_bytes;
        const uint32_t nchunks = (p->nb1 + cb - 1) / cb, per_round = static_cast<uint32_t>(kTilesPerRound) * (static_cast<uint32_t>(kWaves) / cb);
        // (+ 3 per chunk: its two barriers, the diagonal fragments, the project stage -- measured, profiles/r04; + 2 when
        //  stages E and P fetch their fragments from L2 in front of their first MFMA)
        // A stage's tiles are dealt one block per wave, every (waves / cb)-th row tile: rounds x (busy waves per SIMD) / 4 --
        // the four waves of a SIMD share its issue slots, so a stage that leaves half of them idle takes half as long
        // (7 x 7 images: one strip = 16 + 16 tiles on half the chip's CUs, two strips = 16 + 8 on all of them)
        auto stage4 = [&](uint32_t rt) {
          const uint32_t busy = rt * cb < static_cast<uint32_t>(kWaves) ? rt * cb : static_cast<uint32_t>(kWaves);
          return ((rt + per_round - 1) / per_round) * ((busy + 3u) / 4u) * 4u / (static_cast<uint32_t>(kWaves) / 4u);
        };
        const uint32_t cost = nchunks * ((a.has_expand ? stage4(rtE) : 0u) + stage4(rtD) + 4u * 3u);
        if (cost < best_cost) { best_cost = cost; best_cb = cb; best_need = need; best_hid = hid_bytes; best_in = in_bytes; best_wlds = wlds ? 1u : 0u; }
      }
      if (best_cb != 0) {
        const uint32_t cb = best_cb;
        const uint64_t wgs = static_cast<uint64_t>(a.batch) * strips;
        const uint64_t total = ((wgs + cus - 1) / cus) * (best_cost + 4u * 4u);
        if (!found || total < best_total) {
          found = true;
          best_total = total;
          p->rps = rps; p->strips = strips; p->cb = cb; p->nchunks = (p->nb1 + cb - 1) / cb;
          p->hid_pitch = cb * 32u + 16u; p->dw_pitch = p->hid_pitch; p->hid_bytes = best_hid;
          p->w2_off = 0;
          p->b1_off = (9u * p->hidden_pad + 255u) & ~255u;
          p->b2_off = p->b1_off + p->hidden_pad * 4u;
          p->b3_off = p->b2_off + p->hidden_pad * 4u;
          p->in_off = fixed; p->hid_off = fixed + best_in; p->dw_off = fixed + best_in + best_hid;
          {
            const uint32_t dw_bytes = (rtD * 32u * p->hid_pitch + 255u) & ~255u;
            p->wlds = best_wlds;
            p->we_off = p->dw_off + dw_bytes;
            p->wp_off = p->we_off + (a.has_expand ? cb * p->kb1 : 0u) * 1024u;
          }
          *lds_bytes = best_need;
        }
      }
    }
    if (forced != 0) break;
  }
  return found;
}

typedef void (*strip_kernel_t)(const StripParams);

// The kernel of a plan: one per (expand K blocks, common rounding mode); -1 = per-stage switch inside. Fills the
// requantization parameters and stage modes of `p` on the way (they select the instantiation).
strip_kernel_t pick_kernel(const qnnp_hip_fused_strip_args& a, StripParams* pp)
{
  StripParams& p = *pp;
  bool unused = false;
  if (a.has_expand) { p.rq1 = make_requant_dev(a.expand_rq); p.mode1 = stage_mode(p.rq1, &unused); }
  p.rq2 = make_requant_dev(a.dw_rq); p.mode2 = stage_mode(p.rq2, &unused);
  p.rq3 = make_requant_dev(a.project_rq); p.mode3 = stage_mode(p.rq3, &unused);
  const int mode = (p.mode2 == p.mode3 && (!a.has_expand || p.mode1 == p.mode2) && (p.mode2 == 0 || p.mode2 == 3)) ? static_cast<int>(p.mode2) : -1;
  strip_kernel_t kernel = nullptr;
#ifdef QNNP_ENABLE_ABLATION
#define QNNP_STRIP_PICK(KB)                                                                           \
  kernel = p.wlds != 0                                                                                  \
      ? (mode == 0 ? &q8_fused_strip_kernel<KB, 0, true, true> : (mode == 3 ? &q8_fused_strip_kernel<KB, 3, true, true> : &q8_fused_strip_kernel<KB, -1, true, true>)) \
      : (mode == 0 ? &q8_fused_strip_kernel<KB, 0, true> : (mode == 3 ? &q8_fused_strip_kernel<KB, 3, true> : &q8_fused_strip_kernel<KB, -1, true>))
#else
#define QNNP_STRIP_PICK(KB)                                                                           \
  kernel = mode == 0 ? &q8_fused_strip_kernel<KB, 0, true> : (mode == 3 ? &q8_fused_strip_kernel<KB, 3, true> : &q8_fused_strip_kernel<KB, -1, true>)
#endif
  if (!a.has_expand) {
    kernel = mode == 0 ? &q8_fused_strip_kernel<1, 0, false> : (mode == 3 ? &q8_fused_strip_kernel<1, 3, false> : &q8_fused_strip_kernel<1, -1, false>);
#ifdef QNNP_ENABLE_ABLATION
    if (p.wlds != 0) kernel = mode == 0 ? &q8_fused_strip_kernel<1, 0, false, true> : (mode == 3 ? &q8_fused_strip_kernel<1, 3, false, true> : &q8_fused_strip_kernel<1, -1, false, true>);
#endif
  } else {
    switch (p.kb1) {
      case 1: QNNP_STRIP_PICK(1); break;
      case 2: QNNP_STRIP_PICK(2); break;
      case 3: QNNP_STRIP_PICK(3); break;
      case 4: QNNP_STRIP_PICK(4); break;
      default: QNNP_STRIP_PICK(5); break;
    }
  }
#undef QNNP_STRIP_PICK
  return kernel;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-device property of ONE kernel: made once per (kernel, device),
// and the outcome is remembered -- a device or runtime that refuses the opt-in makes the plan unsupported at SETUP
// (qnnp_hip_fused_strip_supported), where the block can still take the tile kernel or stay with its stand-alone
// operators, instead of failing every run with a launch error.
bool lds_optin(strip_kernel_t kernel)
{
  struct Entry { strip_kernel_t kernel; uint32_t ok, refused; };
  static std::mutex mu;
  static Entry table[64];
  static int used = 0;
  const int device = qnnp_hip_device();
  const uint32_t bit = 1u << (static_cast<uint32_t>(device < 0 ? 0 : device) & 31u);
  std::lock_guard<std::mutex> lock(mu);
  Entry* e = nullptr;
  for (int i = 0; i < used; i++) if (table[i].kernel == kernel) { e = &table[i]; break; }
  if (e == nullptr && used < 64) { e = &table[used++]; e->kernel = kernel; e->ok = e->refused = 0; }
  if (e != nullptr) {
    if ((e->ok & bit) != 0) return true;
    if ((e->refused & bit) != 0) return false;
  }
  const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsLimit) == hipSuccess;
  if (!ok) (void) hipGetLastError();
  if (e != nullptr) (ok ? e->ok : e->refused) |= bit;
  return ok;
}

}  // namespace

}  // namespace qnnp

extern "C" int qnnp_hip_fused_strip_bias_offset(const struct qnnp_hip_requant* rq)
{
  bool offset = false;
  (void) qnnp::stage_mode(qnnp::make_requant_dev(*rq), &offset);
  return offset ? 1 : 0;
}

extern "C" int qnnp_hip_fused_strip_supported(const struct qnnp_hip_fused_strip_args* a)
{
  qnnp::StripParams p{};
  uint32_t lds_bytes = 0;
  if (a == nullptr || !qnnp::plan(*a, &p, &lds_bytes)) return 0;
  // a plan above 64 KiB of LDS is only as good as the device's answer to the opt-in of ITS kernel
  return lds_bytes <= 64u * 1024u || qnnp::lds_optin(qnnp::pick_kernel(*a, &p)) ? 1 : 0;
}

extern "C" int qnnp_hip_fused_strip_run(const struct qnnp_hip_fused_strip_args* a, const char** kernel_name)
{
  using namespace qnnp;
  if (a == nullptr || a->input == nullptr || a->output == nullptr) return QNNP_HIP_EINVAL;
  if (a->batch == 0) return QNNP_HIP_OK;
  StripParams p{};
  uint32_t lds_bytes = 0;
  if (!plan(*a, &p, &lds_bytes)) return QNNP_HIP_EINVAL;
#ifdef QNNP_ENABLE_ABLATION
  if (getenv("QNNP_GFX950_PRINT_PLAN") != nullptr) {
    fprintf(stderr, "fused strip plan: %ux%ux%u -> %u -> %u stride %u: rows/strip %u strips %u chunk blocks %u chunks %u lds %u weights in LDS %u\n",
            a->input_height, a->input_width, a->input_channels, a->hidden_channels, a->output_channels, a->stride,
            p.rps, p.strips, p.cb, p.nchunks, lds_bytes, p.wlds);
  }
#endif
  p.input = a->input; p.output = a->output;
  p.batch = a->batch;
  p.H = a->input_height; p.W = a->input_width; p.OH = a->output_height; p.OW = a->output_width;
  p.cin = a->input_channels; p.ch = a->hidden_channels; p.cout = a->output_channels;
  p.in_stride = a->input_stride; p.out_stride = a->output_stride; p.stride = a->stride;
  p.has_expand = a->has_expand; p.has_res = a->has_residual;
  p.inv_w = magic32(p.W); p.inv_ow = magic32(p.OW);
  p.flip1 = (a->expand_flip & 0xFFu) * 0x01010101u;
  p.flip2 = (a->dw_flip & 0xFFu) * 0x01010101u;
  p.flip3 = (a->project_flip & 0xFFu) * 0x01010101u;
  p.hid_pad4 = (a->dw_pad & 0xFFu) * 0x01010101u;
  {
    const uintptr_t in_addr = reinterpret_cast<uintptr_t>(a->input);
    p.in_piece = 4;
    if (p.cin % 16 == 0 && p.in_stride % 16 == 0 && in_addr % 16 == 0) p.in_piece = 16;
    else if (p.cin % 8 == 0 && p.in_stride % 8 == 0 && in_addr % 8 == 0) p.in_piece = 8;
    p.ppp_magic = magic32(p.cin / p.in_piece);
  }
  const uintptr_t out_addr = reinterpret_cast<uintptr_t>(a->output);
  p.store_mode = 0;
  if (a->output_channels % 16 == 0 && a->output_stride % 16 == 0 && out_addr % 16 == 0) p.store_mode = 2;
  else if (a->output_channels % 4 == 0 && a->output_stride % 4 == 0 && out_addr % 4 == 0) p.store_mode = 1;
  p.w1 = a->expand_w; p.b1 = a->expand_bias;
  p.w2 = a->dw_w; p.b2 = a->dw_bias;
  p.w3 = a->project_w; p.b3 = a->project_bias;
  p.add = a->add;
  p.trace = nullptr;
#ifdef QNNP_ENABLE_ABLATION
  p.trace = static_cast<unsigned long long*>(qnnp_hip_trace_buffer());
#endif
  if (p.hidden_pad != a->hidden_pad || p.output_pad != a->output_pad) return QNNP_HIP_EINVAL;
  const strip_kernel_t kernel = pick_kernel(*a, &p);
  // (dynamic LDS above 64 KiB needs the opt-in, per device and per kernel: made once, its outcome remembered -- the
  //  `supported` probe setup calls has already made it, so a refusal was answered there and setup chose another kernel)
  if (lds_bytes > 64u * 1024u && !lds_optin(kernel)) return QNNP_HIP_EINVAL;
  const uint64_t blocks = static_cast<uint64_t>(a->batch) * p.strips;
  if (blocks > 0x7FFFFFFFull) return QNNP_HIP_EINVAL;
  hipStream_t stream = reinterpret_cast<hipStream_t>(qnnp_hip_get_stream());
  hipLaunchKernelGGL(kernel, dim3(static_cast<uint32_t>(blocks)), dim3(kThreads), lds_bytes, stream, p);
  if (kernel_name != nullptr) *kernel_name = "q8_fused_strip";
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}
