/*
 * q8convpatch.hip -- dense 3x3 convolution with MANY channels (ResNet's 128 / 256 / 512-channel layers, stride 1 or 2) on
 * the matrix cores: the input patch of a workgroup's output positions stays in LDS for the whole reduction, the weights
 * stream through an LDS-DMA ring.
 *
 * Same operator and arithmetic as the other implicit-GEMM kernels (replaces q8conv_ukernel_4x4c2__sse2,
 * src/q8conv/4x4c2-sse2.c:14-273, + compute_q8conv, src/operator-run.c:183-217, 837-842, + the indirection buffer,
 * src/indirection.c:18-79) for the dense rows of bench/convolution.cc:642-718 (ResNet-18 / ResNet-50: 3x3 with
 * 128 -> 128 at 28x28, 256 -> 256 at 14x14, 512 -> 512 at 7x7, and their stride-2 entries).
 *
 * Why another kernel (round 5, profiles/r05/conv_lists_*): these layers are 29.6 GOP each -- 5.9 us of matrix pipe --
 * and ran 43-78 us. The offset-table flavour of the 256 x 256 GEMM kernel gathers its activation tile tap by tap
 * through the table (a dependent load and 64-bit address arithmetic per 16-byte piece, every input byte fetched nine
 * times from L2) and its 256-row tiles leave most CUs idle at 14x14 / 7x7; the weight-stationary kernel of
 * q8convwave.hip keeps 9 * C * N weight bytes in registers, which ends at 64 channels. Here:
 *   - a workgroup (8 waves) owns P <= 128 (N-tile 256) or <= 256 (N-tile 128) output positions -- whole output rows of
 *     one image, or whole small images -- and an N-tile of output channels. Their input patch, halo included
 *     (out-of-image pixels = the input zero point), goes to LDS ONCE by LDS-DMA with a PADDED pixel pitch: patch pixel q
 *     starts at q * (C + 16) bytes -- an odd number of 16-byte chunks, so the lanes of a ds_read_b128 group, which read the
 *     same chunk of consecutive pixels, walk the bank quads with an odd stride (no XOR swizzle: the first build had one and
 *     it is gone; what conflicts remain -- SQ_LDS_BANK_CONFLICT 28 % of the LDS-active cycles at 14 x 14 x 256, profiles/r05/
 *     conv_patch_stamps_and_pmc_r05ptrace3.txt -- come from lane groups that straddle an output row's end);
 *   - one pass over the landed patch re-centres it in place (a ^ 0x80) and leaves per-pixel channel sums beside it
 *     (v_sad_u8; the kernel-zero-point row term is their sum over the window, taken in the epilogue: 9 LDS reads per
 *     position instead of row-sum work inside the K loop);
 *   - K = 9 taps x C channels advances 64 bytes per step; a step's weight fragments (N-tile x 64 bytes = 8 / 16 KiB,
 *     copied verbatim from the packed image) arrive through a four-slot LDS ring, two steps ahead, counted vmcnt + one
 *     raw s_barrier per step (placed between the step's two 32-deep halves, so the fragment reads of the next half are
 *     always in flight under the MFMAs of the current one);
 *   - each wave multiplies 2 position blocks x 2 channel blocks (four 32 x 32 accumulators): per 32-deep half two
 *     activation fragments straight from the patch (address = pixel of (position, tap) + swizzled chunk) and two weight
 *     fragments from the ring, four MFMAs;
 *   - epilogue as the weight-stationary kernel: row term, Q31 requantization in registers (lane / offset forms),
 *     v_permlane32_swap, one 16-byte store per lane and channel block.
 * L2 -> LDS traffic per MAC is the weights' only: 16 (8) KiB per 128 (256) positions x 256 (128) channels x 64 deep =
 * 32 B/clk/CU at the full matrix rate, against 48-64 for a tap-by-tap gather of the same tile.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "igemm_params.h"
#include "per_device.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr uint32_t pt_stages(int kstep) { return kstep == 128 ? 2u : 4u; }   // ring slots by bytes of K per step
constexpr uint32_t kPtFlip = 0x80808080u;
constexpr uint32_t kPtLdsLimit = 160 * 1024;

struct PatchArgs {
  uint32_t batch;
  uint32_t imgs, rows;        // a tile: `imgs` images x `rows` output rows x OW positions
  uint32_t pos;               // imgs * rows * OW (<= 32 * position blocks of the flavour)
  uint32_t pr, pw, pimg;      // patch rows / columns / pixels per image
  uint32_t ppix;              // imgs * pimg
  uint32_t tiles_r;           // row tiles per image (1 when imgs > 1)
  uint32_t tiles_m, tiles_n;
  uint32_t inv_ow, inv_rw, inv_pw, inv_pimg, inv_tiles_r, inv_tiles_n;   // ceil(2^32 / d) (0: d == 1), exact for the operands used
  uint32_t ps;                // bytes between patch pixels in LDS: C + 16 (an odd number of 16-byte chunks: consecutive pixels'
                              // copies of one chunk fall into different bank groups, and every fragment address is pixel + immediate)
  uint32_t chunks;            // 16-byte chunks of the patch = ppix * (C / 16 + 1)
  uint32_t inv_cpp1;          // ceil(2^32 / (C / 16 + 1))
  uint32_t pix_off, bias_off, ring_off;   // LDS byte offsets (the patch is at 0)
  uint32_t ksteps, csteps;    // K steps of the flavour (64 or 128 bytes): 9 * C / KSTEP; steps per tap
  uint32_t nch;               // CHUNK flavour: the patch holds 128 of the C channels at a time, C / 128 chunks (1: the whole patch)
  uint32_t abl;               // measurement builds only (QNNP_PATCH_ABL): 1 = no ring requests in the loop, 2 = no per-step barrier,
                              // 4 = no fragment reads in the loop, 8 = no MFMA, 16 = both waves of a SIMD request in the same half,
                              // 32 = no patch pass (no re-centring, no sums), 64 = no epilogue stores
};

__device__ __forceinline__ uint32_t pt_div(uint32_t n, uint32_t inv) { return inv != 0u ? __umulhi(n, inv) : n; }
inline uint32_t pt_magic(uint32_t d) { return d > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + d - 1) / d) : 0u; }

__device__ __forceinline__ uint32_t pt_lds_off(const void* p)
{
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*) p));
}

/* LDS-DMA, 16 bytes per lane, flat per-lane source address; inline asm so that hipcc does not guard later LDS accesses
 * with vmcnt(0) (q8convwave.hip dma16). M0 is written and left: nothing else in this kernel reads it
 * (tests/test_kernel_resources.py). */
__device__ __forceinline__ void pt_dma16(const uint8_t* src, uint32_t lds_dst_uniform)
{
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst_uniform);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
}
/* the saddr form: wave-uniform 64-bit base + 32-bit lane offset (the weight stream: no per-piece vector arithmetic) */
__device__ __forceinline__ uint64_t pt_scalar64(uint64_t v)          // a wave-uniform value, in scalar registers for good
{
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return (static_cast<uint64_t>(hi) << 32) | lo;
}
__device__ __forceinline__ void pt_dma16_saddr(uint64_t base_scalar, uint32_t lane_offset, uint32_t lds_dst_scalar)
{
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_offset), "s"(base_scalar), "s"(lds_dst_scalar) : "memory");
}
template <int N>
__device__ __forceinline__ void pt_wait_vmcnt()
{
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
/* Fragment reads the compiler does not track, with counted waits that are tied to the registers they guard (the "+v"
 * operands make the MFMAs that consume them depend on the wait). Why by hand: with the reads as plain loads hipcc put
 * s_waitcnt lgkmcnt(1) / (0) in front of a half's MFMAs AFTER the next half's reads had been issued -- every step waited for
 * the LDS round trip of fragments it would only need 128 cycles later. A wave's LDS operations return in order. */
template <int OFF>
__device__ __forceinline__ v4i pt_ds_read16(uint32_t addr)
{
  v4i x;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(OFF) : "memory");
  return x;
}
template <int N>
__device__ __forceinline__ void pt_wait_lgkm(v4i& x0, v4i& x1, v4i& x2, v4i& x3)
{
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "n"(N) : "memory");
}
__device__ __forceinline__ void pt_ds_write16(uint32_t off, v4i x)
{
  asm volatile("ds_write_b128 %0, %1" :: "v"(off), "v"(x) : "memory");
}
__device__ __forceinline__ void pt_ds_write4(uint32_t off, int32_t v)
{
  asm volatile("ds_write_b32 %0, %1" :: "v"(off), "v"(v) : "memory");
}

/* NTB = 32-channel blocks of the N-tile: 8 (waves as 2 position pairs x 4 channel pairs, <= 128 positions) or
 *       4 (4 x 2, <= 256 positions). */
/* KSTEP = bytes of K per ring step and barrier: 128 (two ring slots, one step ahead: 16 MFMAs per wave between barriers) or
 *         64 (four slots, three steps ahead; the only choice for 64 input channels). */
/* MB = 32-position blocks of the tile (4 or 8). The workgroup has MB * NTB / 4 waves, each 2 x 2 blocks: 8 waves for
 *      (4, 8) and (8, 4); FOUR waves for (4, 4) -- half the ring, so that two workgroups share a CU where one 8-wave
 *      workgroup with a 64 KiB ring would be alone: one's prologue, barriers and epilogue run under the other's multiplies. */
/* CHUNK (KSTEP 128 only; round 5): the patch of all C channels does not fit beside the ring for a full tile of positions
 *       (stride 2 with 256 / 512 channels: four patch pixels per position) -- the non-chunked plan then shrinks the tile to
 *       28-56 positions and the MFMAs run 22-44 % full. Here the patch holds 128 channels at a time: K runs chunk-major
 *       (chunk, tap), the patch is re-staged and re-centred between chunks (three barriers, the pipeline restarts), the pixel
 *       sums accumulate over the chunks, and the weight stream jumps through the packed image (tap-major there). */
template <int MB, int NTB, int KSTEP, int SEQ, bool FULL, bool CHUNK = false>
__global__ __launch_bounds__(MB * NTB * 16, 2)
void q8_conv_patch_kernel(const IgemmParams p, const ConvGeom g, const PatchArgs a)
{
  static_assert(!CHUNK || KSTEP == 128, "channel chunks are one 128-byte step per tap");
  constexpr int kPtWaves = MB * NTB / 4;
  constexpr int kPtThreads = kPtWaves * 64;
  static_assert((MB == 4 || MB == 8) && (NTB == 4 || NTB == 8) && kPtWaves <= 8, "tile flavours");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [patch][pixel sums][bias][ring of weight steps]
  constexpr uint32_t kWNG = NTB / 2;                  // waves along channels
  constexpr int kSub = KSTEP / 32;                    // 32-deep sub-steps (MFMA K) per step: 4 or 2
  constexpr uint32_t kStages = pt_stages(KSTEP);      // ring slots; kStages - 1 steps are requested ahead
  constexpr int kAhead = static_cast<int>(kStages) - 1;
  constexpr uint32_t kStepBytes = NTB * kSub * 1024u; // weight fragments of one K step
  constexpr uint32_t kRingMask = kStages * kStepBytes - 1u;
  constexpr int kPpw = NTB * kSub / kPtWaves;         // ring pieces (1 KiB) per wave and step: 1, 2 or 4
  static_assert(kPpw >= 1 && (kStages & (kStages - 1u)) == 0u && kSub % 2 == 0, "ring: whole pieces per wave, a power-of-two slot count");

  // (measurement builds: cycle stamps (item 0) and 100 MHz wall-clock stamps (item 1) of wave 0 -- entry, requests issued,
  //  patch landed, patch re-centred, K loop done, stores issued; tools/trace_patch.py)
#define PT_STAMP(slot) do { QNNP_TRACE(p, blockIdx.x, 0, slot); QNNP_TRACE_WALL(p, blockIdx.x, 1, slot); } while (0)
  PT_STAMP(0);
#ifdef QNNP_ENABLE_ABLATION
  const uint32_t abl = a.abl;
#else
  constexpr uint32_t abl = 0;
#endif
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t khalf = lane >> 5;
  const uint32_t wm = wave / kWNG, wn = wave % kWNG;  // this wave: position blocks 2 wm, 2 wm + 1; channel blocks 2 wn, 2 wn + 1

  const uint32_t tile_m = pt_div(blockIdx.x, a.inv_tiles_n);
  const uint32_t tile_n = blockIdx.x - tile_m * a.tiles_n;
  uint32_t img0, row0;
  if (a.imgs > 1) { img0 = tile_m * a.imgs; row0 = 0; }
  else { img0 = pt_div(tile_m, a.inv_tiles_r); row0 = (tile_m - img0 * a.tiles_r) * a.rows; }

  const uint32_t C = CHUNK ? 128u : p.kc;             // channels the patch holds
  const uint32_t cpp = C >> 4;
  const uint32_t ps = a.ps;                           // bytes between patch pixels in LDS: C + 16
  const uint32_t kblocks = p.k_pad >> 5;
  const uint32_t nb_tile = tile_n * NTB;              // first 32-channel block of the tile

  const uint32_t lds0 = pt_lds_off(lds);              // (LDS-DMA destinations and the asm accesses address the allocation itself)
  int32_t* pix = reinterpret_cast<int32_t*>(lds + a.pix_off);
  int32_t* bias_lds = reinterpret_cast<int32_t*>(lds + a.bias_off);

  // ---- the weight stream: piece j of a step = fragment (channel block j / kSub, sub-step j % kSub); wave w owns j = w (+ 8 ..).
  //      Wave-uniform 64-bit sources in scalar registers, one add-with-carry per piece and step.
  uint64_t w_src[kPpw];
  uint32_t w_dst[kPpw];
#pragma unroll
  for (int i = 0; i < kPpw; i++) {
    const uint32_t j = wave + i * kPtWaves;
    w_src[i] = pt_scalar64(reinterpret_cast<uint64_t>(p.packed_w) + (static_cast<uint64_t>(nb_tile + j / kSub) * kblocks + j % kSub) * 1024u);
    w_dst[i] = __builtin_amdgcn_readfirstlane(lds0 + a.ring_off + j * 1024u);
  }
  const uint32_t lane16 = lane * 16u;
  uint32_t wr_off = 0;                               // ring offset of the next step to request
  uint32_t st_tap = 0, st_ch = 0;                    // CHUNK: (tap, chunk) of the next step to request; its K offset in the
  auto stage = [&]() __attribute__((always_inline)) { //        packed image is (tap * nch + chunk) * 128 bytes
#pragma unroll
    for (int i = 0; i < kPpw; i++) {
      if constexpr (CHUNK) {
        pt_dma16_saddr(w_src[i] + (st_tap * a.nch + st_ch) * (kSub * 1024u), lane16, w_dst[i] + wr_off);
      } else {
        pt_dma16_saddr(w_src[i], lane16, w_dst[i] + wr_off);
        w_src[i] += kSub * 1024u;
      }
    }
    if constexpr (CHUNK) { if (++st_tap == 9u) { st_tap = 0; st_ch++; } }
    wr_off = (wr_off + kStepBytes) & kRingMask;
  };
#pragma unroll
  for (int i = 0; i < kAhead; i++) stage();           // (ksteps >= 9 > kAhead)

  // ---- the patch: LDS chunk v = pixel q * (cpp + 1) + c <- chunk c of input pixel (img0 + il, iy, ix), or the zero-point line
  //      of the fill table (pixels outside the image; c == cpp is the pixel's padding chunk). `ch`: the channel chunk (CHUNK).
  // CHUNK: the source of a lane's first kCache pieces is worked out once (chunk 0) and kept -- the later chunks are the same
  // pixels 128 bytes further (stamps of the first build, 14x14 512 -> 512: 6.1 k cycles of address arithmetic per chunk for
  // four waves, a third of a chunk's nine multiply steps)
  constexpr int kCache = CHUNK ? 16 : 1;
  const uint8_t* src_cache[kCache];
  uint32_t src_inside = 0;                           // bit i: piece i of this lane reads the image (not the zero-point line)
  auto load_patch = [&](uint32_t ch) __attribute__((always_inline)) {
    const uint8_t* zp_line = p.fill_table + (p.izp_fill & 0xFFu) * 16u;
    const uint32_t pieces = (a.chunks + 63u) >> 6;
    auto source = [&](uint32_t piece, bool* inside_out) __attribute__((always_inline)) -> const uint8_t* {
      const uint32_t v = min(piece * 64u + lane, a.chunks - 1u);
      const uint32_t q = pt_div(v, a.inv_cpp1);
      const uint32_t c = v - q * (cpp + 1u);
      const uint32_t il = pt_div(q, a.inv_pimg);
      const uint32_t rem = q - il * a.pimg;
      const uint32_t prr = pt_div(rem, a.inv_pw);
      const uint32_t pc = rem - prr * a.pw;
      const int32_t iy = static_cast<int32_t>(row0 * g.sh + prr) - static_cast<int32_t>(g.pad_top);
      const int32_t ix = static_cast<int32_t>(pc) - static_cast<int32_t>(g.pad_left);
      const uint32_t img = img0 + il;
      const bool inside = c < cpp && img < a.batch && static_cast<uint32_t>(iy) < g.H && static_cast<uint32_t>(ix) < g.W;
      *inside_out = inside;
      return inside
          ? p.input + static_cast<uint64_t>(img) * p.image_stride +
                static_cast<uint64_t>(static_cast<uint32_t>(iy) * g.W + static_cast<uint32_t>(ix)) * p.input_stride + (c << 4)
          : zp_line;
    };
    uint32_t first = wave;
    if constexpr (CHUNK) {
#pragma unroll
      for (int i = 0; i < kCache; i++) {
        const uint32_t piece = wave + static_cast<uint32_t>(i) * kPtWaves;
        if (piece < pieces) {                          // (wave-uniform)
          if (ch == 0) {
            bool inside;
            src_cache[i] = source(piece, &inside);
            src_inside |= inside ? (1u << i) : 0u;
          }
          pt_dma16(src_cache[i] + (((src_inside >> i) & 1u) != 0u ? ch * 128u : 0u), lds0 + piece * 1024u);
        }
      }
      first = wave + static_cast<uint32_t>(kCache) * kPtWaves;
    }
    for (uint32_t piece = first; piece < pieces; piece += kPtWaves) {
      bool inside;
      const uint8_t* src = source(piece, &inside);
      pt_dma16(src + (inside ? ch * 128u : 0u), lds0 + piece * 1024u);
    }
    if (ch == 0 && wave == kPtWaves - 1 && lane < NTB * 8u) {
      const int32_t* b = (rq_is_lane<SEQ>() ? p.bias2u : p.bias2) + nb_tile * 32u;
      pt_dma16(reinterpret_cast<const uint8_t*>(b) + lane * 16u, lds0 + a.bias_off);
    }
  };
  load_patch(0);
  PT_STAMP(1);
  pt_wait_vmcnt<0>();
  asm volatile("s_barrier" ::: "memory");
  PT_STAMP(2);

  // ---- one pass over the landed patch, four threads per pixel (a quarter of its chunks each): re-centre in place (a ^ 0x80),
  //      the pixel's channel sum (of a') beside it
  auto patch_pass = [&](uint32_t ch) __attribute__((always_inline)) {
    if (abl & 32u) return;
    const bool sums = p.row_coeff != 0;
    if constexpr (CHUNK) {
      // two chunks per thread; four tasks per trip with all eight reads in front (the plain loop below is one LDS round trip per
      // task: 3.8 k cycles per chunk for the four-wave flavour by the stamps)
      const uint32_t total = a.ppix * 4u;
      for (uint32_t t0 = tid; t0 < total; t0 += kPtThreads * 4u) {
        v4i x[4][2];
        uint32_t mine[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = min(t0 + static_cast<uint32_t>(u) * kPtThreads, total - 1u);
          mine[u] = (t >> 2) * ps + (t & 3u) * 32u;
          x[u][0] = *reinterpret_cast<const v4i*>(lds + mine[u]);
          x[u][1] = *reinterpret_cast<const v4i*>(lds + mine[u] + 16u);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const uint32_t t = t0 + static_cast<uint32_t>(u) * kPtThreads;
          if (t < total) {                            // (whole quads: total is a multiple of four)
            uint32_t s = 0;
#pragma unroll
            for (int c = 0; c < 2; c++) {
              v4i y = x[u][c];
              if (sums) {
                s = __builtin_amdgcn_sad_u8(y.x, 0u, s);
                s = __builtin_amdgcn_sad_u8(y.y, 0u, s);
                s = __builtin_amdgcn_sad_u8(y.z, 0u, s);
                s = __builtin_amdgcn_sad_u8(y.w, 0u, s);
              }
              y.x ^= static_cast<int>(kPtFlip); y.y ^= static_cast<int>(kPtFlip); y.z ^= static_cast<int>(kPtFlip); y.w ^= static_cast<int>(kPtFlip);
              pt_ds_write16(lds0 + mine[u] + c * 16u, y);
            }
            if (sums) {
              s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0xB1, 0xF, 0xF, false));
              s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0x4E, 0xF, 0xF, false));
              if ((t & 3u) == 0u) {
                int32_t sum = static_cast<int32_t>(s) - 128 * 128;
                if (ch != 0) sum += pix[t >> 2];      // (the same thread wrote it a chunk ago)
                pt_ds_write4(lds0 + a.pix_off + (t >> 2) * 4u, sum);
              }
            }
          }
        }
      }
      return;
    }
    const uint32_t per = cpp >> 2;                    // chunks per thread: 1 .. 8
    for (uint32_t t = tid; t < a.ppix * 4u; t += kPtThreads) {
      const uint32_t q = t >> 2;
      const uint32_t mine = q * ps + (t & 3u) * per * 16u;
      uint32_t s = 0;
      for (uint32_t c = 0; c < per; c++) {
        v4i x = *reinterpret_cast<const v4i*>(lds + mine + c * 16u);
        if (sums) {
          s = __builtin_amdgcn_sad_u8(x.x, 0u, s);
          s = __builtin_amdgcn_sad_u8(x.y, 0u, s);
          s = __builtin_amdgcn_sad_u8(x.z, 0u, s);
          s = __builtin_amdgcn_sad_u8(x.w, 0u, s);
        }
        x.x ^= static_cast<int>(kPtFlip); x.y ^= static_cast<int>(kPtFlip); x.z ^= static_cast<int>(kPtFlip); x.w ^= static_cast<int>(kPtFlip);
        pt_ds_write16(lds0 + mine + c * 16u, x);
      }
      if (sums) {
        s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
        s += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(s), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
        if ((t & 3u) == 0u) {
          pt_ds_write4(lds0 + a.pix_off + q * 4u, static_cast<int32_t>(s) - 128 * static_cast<int32_t>(C));
        }
      }
    }
  };
  patch_pass(0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  PT_STAMP(3);

  // ---- this lane's positions (one per position block): patch pixel of tap (0, 0), output offset
  // (round 6) the stores: after the requantization this lane holds 16 channels of ITS position for each of the wave's two channel blocks --
  // 32-byte pieces per position and instruction. The 16-lane rows trade pieces instead (v_permlane16_swap, q8convc3.hip c3_store_unit):
  // the first instruction of a position block then writes positions 0..15 with all 64 channels of the wave, the second one 16..31, so
  // `out_off[mi][half]` is the offset of position 16 half + (lane & 15) plus this lane's piece: 32 (row & 1) + 16 (row >> 1), row = lane >> 4
  uint32_t q0[2], out_off[2][2], abase0[2];
#pragma unroll
  for (int mi = 0; mi < 2; mi++) {
    const uint32_t pos = (wm * 2u + mi) * 32u + (lane & 31u);
    const uint32_t il = pt_div(pos, a.inv_rw);
    const uint32_t rem = pos - il * (a.rows * g.OW);
    const uint32_t r = pt_div(rem, a.inv_ow);
    const uint32_t x = rem - r * g.OW;
    q0[mi] = pos < a.pos ? il * a.pimg + r * g.sh * a.pw + x * g.sw : 0u;
    abase0[mi] = lds0 + q0[mi] * ps + khalf * 16u;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const uint32_t spos = (wm * 2u + mi) * 32u + half * 16u + (lane & 15u);
      const uint32_t sil = pt_div(spos, a.inv_rw);
      const uint32_t srem = spos - sil * (a.rows * g.OW);
      const uint32_t sr = pt_div(srem, a.inv_ow);
      const uint32_t sx = srem - sr * g.OW;
      const bool ok = spos < a.pos && img0 + sil < a.batch && row0 + sr < g.OH;
      const uint32_t row16 = lane >> 4;
      out_off[mi][half] = ok ? (((img0 + sil) * g.OH + row0 + sr) * g.OW + sx) * p.output_stride + nb_tile * 32u + wn * 64u +
                                   (row16 & 1u) * 32u + (row16 >> 1) * 16u
                             : 0xFFFFFF00u;         // (beyond the descriptor: the store is dropped)
    }
  }
  // ---- accumulators start at the folded bias
  v16i acc[2][2];
#pragma unroll
  for (int ni = 0; ni < 2; ni++)
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const v4i b = *reinterpret_cast<const v4i*>(bias_lds + (wn * 2u + ni) * 32u + rg * 8 + khalf * 4);
#pragma unroll
      for (int mi = 0; mi < 2; mi++) {
        acc[mi][ni][rg * 4 + 0] = b.x; acc[mi][ni][rg * 4 + 1] = b.y; acc[mi][ni][rg * 4 + 2] = b.z; acc[mi][ni][rg * 4 + 3] = b.w;
      }
    }

  // ---- K loop. Fragment addresses are a register plus an immediate: activations = acur[mi] + 32 * sub-step, acur = this lane's
  //      pixel of the tap + KSTEP bytes per step; weights = bcur + (channel block, sub-step) KiB, bcur = the ring slot
  struct Frags { v4i a[2], b[2]; };
  const uint32_t bbase = lds0 + a.ring_off + (wn * 2u * kSub) * 1024u + lane16;     // channel block 2 wn, sub-step 0 of slot 0
  uint32_t acur[2] = {abase0[0], abase0[1]};
  uint32_t bcur = bbase;
  uint32_t rd_off = 0, kc = 0, tap = 0;              // ring offset / channel step / tap of the step whose fragments are read next
  auto load = [&](Frags& f, auto sub_c) __attribute__((always_inline)) {
    constexpr uint32_t sub = decltype(sub_c)::value;
    if (abl & 4u) return;
    f.a[0] = pt_ds_read16<sub * 32>(acur[0]);
    f.a[1] = pt_ds_read16<sub * 32>(acur[1]);
    f.b[0] = pt_ds_read16<sub * 1024>(bcur);
    f.b[1] = pt_ds_read16<sub * 1024 + kSub * 1024>(bcur);
  };
  // The same four reads WITH their wait, as one asm block: for the places where the registers cross an irregular edge of the
  // control flow (the chunk loop's header) -- hipcc is free to copy an asm output the moment the asm ends, and a copy of a
  // register whose ds_read is still in flight copies the OLD value (the first chunked build re-started every chunk with the
  // previous step's fragments that way). Inside the regular step loop the deferred waits are checked by the parity tests.
  auto load_now = [&](Frags& f) __attribute__((always_inline)) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:%7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(f.a[0]), "=&v"(f.a[1]), "=&v"(f.b[0]), "=&v"(f.b[1])
                 : "v"(acur[0]), "v"(acur[1]), "v"(bcur), "n"(kSub * 1024) : "memory");
  };
  auto advance = [&]() __attribute__((always_inline)) {       // to the next step: KSTEP bytes further, or the next tap's pixel
    rd_off = (rd_off + kStepBytes) & kRingMask;
    bcur = bbase + rd_off;
    if (CHUNK || ++kc == a.csteps) {
      kc = 0;
      tap = min(tap + 1u, 8u);
      const uint32_t ky = (tap * 11u) >> 5;          // tap / 3 for tap < 9
      const uint32_t tap_bytes = (ky * a.pw + tap - ky * 3u) * ps;
      acur[0] = abase0[0] + tap_bytes;
      acur[1] = abase0[1] + tap_bytes;
    } else {
      acur[0] += KSTEP;
      acur[1] += KSTEP;
    }
  };
  auto mfma1 = [&](v16i& c, const v4i& w, const v4i& x) __attribute__((always_inline)) {
    if (!(abl & 8u)) c = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, x, c, 0, 0, 0);
  };

  Frags fa, fb;
  fa.a[0] = fa.a[1] = fa.b[0] = fa.b[1] = fb.a[0] = fb.a[1] = fb.b[0] = fb.b[1] = v4i{0, 0, 0, 0};
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the bias reads above are the compiler's: none may be pending beside the counted ones)
  if constexpr (CHUNK) load_now(fa); else load(fa, std::integral_constant<uint32_t, 0>{});
  // One sub-step: the MFMAs of `cur`, whose fragments were requested a sub-step ago, while `nxt` receives the next sub-step's.
  // Before the LAST sub-step of a step: counted wait + barrier (the next step's fragments are in their slot, everyone has left
  // the previous one's), then the next step's first fragments. STAGE (first sub-step): request the step kAhead ahead into the
  // slot the previous step has left, under the MFMAs. VM: this wave's ring requests that may stay in flight at the barrier.
  auto sub_step = [&](Frags& cur, Frags& nxt, auto sub_c, auto stage_c, auto vm_c, auto last_c) __attribute__((always_inline)) {
    constexpr int sub = decltype(sub_c)::value;
    constexpr bool kStage = decltype(stage_c)::value;
    constexpr int kVm = decltype(vm_c)::value;
    constexpr bool kLast = decltype(last_c)::value;
    bool more = true;                                 // `nxt` is being loaded behind `cur`
    if constexpr (sub + 1 < kSub) {
      load(nxt, std::integral_constant<uint32_t, sub + 1>{});
    } else {
      pt_wait_vmcnt<kVm>();
      if (!(abl & 2u)) asm volatile("s_barrier" ::: "memory");
      if constexpr (!kLast) { advance(); load(nxt, std::integral_constant<uint32_t, 0>{}); } else { more = false; }
    }
    if constexpr (sub + 1 < kSub || !kLast) pt_wait_lgkm<4>(cur.a[0], cur.a[1], cur.b[0], cur.b[1]);
    else pt_wait_lgkm<0>(cur.a[0], cur.a[1], cur.b[0], cur.b[1]);
    (void) more;
    __builtin_amdgcn_sched_barrier(0);
    mfma1(acc[0][0], cur.b[0], cur.a[0]);
    if constexpr (sub == 0 && kStage) {
      __builtin_amdgcn_sched_barrier(0);
      if (!(abl & 1u)) stage();
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma1(acc[0][1], cur.b[1], cur.a[0]);
    mfma1(acc[1][0], cur.b[0], cur.a[1]);
    mfma1(acc[1][1], cur.b[1], cur.a[1]);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto step_body = [&](auto stage_c, auto vm_c, auto last_c) __attribute__((always_inline)) {
    sub_step(fa, fb, std::integral_constant<int, 0>{}, stage_c, vm_c, last_c);
    sub_step(fb, fa, std::integral_constant<int, 1>{}, stage_c, vm_c, last_c);
    if constexpr (kSub == 4) {
      sub_step(fa, fb, std::integral_constant<int, 2>{}, stage_c, vm_c, last_c);
      sub_step(fb, fa, std::integral_constant<int, 3>{}, stage_c, vm_c, last_c);
    }
  };
  using T = std::true_type;
  using F = std::false_type;
  if constexpr (CHUNK) {
    // chunk-major: nine steps (taps) per chunk; a chunk's last step reads nothing ahead (the patch is about to change) but
    // still requests the next chunk's first weights, which land under the re-staging
    using V0 = std::integral_constant<int, 0>;
    for (uint32_t ch = 0; ch < a.nch; ch++) {
      if (ch != 0) {
        asm volatile("s_barrier" ::: "memory");       // every wave has its last fragments of the previous chunk
        load_patch(ch);
        pt_wait_vmcnt<0>();
        asm volatile("s_barrier" ::: "memory");
        patch_pass(ch);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        rd_off = (rd_off + kStepBytes) & kRingMask;   // (the last step did not advance)
        bcur = bbase + rd_off;
        tap = 0;
        acur[0] = abase0[0];
        acur[1] = abase0[1];
        load_now(fa);
      }
      for (uint32_t t = 0; t < 8u; t++) step_body(T{}, V0{}, F{});
      if (ch + 1u < a.nch) step_body(T{}, V0{}, T{}); else step_body(F{}, V0{}, T{});
    }
  } else {
  // at the barrier of step s the requests of steps s + 2 .. s + kAhead may stay in flight; the last kAhead steps request nothing
  for (uint32_t step = 0; step + kAhead < a.ksteps; step++) step_body(T{}, std::integral_constant<int, (kAhead - 1) * kPpw>{}, F{});
  if constexpr (kAhead == 3) {
    step_body(F{}, std::integral_constant<int, kPpw>{}, F{});
    step_body(F{}, std::integral_constant<int, 0>{}, F{});
  }
  step_body(F{}, std::integral_constant<int, 0>{}, T{});
  }
  PT_STAMP(4);

  // ---- fused epilogue: row term (sum of the window's pixel sums), requantization in registers, 16-byte stores
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>((p.rows - 1u) * p.output_stride + p.n), 0x00020000);   // (launcher: < 2^31)
#pragma unroll
  for (int mi = 0; mi < 2; mi++) {
    int32_t s = 0;
    if (p.row_coeff != 0) {
      const int32_t* pq = pix + q0[mi];
#pragma unroll
      for (int t = 0; t < 9; t++) s += pq[(t / 3) * a.pw + (t % 3)];
    }
    const int32_t rowterm = with_rq_offset<SEQ>(p.row_coeff * s);
    uint64_t row_addend = 0;
    if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
    v4i outv[2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++) {
      uint32_t pk[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        if constexpr (rq_is_lane<SEQ>()) {
          pk[rg] = q31_requantize_pack4_lane<SEQ, FULL>(
              static_cast<uint32_t>(acc[mi][ni][rg * 4 + 0]), static_cast<uint32_t>(acc[mi][ni][rg * 4 + 1]),
              static_cast<uint32_t>(acc[mi][ni][rg * 4 + 2]), static_cast<uint32_t>(acc[mi][ni][rg * 4 + 3]), row_addend, p.lane, p.rq);
        } else {
          pk[rg] = q31_requantize_pack4<SEQ, FULL, false>(
              add_wrap(acc[mi][ni][rg * 4 + 0], rowterm), add_wrap(acc[mi][ni][rg * 4 + 1], rowterm),
              add_wrap(acc[mi][ni][rg * 4 + 2], rowterm), add_wrap(acc[mi][ni][rg * 4 + 3], rowterm), p.rq);
        }
      }
      const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
      // this lane now holds 16 consecutive channels of ITS position: (2 wn + ni) * 32 + khalf * 16 .. + 15
      outv[ni] = v4i{static_cast<int>(s02[0]), static_cast<int>(s02[1]), static_cast<int>(s13[0]), static_cast<int>(s13[1])};
    }
    // row 1 of block 0 <-> row 0 of block 1, row 3 <-> row 2: outv[0] = positions 0..15 of the block x the wave's 64 channels, outv[1] = 16..31
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const auto sw = __builtin_amdgcn_permlane16_swap(static_cast<uint32_t>(outv[0][d]), static_cast<uint32_t>(outv[1][d]), false, false);
      outv[0][d] = static_cast<int>(sw[0]); outv[1][d] = static_cast<int>(sw[1]);
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
      __builtin_amdgcn_raw_buffer_store_b128(
          __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, outv[half]), out_rsrc,
          (abl & 64u) ? 0xFFFFFF00u : out_off[mi][half], 0, 0);
    }
  }
  PT_STAMP(5);
#undef PT_STAMP
}

template <int MB, int NTB, int KSTEP, int SEQ, bool FULL, bool CHUNK = false>
int launch_patch_as(const IgemmParams& p, const ConvGeom& g, const PatchArgs& a, uint32_t lds_bytes, hipStream_t stream)
{
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_patch_kernel<MB, NTB, KSTEP, SEQ, FULL, CHUNK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kPtLdsLimit)) != hipSuccess) {
      (void) hipGetLastError();
    }
  }
  hipLaunchKernelGGL((q8_conv_patch_kernel<MB, NTB, KSTEP, SEQ, FULL, CHUNK>), dim3(a.tiles_m * a.tiles_n), dim3(MB * NTB * 16), lds_bytes, stream, p, g, a);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/* Tile geometry and LDS plan of one flavour (mb position blocks x ntb channel blocks), or false when it does not fit. */
bool plan_flavour(const IgemmParams& p, const ConvGeom& g, uint32_t batch, uint32_t mb, uint32_t ntb, PatchArgs* a, uint32_t* lds_bytes,
                  bool chunked = false)
{
  if (chunked && (p.kc % 128u != 0 || p.kc < 256u)) return false;
  const uint32_t C = chunked ? 128u : p.kc;            // channels the patch holds at a time
  const uint32_t pmax = mb * 32u;
  const uint32_t cpp = C >> 4;
  if (p.n % (ntb * 32u) != 0 || g.OW > pmax) return false;
  uint32_t imgs = 1, rows = g.OH;
  if (g.OH * g.OW <= pmax) {
    imgs = pmax / (g.OH * g.OW);
    if (imgs > batch) imgs = batch;
  } else {
    rows = pmax / g.OW;
    const uint32_t tiles_r = (g.OH + rows - 1) / rows;
    rows = (g.OH + tiles_r - 1) / tiles_r;            // even rows per tile
  }
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_PATCH_ROWS")) {   // measurement builds: at most this many output rows per tile
    const uint32_t cap = static_cast<uint32_t>(atoi(env));
    if (cap != 0 && imgs == 1 && rows > cap) rows = cap;
  }
#endif
  const uint32_t kstep = p.kc % 128u == 0 ? 128u : 64u;
  const uint32_t ring = pt_stages(static_cast<int>(kstep)) * ntb * (kstep / 32u) * 1024u;
  for (;;) {
    a->imgs = imgs; a->rows = rows;
    a->pr = (rows - 1u) * g.sh + 3u;
    a->pw = (g.OW - 1u) * g.sw + 3u;
    a->pimg = a->pr * a->pw;
    a->ppix = imgs * a->pimg;
    a->chunks = a->ppix * (cpp + 1u);
    const uint32_t patch_bytes = ((a->chunks + 63u) / 64u) * 1024u;
    a->pix_off = patch_bytes;
    a->bias_off = a->pix_off + ((a->ppix * 4u + 15u) & ~15u);
    a->ring_off = (a->bias_off + ntb * 128u + 1023u) & ~1023u;
    *lds_bytes = a->ring_off + ring;
    if (*lds_bytes <= kPtLdsLimit) break;
    if (imgs > 1) imgs--;                              // fewer images, then fewer rows, per tile
    else if (rows > 1) rows = (rows + 1u) / 2u;
    else return false;
  }
  a->batch = batch;
  a->pos = a->imgs * a->rows * g.OW;
  a->tiles_r = a->imgs > 1 ? 1u : (g.OH + a->rows - 1u) / a->rows;
  a->tiles_m = a->imgs > 1 ? (batch + a->imgs - 1u) / a->imgs : batch * a->tiles_r;
  a->tiles_n = p.n / (ntb * 32u);
  if (static_cast<uint64_t>(a->tiles_m) * a->tiles_n >= (UINT64_C(1) << 31)) return false;
  // the reciprocal divisions are exact while dividend * divisor < 2^32: the operands are tile ids, positions (< 256) and
  // patch chunks (< 2^16)
  if (static_cast<uint64_t>(a->tiles_m) * a->tiles_n * a->tiles_n >= (UINT64_C(1) << 32)) return false;
  if (static_cast<uint64_t>(a->tiles_m) * a->tiles_r >= (UINT64_C(1) << 32)) return false;
  if (a->chunks >= 65536u) return false;
  a->inv_ow = pt_magic(g.OW);
  a->inv_rw = pt_magic(a->rows * g.OW);
  a->inv_pw = pt_magic(a->pw);
  a->inv_pimg = pt_magic(a->pimg);
  a->inv_tiles_r = pt_magic(a->tiles_r);
  a->inv_tiles_n = pt_magic(a->tiles_n);
  a->ps = C + 16u;
  a->inv_cpp1 = pt_magic(cpp + 1u);
  a->csteps = C / kstep;
  a->ksteps = 9u * (p.kc / kstep);
  a->nch = chunked ? p.kc / 128u : 1u;
  a->abl = 0;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_PATCH_ABL")) a->abl = static_cast<uint32_t>(atoi(env));
#endif
  return true;
}

/* The flavour of a shape: (8, 4) for output channels in odd multiples of 128; otherwise (4, 8), or the four-wave (4, 4) when
 * (4, 8) would launch fewer workgroups than the chip has CUs. */
bool plan_patch(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch, PatchArgs* a, uint32_t* mb,
                uint32_t* ntb, uint32_t* lds_bytes)
{
  if (groups != 1 || vec != 16 || p.fill_table == nullptr || p.offsets == nullptr || batch == 0) return false;
  if (g.KH != 3 || g.KW != 3 || g.dh != 1 || g.dw != 1 || g.sh != g.sw || g.sh == 0 || g.sh > 2) return false;
  const uint32_t C = p.kc;
  if (!(C == 64 || C == 128 || C == 256 || C == 512)) return false;
  if (p.k_total != 9u * C || p.k_pad != p.k_total) return false;
  if (p.n != p.n_pad || p.n % 128u != 0 || p.store_mode != 2) return false;
  if (g.OW == 0 || g.OH == 0 || p.rows != batch * g.OH * g.OW) return false;
  const uint64_t out_bytes = static_cast<uint64_t>(p.rows - 1u) * p.output_stride + p.n;
  if (out_bytes >= (UINT64_C(1) << 31) - 512u) return false;
  uint32_t forced = 0;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_PATCH_TILE")) forced = static_cast<uint32_t>(atoi(env));   // 84, 44 or 48 = (mb, ntb)
#endif
  // the whole patch, or 128 channels of it at a time where that buys a fuller tile of positions (the LDS limit shrank the
  // whole-patch tile: stride 2 with 256 / 512 channels)
  int chunk_mode = -1;                                 // -1 automatic, 0 never, 1 wherever it plans
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_PATCH_CHUNK")) chunk_mode = atoi(env);
#endif
  auto plan_flavour = [&](const IgemmParams& pp, const ConvGeom& gg, uint32_t b, uint32_t m, uint32_t n, PatchArgs* out, uint32_t* lds) {
    PatchArgs whole, part;
    uint32_t lds_whole = 0, lds_part = 0;
    const bool ok_whole = qnnp::plan_flavour(pp, gg, b, m, n, &whole, &lds_whole, false);
    const bool ok_part = chunk_mode != 0 && qnnp::plan_flavour(pp, gg, b, m, n, &part, &lds_part, true);
    if (ok_part && (!ok_whole || chunk_mode == 1 || part.pos > whole.pos)) { *out = part; *lds = lds_part; return true; }
    if (ok_whole) { *out = whole; *lds = lds_whole; return true; }
    return false;
  };
  if (forced != 0) {
    *mb = forced / 10u; *ntb = forced % 10u;
    return ((*mb == 8 && *ntb == 4) || (*mb == 4 && (*ntb == 4 || *ntb == 8))) && plan_flavour(p, g, batch, *mb, *ntb, a, lds_bytes);
  }
  if (p.n % 256u != 0) { *mb = 8; *ntb = 4; return plan_flavour(p, g, batch, 8, 4, a, lds_bytes); }
  // (same box, interleaved -- profiles/r05/conv_patch_tile_flavours_ab_r05piter4.txt: two four-wave workgroups per CU LOSE to one
  //  eight-wave workgroup on 14x14 256 -> 256, 22.4 against 24.3 us, and on the stride-2 rows; they win where the eight-wave
  //  tiling leaves CUs without a workgroup: 7x7 512 -> 512, 128 workgroups on 256 CUs, 32.8 -> 28.8 us)
  *mb = 4; *ntb = 8;
  const bool wide = plan_flavour(p, g, batch, 4, 8, a, lds_bytes);
  if (wide && static_cast<uint64_t>(a->tiles_m) * a->tiles_n >= p.cu_count) return true;
  PatchArgs narrow;
  uint32_t narrow_lds = 0;
  if (plan_flavour(p, g, batch, 4, 4, &narrow, &narrow_lds)) { *a = narrow; *lds_bytes = narrow_lds; *ntb = 4; return true; }
  return wide;
}

}  // namespace

/* dense 3x3 (stride 1 or 2, dilation 1), one group, 64 / 128 / 256 / 512 input channels, output channels a multiple of
 * 128 stored densely in whole 16-byte pieces, an output row of at most 128 positions */
bool convpatch_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch)
{
  PatchArgs a;
  uint32_t mb = 0, ntb = 0, lds_bytes = 0;
  return plan_patch(p, g, groups, vec, batch, &a, &mb, &ntb, &lds_bytes);
}

int convpatch_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name)
{
  PatchArgs a;
  uint32_t mb = 0, ntb = 0, lds_bytes = 0;
  if (!plan_patch(p, g, 1, 16, batch, &a, &mb, &ntb, &lds_bytes)) return QNNP_HIP_EINVAL;
  *name = "q8_conv_patch_mfma";
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    constexpr bool kFull = decltype(full)::value;
    if (a.nch > 1u) {
      if (mb == 8u) rc = launch_patch_as<8, 4, 128, kSeq, kFull, true>(p, g, a, lds_bytes, stream);
      else if (ntb == 8u) rc = launch_patch_as<4, 8, 128, kSeq, kFull, true>(p, g, a, lds_bytes, stream);
      else rc = launch_patch_as<4, 4, 128, kSeq, kFull, true>(p, g, a, lds_bytes, stream);
    } else if (p.kc % 128u == 0) {
      if (mb == 8u) rc = launch_patch_as<8, 4, 128, kSeq, kFull>(p, g, a, lds_bytes, stream);
      else if (ntb == 8u) rc = launch_patch_as<4, 8, 128, kSeq, kFull>(p, g, a, lds_bytes, stream);
      else rc = launch_patch_as<4, 4, 128, kSeq, kFull>(p, g, a, lds_bytes, stream);
    } else {
      if (mb == 8u) rc = launch_patch_as<8, 4, 64, kSeq, kFull>(p, g, a, lds_bytes, stream);
      else if (ntb == 8u) rc = launch_patch_as<4, 8, 64, kSeq, kFull>(p, g, a, lds_bytes, stream);
      else rc = launch_patch_as<4, 4, 64, kSeq, kFull>(p, g, a, lds_bytes, stream);
    }
  });
  return rc;
}

}  // namespace qnnp
