/*
 * q8gemm256.hip -- the large-problem uint8 GEMM / implicit-GEMM kernel:
 * 256x256 output tile per workgroup, operands staged by LDS-DMA, int8 MFMA.
 *
 * Same arithmetic and operand roles as q8igemm.hip (which see): it replaces the
 * same reference microkernels (src/q8gemm/4x4c2-sse2.c:14-318,
 * src/q8conv/4x4c2-sse2.c:14-273) for shapes big enough to be MFMA-bound, i.e.
 * BASELINE.json configs[1] (q8gemm M=N=K=4096).
 *
 * Structure (per workgroup: one 256 x 256 output tile; 8 waves as 4 (rows) x 2 (channels), 64 x 128
 * outputs per wave -- or, A/B flavour, 4 waves as 2 x 2 with 128 x 128 per wave):
 *   - K advances 64 bytes per tile; a ring of four LDS stages of {activations 256x64 B, weights
 *     256x64 B} = 128 KiB, filled with global_load_lds (16 B per lane, no VGPR round trip). One raw
 *     s_barrier per K tile behind a counted s_waitcnt vmcnt, so up to three tiles stay in flight;
 *     the DMA of tile kt+4 goes into the slot of tile kt from that tile's mid-point barrier on.
 *   - activations keep the caller's row-major image; the 16-byte chunk index is XOR-swizzled with
 *     (row >> 2) & 3 -- applied to the DMA SOURCE address and to the ds_read_b128 address (the
 *     LDS-DMA destination is lane-linear by construction), so fragment reads are conflict free.
 *   - weights arrive already as MFMA fragments (pack.h), copied verbatim: fragment reads are linear.
 *   - uint8 -> int8 re-centring of activations is one v_xor per fragment dword after the LDS read;
 *     the per-row sum needed for the kernel-zero-point term is taken over the RAW bytes with
 *     v_sad_u8 (v_dot4c costs ~7 cycles beside MFMAs, v_sad_u8 ~0.5), each of the 2 channel-waves
 *     doing one half of K (the wave-specific rotation of the K sub-step order makes that
 *     branch-free), combined through LDS at the end.
 *   - two software-pipelined phases per tile, instruction order pinned with sched_barrier (see the
 *     comment at `iteration`).
 *   - convolution: each lane's DMA source comes from the device offset table; padding taps and
 *     K padding read constant 16-byte lines of the fill table.
 *   - epilogue fused in registers: + bias2 + row term -> Q31 requantize -> clamp -> 4 channels
 *     per dword.
 *
 * What bounds it on MI355X (measured, DESIGN.md "GEMM headroom"): under this kernel the chip clocks at
 * 1.3-1.5 GHz (power management; a bare random-operand MFMA loop holds 1.78 GHz = 3.5 POP/s, the
 * vendor int8 GEMM of hipBLASLt reaches 2.0-2.1 POP/s on the same problem), and the L2 -> LDS
 * stream alone accounts for 15-20 % of the clock.
 *
 * Requirements (checked by gemm256_supported): 16-byte aligned activations with
 * group_input_channels % 16 == 0 and pixel stride % 16 == 0.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "igemm_epilogue.hip.h"
#include "igemm_params.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kBN = 256;
constexpr int kBK = 64;                        // bytes of K per tile (two 32-deep MFMA sub-steps)
constexpr int kWTile = kBN * kBK;              // 16 KiB
constexpr uint32_t kFlip = 0x80808080u;

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#ifndef QNNP_DMA_AUX
#define QNNP_DMA_AUX 0
#endif
__device__ __forceinline__ void dma16(const uint8_t* src, uint8_t* lds_wave_base)
{
  // 16 bytes per lane, LDS destination = wave-uniform base + lane * 16
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*) src,
      (__attribute__((address_space(3))) void*) lds_wave_base, 16, 0, QNNP_DMA_AUX);
}

/* a wave-uniform pointer, in scalar registers for good (a uniform value the compiler happened to compute with vector
 * instructions -- a 64-bit multiply, say -- would reach an "s" asm operand as a VGPR pair: an assembler error) */
__device__ __forceinline__ const uint8_t* scalar_ptr(const uint8_t* ptr)
{
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}

/* The saddr form: 64-bit wave-uniform base in an SGPR pair + 32-bit lane offset. hipcc selects the VGPR-pair form for
 * the builtin whatever the shape of the address expression (one v_lshl_add_u64 per piece), hence the instruction itself;
 * m0 = LDS destination of lane 0, as the builtin sets it. (m0 is a reserved register: hipcc ignores it in a clobber list.
 * Nothing else in the lean kernels touches it -- tests/test_kernel_resources.py disassembles them and checks.) */
__device__ __forceinline__ void dma16_saddr(const uint8_t* base, uint32_t lane_offset, uint8_t* lds_wave_base)
{
  const uint32_t lds_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) uint8_t*) lds_wave_base));
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(lane_offset), "s"(base), "s"(lds_addr));
}
/* The same in two halves, for the main loop: m0 is written one MFMA ahead of the load, which is the wait state the
 * pair needs (no s_nop). Nothing else in that loop touches m0. */
__device__ __forceinline__ void dma16_set_m0(uint8_t* lds_wave_base)
{
  const uint32_t lds_addr = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) uint8_t*) lds_wave_base));
  asm volatile("s_mov_b32 m0, %0" : : "s"(lds_addr));
}
__device__ __forceinline__ void dma16_saddr_m0_set(const uint8_t* base, uint32_t lane_offset)
{
  asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(lane_offset), "s"(base));
}

constexpr int kWN = 2;                         // waves along channels (each 128 channels = 4 MFMA tiles)
constexpr int kTN = kBN / (kWN * 32);
static_assert(kWN == 2 && kBK == 64, "row-sum split: each channel-wave owns one of the two K sub-steps");

// ABL: measurement-only ablation mask (builds with -DQNNP_ENABLE_ABLATION, env QNNP_GFX950_ABLATE);
// 0 in the product. 1 = no requantization in the epilogue, 2 = no recentring / row sums,
// 4 = no MFMA, 8 = no LDS-DMA after the prologue, 16 = no fragment reads after the first tile,
// 32 = no per-tile wait + barrier; 64 = experiment: static s_setprio 1 for the younger half of the workgroup
// (waves 4-7) before the main loop (guide T5, static form).
// WM = waves along rows, BM = rows of the workgroup's tile:
//   WM 4, BM 256 -> 8 waves (two per SIMD), 64 x 128 outputs per wave, 12 fragment reads per 16 MFMAs, 4-slot ring;
//   WM 2, BM 256 -> 4 waves (one per SIMD, the whole register file), 128 x 128 outputs per wave,
//                   16 fragment reads per 32 MFMAs;
//   WM 2, BM 128 -> 4 waves of 64 x 128 outputs, 128 x 256 tile, 3-slot ring of 24 KiB stages = 72 KiB, so TWO
//                   workgroups share a CU: one's barriers, prologue and epilogue run under the other's multiplies
//                   (the per-wave loop is the 8-wave flavour's; the L2 -> LDS stream per MFMA grows by half).
//   PP (WM 4, BM 256 only): the PING-PONG schedule. The two waves of a SIMD (wave w and w + 4) take turns: one issues
//                   nothing but the 8 MFMAs of a K sub-step (s_setprio 1) while the other reads its next fragments,
//                   issues its LDS-DMA pieces and does its re-centring / row-sum VALU work; a raw s_barrier swaps the
//                   roles. The DMA issue stalls (100-185 cycles per piece on a busy CU) and the LDS waits then sit in a
//                   phase whose wave has no MFMA to issue, beside a partner that has nothing else. One fragment set.
//   LEAN (plain GEMM, WM 4, BM 256 only; the launcher checks K % 64 == 0, N % 256 == 0 and the 32-bit offset ranges):
//                   the same schedule with fewer instructions between the MFMAs. LDS-DMA sources are a wave-uniform
//                   64-bit base (SALU: one add-with-carry per K tile) plus a loop-invariant 32-bit lane offset -- the
//                   saddr form of global_load_lds -- instead of per-piece 64-bit VALU adds and the K / N padding
//                   selects; the steady state is unrolled over the four ring slots, so fragment and DMA destination
//                   addresses are loop-invariant registers plus immediates.
template <bool IS_CONV, int WM, int BM = 256, int ABL = 0, bool PP = false, bool LEAN = false>
__global__ __launch_bounds__(WM * kWN * 64, (WM == 4 || BM == 128) ? 2 : 1)
void q8_gemm_mfma_256x256_kernel(const IgemmParams p)
{
  static_assert(!PP || (WM == 4 && BM == 256), "ping-pong schedule: 8 waves, 256 x 256 tile");
  static_assert(!LEAN || (!IS_CONV && !PP && BM == 256 && ABL == 0), "lean flavour: plain GEMM, 256 x 256 tiles");
  constexpr int kBM = BM;
  constexpr int kStages = BM == 256 ? 4 : 3;     // LDS ring: tile t + kStages - 1 is being fetched while tile t is multiplied
  constexpr int kATile = kBM * kBK;              // 16 / 8 KiB
  constexpr int kStage = kATile + kWTile;        // 32 / 24 KiB
  constexpr int kWM = WM;
  constexpr int kThreads = WM * kWN * 64;
  constexpr int kTM = kBM / (kWM * 32);          // MFMA tiles of 32 rows per wave
  constexpr int kAChunks = (kBM * 4) / kThreads; // 16-byte activation chunks per thread per tile
  constexpr int kWFrags = 16 / (kWM * kWN);      // 1 KiB weight fragments per wave per tile
  constexpr int kDma = kAChunks + kWFrags;       // LDS-DMA instructions per thread per tile
  constexpr int kMma = kTM * kTN;                // MFMAs per K sub-step
  constexpr int kParts = 2 * kTM;                // recentring parts per fragment set
  static_assert(kDma % 2 == 0 && kDma * (kStages - 2) < 64, "the DMA of a tile is issued in two halves; vmcnt is 6 bits");

  // single LDS object: ring of {A, W} tiles, then kWN x 256 partial row sums
  __shared__ __attribute__((aligned(16))) uint8_t lds[kStages * kStage + kWN * kBM * 4];
  int32_t* lds_rowsum = reinterpret_cast<int32_t*>(lds + kStages * kStage);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave / kWN;      // row slice of this wave
  const uint32_t wn = wave % kWN;      // 0..1: 128-channel half
  const uint32_t g = blockIdx.y;

  // Workgroup -> tile: hardware places block b on XCD b % 8, so give each XCD a contiguous run of
  // logical ids, and walk logical ids in bands of 4 row-tiles (channel-tile fastest inside a band)
  // so the ~32 co-resident tiles of an XCD form a compact patch that shares row panels and weight
  // panels in that XCD's L2.
  QNNP_TRACE(p, blockIdx.x, 0, 0);
  QNNP_TRACE_WALL(p, blockIdx.x, 3, 0);
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  uint32_t m_tile, n_tile;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    const uint32_t logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr uint32_t kBand = 1024 / kBM;      // an XCD's 32 co-resident 256-row (64 128-row) tiles: 1024 rows x 2048 channels
    const uint32_t per_band = kBand * tiles_n;
    const uint32_t band = logical / per_band;
    const uint32_t within = logical - band * per_band;
    const uint32_t rows_in_band = min(kBand, tiles_m - band * kBand);
    m_tile = band * kBand + within % rows_in_band;
    n_tile = within / rows_in_band;
  }

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t ktiles = p.k_pad / kBK;                                 // k_pad is a multiple of 64
  const uint8_t* pad_k = p.fill_table + 0x80 * 16;                       // a' == 0
  const uint8_t* pad_zp = p.fill_table + (p.izp_fill & 0xFFu) * 16;      // a == input zero point
  const uint8_t* pad_w = p.fill_table;                                   // w' == 0

  // ---- activation DMA assignment: kAChunks chunks per thread, LDS linear index L = i*kThreads + tid ----
  // LDS image of an activation tile: [256 rows][4 chunks of 16 B], chunk slot s of row r holds logical
  // chunk s ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragment reads (4 rows share a 256-byte bank row).
  const uint8_t* a_row[kAChunks];   // gemm: row base (+ group); conv: image base (+ group)
  const int32_t* a_offs[kAChunks];  // conv: offset-table row of this pixel
  uint32_t a_chunk[kAChunks];       // logical 16-byte chunk (0..3) this lane fetches for its slot
#pragma unroll
  for (int i = 0; i < kAChunks; i++) {
    const uint32_t L = i * kThreads + tid;
    const uint32_t r = L >> 2;
    const uint32_t s = L & 3u;
    a_chunk[i] = s ^ ((r >> 2) & 3u);
    uint32_t m = m_tile * kBM + r;
    if (m >= p.rows) m = p.rows - 1;           // clamp: results of those rows are never stored
    if constexpr (IS_CONV) {
      const uint32_t img = m / p.rows_per_image;
      const uint32_t pix = m - img * p.rows_per_image;
      a_row[i] = p.input + static_cast<uint64_t>(img) * p.image_stride + static_cast<uint64_t>(g) * p.kc;
      a_offs[i] = p.offsets + static_cast<uint64_t>(pix) * p.ks;
    } else {
      a_row[i] = p.input + static_cast<uint64_t>(m) * p.input_stride + static_cast<uint64_t>(g) * p.kc;
      a_offs[i] = nullptr;
    }
  }

  // ---- weight DMA assignment: 16 fragments of 1 KiB per tile, kWFrags per wave ----
  const uint32_t nb0 = n_tile * (kBN / 32);
  const uint8_t* w_group = reinterpret_cast<const uint8_t*>(p.packed_w) +
      static_cast<uint64_t>(g) * nblocks * kblocks * 1024 + lane * 16;

  const uint8_t* w_src0[kWFrags];   // source of this wave's fragment i in K tile 0 ...
  uint32_t w_kstep[kWFrags];        // ... and its advance per K tile (0 when the block is N padding)
#pragma unroll
  for (int i = 0; i < kWFrags; i++) {
    const uint32_t F = i * (kWM * kWN) + wave;
    const uint32_t nb = nb0 + (F >> 1);
    const bool valid = nb < nblocks;
    w_src0[i] = valid ? w_group + (static_cast<uint64_t>(nb) * kblocks + (F & 1u)) * 1024 : pad_w;
    w_kstep[i] = valid ? 2048u : 0u;
  }

  // lean flavour: wave-uniform bases + loop-invariant 32-bit lane offsets (every K tile and channel block exists)
  const uint8_t* a_base = nullptr;
  const uint8_t* w_base = nullptr;
  uint32_t a_voff[kAChunks];
  uint32_t w_voff[kWFrags];
  if constexpr (LEAN) {
    a_base = scalar_ptr(p.input + static_cast<uint64_t>(m_tile * kBM) * p.input_stride + static_cast<uint64_t>(g) * p.kc);
#pragma unroll
    for (int i = 0; i < kAChunks; i++) {
      const uint32_t r = (i * kThreads + tid) >> 2;
      uint32_t m = m_tile * kBM + r;
      if (m >= p.rows) m = p.rows - 1;
      a_voff[i] = (m - m_tile * kBM) * p.input_stride + a_chunk[i] * 16;
    }
    // fragment F = i * 8 + wave: channel block nb0 + i * 4 + (wave >> 1), K block (wave & 1) of the tile's two
    w_base = scalar_ptr(reinterpret_cast<const uint8_t*>(p.packed_w) + static_cast<uint64_t>(g) * nblocks * kblocks * 1024 +
        (static_cast<uint64_t>(nb0 + (wave >> 1)) * kblocks + (wave & 1u)) * 1024);
#pragma unroll
    for (int i = 0; i < kWFrags; i++) w_voff[i] = lane * 16 + static_cast<uint32_t>(i) * (kWM * kWN / 2) * kblocks * 1024;
  }

  // One K tile = exactly kDma LDS-DMA instructions per thread (the vmcnt arithmetic depends on it):
  // pieces 0..kAChunks-1 = activation chunks, the rest = weight fragments. `slot` = kt % kStages (a literal in the
  // unrolled steady state of the lean flavour).
  auto stage_piece = [&](uint32_t kt, int piece, uint32_t slot) __attribute__((always_inline)) {
    uint8_t* a_dst = lds + slot * kStage;
    uint8_t* w_dst = a_dst + kATile;
    if constexpr (LEAN) {
      if (piece < kAChunks) {
        const uint8_t* tile = a_base + static_cast<uint64_t>(kt) * kBK;            // scalar
        dma16_saddr(tile, a_voff[piece], a_dst + (piece * kThreads + wave * 64) * 16);
      } else {
        const int i = piece - kAChunks;
        const uint32_t F = i * (kWM * kWN) + wave;
        const uint8_t* tile = w_base + static_cast<uint64_t>(kt) * 2048;           // scalar
        dma16_saddr(tile, w_voff[i], w_dst + F * 1024);
      }
      return;
    }
    if (piece < kAChunks) {
      const int i = piece;
      const uint32_t kk = kt * kBK + a_chunk[i] * 16;
      const uint8_t* src;
      if constexpr (IS_CONV) {
        src = pad_k;
        if (kk < p.k_total) {
          const uint32_t tap = kk / p.kc;
          const uint32_t ch = kk - tap * p.kc;
          const int32_t off = a_offs[i][tap];
          src = off >= 0 ? a_row[i] + off + ch : pad_zp;
        }
      } else {
        src = kk < p.k_total ? a_row[i] + kk : pad_k;
      }
      dma16(src, a_dst + (i * kThreads + wave * 64) * 16);
    } else {
      const int i = piece - kAChunks;
      const uint32_t F = i * (kWM * kWN) + wave;   // fragment slot: (32-channel block, 32-deep K block)
      const uint8_t* src = w_src0[i] + static_cast<uint64_t>(kt) * w_kstep[i];   // branch-free: step 0 on the zero chunk
      dma16(src, w_dst + F * 1024);
    }
  };
  auto stage = [&](uint32_t kt) {
#pragma unroll
    for (int piece = 0; piece < kDma; piece++) stage_piece(kt, piece, kt % kStages);
  };
  // lean flavour, main loop: stage_piece in two halves around an MFMA (dma16_set_m0)
  auto piece_m0 = [&](int piece, uint32_t slot) __attribute__((always_inline)) {
    uint8_t* a_dst = lds + slot * kStage;
    if (piece < kAChunks) dma16_set_m0(a_dst + (piece * kThreads + wave * 64) * 16);
    else dma16_set_m0(a_dst + kATile + ((piece - kAChunks) * (kWM * kWN) + wave) * 1024);
  };
  auto piece_load = [&](uint32_t kt, int piece) __attribute__((always_inline)) {
    if (piece < kAChunks) dma16_saddr_m0_set(a_base + static_cast<uint64_t>(kt) * kBK, a_voff[piece]);
    else dma16_saddr_m0_set(w_base + static_cast<uint64_t>(kt) * 2048, w_voff[piece - kAChunks]);
  };

  // Accumulators start at the folded bias (the MFMAs add into them): the loads ride under the prologue's DMA
  // wait instead of sitting exposed after the main loop, and the epilogue saves one add per value.
  // (lean flavour, whole-line stores: where the lane forms of the requantization apply, from the bias + 2^31 half of
  //  the pair table, and the epilogue spends no add on the row term either -- requant.hip.h)
  bool lane_rq = false;
  if constexpr (LEAN) {
    lane_rq = p.store_mode == 2 && p.bias2u != nullptr && (p.lane.kind == 1 || (p.lane.kind == 2 && p.rq.full_range != 0));
  }
  const int32_t* bias_tab = lane_rq ? p.bias2u : p.bias2;
  v16i acc[kTM][kTN];
#pragma unroll
  for (int tn = 0; tn < kTN; tn++) {
    uint32_t nbb = n_tile * (kBN / 32) + wn * kTN + tn;
    if (nbb >= nblocks) nbb = nblocks - 1;       // clamped blocks are never stored
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const uint32_t ncol = nbb * 32 + rg * 8 + (lane >> 5) * 4;
      const int4 b = *reinterpret_cast<const int4*>(bias_tab + static_cast<uint64_t>(g) * p.n_pad + ncol);
#pragma unroll
      for (int tm = 0; tm < kTM; tm++) {
        acc[tm][tn][rg * 4 + 0] = b.x;
        acc[tm][tn][rg * 4 + 1] = b.y;
        acc[tm][tn][rg * 4 + 2] = b.z;
        acc[tm][tn][rg * 4 + 3] = b.w;
      }
    }
  }
  uint32_t rs[kTM];
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) rs[tm] = 0;

  const uint32_t frag_row0 = wm * (kTM * 32) + (lane & 31u);
  const uint32_t frag_khalf = lane >> 5;
  // swizzled activation fragment address: row*64 + (((ksub*2 + khalf) ^ ((row >> 2) & 3)) << 4)
  //                                     = a_fbase ^ (ksub << 5)
  uint32_t a_fbase[kTM];
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
    const uint32_t row = frag_row0 + tm * 32;
    a_fbase[tm] = row * kBK + ((frag_khalf ^ ((row >> 2) & 3u)) << 4);
  }
  const uint32_t w_fbase = kATile + (wn * kTN * 2) * 1024 + lane * 16;   // + (tn*2 + ksub)*1024

  struct Frags {
    v4i a[kTM];
    v4i w[kTN];
  };
  // activation fragments first: they are recentred (VALU) before the weight fragments are needed
  auto read_frags = [&](const uint8_t* st, uint32_t ksub, Frags& f) __attribute__((always_inline)) {
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      f.a[tm] = *reinterpret_cast<const v4i*>(st + (a_fbase[tm] ^ (ksub << 5)));
    }
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      f.w[tn] = *reinterpret_cast<const v4i*>(st + w_fbase + (tn * 2 + ksub) * 1024);
    }
  };
  // lean flavour, ring slot known at compile time: loop-invariant address registers (the ds_read immediate reaches
  // 64 KiB, so one register per half of the ring) + immediates; sub 0 = this wave's first K sub-step, 1 = its second
  uint32_t a_off[2][kTM][2];
  uint32_t w_off[2][2];
  if constexpr (LEAN) {
#pragma unroll
    for (int sub = 0; sub < 2; sub++) {
      const uint32_t ksub = sub == 0 ? wn : (wn ^ 1u);
#pragma unroll
      for (int hi = 0; hi < 2; hi++) {
#pragma unroll
        for (int tm = 0; tm < kTM; tm++) {
          a_off[sub][tm][hi] = (a_fbase[tm] ^ (ksub << 5)) + hi * 2 * kStage;
          asm volatile("" : "+v"(a_off[sub][tm][hi]));
        }
        w_off[sub][hi] = w_fbase + ksub * 1024 + hi * 2 * kStage;
        asm volatile("" : "+v"(w_off[sub][hi]));
      }
    }
  }
  auto read_frags_slot = [&](uint32_t slot, int sub, Frags& f) __attribute__((always_inline)) {
    const uint32_t hi = slot >> 1, imm = (slot & 1u) * kStage;
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      f.a[tm] = *reinterpret_cast<const v4i*>(lds + a_off[sub][tm][hi] + imm);
    }
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      f.w[tn] = *reinterpret_cast<const v4i*>(lds + w_off[sub][hi] + imm + tn * 2048);
    }
  };
  auto flip = [&](Frags& f) {
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      f.a[tm].x ^= static_cast<int>(kFlip);
      f.a[tm].y ^= static_cast<int>(kFlip);
      f.a[tm].z ^= static_cast<int>(kFlip);
      f.a[tm].w ^= static_cast<int>(kFlip);
    }
  };
  // Row sums are taken over the RAW uint8 bytes (before recentring) with v_sad_u8 -- |a - 0| summed over the
  // 4 bytes of a dword plus an accumulator, one plain VALU op; v_dot4c beside MFMAs costs ~7 cycles each on
  // the matrix pipe (tools/ubench_gap.hip). sum(a') = sum(a) - 128 * k_pad is applied once in the epilogue
  // (K padding is staged as a = 0x80, i.e. a' = 0, so it is covered by the same constant).
  auto rowsum = [&](const Frags& f) {
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      uint32_t s = rs[tm];
      s = __builtin_amdgcn_sad_u8(f.a[tm].x, 0u, s);
      s = __builtin_amdgcn_sad_u8(f.a[tm].y, 0u, s);
      s = __builtin_amdgcn_sad_u8(f.a[tm].z, 0u, s);
      s = __builtin_amdgcn_sad_u8(f.a[tm].w, 0u, s);
      rs[tm] = s;
    }
  };
  auto mma = [&](const Frags& f, int i) {       // i = 0..kMma-1 -> (tm, tn)
    const int tm = i / kTN, tn = i % kTN;
    acc[tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[tn], f.a[tm], acc[tm][tn], 0, 0, 0);
  };
  // half h (0..3) of the recentring of one fragment set: 2 of its 8 dwords
  // (the empty asm makes the result opaque HERE: without it the compiler sinks the v_xor to the
  //  MFMA that consumes it, a barrier later, and the LDS wait lands on the critical path)
  auto flip_part = [&](Frags& f, int h) {
    const int tm = h >> 1;
    if (h & 1) {
      f.a[tm].z ^= static_cast<int>(kFlip);
      f.a[tm].w ^= static_cast<int>(kFlip);
      asm volatile("" : "+v"(f.a[tm].z), "+v"(f.a[tm].w));
    } else {
      f.a[tm].x ^= static_cast<int>(kFlip);
      f.a[tm].y ^= static_cast<int>(kFlip);
      asm volatile("" : "+v"(f.a[tm].x), "+v"(f.a[tm].y));
    }
  };
  // forces the LDS wait for a fragment set's weight registers to THIS point (where they have long
  // landed) instead of in front of their first MFMA, behind the next set's freshly issued reads
  auto settle_w = [&](Frags& f) {
    asm volatile("" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]));
  };
  auto rowsum_part = [&](const Frags& f, int h) {     // on the RAW bytes: call BEFORE flip_part(f, h)
    const int tm = h >> 1;
    if (h & 1) {
      rs[tm] = __builtin_amdgcn_sad_u8(f.a[tm].z, 0u, rs[tm]);
      rs[tm] = __builtin_amdgcn_sad_u8(f.a[tm].w, 0u, rs[tm]);
    } else {
      rs[tm] = __builtin_amdgcn_sad_u8(f.a[tm].x, 0u, rs[tm]);
      rs[tm] = __builtin_amdgcn_sad_u8(f.a[tm].y, 0u, rs[tm]);
    }
  };
#define QNNP_PIN() __builtin_amdgcn_sched_barrier(0)

  // Counted wait: tile `kt` has landed when at most the LDS-DMA groups of the tiles issued after it
  // (kDma instructions each, completing in issue order) are still outstanding.
  auto wait_tile = [&](uint32_t later_tiles_in_flight) {
    if (kStages > 3 && later_tiles_in_flight >= 2) {
      wait_vmcnt<2 * kDma>();
    } else if (later_tiles_in_flight >= 1) {
      wait_vmcnt<kDma>();
    } else {
      wait_vmcnt<0>();
    }
  };

  // Each wave walks the two K sub-steps of a tile starting at sub-step wn (integer sums commute),
  // so the sub-step whose row sums it owns is always its FIRST one: the split costs no branch.
  const uint32_t s_first = wn;
  const uint32_t s_second = wn ^ 1u;

  if constexpr (PP) {
    // ---- ping-pong main loop. Interval j lies between barriers j and j + 1; group 0 (waves 0-3, one per SIMD) reads
    // unit u in interval 2u and multiplies it in 2u + 1, group 1 (waves 4-7) runs one interval behind. A unit is one
    // K sub-step of a tile: tile kt = u / 2. Tile kt is last read in interval 4 kt + 3 (group 1), so its ring slot is
    // free from barrier 4 kt + 4 on: the read phases of tile kt + 1 (intervals >= 4 kt + 4) fetch tile kt + 4 ... no:
    // they fetch tile (kt + 1) + 3 into slot kt % 4. The wait that retires tile kt + 1 sits at the end of the read
    // phase of (kt, second sub-step), in front of a barrier every reader of tile kt + 1 passes first.
    const uint32_t group = wave >> 2;
#pragma unroll
    for (int t = 0; t < kStages - 1; t++) {
      if (static_cast<uint32_t>(t) < ktiles) stage(t);
    }
    wait_tile(min(static_cast<uint32_t>(kStages - 2), ktiles - 1));
    __builtin_amdgcn_s_barrier();
    if (group == 1u) __builtin_amdgcn_s_barrier();          // stagger: group 1 starts one interval late
    Frags f;
    auto read_phase = [&](auto fetch_c, uint32_t kt, int half) __attribute__((always_inline)) {
      constexpr bool FETCH = decltype(fetch_c)::value;
      const uint8_t* st = lds + (kt % kStages) * kStage;
      QNNP_PIN();
      read_frags(st, half == 0 ? s_first : s_second, f);
      QNNP_PIN();
      if constexpr (FETCH) {
#pragma unroll
        for (int i = 0; i < kDma / 2; i++) stage_piece(kt + kStages - 1, half * (kDma / 2) + i, (kt + kStages - 1) % kStages);
        QNNP_PIN();
      }
      if (half == 0) rowsum(f);                               // this wave owns the row sums of its first sub-step
      flip(f);
      asm volatile("" : "+v"(f.a[0]), "+v"(f.a[1]));
      settle_w(f);                                            // the LDS wait for the weight fragments belongs HERE
      QNNP_PIN();
    };
    auto multiply_phase = [&]() __attribute__((always_inline)) {
      QNNP_PIN();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < kMma; i++) mma(f, i);
      __builtin_amdgcn_s_setprio(0);
      QNNP_PIN();
    };
    uint32_t kt = 0;
    for (; kt + kStages - 1 < ktiles; kt++) {                 // steady state: tile kt + 3 exists
      read_phase(std::true_type{}, kt, 0);
      __builtin_amdgcn_s_barrier();
      multiply_phase();
      __builtin_amdgcn_s_barrier();
      read_phase(std::true_type{}, kt, 1);
      wait_vmcnt<(kStages - 2) * kDma>();                     // tile kt + 1 resident (kt + 2, kt + 3 may be in flight)
      __builtin_amdgcn_s_barrier();
      multiply_phase();
      __builtin_amdgcn_s_barrier();
    }
    for (; kt < ktiles; kt++) {                               // drain: nothing left to fetch
      read_phase(std::false_type{}, kt, 0);
      __builtin_amdgcn_s_barrier();
      multiply_phase();
      __builtin_amdgcn_s_barrier();
      read_phase(std::false_type{}, kt, 1);
      if (kt + 1 < ktiles) wait_tile(ktiles - 1 - (kt + 1));
      __builtin_amdgcn_s_barrier();
      multiply_phase();
      if (!(kt + 1 == ktiles && group == 1u)) __builtin_amdgcn_s_barrier();   // group 1 made one extra at the start
    }
  } else {
  // ---- prologue: tiles 0..2 in flight, tile 0 resident, its first fragments recentred ----
#pragma unroll
  for (int t = 0; t < kStages - 1; t++) {
    if (static_cast<uint32_t>(t) < ktiles) stage(t);
  }
  wait_tile(min(static_cast<uint32_t>(kStages - 2), ktiles - 1));
  __builtin_amdgcn_s_barrier();
  Frags fa, fb;
  read_frags(lds, s_first, fa);
  if (!(ABL & 2)) {
    rowsum(fa);
    flip(fa);
  }
  settle_w(fa);
  QNNP_TRACE(p, blockIdx.x, 0, 1);

  /*
   * Software pipeline: two phases per K tile (one per 32-deep K sub-step), every phase = kMma MFMAs on one
   * fragment set while the other set is read from LDS and recentred:
   *   phase 1: multiply fa = (tile kt, first sub-step)  | read + recentre fb = (tile kt, second sub-step)
   *   -- counted vmcnt + raw barrier: tile kt+1 resident, ring slot of tile kt free for the DMA --
   *   phase 2: multiply fb                               | read + recentre + row-sum fa = (tile kt+1, first sub-step)
   * A __syncthreads() would drain the DMA queue (vmcnt(0)); the raw barrier keeps two tiles in flight.
   * Details that matter (each measured, tools/ubench_gap.hip and the ablation build):
   *  - the LDS-DMA of a tile is spread over BOTH phases (half the pieces each, one piece every few
   *    MFMAs) instead of one burst in which all waves of the CU queue on the texture-address path;
   *  - the ring is used to its full depth: slot kt%4 is free from the mid-tile barrier of tile kt on, so
   *    phase 2 of tile kt already fetches tile kt+4 (first half) and phase 1 of tile kt+1 the second half;
   *  - activation fragments are read first and recentred in the shadow of the LAST MFMAs of a phase, so
   *    the LDS wait sits a whole phase after the reads were issued.
   * P1F: phase 1 issues the second half of tile kt+3; MORE: tile kt+1 exists; P2F: tile kt+4 exists.
   * (With the 3-slot ring of the 128-row flavour read kt+2 / kt+3: the distances are kStages - 1 and kStages.)
   */
  constexpr int kHalf = kDma / 2;                 // LDS-DMA pieces per phase
  constexpr int kDmaGap = kMma / kHalf;           // one piece every kDmaGap MFMAs
  constexpr int kFlipMma = kParts / 2;            // the last kFlipMma MFMAs of a phase carry 2 recentring parts each
  auto iteration = [&](auto p1f_c, auto more_c, auto p2f_c, uint32_t kt, uint32_t slot, auto known_c) __attribute__((always_inline)) {
    constexpr bool KNOWN = decltype(known_c)::value;     // `slot` is a literal (lean flavour's unrolled steady state)
    constexpr bool P1F = decltype(p1f_c)::value;
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool P2F = decltype(p2f_c)::value;
    const uint8_t* st = lds + slot * kStage;              // slot == kt % kStages

    QNNP_PIN();
    if constexpr (KNOWN) read_frags_slot(slot, 1, fb);
    else if (!(ABL & 16)) read_frags(st, s_second, fb);
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      if constexpr (LEAN && P1F) {
        if (i % kDmaGap == 0 && i / kDmaGap < kHalf) piece_m0(kHalf + i / kDmaGap, (slot + kStages - 1) % kStages);
        QNNP_PIN();
      }
      if (!(ABL & 4)) mma(fa, i);
      QNNP_PIN();
      if constexpr (LEAN && P1F) {
        if (i % kDmaGap == 0 && i / kDmaGap < kHalf) piece_load(kt + kStages - 1, kHalf + i / kDmaGap);
        QNNP_PIN();
      } else if constexpr (P1F) {
        if (i % kDmaGap == 0 && i / kDmaGap < kHalf && !(ABL & 8)) stage_piece(kt + kStages - 1, kHalf + i / kDmaGap, (slot + kStages - 1) % kStages);
        QNNP_PIN();
      }
      if (i >= kMma - kFlipMma) {
        const int h = (i - (kMma - kFlipMma)) * 2;
        if constexpr (LEAN) {
          if (h == 0) { __builtin_amdgcn_s_waitcnt(0xC07F); QNNP_PIN(); }   // lgkmcnt(0) once: the reads were issued a phase ago
        }
        if (!(ABL & 2)) {
          flip_part(fb, h);
          flip_part(fb, h + 1);
        }
        QNNP_PIN();
      }
    }
    if (!(ABL & 16)) settle_w(fb);
    QNNP_PIN();
#ifdef QNNP_ENABLE_ABLATION
    constexpr bool kStamp = P1F && MORE && P2F;
    const bool stamp = kStamp && (kt == 20 || kt == 21);
    if (stamp) QNNP_TRACE_WAVE(p, 1024 + blockIdx.x, wave, (kt - 20) * 4 + 0);
    QNNP_PIN();
#endif

    if constexpr (MORE) {
      // tile kt+1 must be resident; issued after it so far: tiles kt+2 .. min(kt+3, ktiles-1), complete
      if constexpr (P1F || P2F) {
        wait_vmcnt<(kStages - 2) * kDma>();
#ifdef QNNP_ENABLE_ABLATION
        QNNP_PIN();
        if (stamp) QNNP_TRACE_WAVE(p, 1024 + blockIdx.x, wave, (kt - 20) * 4 + 1);
        QNNP_PIN();
#endif
      } else {
        const uint32_t last = min(kt + kStages - 1, ktiles - 1);
        wait_tile(last - (kt + 1));
      }
      __builtin_amdgcn_s_barrier();
    }
    QNNP_PIN();
#ifdef QNNP_ENABLE_ABLATION
    if (stamp) QNNP_TRACE_WAVE(p, 1024 + blockIdx.x, wave, (kt - 20) * 4 + 2);
    QNNP_PIN();
#endif

    if constexpr (MORE) {
      if constexpr (KNOWN) read_frags_slot((slot + 1) % kStages, 0, fa);
      else if (!(ABL & 16)) read_frags(lds + ((slot + 1) % kStages) * kStage, s_first, fa);
    }
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      if constexpr (LEAN && P2F) {
        if (i % kDmaGap == 0 && i / kDmaGap < kHalf) piece_m0(i / kDmaGap, slot);
        QNNP_PIN();
      }
      if (!(ABL & 4)) mma(fb, i);
      QNNP_PIN();
      if constexpr (LEAN && P2F) {
        if (i % kDmaGap == 0 && i / kDmaGap < kHalf) piece_load(kt + kStages, i / kDmaGap);
        QNNP_PIN();
      } else if constexpr (P2F) {
        if (i % kDmaGap == 0 && i / kDmaGap < kHalf && !(ABL & 8)) stage_piece(kt + kStages, i / kDmaGap, slot);
        QNNP_PIN();
      }
      if constexpr (MORE) {
        if (i >= kMma - kFlipMma) {
          const int h = (i - (kMma - kFlipMma)) * 2;
          if constexpr (LEAN) {
            if (h == 0) { __builtin_amdgcn_s_waitcnt(0xC07F); QNNP_PIN(); }
          }
          if (!(ABL & 2)) {
            rowsum_part(fa, h);
            flip_part(fa, h);
            rowsum_part(fa, h + 1);
            flip_part(fa, h + 1);
          }
          QNNP_PIN();
        }
      }
    }
    if constexpr (MORE) { if (!(ABL & 16)) settle_w(fa); }
    QNNP_PIN();
#ifdef QNNP_ENABLE_ABLATION
    if (stamp) QNNP_TRACE_WAVE(p, 1024 + blockIdx.x, wave, (kt - 20) * 4 + 3);
    QNNP_PIN();
#endif
  };

  // (the prologue above staged tiles 0..kStages-2; one more completes the ring)
  if (ktiles > static_cast<uint32_t>(kStages - 1)) stage(kStages - 1);
  if constexpr ((ABL & 64) != 0) {
    if (wave >= static_cast<uint32_t>(kWM * kWN / 2)) __builtin_amdgcn_s_setprio(1);
  }
  uint32_t kt = 0;
  if (ktiles > static_cast<uint32_t>(kStages)) {
    iteration(std::false_type{}, std::true_type{}, std::true_type{}, 0u, 0u, std::false_type{});
    kt = 1;
    if constexpr (LEAN) {
      static_assert(kStages == 4, "unrolled over the ring");
      for (; kt + 3 + kStages < ktiles; kt += 4) {         // steady state, ring slots as literals
        iteration(std::true_type{}, std::true_type{}, std::true_type{}, kt, 1u, std::true_type{});
        iteration(std::true_type{}, std::true_type{}, std::true_type{}, kt + 1, 2u, std::true_type{});
        iteration(std::true_type{}, std::true_type{}, std::true_type{}, kt + 2, 3u, std::true_type{});
        iteration(std::true_type{}, std::true_type{}, std::true_type{}, kt + 3, 0u, std::true_type{});
      }
    }
    for (; kt + kStages < ktiles; kt++) {                  // steady state
      iteration(std::true_type{}, std::true_type{}, std::true_type{}, kt, kt % kStages, std::false_type{});
    }
    iteration(std::true_type{}, std::true_type{}, std::false_type{}, kt, kt % kStages, std::false_type{});   // kt == ktiles - kStages
    kt++;
  }
  for (; kt + 1 < ktiles; kt++) {                          // drain: nothing left to fetch
    iteration(std::false_type{}, std::true_type{}, std::false_type{}, kt, kt % kStages, std::false_type{});
  }
  iteration(std::false_type{}, std::false_type{}, std::false_type{}, kt, kt % kStages, std::false_type{});   // last tile
  }  // !PP
#undef QNNP_PIN

  QNNP_TRACE(p, blockIdx.x, 0, 2);
  const int4 no_bias[4] = {};                 // (the bias is already in the accumulators)

  // ---- combine the row sums: 2 K halves (lane, lane+32) then the 2 channel-waves ----
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
    uint32_t s = rs[tm];
    s += __shfl_xor(s, 32);
    if (lane < 32) lds_rowsum[wn * kBM + wm * (kTM * 32) + tm * 32 + lane] = static_cast<int32_t>(s);
  }
  __syncthreads();
  QNNP_TRACE(p, blockIdx.x, 0, 3);

  // ---- fused epilogue (igemm_epilogue.hip.h); the requantization flavour is chosen once ----
  const uint32_t raw_to_centred = 128u * p.k_pad;      // sum(a') = sum(a) - 128 * k_pad
  if constexpr (LEAN) {
    if (lane_rq) {
      // the store_mode 2 branch below with the lane forms: the row term enters as the multiply-add's addend
      auto epilogue_lane = [&](auto seq_c, auto full_c) __attribute__((always_inline)) {
        constexpr int kSeq = decltype(seq_c)::value;
        constexpr bool kFull = decltype(full_c)::value;
        constexpr uint32_t kPitch = kTN * 32 + 16;
        uint8_t* image = lds + wave * (kTM * 32 * kPitch);
        uint64_t row_addend[kTM];
#pragma unroll
        for (int tm = 0; tm < kTM; tm++) {
          const uint32_t row = frag_row0 + tm * 32;
          row_addend[tm] = lane_addend(p.row_coeff * static_cast<int32_t>(
              static_cast<uint32_t>(lds_rowsum[row]) + static_cast<uint32_t>(lds_rowsum[kBM + row]) - raw_to_centred), p.lane);
        }
#pragma unroll
        for (int tn = 0; tn < kTN; tn++) {
#pragma unroll
          for (int tm = 0; tm < kTM; tm++) {
            igemm_stage_tile_lane<kSeq, kFull>(acc[tm][tn], row_addend[tm], image + (tm * 32 + (lane & 31u)) * kPitch, tn * 32, frag_khalf, p);
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
        const uint32_t m0 = m_tile * kBM + wm * (kTM * 32);
        const uint32_t n0 = (nb0 + wn * kTN) * 32;
        uint8_t* out0 = p.output + static_cast<uint64_t>(m0) * p.output_stride + static_cast<uint64_t>(g) * p.n + n0;
#pragma unroll
        for (int i = 0; i < (kTM * 32 * kTN * 2) / 64; i++) {
          const uint32_t idx = i * 64 + lane;
          const uint32_t r = idx / (kTN * 2);
          const uint32_t c = idx % (kTN * 2);
          const uint4 v = *reinterpret_cast<const uint4*>(image + r * kPitch + c * 16);
          if (m0 + r < p.rows && n0 + c * 16 < p.n) {
            typedef int nt_v4i __attribute__((ext_vector_type(4)));     // whole lines, written once: streaming hint
            const nt_v4i x = {static_cast<int>(v.x), static_cast<int>(v.y), static_cast<int>(v.z), static_cast<int>(v.w)};
            nt_v4i* dst = reinterpret_cast<nt_v4i*>(out0 + static_cast<uint64_t>(r) * p.output_stride + c * 16);
            if (p.stream_out) {                                    // ("streaming_stores", igemm_params.h)
              // (as an instruction: with the builtin, hipcc merges the two stores of this branch and drops the hint)
              asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(x) : "memory");
            } else {
              *dst = x;
            }
          }
        }
      };
      if (p.lane.kind == 1) {
        if (p.rq.full_range) epilogue_lane(std::integral_constant<int, kRqShift0Lane>{}, std::true_type{});
        else epilogue_lane(std::integral_constant<int, kRqShift0Lane>{}, std::false_type{});
      } else if (p.lane.shift <= QNNP_REQUANT_LANE_PK_MAX_SHIFT) {
        epilogue_lane(std::integral_constant<int, kRqBoundedLanePk>{}, std::true_type{});     // (round 5: the packed tail)
      } else {
        epilogue_lane(std::integral_constant<int, kRqBoundedLane>{}, std::true_type{});
      }
      return;
    }
  }
  requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
    // (+ 2^31 for the offset rounding sequences, requant.hip.h: rides on the row term)
    constexpr int kSeq = decltype(shift0)::value;
    if (p.store_mode == 2) {
      // 16-byte aligned rows: a wave's (kTM*32) x 128-byte sub-tile is requantized into a private image in the
      // (now idle) LDS ring and leaves as WHOLE 128-byte lines, eight rows per store instruction. Direct
      // 16-byte stores would put 32-byte partial-line writes on the L2 (2.8 TB/s against 5+ for whole lines,
      // measured on the pointwise kernel).
      constexpr uint32_t kPitch = kTN * 32 + 16;           // +16: the 8-lane ds_write_b128 groups hit distinct banks
      uint8_t* image = lds + wave * (kTM * 32 * kPitch);
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) {
        const uint32_t nb = nb0 + wn * kTN + tn;
        if (nb >= nblocks) continue;                       // wave-uniform
#pragma unroll
        for (int tm = 0; tm < kTM; tm++) {
          const uint32_t row = frag_row0 + tm * 32;
          const int32_t rowterm = with_rq_offset<kSeq>(p.row_coeff *
              static_cast<int32_t>(static_cast<uint32_t>(lds_rowsum[row]) + static_cast<uint32_t>(lds_rowsum[kBM + row]) - raw_to_centred));
          igemm_stage_tile<decltype(shift0)::value, decltype(full)::value, (ABL & 1) != 0, 2>(
              acc[tm][tn], no_bias, rowterm, image + (tm * 32 + (lane & 31u)) * kPitch, tn * 32, frag_khalf, p);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
      const uint32_t m0 = m_tile * kBM + wm * (kTM * 32);
      const uint32_t n0 = (nb0 + wn * kTN) * 32;
      uint8_t* out0 = p.output + static_cast<uint64_t>(m0) * p.output_stride + static_cast<uint64_t>(g) * p.n + n0;
#pragma unroll
      for (int i = 0; i < (kTM * 32 * kTN * 2) / 64; i++) {
        const uint32_t idx = i * 64 + lane;
        const uint32_t r = idx / (kTN * 2);
        const uint32_t c = idx % (kTN * 2);
        const uint4 v = *reinterpret_cast<const uint4*>(image + r * kPitch + c * 16);
        if (m0 + r < p.rows && n0 + c * 16 < p.n) {
          *reinterpret_cast<uint4*>(out0 + static_cast<uint64_t>(r) * p.output_stride + c * 16) = v;
        }
      }
      return;
    }
    if constexpr (kWM == 4) {
#pragma unroll
      for (int tm = 0; tm < kTM; tm++) {
        const uint32_t row = frag_row0 + tm * 32;
        const uint32_t m = m_tile * kBM + row;
        const int32_t rowterm = with_rq_offset<kSeq>(p.row_coeff *
            static_cast<int32_t>(static_cast<uint32_t>(lds_rowsum[row]) + static_cast<uint32_t>(lds_rowsum[kBM + row]) - raw_to_centred));
        uint8_t* out_row = p.output + static_cast<uint64_t>(m) * p.output_stride + static_cast<uint64_t>(g) * p.n;
#pragma unroll
        for (int tn = 0; tn < kTN; tn++) {
          const uint32_t nb = nb0 + wn * kTN + tn;
          if (nb >= nblocks) continue;       // wave-uniform
          igemm_store_tile<decltype(shift0)::value, decltype(full)::value, (ABL & 1) != 0, 2>(
              acc[tm][tn], no_bias, rowterm, out_row, nb * 32, frag_khalf, m < p.rows, p);
        }
      }
    } else {
      int32_t rowterm[kTM];
#pragma unroll
      for (int tm = 0; tm < kTM; tm++) {
        const uint32_t row = frag_row0 + tm * 32;
        rowterm[tm] = with_rq_offset<kSeq>(p.row_coeff *
            static_cast<int32_t>(static_cast<uint32_t>(lds_rowsum[row]) + static_cast<uint32_t>(lds_rowsum[kBM + row]) - raw_to_centred));
      }
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) {
        const uint32_t nb = nb0 + wn * kTN + tn;
        if (nb >= nblocks) continue;         // wave-uniform
#pragma unroll
        for (int tm = 0; tm < kTM; tm++) {
          const uint32_t m = m_tile * kBM + frag_row0 + tm * 32;
          uint8_t* out_row = p.output + static_cast<uint64_t>(m) * p.output_stride + static_cast<uint64_t>(g) * p.n;
          igemm_store_tile<decltype(shift0)::value, decltype(full)::value, (ABL & 1) != 0, 2>(
              acc[tm][tn], no_bias, rowterm[tm], out_row, nb * 32, frag_khalf, m < p.rows, p);
        }
      }
    }
  });
  QNNP_TRACE(p, blockIdx.x, 0, 4);
  QNNP_TRACE_WALL(p, blockIdx.x, 3, 1);
}

}  // namespace

bool gemm256_supported(const IgemmParams& p, uint32_t vec)
{
  return vec == 16 && p.fill_table != nullptr && p.rows >= 1 && p.k_total % 16 == 0;
}

template <bool IS_CONV, int WM, int BM = 256>
static int launch256(const IgemmParams& p, const dim3& grid, hipStream_t stream)
{
  hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<IS_CONV, WM, BM>), grid, dim3(WM * kWN * 64), 0, stream, p);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/* waves4 = false: 8 waves (two per SIMD), 64 x 128 outputs per wave -- the default;
 * waves4 = true:  4 waves (one per SIMD, the whole register file), 128 x 128 per wave: a third less LDS read
 *                 traffic and a higher sustained clock, but every issue stall is exposed (A/B flavour). */
/* lean flavour (same schedule, fewer instructions around the MFMAs): plain GEMM whose K tiles and channel blocks all
 * exist, lane offsets within 32 bits */
bool gemm256_lean_supported(const IgemmParams& p)
{
  return p.offsets == nullptr && p.k_total == p.k_pad && p.k_pad % kBK == 0 && p.n_pad % kBN == 0 &&
         p.k_pad <= (1u << 22) && static_cast<uint64_t>(p.input_stride) * 256u < (1ull << 32);
}

int gemm256_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, bool waves4, bool rows128, bool pingpong, int lean)
{
  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  const bool conv = p.offsets != nullptr;
  // 128 x 256 tiles (four waves, two workgroups per CU): lost the A/B on the 4096^3 GEMM, but a problem whose 256-row
  // tiling does not even give every CU one workgroup (ResNet's 14x14 / 7x7 layers at batch 128: 98 or 50 tiles on 256 CUs)
  // is bounded by that, not by the loop -- round 5, profiles/r05: 3x3 256 -> 256 at 14x14 43 -> 2x us
  bool underfilled = false;
#ifdef QNNP_ENABLE_ABLATION
  const char* env_fill = getenv("QNNP_GEMM_ROWS128_AUTO");
  const bool fill_auto = env_fill == nullptr || atoi(env_fill) != 0;
#else
  const bool fill_auto = true;
#endif
  // (lean == 1: the automatic choice -- a forced "gemm_kernel" keeps its tile; plain GEMMs the lean flavour takes keep it)
  if (fill_auto && !waves4 && !pingpong && !rows128 && lean == 1 && !(gemm256_lean_supported(p))) {
    const uint64_t tiles256 = static_cast<uint64_t>((p.rows + 255u) / 256u) * tiles_n * groups;
    underfilled = tiles256 < p.cu_count && p.rows > 128u;
  }
  const uint32_t bm = (rows128 || underfilled) ? 128u : 256u;
  const uint32_t tiles_m = (p.rows + bm - 1) / bm;
  const dim3 grid(tiles_m * tiles_n, groups, 1);
  if (underfilled) {
    *name = conv ? "q8_gemm_mfma_128x256_conv" : "q8_gemm_mfma_128x256";
    return conv ? launch256<true, 2, 128>(p, grid, stream) : launch256<false, 2, 128>(p, grid, stream);
  }
#ifndef QNNP_ENABLE_ABLATION
  // The structures that LOST their A/B (profiles/r03/gemm_structures_ab_r03a.txt: 4 waves of 128 x 128, forced 128 x 256
  // tiles, the ping-pong schedule) are evidence, not product: measurement builds only.
  if (waves4 || rows128 || pingpong) return QNNP_HIP_EINVAL;
#else
  if (rows128) {
    *name = conv ? "q8_gemm_mfma_128x256_conv" : "q8_gemm_mfma_128x256";
    return conv ? launch256<true, 2, 128>(p, grid, stream) : launch256<false, 2, 128>(p, grid, stream);
  }
  if (pingpong) {
    *name = conv ? "q8_gemm_mfma_256x256_pp_conv" : "q8_gemm_mfma_256x256_pp";
    if (conv) hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<true, 4, 256, 0, true>), grid, dim3(512), 0, stream, p);
    else hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<false, 4, 256, 0, true>), grid, dim3(512), 0, stream, p);
    return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  }
  if (!conv) {
    const char* env = getenv("QNNP_GFX950_ABLATE");
    const int abl = env != nullptr ? atoi(env) : 0;
    *name = waves4 ? "q8_gemm_mfma_256x256_w4" : "q8_gemm_mfma_256x256";
#define QNNP_ABL_CASE(V) case V: if (waves4) hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<false, 2, 256, V>), grid, dim3(256), 0, stream, p); \
        else hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<false, 4, 256, V>), grid, dim3(512), 0, stream, p); \
        return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
    switch (abl) {
      case 0: break;
      QNNP_ABL_CASE(1) QNNP_ABL_CASE(2) QNNP_ABL_CASE(3) QNNP_ABL_CASE(4) QNNP_ABL_CASE(8) QNNP_ABL_CASE(16) QNNP_ABL_CASE(24) QNNP_ABL_CASE(26)
      QNNP_ABL_CASE(27) QNNP_ABL_CASE(64)
      default: break;
    }
#undef QNNP_ABL_CASE
  }
  if (waves4) {
    if (lean != 0 && gemm256_lean_supported(p)) {
      *name = "q8_gemm_mfma_256x256_w4_lean";
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<false, 2, 256, 0, false, true>), grid, dim3(256), 0, stream, p);
      return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
    }
    if (lean > 1) return QNNP_HIP_EINVAL;
    *name = conv ? "q8_gemm_mfma_256x256_w4_conv" : "q8_gemm_mfma_256x256_w4";
    return conv ? launch256<true, 2>(p, grid, stream) : launch256<false, 2>(p, grid, stream);
  }
#endif
  if (lean != 0 && gemm256_lean_supported(p)) {
    *name = "q8_gemm_mfma_256x256_lean";
    hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<false, 4, 256, 0, false, true>), grid, dim3(512), 0, stream, p);
    return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  }
  if (lean > 1) return QNNP_HIP_EINVAL;        // forced and not applicable
  *name = conv ? "q8_gemm_mfma_256x256_conv" : "q8_gemm_mfma_256x256";
  return conv ? launch256<true, 4>(p, grid, stream) : launch256<false, 4>(p, grid, stream);
}

}  // namespace qnnp
