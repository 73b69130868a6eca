/*
 * q8gemm256x.hip -- the zero-point-centred 256 x 256 uint8 GEMM on v_mfma_i32_16x16x64_i8 (round 6; BASELINE.json configs[1]).
 *
 * Same role, same algebra, same weight image, same LDS-DMA ring, same tile -> XCD map as q8gemm256c.hip (it replaces
 * q8gemm_ukernel_4x4c2__sse2, reference src/q8gemm/4x4c2-sse2.c:14-318, and its tiler compute_q8gemm,
 * src/operator-run.c:39-70, 797-802, for MFMA-bound problems whose kernel zero point is 127 or 128). What changes is the
 * matrix instruction, and with it the fragment geometry and the epilogue.
 *
 * WHY. This GEMM is bound by the chip's power budget, not by issue slots (the matrix pipe is busy ~80 % of
 * the launch at ~1.4 GHz of 2.4; removing stalls returns as a lower clock). tools/ubench_mfma2.hip (profiles/r06/) measures
 * what the two int8 shapes sustain on RANDOM operands with nothing else in the loop: 32x32x32 3.41-3.47 PetaOP/s at 1.74 GHz,
 * 16x16x64 4.06-4.09 at 2.05 GHz -- the 16 x 16 shape reads and writes 4 accumulator registers per 16 K MACs (K = 64 per
 * instruction) where the 32 x 32 one moves 16 per 32 K MACs: half the accumulator traffic per MAC, +18 % rate under the same
 * power cap. LDS fragment traffic per MAC is the same for a 64 x 128 wave tile (12 ds_read_b128 per 64-byte K tile).
 *
 * Geometry. 8 waves = 4 (rows) x 2 (channels); a wave owns 64 rows x 128 channels = 4 x 8 tiles of 16 x 16, 128 accumulator
 * registers. Weights are operand A, activations operand B:
 *   operand lane l: row / channel (l & 15) of the tile, K bytes [16 g, 16 g + 16) of the 64-byte K tile, g = l >> 4
 *   result  lane l, register r: activation row (l & 15), channel 4 (l >> 4) + r  -- four consecutive channels = one output dword
 * The activation tile image in LDS stays row-major [256 rows][4 slots of 16 B] as LDS-DMA lays it (lane L of a piece -> row
 * L >> 2, slot L & 3: four consecutive lanes fetch one row's 64 contiguous bytes); slot s of row r holds K chunk s ^ f(r),
 * f(r) = 3 if r & 8 else 0, which makes the 16x16x64 fragment read (lane -> row l & 15, chunk l >> 4) conflict-free for
 * ds_read_b128's 16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MI355X_MICROARCH.md, LDS): every group touches
 * the sixteen 16-byte bank quads once. The weight image is the one pack.h makes for the 32x32x32 kernels -- 1 KiB fragments
 * [32 channels][32 K bytes], position ((c & 31) + 32 ((k & 31) >> 4)) * 16 -- read with other lane addresses: lane l of
 * 16-channel tile t takes fragment (t >> 1, g >> 1), position (16 (t & 1) + (l & 15) + 32 (g & 1)) * 16; conflict-free as well.
 *
 * Pipeline per K tile (one instruction covers the whole 64 bytes of K, so every accumulator is touched once per tile):
 *   phase 1: 16 MFMAs  a[0..3] x wl[0..3]   | read wh[0..3] (this tile) | LDS-DMA pieces 2, 3 of tile kt + RING - 1
 *   -- counted vmcnt + barrier: tile kt + 1 resident, every read of tile kt done --
 *   phase 2: 16 MFMAs  a[0..3] x wh[0..3]   | read wl[0..3], a[0..3] of tile kt + 1, a[tm] right behind its last use
 *                                           | re-centre them | LDS-DMA pieces 0, 1 of tile kt + RING
 * 48 fragment registers, no double buffer: in both phases the activation operand stays for four MFMAs and the weight operand
 * walks 0123 3210 ... (one operand changes per MFMA: the order the microbenchmark prices highest).
 *
 * Epilogue without LDS: requantize four accumulators -> one dword; two 4 x 4 dword transposes over the four 16-lane rows
 * (v_permlane32_swap + v_permlane16_swap, 8 instructions per 16-row block) give every lane 16 consecutive channels of its
 * row; one DPP row rotate by 8 under a bank mask pairs them so that a store instruction writes eight whole 128-byte lines.
 *
 * Requirements: as q8gemm256c.hip (gemm256c_supported).
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "igemm_params.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kBM = 256;
constexpr int kBN = 256;
constexpr int kBK = 64;                        // bytes of K per tile = one 16x16x64 step
constexpr int kATile = kBM * kBK;              // 16 KiB
constexpr int kWTile = kBN * kBK;              // 16 KiB
constexpr int kStage = kATile + kWTile;        // 32 KiB
constexpr int kThreads = 512;                  // 8 waves: 4 (rows) x 2 (channels), 64 x 128 outputs per wave
constexpr int kTM = 4;                         // 16-row MFMA tiles per wave
constexpr int kTN = 8;                         // 16-channel MFMA tiles per wave
constexpr int kHalf = kTN / 2;                 // weight fragments per phase
constexpr int kDma = 4;                        // LDS-DMA instructions per thread and K tile: 2 activation + 2 weight pieces
constexpr int kMma = kTM * kHalf;              // MFMAs per phase
constexpr int kRing = 4;                       // LDS stages of 32 KiB; the remaining 32 KiB hold the waves' bias lines
constexpr int kBiasArea = kRing * kStage;      // 8 waves x 512 bytes

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

/* a wave-uniform pointer, in scalar registers for good */
__device__ __forceinline__ const uint8_t* scalar_ptr(const uint8_t* ptr)
{
  const uint64_t v = reinterpret_cast<uint64_t>(ptr);
  const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v));
  const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(v >> 32));
  return reinterpret_cast<const uint8_t*>((static_cast<uint64_t>(hi) << 32) | lo);
}

__device__ __forceinline__ uint32_t lds_address(uint8_t* lds_ptr)
{
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint8_t*) lds_ptr));
}

/* LDS-DMA, saddr form: 16 bytes per lane from base + lane_offset to m0 + lane * 16 */
__device__ __forceinline__ void dma16_saddr(const uint8_t* base, uint32_t lane_offset, uint8_t* lds_wave_base)
{
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               : : "v"(lane_offset), "s"(base), "s"(lds_address(lds_wave_base)));
}
__device__ __forceinline__ void dma16_set_m0(uint8_t* lds_wave_base)
{
  asm volatile("s_mov_b32 m0, %0" : : "s"(lds_address(lds_wave_base)));
}
__device__ __forceinline__ void dma16_saddr_m0_set(const uint8_t* base, uint32_t lane_offset)
{
  asm volatile("global_load_lds_dwordx4 %0, %1" : : "v"(lane_offset), "s"(base));
}

/* chunk swizzle of the activation image: rows 8..15 of every 16 keep their K chunks in slots c ^ 3 */
__device__ __forceinline__ uint32_t a_swizzle(uint32_t row) { return (row & 8u) != 0 ? 3u : 0u; }

#define QNNP_PIN() __builtin_amdgcn_sched_barrier(0)

// measurement builds: cycle stamps of wave 0 (item 0) and wave 4 (item 1) of every workgroup; items 2, 3 = wall clock
#ifdef QNNP_ENABLE_ABLATION
#define QNNP_X_STAMP(slot)                                                                                  \
  do {                                                                                                       \
    if (p.trace != nullptr && lane == 0 && (wave & 3u) == 0)                                                 \
      p.trace[(blockIdx.x * 4 + (wave >> 2)) * 8 + (slot)] = __builtin_readcyclecounter();                  \
  } while (0)
#define QNNP_X_STAMP_WALL(slot)                                                                             \
  do {                                                                                                       \
    if (p.trace != nullptr && lane == 0 && (wave & 3u) == 0)                                                 \
      p.trace[(blockIdx.x * 4 + 2 + (wave >> 2)) * 8 + (slot)] = wall_clock64();                            \
  } while (0)
#else
#define QNNP_X_STAMP(slot) do { } while (0)
#define QNNP_X_STAMP_WALL(slot) do { } while (0)
#endif

/*
 * SEQ / CLAMP: rounding sequence and clamp class of the requantization (requant.hip.h), chosen by the launcher.
 * ALIGNED: the K tiles are a multiple of the ring (K % 256 == 0): every ring slot is a literal, the drain included.
 * ABL: measurement-only ablation mask (builds with -DQNNP_ENABLE_ABLATION, env QNNP_GFX950_ABLATE); 0 in the product.
 * 1 = no epilogue, 2 = no recentring, 4 = no MFMA, 8 = no LDS-DMA after the prologue, 16 = no fragment reads after the
 * prologue, 32 = no per-tile wait + barrier, 64 = no global stores.
 */
/* ROWSUM (round 6): any OTHER kernel zero point, on the standard image (w - 128, activations ^ 0x80): the kernel-zero-point row term
 * (128 - kzp) * sum_k (a(m,k) - 128) of q8gemm256.hip is added to the accumulators in front of the requantization. The sums are
 * taken over the RAW fragment bytes with v_sad_u8 (one per dword, beside the re-centring XOR); a lane holds chunk g of its row, the
 * four lanes of a row are summed once in the epilogue -- and the result lane of a row is its operand lane, nothing moves. */
/* Inside a phase the weight operand walks its four fragments in SNAKE order (0123 3210 0123 3210): only ONE operand changes per MFMA.
 * tools/ubench_mfma2.hip: 4.27 POP/s against 4.18 for "activation held for four, weights 0123 0123" (and 4.20 for holding it for eight);
 * this kernel, interleaved on one box: 49.96 -> 49.36 us (profiles/r06/gemm_ab_c16_snake_r06k.txt).
 * OPT (measurement builds, env QNNP_C16_OPT; 0 in the product): 1 = the plain order, for that A/B. */
template <int SEQ, int CLAMP, bool ALIGNED, int ABL = 0, bool ROWSUM = false, int OPT = 0>
__global__ __launch_bounds__(kThreads, 2)
void q8_gemm_mfma_256x256_c16_kernel(const IgemmParams p)
{
  static_assert(SEQ == kRqShift0Ofs || SEQ == kRqBoundedOfs || SEQ == kRqGeneral, "offset forms, or the general one");
  constexpr int RING = kRing;
  constexpr int kGroups = (RING + 1) / 2;        // address registers per fragment: a ds_read immediate reaches 64 KiB = 2 stages

  __shared__ __attribute__((aligned(16))) uint8_t lds[kRing * kStage + 8 * 512];    // the ONE LDS object

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave >> 1;       // 64-row slice
  const uint32_t wn = wave & 1u;       // 128-channel half
  const uint32_t g = blockIdx.y;
  QNNP_X_STAMP(0);
  QNNP_X_STAMP_WALL(0);

  // Workgroup -> tile (q8gemm256.hip): contiguous logical ids per XCD, bands of four row tiles -- an XCD's 32 workgroups of a
  // 16 x 16-tile launch cover 4 row tiles x 8 channel tiles, the most compact block 32 tiles allow (4 + 8 operand panels).
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = p.n_pad / kBN;
  uint32_t m_tile, n_tile;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    const uint32_t logical = xcd * q + min(xcd, r) + idx;
    constexpr uint32_t kBand = 4;
    const uint32_t band = p.tiles_n_magic != 0 ? __umulhi(logical >> 2, p.tiles_n_magic) : logical >> 2;   // logical / (4 * tiles_n)
    const uint32_t within = logical - band * kBand * tiles_n;
    const uint32_t rows_in_band = min(kBand, tiles_m - band * kBand);
    if (rows_in_band == kBand) {
      m_tile = band * kBand + (within & 3u);
      n_tile = within >> 2;
    } else {
      m_tile = band * kBand + within % rows_in_band;
      n_tile = within / rows_in_band;
    }
  }

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t ktiles = p.k_pad / kBK;
  const uint32_t nb0 = n_tile * (kBN / 32);

  // ---- LDS-DMA sources: wave-uniform bases + loop-invariant 32-bit lane offsets ----
  // (strided 1x1 convolutions -- igemm_params.h `offsets_dense`: a row's address comes from the operator's table, the lane
  //  offsets are then absolute; the launcher checks that the tensor ends below 2^32)
  const bool table_rows = p.offsets_dense != 0;
  const uint8_t* a_base = scalar_ptr(table_rows ? p.input + static_cast<uint64_t>(g) * p.kc
      : p.input + static_cast<uint64_t>(m_tile * kBM) * p.input_stride + static_cast<uint64_t>(g) * p.kc);
  uint32_t a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const uint32_t L = i * kThreads + tid;
    const uint32_t r = L >> 2;
    const uint32_t chunk = (L & 3u) ^ a_swizzle(r);
    uint32_t m = m_tile * kBM + r;
    if (m >= p.rows) m = p.rows - 1;             // clamp: results of those rows are never stored
    a_voff[i] = (m - m_tile * kBM) * p.input_stride + chunk * 16;
    if (table_rows) {
      const uint32_t img = p.rpi_magic != 0 ? __umulhi(m, p.rpi_magic) : m / p.rows_per_image;
      const uint32_t pix = m - img * p.rows_per_image;
      a_voff[i] = img * static_cast<uint32_t>(p.image_stride) + static_cast<uint32_t>(p.offsets[pix]) + chunk * 16;
    }
  }
  // weight fragment F = i * 8 + wave: channel block nb0 + i * 4 + (wave >> 1), K block (wave & 1) of the tile's two
  const uint8_t* w_base = scalar_ptr(reinterpret_cast<const uint8_t*>(p.packed_w) + static_cast<uint64_t>(g) * nblocks * kblocks * 1024 +
      (static_cast<uint64_t>(nb0 + (wave >> 1)) * kblocks + (wave & 1u)) * 1024);
  uint32_t w_voff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) w_voff[i] = lane * 16 + static_cast<uint32_t>(i) * 4u * kblocks * 1024u;

  auto piece_dst = [&](int piece, uint32_t slot) __attribute__((always_inline)) -> uint8_t* {
    uint8_t* a_dst = lds + slot * kStage;
    if (piece < 2) return a_dst + (piece * kThreads + wave * 64) * 16;
    return a_dst + kATile + ((piece - 2) * 8 + wave) * 1024;
  };
  auto piece_src = [&](uint32_t kt, int piece) __attribute__((always_inline)) -> const uint8_t* {
    return piece < 2 ? a_base + static_cast<uint64_t>(kt) * kBK : w_base + static_cast<uint64_t>(kt) * 2048;
  };
  auto piece_off = [&](int piece) __attribute__((always_inline)) -> uint32_t {
    return piece < 2 ? a_voff[piece] : w_voff[piece - 2];
  };
  auto stage_piece = [&](uint32_t kt, int piece, uint32_t slot) __attribute__((always_inline)) {
    dma16_saddr(piece_src(kt, piece), piece_off(piece), piece_dst(piece, slot));
  };

  // ---- prologue, part 1: the first tile's DMA, the folded bias, the rest of the ring ----
#pragma unroll
  for (int piece = 0; piece < kDma; piece++) stage_piece(0, piece, 0);

  // the wave's 128 folded biases (+ 2^31 for the offset forms): ONE LDS-DMA instruction, lanes 0..31, 512 bytes
  const int32_t* bias_tab = SEQ == kRqGeneral ? p.bias2 : p.bias2u;
  uint8_t* bias_line = lds + kBiasArea + wave * 512;
  if (lane < 32) {
    dma16_saddr(scalar_ptr(reinterpret_cast<const uint8_t*>(bias_tab + static_cast<uint64_t>(g) * p.n_pad + (nb0 + wn * 4) * 32)),
                   lane * 16, bias_line);
  }

#pragma unroll
  for (int t = 1; t < RING; t++) {
#pragma unroll
    for (int piece = 0; piece < kDma; piece++) stage_piece(t, piece, t);
  }

  // ---- fragment addresses ----
  const uint32_t frow = lane & 15u;              // row / channel of a 16 x 16 tile
  const uint32_t fg = lane >> 4;                 // K chunk of the operand; channel quad of the result
  uint32_t a_off[kGroups];                       // + tm * 1024 + (slot & 1) * kStage
  uint32_t w_off[kGroups];                       // + (tn >> 1) * 2048 + (tn & 1) * 256 + (slot & 1) * kStage
#pragma unroll
  for (int h = 0; h < kGroups; h++) {
    a_off[h] = (wm * 64 + frow) * kBK + ((fg ^ a_swizzle(frow)) << 4) + h * 2 * kStage;
    w_off[h] = kATile + (wn * 8 + (fg >> 1)) * 1024 + (frow + 32 * (fg & 1u)) * 16 + h * 2 * kStage;
    asm volatile("" : "+v"(a_off[h]), "+v"(w_off[h]));
  }
  v4i fa[kTM];                                   // activation fragments of the current K tile (operand B)
  v4i wl[kHalf], wh[kHalf];                      // weight fragments: channel tiles 0..3 / 4..7 of the wave (operand A)

  // slot known at compile time: address register of its pair of stages + immediates; run-time slot: one add
  auto read_a = [&](auto known_c, uint32_t slot, int tm) __attribute__((always_inline)) {
    if constexpr (decltype(known_c)::value) {
      fa[tm] = *reinterpret_cast<const v4i*>(lds + a_off[slot >> 1] + (slot & 1u) * kStage + tm * 1024);
    } else {
      fa[tm] = *reinterpret_cast<const v4i*>(lds + slot * kStage + a_off[0] + tm * 1024);
    }
  };
  auto read_w = [&](auto known_c, uint32_t slot, int tn, v4i& dst) __attribute__((always_inline)) {
    const uint32_t imm = (tn >> 1) * 2048 + (tn & 1) * 256;
    if constexpr (decltype(known_c)::value) {
      dst = *reinterpret_cast<const v4i*>(lds + w_off[slot >> 1] + (slot & 1u) * kStage + imm);
    } else {
      dst = *reinterpret_cast<const v4i*>(lds + slot * kStage + w_off[0] + imm);
    }
  };

  const uint32_t flip = p.a_flip;                // 0x80808080 (kzp 128) or 0x7F7F7F7F (kzp 127), scalar
  uint32_t rs[kTM] = {0u, 0u, 0u, 0u};           // ROWSUM: sum of the raw bytes of this lane's chunks of rows 16 tm + (lane & 15)
  // half (0 / 1) of the recentring of one activation fragment: 2 of its 4 dwords
  auto flip_half = [&](int tm, int half) __attribute__((always_inline)) {
    if constexpr (ROWSUM) {
      if (half) {
        rs[tm] = __builtin_amdgcn_sad_u8(static_cast<uint32_t>(fa[tm].z), 0u, rs[tm]);
        rs[tm] = __builtin_amdgcn_sad_u8(static_cast<uint32_t>(fa[tm].w), 0u, rs[tm]);
      } else {
        rs[tm] = __builtin_amdgcn_sad_u8(static_cast<uint32_t>(fa[tm].x), 0u, rs[tm]);
        rs[tm] = __builtin_amdgcn_sad_u8(static_cast<uint32_t>(fa[tm].y), 0u, rs[tm]);
      }
      // (pinned HERE: left alone, hipcc sinks the sums of the unrolled drain tiles to the epilogue, where their result is first
      //  needed, and keeps -- spills -- the raw fragments until then)
      asm volatile("" : "+v"(rs[tm]));
    }
    if constexpr ((ABL & 2) != 0) {
      asm volatile("" : "+v"(fa[tm]));
    } else if (half) {
      fa[tm].z ^= static_cast<int>(flip);
      fa[tm].w ^= static_cast<int>(flip);
      asm volatile("" : "+v"(fa[tm].z), "+v"(fa[tm].w));
    } else {
      fa[tm].x ^= static_cast<int>(flip);
      fa[tm].y ^= static_cast<int>(flip);
      asm volatile("" : "+v"(fa[tm].x), "+v"(fa[tm].y));
    }
  };

  v4i acc[kTM][kTN];
  auto mma = [&](const v4i& w, int tm, int tn) __attribute__((always_inline)) {
    if constexpr ((ABL & 4) != 0) return;
    acc[tm][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, fa[tm], acc[tm][tn], 0, 0, 0);
  };

  // ---- prologue, part 2: tile 0 and the bias line have landed (loads complete in issue order) ----
  wait_vmcnt<(RING - 1) * kDma>();
  __builtin_amdgcn_s_barrier();
  QNNP_X_STAMP(1);
  using T = std::true_type;
  using F = std::false_type;
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) read_a(T{}, 0u, tm);
#pragma unroll
  for (int tn = 0; tn < kHalf; tn++) read_w(T{}, 0u, tn, wl[tn]);
  // accumulators: lane l holds, in register r of tile tn, channel (nb0 + wn * 4) * 32 + tn * 16 + 4 * (l >> 4) + r
  // (its own wave's DMA: visible behind the vmcnt wait above; four addresses per read, broadcasts)
#pragma unroll
  for (int tn = 0; tn < kTN; tn++) {
    const v4i b = *reinterpret_cast<const v4i*>(bias_line + tn * 64 + fg * 16);
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) acc[tm][tn] = b;
  }
  // fa[0], fa[1] re-centred here, fa[2], fa[3] left raw: the state every tile's phase 1 starts in (it re-centres the last two)
#pragma unroll
  for (int tm = 0; tm < 2; tm++) { flip_half(tm, 0); flip_half(tm, 1); }
  asm volatile("" : "+v"(wl[0]), "+v"(wl[1]), "+v"(wl[2]), "+v"(wl[3]));
  if constexpr ((ABL & 16) != 0) { wh[0] = wl[0]; wh[1] = wl[1]; wh[2] = wl[2]; wh[3] = wl[3]; }

  /*
   * One K tile (see the file comment). SYNC: 1 = counted wait + barrier; 2 = the FINAL one, vmcnt(0): every tile is resident
   * afterwards; 0 = none (the tail). P1F / P2F: this tile's phases still issue LDS-DMA. MORE: a next tile exists.
   * KNOWN: `slot` is a literal.
   */
  auto iteration = [&](auto p1f_c, auto more_c, auto p2f_c, auto sync_c, auto known_c, uint32_t kt, uint32_t slot) __attribute__((always_inline)) {
    constexpr bool P1F = decltype(p1f_c)::value && (ABL & 8) == 0;
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool P2F = decltype(p2f_c)::value && (ABL & 8) == 0;
    constexpr int SYNC = (ABL & 32) != 0 ? 0 : decltype(sync_c)::value;
    constexpr bool KNOWN = decltype(known_c)::value;
    constexpr bool READS = (ABL & 16) == 0;
    const uint32_t prev_slot = slot == 0 ? RING - 1 : slot - 1;
    const uint32_t next_slot = slot + 1 == RING ? 0 : slot + 1;

    // ---- phase 1: fa x wl; the tile's second four weight fragments arrive, one behind each of the first MFMAs ----
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      const int tm = i / kHalf, tn = ((OPT & 1) == 0 && (tm & 1) != 0) ? kHalf - 1 - i % kHalf : i % kHalf;
      if constexpr (P1F && KNOWN) {
        if (i % 8 == 0) { dma16_set_m0(piece_dst(2 + i / 8, prev_slot)); QNNP_PIN(); }
      }
      mma(wl[tn], tm, tn);
      QNNP_PIN();
      if constexpr (READS) {
        if (i < kHalf) { read_w(known_c, slot, kHalf + i, wh[i]); QNNP_PIN(); }
      }
      if constexpr (P1F && KNOWN) {
        if (i % 8 == 0) { dma16_saddr_m0_set(piece_src(kt + RING - 1, 2 + i / 8), piece_off(2 + i / 8)); QNNP_PIN(); }
      } else if constexpr (P1F) {
        if (i % 8 == 0) { stage_piece(kt + RING - 1, 2 + i / 8, prev_slot); QNNP_PIN(); }
      }
      // the last two activation fragments of this tile were read at the end of the previous phase 2: re-centre them ahead of
      // their first use (MFMA 8 and 12)
      // (LDS reads return in order: behind fa[2] came fa[3] and wh[0..2] by now, behind fa[3] the four wh reads)
      if (i == 2) { __builtin_amdgcn_s_waitcnt(0xC47F); QNNP_PIN(); }     // lgkmcnt(4): fa[2] has landed
      if (i == 2 || i == 3) { flip_half(2, i - 2); QNNP_PIN(); }
      if (i == 6) { __builtin_amdgcn_s_waitcnt(0xC47F); QNNP_PIN(); }     // lgkmcnt(4): fa[3] has landed
      if (i == 6 || i == 7) { flip_half(3, i - 6); QNNP_PIN(); }
      if (i == kMma - 2) { __builtin_amdgcn_s_waitcnt(0xC07F); QNNP_PIN(); }   // wh complete
    }
    asm volatile("" : "+v"(wh[0]), "+v"(wh[1]), "+v"(wh[2]), "+v"(wh[3]));
    QNNP_PIN();

    if constexpr (SYNC == 1) {
      wait_vmcnt<(RING - 2) * kDma>();
      __builtin_amdgcn_s_barrier();
    } else if constexpr (SYNC == 2) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();         // from here on nothing synchronizes the waves
    }
    QNNP_PIN();

    // ---- phase 2: fa x wh; the next tile's first four weight fragments and its activation fragments arrive, fa[tm] right
    //      behind the last MFMA that reads the old one ----
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      const int tm = i / kHalf, tn = ((OPT & 1) == 0 && (tm & 1) != 0) ? kHalf - 1 - i % kHalf : i % kHalf;
      if constexpr (P2F && KNOWN) {
        if (i % 8 == 0) { dma16_set_m0(piece_dst(i / 8, slot)); QNNP_PIN(); }
      }
      mma(wh[tn], tm, kHalf + tn);
      QNNP_PIN();
      if constexpr (MORE && READS) {
        if (tm == 0) { read_w(known_c, next_slot, tn, wl[tn]); QNNP_PIN(); }
        if (i % kHalf == kHalf - 1) { read_a(known_c, next_slot, tm); QNNP_PIN(); }
      }
      if constexpr (P2F && KNOWN) {
        if (i % 8 == 0) { dma16_saddr_m0_set(piece_src(kt + RING, i / 8), piece_off(i / 8)); QNNP_PIN(); }
      } else if constexpr (P2F) {
        if (i % 8 == 0) { stage_piece(kt + RING, i / 8, slot); QNNP_PIN(); }
      }
      if constexpr (MORE) {
        // fa[0] (read behind MFMA 3) re-centred behind MFMAs 9, 10; fa[1] (behind MFMA 7) behind MFMAs 13, 14
        if (i == 9) { __builtin_amdgcn_s_waitcnt(0xC17F); QNNP_PIN(); }      // lgkmcnt(1): all but fa[1]
        if (i == 9 || i == 10) { flip_half(0, i - 9); QNNP_PIN(); }
        if (i == 13) { __builtin_amdgcn_s_waitcnt(0xC17F); QNNP_PIN(); }     // lgkmcnt(1): all but fa[2]
        if (i == 13 || i == 14) { flip_half(1, i - 13); QNNP_PIN(); }
      }
    }
    if constexpr (MORE) asm volatile("" : "+v"(wl[0]), "+v"(wl[1]), "+v"(wl[2]), "+v"(wl[3]));
    QNNP_PIN();
  };

  using Sync0 = std::integral_constant<int, 0>;
  using Sync1 = std::integral_constant<int, 1>;
  using Sync2 = std::integral_constant<int, 2>;

  // (the launcher guarantees ktiles >= 2 * RING)
  iteration(F{}, T{}, T{}, Sync1{}, T{}, 0u, 0u);   // the prologue staged pieces 2, 3 of tile RING - 1 already
  QNNP_X_STAMP(2);
  uint32_t kt = 1;
  for (; kt + (RING - 1) + RING < ktiles; kt += RING) {   // steady state, ring slots as literals (kt % RING == 1 here)
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt, 1u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 1, 2u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 2, 3u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 3, 0u);
  }
  if constexpr (ALIGNED) {
    // ktiles % 4 == 0: the loop above stopped at kt == ktiles - 7 (slot 1); the rest of the tiles with literal slots
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt, 1u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 1, 2u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 2, 3u);
    QNNP_X_STAMP(3);
    iteration(T{}, T{}, F{}, Sync1{}, T{}, kt + 3, 0u);     // ktiles - 4: the last pieces of the last tile
    iteration(F{}, T{}, F{}, Sync2{}, T{}, kt + 4, 1u);     // ktiles - 3: the final wait + barrier
    QNNP_X_STAMP(4);
    iteration(F{}, T{}, F{}, Sync0{}, T{}, kt + 5, 2u);     // tail: everything resident, no barriers
    iteration(F{}, F{}, F{}, Sync0{}, T{}, kt + 6, 3u);     // last tile
  } else {
    uint32_t slot = 1;                                      // == kt % RING
    auto advance = [&]() __attribute__((always_inline)) { kt++; slot = slot + 1 == RING ? 0 : slot + 1; };
    while (kt + RING < ktiles) {                            // steady state, run-time slot
      iteration(T{}, T{}, T{}, Sync1{}, F{}, kt, slot);
      advance();
    }
    QNNP_X_STAMP(3);
    iteration(T{}, T{}, F{}, Sync1{}, F{}, kt, slot); // kt == ktiles - RING: the last pieces of the last tile
    advance();
    iteration(F{}, T{}, F{}, Sync2{}, F{}, kt, slot); // kt == ktiles - RING + 1: the final wait + barrier
    advance();
    QNNP_X_STAMP(4);
    while (kt + 1 < ktiles) {                               // tail: everything resident, no barriers
      iteration(F{}, T{}, F{}, Sync0{}, F{}, kt, slot);
      advance();
    }
    iteration(F{}, F{}, F{}, Sync0{}, F{}, kt, slot); // last tile
  }
  QNNP_X_STAMP(5);

  // ---- fused epilogue: Q31 requantize in registers -> lane transposes -> whole 128-byte lines, no LDS ----
  if constexpr ((ABL & 1) != 0) {
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) asm volatile("" : : "v"(acc[tm][tn]));
    }
    return;
  }
  const uint32_t m0 = m_tile * kBM + wm * 64;
  const uint32_t n0 = (nb0 + wn * 4) * 32;
  uint8_t* out0 = p.output + static_cast<uint64_t>(m0) * p.output_stride + static_cast<uint64_t>(g) * p.n + n0;
  // a store instruction of block tm writes rows (frow & 7) [+ 8], bytes (frow >> 3) * 64 + fg * 16 .. + 16 of the wave's 128
  const uint32_t st_row = frow & 7u;
  const uint32_t st_col = (frow >> 3) * 64 + fg * 16;
  const bool col_ok = n0 + st_col < p.n;
  // 4 x 4 dword transpose over the four 16-lane rows: in  q[j] at lane row g = element (g, j);  out q[j] at lane row g = (j, g)
  auto transpose4 = [&](uint32_t (&q)[4]) __attribute__((always_inline)) {
    const auto s02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);   // {q0.r0 q0.r1 q2.r0 q2.r1}, {q0.r2 q0.r3 q2.r2 q2.r3}
    const auto s13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    const auto lo = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);  // {q0.r0 q1.r0 q2.r0 q3.r0}, {q0.r1 q1.r1 q2.r1 q3.r1}
    const auto hi = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);  // {.. r2 ..}, {.. r3 ..}
    q[0] = lo[0]; q[1] = lo[1]; q[2] = hi[0]; q[3] = hi[1];
  };
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
    if constexpr (ROWSUM) {
      // the row's four chunk sums (lanes l, l ^ 16, l ^ 32, l ^ 48), then (128 - kzp) * (sum of a - 128 K) onto every accumulator of
      // the row (wrapping: the accumulators may carry the 2^31 offset of the offset forms)
      uint32_t t = rs[tm];
      t += static_cast<uint32_t>(__shfl_xor(static_cast<int>(t), 16));
      t += static_cast<uint32_t>(__shfl_xor(static_cast<int>(t), 32));
      const uint32_t term = static_cast<uint32_t>(p.row_coeff) * (t - 128u * p.k_pad);
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) {
#pragma unroll
        for (int r = 0; r < 4; r++) acc[tm][tn][r] = static_cast<int>(static_cast<uint32_t>(acc[tm][tn][r]) + term);
      }
    }
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      lo[j] = q31_requantize_pack4_clamp<SEQ, CLAMP>(acc[tm][j][0], acc[tm][j][1], acc[tm][j][2], acc[tm][j][3], p.rq);
      hi[j] = q31_requantize_pack4_clamp<SEQ, CLAMP>(acc[tm][4 + j][0], acc[tm][4 + j][1], acc[tm][4 + j][2], acc[tm][4 + j][3], p.rq);
    }
    transpose4(lo);        // lane (row, g): channels 16 g .. 16 g + 15 of its row (bytes 0..63 of the wave's 128)
    transpose4(hi);        // ... and channels 64 + 16 g ..
    // whole lines: rows 0..7 of the block keep their low halves and take the high halves of the SAME rows from lanes row + 8
    typedef int nt_v4i __attribute__((ext_vector_type(4)));
    nt_v4i x, y;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      // row_ror:8 = the other half of the 16-lane row; bank mask 0xC = lanes 8..15 of a row take the source, 0x3 = lanes 0..7
      x[j] = __builtin_amdgcn_update_dpp(static_cast<int>(lo[j]), static_cast<int>(hi[j]), 0x128, 0xF, 0xC, false);
      y[j] = __builtin_amdgcn_update_dpp(static_cast<int>(hi[j]), static_cast<int>(lo[j]), 0x128, 0xF, 0x3, false);
    }
    const uint32_t rx = tm * 16 + st_row, ry = rx + 8;
    nt_v4i* dx = reinterpret_cast<nt_v4i*>(out0 + static_cast<uint64_t>(rx) * p.output_stride + st_col);
    nt_v4i* dy = reinterpret_cast<nt_v4i*>(out0 + static_cast<uint64_t>(ry) * p.output_stride + st_col);
    const bool stores = (ABL & 64) == 0 || p.rows == 0xFFFFFFFFu;
    if (m0 + rx < p.rows && col_ok && stores) {
      if (p.stream_out) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dx), "v"(x) : "memory");
      else *dx = x;
    }
    if (m0 + ry < p.rows && col_ok && stores) {
      if (p.stream_out) asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dy), "v"(y) : "memory");
      else *dy = y;
    }
    if (tm == 1) QNNP_X_STAMP(6);
  }
  QNNP_X_STAMP(7);
#ifdef QNNP_ENABLE_ABLATION
  if (p.trace != nullptr) {                      // when the stores have left the wave
    wait_vmcnt<0>();
    QNNP_X_STAMP_WALL(1);
  }
#endif
}
#undef QNNP_PIN

template <bool ALIGNED, bool ROWSUM>
int launch_x(const IgemmParams& p, const dim3& grid, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
#ifdef QNNP_ENABLE_ABLATION
  if constexpr (ALIGNED && !ROWSUM) {
    const char* env = getenv("QNNP_GFX950_ABLATE");
    const int abl = env != nullptr ? atoi(env) : 0;
#define QNNP_ABL_CASE(V) case V: hipLaunchKernelGGL((q8_gemm_mfma_256x256_c16_kernel<kRqShift0Ofs, 1, true, V>), grid, dim3(kThreads), 0, stream, p); \
        return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
    switch (abl) {
      QNNP_ABL_CASE(1) QNNP_ABL_CASE(2) QNNP_ABL_CASE(4) QNNP_ABL_CASE(8) QNNP_ABL_CASE(16) QNNP_ABL_CASE(24)
      QNNP_ABL_CASE(27) QNNP_ABL_CASE(32) QNNP_ABL_CASE(59) QNNP_ABL_CASE(64)
      default: break;
    }
#undef QNNP_ABL_CASE
    const char* oenv = getenv("QNNP_C16_OPT");
    if (oenv != nullptr && atoi(oenv) == 1) {
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_c16_kernel<kRqShift0Ofs, 1, true, 0, false, 1>), grid, dim3(kThreads), 0, stream, p);
      return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
    }
  }
#endif
  if (p.rq.f.shift != 0 && p.rq.f.bounded && p.rq.f.ofs_kind == 2 && !p.rq.full_range) {
    // bounded accumulators, shift >= 1, a clamp other than [0, 255]: the bounded sequence with the clamp class picked here
    if (p.rq.zp_late == 0) hipLaunchKernelGGL((q8_gemm_mfma_256x256_c16_kernel<kRqBoundedOfs, 1, ALIGNED, 0, ROWSUM>), grid, dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL((q8_gemm_mfma_256x256_c16_kernel<kRqBoundedOfs, 2, ALIGNED, 0, ROWSUM>), grid, dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  }
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    if constexpr (decltype(full)::value) {
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_c16_kernel<kSeq, 0, ALIGNED, 0, ROWSUM>), grid, dim3(kThreads), 0, stream, p);
    } else if (p.rq.zp_late == 0) {             // zero point folded (or zero): no add behind the clamp
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_c16_kernel<kSeq, 1, ALIGNED, 0, ROWSUM>), grid, dim3(kThreads), 0, stream, p);
    } else {
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_c16_kernel<kSeq, 2, ALIGNED, 0, ROWSUM>), grid, dim3(kThreads), 0, stream, p);
    }
    rc = hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  });
  return rc;
}

}  // namespace

/* `p` must carry the CENTRED weight image, its bias pair table and a_flip (q8igemm.hip); gemm256c_supported(p) holds.
 * p.row_coeff != 0: the STANDARD image (centred on 128, a_flip 0x80808080) of an operator with another kernel zero point -- the
 * flavour that adds the kernel-zero-point row term (ROWSUM). */
int gemm256x_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name)
{
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = p.n_pad / kBN;
  const dim3 grid(tiles_m * tiles_n, groups, 1);
  IgemmParams pm = p;
  // x / tiles_n == hi32(x * magic) for x < 2^32 / tiles_n (the tile ids); 0 stands for tiles_n == 1
  pm.tiles_n_magic = tiles_n == 1 ? 0u : static_cast<uint32_t>((1ull << 32) / tiles_n) + 1u;
  const bool aligned = (p.k_pad / kBK) % kRing == 0;
  if (p.row_coeff != 0) {
    *name = "q8_gemm_mfma_256x256_r16";
    return aligned ? launch_x<true, true>(pm, grid, stream) : launch_x<false, true>(pm, grid, stream);
  }
  *name = "q8_gemm_mfma_256x256_c16";
  return aligned ? launch_x<true, false>(pm, grid, stream) : launch_x<false, false>(pm, grid, stream);
}

}  // namespace qnnp
