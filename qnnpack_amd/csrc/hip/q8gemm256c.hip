/*
 * q8gemm256c.hip -- the zero-point-CENTRED flavour of the 256 x 256 uint8 GEMM (BASELINE.json configs[1]).
 *
 * Same role as q8gemm256.hip (it replaces q8gemm_ukernel_4x4c2__sse2, reference src/q8gemm/4x4c2-sse2.c:14-318, and
 * its tiler compute_q8gemm, src/operator-run.c:39-70, 797-802, for MFMA-bound problems), same tile, same LDS-DMA ring,
 * same software pipeline -- for the operators whose kernel zero point lets the weight image be centred ON that zero
 * point instead of on 128:
 *
 *   kzp == 128:  w'' = w - 128 = w ^ 0x80 (the standard image),   a'' = a ^ 0x80 = a - 128
 *   kzp == 127:  w'' = 127 - w = w ^ 0x7F (its bitwise NOT),       a'' = a ^ 0x7F = 127 - a
 *
 * Either way  sum_k a''(m,k) w''(n,k) == sum_k (a - kzp)(w - kzp)  exactly (int8 operands, no overflow: both factors
 * lie in [-128, 127]), hence
 *
 *   bias + sum (a - izp)(w - kzp) = biasc[n] + sum a'' w'',   biasc[n] = bias[n] + (kzp - izp) * sum_k (w(n,k) - kzp)
 *
 * and the kernel-zero-point ROW term of q8gemm256.hip -- (128 - kzp) * sum_k a'(m,k): eight v_sad_u8 per K tile and
 * wave, an exchange of the partial sums through LDS behind the main loop, a per-lane 64-bit addend in the epilogue --
 * does not exist. (It is the trick pack.h already plays for the depthwise dot-product image, `dw_wrange`; and the
 * reference's own benchmarks and PyTorch's symmetric weights use exactly these two zero points, bench/q8gemm.cc:60-64.)
 * Every other zero point keeps q8gemm256.hip.
 *
 * What else differs from the lean flavour there (each measured in round 4, DESIGN.md section 4.1, profiles/r04/):
 *   - the requantization sequence and the clamp class are TEMPLATE arguments chosen by the launcher (the kernel has no
 *     branch on them); with no row term the accumulators start from biasc + 2^31 and the offset forms of requant_math.h
 *     apply as they are; a clamp other than [0, 255] with a folded zero point costs one v_med3 per value and a 3-op pack;
 *   - the folded bias arrives by ONE LDS-DMA instruction per wave (512 bytes into the 32 KiB of LDS the ring leaves
 *     free) right behind the first tile's pieces and is read back as broadcast ds_read_b128: sixteen global loads per
 *     wave -- as many texture-path instructions as the whole ring fill -- sat in front of the first MFMA before;
 *   - the weight-fragment reads of a phase are issued one per MFMA instead of in one burst behind the barrier (the LDS
 *     command FIFO was full for 1.2 M of 45 M wave-cycles);
 *   - a barrier-free TAIL: the last three K tiles are resident once the last LDS-DMA has landed, so one final
 *     vmcnt(0) + barrier releases the waves to run to the end on their own, and the output tile is staged through the
 *     two ring slots that are free from then on, half a wave tile at a time.
 * Measured and dropped (profiles/r04/gemm_centred_ring_skew_ab_r04a.txt, gemm_centred_cycle_stamps_r04c.txt): a fifth
 * ring stage (59.3 against 58.9 us), s_setprio for the older wave of each SIMD in the tail (the older wave wins the
 * matrix pipe anyway: its tail takes 1.5 k cycles against the younger one's 2.7 k with or without it) and, the blunt form
 * of the same idea, putting the younger waves to sleep for 1024 / 1536 cycles behind the final barrier (59.64 / 59.68
 * against 59.53 us, gemm_centred_aligned_sleep_ab_r04e.txt), a wave-dependent issue position for the LDS-DMA pieces (four
 * copies of the loop: 210 spilled registers), nt / sc1 cache policy on the LDS-DMA loads (61.0 / 55.7 against 55.5 us).
 *
 * Requirements (gemm256c_supported): plain GEMM, K % 64 == 0, K >= 512, N padded to 256, 16-byte aligned rows and
 * outputs (store_mode 2), a bias pair table.
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

constexpr int kBM = 256;
constexpr int kBN = 256;
constexpr int kBK = 64;                        // bytes of K per tile (two 32-deep MFMA sub-steps)
constexpr int kATile = kBM * kBK;              // 16 KiB
constexpr int kWTile = kBN * kBK;              // 16 KiB
constexpr int kStage = kATile + kWTile;        // 32 KiB
constexpr int kThreads = 512;                  // 8 waves: 4 (rows) x 2 (channels), 64 x 128 outputs per wave
constexpr int kTM = 2;                         // 32-row MFMA tiles per wave
constexpr int kTN = 4;                         // 32-channel MFMA tiles per wave
constexpr int kDma = 4;                        // LDS-DMA instructions per thread and K tile: 2 activation + 2 weight pieces
constexpr int kMma = kTM * kTN;                // MFMAs per K sub-step
constexpr int kRing = 4;                       // LDS stages of 32 KiB; the remaining 32 KiB hold the waves' bias lines
constexpr int kBiasArea = kRing * kStage;      // 8 waves x 512 bytes
constexpr uint32_t kImagePitch = kTN * 32 + 16;          // +16: the 8-lane ds_write_b128 groups hit distinct banks
constexpr uint32_t kImageBytes = 32 * kImagePitch;       // one 32-row half of a wave's 64 x 128 output tile

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

/* a wave-uniform pointer, in scalar registers for good (q8gemm256.hip) */
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

/* LDS-DMA, saddr form (q8gemm256.hip): 16 bytes per lane from base + lane_offset to m0 + lane * 16.
 * (Cache policy bits on these loads, measured in round 4: sc1 level, nt 10 % slower -- every line is re-read by the
 *  other CUs of the XCD; profiles/r04/gemm_centred_bias_dma_clamp_spread_policy_ab_r04d.txt.) */
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

#define QNNP_PIN() __builtin_amdgcn_sched_barrier(0)

// measurement builds: cycle stamps of wave 0 (item 0) and wave 4 (item 1) of every workgroup; item 3 = wall clock
#ifdef QNNP_ENABLE_ABLATION
#define QNNP_C_STAMP(slot)                                                                                  \
  do {                                                                                                       \
    if (p.trace != nullptr && lane == 0 && (wave & 3u) == 0)                                                 \
      p.trace[(blockIdx.x * 4 + (wave >> 2)) * 8 + (slot)] = __builtin_readcyclecounter();                  \
  } while (0)
#define QNNP_C_STAMP_WALL(slot)                                                                             \
  do {                                                                                                       \
    if (p.trace != nullptr && lane == 0 && (wave & 3u) == 0)                                                 \
      p.trace[(blockIdx.x * 4 + 2 + (wave >> 2)) * 8 + (slot)] = wall_clock64();                            \
  } while (0)
#else
#define QNNP_C_STAMP(slot) do { } while (0)
#define QNNP_C_STAMP_WALL(slot) do { } while (0)
#endif

/*
 * SEQ / CLAMP: rounding sequence and clamp class of the requantization (requant.hip.h), chosen by the launcher.
 * ALIGNED: the K tiles are a multiple of the ring (K % 256 == 0): every ring slot is a literal, the drain included.
 * OPT (A/B structure, "gemm_kernel" 21): 2 = the fragment reads of a phase in one burst behind the barrier (round 3's
 * order) instead of the weight fragments one per MFMA.
 * ABL: measurement-only ablation mask (builds with -DQNNP_ENABLE_ABLATION, env QNNP_GFX950_ABLATE); 0 in the product.
 * 1 = no epilogue (requantization + stores), 2 = no recentring, 4 = no MFMA, 8 = no LDS-DMA after the prologue,
 * 16 = no fragment reads after the prologue, 32 = no per-tile wait + barrier, 64 = no global stores (everything else of the
 * epilogue stays).
 */
template <int SEQ, int CLAMP, bool ALIGNED, int OPT = 0, int ABL = 0>
__global__ __launch_bounds__(kThreads, 2)
void q8_gemm_mfma_256x256_c_kernel(const IgemmParams p)
{
  static_assert(SEQ == kRqShift0Ofs || SEQ == kRqBoundedOfs || SEQ == kRqGeneral, "offset forms, or the general one");
  constexpr int RING = kRing;
  constexpr int kGroups = (RING + 1) / 2;        // address registers per fragment: a ds_read immediate reaches 64 KiB = 2 stages

  __shared__ __attribute__((aligned(16))) uint8_t lds[kRing * kStage + 8 * 512];    // the ONE LDS object (guide 5, trap 4a)

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave >> 1;       // 64-row slice
  const uint32_t wn = wave & 1u;       // 128-channel half
  const uint32_t g = blockIdx.y;
  QNNP_C_STAMP(0);
  QNNP_C_STAMP_WALL(0);

  // Workgroup -> tile (q8gemm256.hip): contiguous logical ids per XCD, bands of four row tiles.
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = p.n_pad / kBN;
  uint32_t m_tile, n_tile;
  {
    // (branch-free up to the band test, division by multiplication: the kernel arguments are fetched in one round
    //  instead of one per basic block -- each round is a scalar-cache miss in front of the first LDS-DMA)
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    const uint32_t logical = xcd * q + min(xcd, r) + idx;       // == (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx
    constexpr uint32_t kBand = 4;
    const uint32_t band = p.tiles_n_magic != 0 ? __umulhi(logical >> 2, p.tiles_n_magic) : logical >> 2;   // logical / (4 * tiles_n)
    const uint32_t within = logical - band * kBand * tiles_n;
    const uint32_t rows_in_band = min(kBand, tiles_m - band * kBand);
    if (rows_in_band == kBand) {                 // (the common case without a division)
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
  // activation tile image: [256 rows][4 chunks of 16 B], chunk slot s of row r holds logical chunk s ^ ((r >> 2) & 3)
  // (round 5: strided 1x1 convolutions -- igemm_params.h `offsets_dense`: a row's address comes from the operator's table, one
  //  valid entry per output pixel; the lane offsets are then absolute, the launcher checks that the tensor ends below 2^32)
  const bool table_rows = p.offsets_dense != 0;
  const uint8_t* a_base = scalar_ptr(table_rows ? p.input + static_cast<uint64_t>(g) * p.kc
      : p.input + static_cast<uint64_t>(m_tile * kBM) * p.input_stride + static_cast<uint64_t>(g) * p.kc);
  uint32_t a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const uint32_t L = i * kThreads + tid;
    const uint32_t r = L >> 2;
    const uint32_t chunk = (L & 3u) ^ ((r >> 2) & 3u);
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

  // The wave's 128 folded biases (+ 2^31 for the offset forms): ONE LDS-DMA instruction, lanes 0..31, 512 bytes into the
  // wave's line of the bias area. (Sixteen 16-byte global loads per wave, as before, are as many texture-path
  // instructions as the whole ring fill, and the first MFMA waited for them.)
  const int32_t* bias_tab = SEQ == kRqGeneral ? p.bias2 : p.bias2u;
  uint8_t* bias_line = lds + kBiasArea + wave * 512;
  if (lane < 32) {
    dma16_saddr(scalar_ptr(reinterpret_cast<const uint8_t*>(bias_tab + static_cast<uint64_t>(g) * p.n_pad + (nb0 + wn * kTN) * 32)),
                   lane * 16, bias_line);
  }

#pragma unroll
  for (int t = 1; t < RING; t++) {
#pragma unroll
    for (int piece = 0; piece < kDma; piece++) stage_piece(t, piece, t);
  }

  // ---- fragment addresses ----
  const uint32_t frag_row0 = wm * (kTM * 32) + (lane & 31u);
  const uint32_t frag_khalf = lane >> 5;
  // swizzled activation fragment address: row * 64 + (((ksub * 2 + khalf) ^ ((row >> 2) & 3)) << 4) = a_fbase ^ (ksub << 5)
  uint32_t a_off[2][kTM][kGroups];
  uint32_t w_off[2][kGroups];
#pragma unroll
  for (int sub = 0; sub < 2; sub++) {
#pragma unroll
    for (int h = 0; h < kGroups; h++) {
#pragma unroll
      for (int tm = 0; tm < kTM; tm++) {
        const uint32_t row = frag_row0 + tm * 32;
        const uint32_t a_fbase = row * kBK + ((frag_khalf ^ ((row >> 2) & 3u)) << 4);
        a_off[sub][tm][h] = (a_fbase ^ (static_cast<uint32_t>(sub) << 5)) + h * 2 * kStage;
        asm volatile("" : "+v"(a_off[sub][tm][h]));
      }
      w_off[sub][h] = kATile + (wn * kTN * 2) * 1024 + lane * 16 + sub * 1024 + h * 2 * kStage;   // + tn * 2048
      asm volatile("" : "+v"(w_off[sub][h]));
    }
  }

  struct Frags {
    v4i a[kTM];
    v4i w[kTN];
  };
  // ring slot known at compile time: address register of its pair of stages + immediates
  auto read_frags_slot = [&](uint32_t slot, int sub, Frags& f) __attribute__((always_inline)) {
    const uint32_t h = slot >> 1, imm = (slot & 1u) * kStage;
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      f.a[tm] = *reinterpret_cast<const v4i*>(lds + a_off[sub][tm][h] + imm);
    }
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      f.w[tn] = *reinterpret_cast<const v4i*>(lds + w_off[sub][h] + imm + tn * 2048);
    }
  };
  // (OPT 2: the activation fragments first, the weight fragments one per MFMA)
  auto read_frags_a_slot = [&](uint32_t slot, int sub, Frags& f) __attribute__((always_inline)) {
    const uint32_t h = slot >> 1, imm = (slot & 1u) * kStage;
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      f.a[tm] = *reinterpret_cast<const v4i*>(lds + a_off[sub][tm][h] + imm);
    }
  };
  auto read_frag_w_slot = [&](uint32_t slot, int sub, Frags& f, int tn) __attribute__((always_inline)) {
    const uint32_t h = slot >> 1, imm = (slot & 1u) * kStage;
    f.w[tn] = *reinterpret_cast<const v4i*>(lds + w_off[sub][h] + imm + tn * 2048);
  };
  // run-time slot (the few iterations outside the unrolled steady state)
  auto read_frags_rt = [&](uint32_t slot, int sub, Frags& f) __attribute__((always_inline)) {
    const uint8_t* st = lds + slot * kStage;
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      f.a[tm] = *reinterpret_cast<const v4i*>(st + a_off[sub][tm][0]);
    }
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      f.w[tn] = *reinterpret_cast<const v4i*>(st + w_off[sub][0] + tn * 2048);
    }
  };

  const uint32_t flip = p.a_flip;                // 0x80808080 (kzp 128) or 0x7F7F7F7F (kzp 127), scalar
  // half h (0..3) of the recentring of one fragment set: 2 of its 8 dwords (opaque HERE: q8gemm256.hip)
  auto flip_part = [&](Frags& f, int h) __attribute__((always_inline)) {
    const int tm = h >> 1;
    if constexpr ((ABL & 2) != 0) {
      asm volatile("" : "+v"(f.a[tm]));
    } else if (h & 1) {
      f.a[tm].z ^= static_cast<int>(flip);
      f.a[tm].w ^= static_cast<int>(flip);
      asm volatile("" : "+v"(f.a[tm].z), "+v"(f.a[tm].w));
    } else {
      f.a[tm].x ^= static_cast<int>(flip);
      f.a[tm].y ^= static_cast<int>(flip);
      asm volatile("" : "+v"(f.a[tm].x), "+v"(f.a[tm].y));
    }
  };
  auto settle_w = [&](Frags& f) __attribute__((always_inline)) {
    asm volatile("" : "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]));
  };

  v16i acc[kTM][kTN];
  auto mma = [&](const Frags& f, int i) __attribute__((always_inline)) {       // i = 0..kMma-1 -> (tm, tn)
    const int tm = i / kTN, tn = i % kTN;
    if constexpr ((ABL & 4) != 0) return;
    acc[tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[tn], f.a[tm], acc[tm][tn], 0, 0, 0);
  };

  // ---- prologue, part 2: tile 0 and the bias line have landed (loads complete in issue order) ----
  wait_vmcnt<(RING - 1) * kDma>();
  __builtin_amdgcn_s_barrier();
  QNNP_C_STAMP(1);
  Frags fa, fb;
  read_frags_slot(0, 0, fa);
  // accumulators: lane l holds, in register r of tile tn, channel (nb0 + wn * 4 + tn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
  // (its own wave's DMA: visible behind the vmcnt wait above; two addresses per read, a broadcast)
#pragma unroll
  for (int tn = 0; tn < kTN; tn++) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const v4i b = *reinterpret_cast<const v4i*>(bias_line + (lane >> 5) * 16 + tn * 128 + rg * 32);
#pragma unroll
      for (int tm = 0; tm < kTM; tm++) {
        acc[tm][tn][rg * 4 + 0] = b.x;
        acc[tm][tn][rg * 4 + 1] = b.y;
        acc[tm][tn][rg * 4 + 2] = b.z;
        acc[tm][tn][rg * 4 + 3] = b.w;
      }
    }
  }
  flip_part(fa, 0); flip_part(fa, 1); flip_part(fa, 2); flip_part(fa, 3);
  settle_w(fa);
  if constexpr ((ABL & 16) != 0) fb = fa;       // (measurement builds: something defined to multiply)

  /*
   * Software pipeline (q8gemm256.hip, "iteration"): two phases per K tile, each = 8 MFMAs on one fragment set while the
   * other set is read from LDS and recentred; the LDS-DMA of a tile is spread over both phases.
   *   phase 1: multiply fa = (tile kt, sub-step 0) | read + recentre fb = (tile kt, sub-step 1)
   *            | pieces 2, 3 of tile kt + RING - 1 -> slot of tile kt - 1 (free since that tile's barrier)
   *   -- counted vmcnt + raw barrier: tile kt + 1 resident, slot of tile kt free --
   *   phase 2: multiply fb | read + recentre fa = (tile kt + 1, sub-step 0) | pieces 0, 1 of tile kt + RING -> slot of tile kt
   * At the wait of tile kt the tiles kt + 2 .. kt + RING - 1 may still be in flight: vmcnt((RING - 2) * 4).
   * SYNC: 1 = that wait + barrier; 2 = the FINAL one, vmcnt(0): every tile is resident afterwards; 0 = none (the tail).
   */
  auto iteration = [&](auto p1f_c, auto more_c, auto p2f_c, auto sync_c, auto known_c, auto stag_c, uint32_t kt, uint32_t slot) __attribute__((always_inline)) {
    constexpr int S = decltype(stag_c)::value;           // MFMA position (mod 4) of the LDS-DMA pieces
    constexpr bool SPREAD = (OPT & 2) == 0;
    constexpr bool P1F = decltype(p1f_c)::value && (ABL & 8) == 0;
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool P2F = decltype(p2f_c)::value && (ABL & 8) == 0;
    constexpr int SYNC = (ABL & 32) != 0 ? 0 : decltype(sync_c)::value;
    constexpr bool KNOWN = decltype(known_c)::value;     // `slot` is a literal
    const uint32_t prev_slot = slot == 0 ? RING - 1 : slot - 1;
    const uint32_t next_slot = slot + 1 == RING ? 0 : slot + 1;

    QNNP_PIN();
    if constexpr ((ABL & 16) == 0) {
      if constexpr (KNOWN && SPREAD) read_frags_a_slot(slot, 1, fb);
      else if constexpr (KNOWN) read_frags_slot(slot, 1, fb);
      else read_frags_rt(slot, 1, fb);
    }
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      if constexpr (P1F && KNOWN) {
        if (i % 4 == S) { dma16_set_m0(piece_dst(2 + i / 4, prev_slot)); QNNP_PIN(); }
      }
      mma(fa, i);
      QNNP_PIN();
      if constexpr (KNOWN && SPREAD && (ABL & 16) == 0) {
        if (i < kTN) { read_frag_w_slot(slot, 1, fb, i); QNNP_PIN(); }
      }
      if constexpr (P1F && KNOWN) {
        if (i % 4 == S) { dma16_saddr_m0_set(piece_src(kt + RING - 1, 2 + i / 4), piece_off(2 + i / 4)); QNNP_PIN(); }
      } else if constexpr (P1F) {
        if (i % 4 == 0) { stage_piece(kt + RING - 1, 2 + i / 4, prev_slot); QNNP_PIN(); }
      }
      if (i >= kMma - 2) {
        const int h = (i - (kMma - 2)) * 2;
        if (h == 0) { __builtin_amdgcn_s_waitcnt(0xC07F); QNNP_PIN(); }   // lgkmcnt(0) once: the reads were issued a phase ago
        flip_part(fb, h);
        flip_part(fb, h + 1);
        QNNP_PIN();
      }
    }
    settle_w(fb);
    QNNP_PIN();

    if constexpr (SYNC == 1) {
      wait_vmcnt<(RING - 2) * kDma>();
      __builtin_amdgcn_s_barrier();
    } else if constexpr (SYNC == 2) {
      wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();         // from here on nothing synchronizes the waves
    }
    QNNP_PIN();

    if constexpr (MORE && (ABL & 16) == 0) {
      if constexpr (KNOWN && SPREAD) read_frags_a_slot(next_slot, 0, fa);
      else if constexpr (KNOWN) read_frags_slot(next_slot, 0, fa);
      else read_frags_rt(next_slot, 0, fa);
    }
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      if constexpr (P2F && KNOWN) {
        if (i % 4 == S) { dma16_set_m0(piece_dst(i / 4, slot)); QNNP_PIN(); }
      }
      mma(fb, i);
      QNNP_PIN();
      if constexpr (MORE && KNOWN && SPREAD && (ABL & 16) == 0) {
        if (i < kTN) { read_frag_w_slot(next_slot, 0, fa, i); QNNP_PIN(); }
      }
      if constexpr (P2F && KNOWN) {
        if (i % 4 == S) { dma16_saddr_m0_set(piece_src(kt + RING, i / 4), piece_off(i / 4)); QNNP_PIN(); }
      } else if constexpr (P2F) {
        if (i % 4 == 0) { stage_piece(kt + RING, i / 4, slot); QNNP_PIN(); }
      }
      if constexpr (MORE) {
        if (i >= kMma - 2) {
          const int h = (i - (kMma - 2)) * 2;
          if (h == 0) { __builtin_amdgcn_s_waitcnt(0xC07F); QNNP_PIN(); }
          flip_part(fa, h);
          flip_part(fa, h + 1);
          QNNP_PIN();
        }
      }
    }
    if constexpr (MORE) settle_w(fa);
    QNNP_PIN();
  };

  using T = std::true_type;
  using F = std::false_type;
  using Sync0 = std::integral_constant<int, 0>;
  using Sync1 = std::integral_constant<int, 1>;
  using Sync2 = std::integral_constant<int, 2>;
  using S0 = std::integral_constant<int, 0>;

  // (the launcher guarantees ktiles >= 2 * RING)
  iteration(F{}, T{}, T{}, Sync1{}, T{}, S0{}, 0u, 0u);   // the prologue staged pieces 2, 3 of tile RING - 1 already
  QNNP_C_STAMP(2);
  auto rest = [&](auto stag_all) __attribute__((always_inline)) {
  uint32_t kt = 1;
  auto steady = [&](auto stag_c) __attribute__((always_inline)) {
    for (; kt + (RING - 1) + RING < ktiles; kt += RING) {   // steady state, ring slots as literals (kt % RING == 1 here)
      iteration(T{}, T{}, T{}, Sync1{}, T{}, stag_c, kt, 1u);
      iteration(T{}, T{}, T{}, Sync1{}, T{}, stag_c, kt + 1, 2u);
      iteration(T{}, T{}, T{}, Sync1{}, T{}, stag_c, kt + 2, 3u);
      iteration(T{}, T{}, T{}, Sync1{}, T{}, stag_c, kt + 3, 0u);
    }
  };
  steady(stag_all);
  if constexpr (ALIGNED) {
    // ktiles % 4 == 0: the loop above stopped at kt == ktiles - 7 (slot 1); the rest of the tiles with literal slots
    iteration(T{}, T{}, T{}, Sync1{}, T{}, S0{}, kt, 1u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, S0{}, kt + 1, 2u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, S0{}, kt + 2, 3u);
    QNNP_C_STAMP(3);
    iteration(T{}, T{}, F{}, Sync1{}, T{}, S0{}, kt + 3, 0u);     // ktiles - 4: the last pieces of the last tile
    iteration(F{}, T{}, F{}, Sync2{}, T{}, S0{}, kt + 4, 1u);     // ktiles - 3: the final wait + barrier
    QNNP_C_STAMP(4);
    iteration(F{}, T{}, F{}, Sync0{}, T{}, S0{}, kt + 5, 2u);     // tail: everything resident, no barriers
    iteration(F{}, F{}, F{}, Sync0{}, T{}, S0{}, kt + 6, 3u);     // last tile
  } else {
  uint32_t slot = 1;                                      // == kt % RING
  auto advance = [&]() __attribute__((always_inline)) { kt++; slot = slot + 1 == RING ? 0 : slot + 1; };
  while (kt + RING < ktiles) {                            // steady state, run-time slot
    iteration(T{}, T{}, T{}, Sync1{}, F{}, S0{}, kt, slot);
    advance();
  }
  QNNP_C_STAMP(3);
  iteration(T{}, T{}, F{}, Sync1{}, F{}, S0{}, kt, slot); // kt == ktiles - RING: the last pieces of the last tile
  advance();
  iteration(F{}, T{}, F{}, Sync2{}, F{}, S0{}, kt, slot); // kt == ktiles - RING + 1: the final wait + barrier
  advance();
  QNNP_C_STAMP(4);
  while (kt + 1 < ktiles) {                               // tail: everything resident, no barriers
    iteration(F{}, T{}, F{}, Sync0{}, F{}, S0{}, kt, slot);
    advance();
  }
  iteration(F{}, F{}, F{}, Sync0{}, F{}, S0{}, kt, slot); // last tile
  }
  QNNP_C_STAMP(5);

  // ---- fused epilogue: Q31 requantize in registers -> half a wave tile at a time through LDS -> whole 128-byte lines ----
  if constexpr ((ABL & 1) != 0) {
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) asm volatile("" : : "v"(acc[tm][tn]));
    }
    return;
  }
  // The two ring slots nobody reads after the final barrier: those of tiles kf = ktiles - RING + 1 and kf - 1.
  const uint32_t kf = ktiles - RING + 1;
  const uint32_t slot_a = ALIGNED ? 1u : kf % RING;
  const uint32_t slot_b = ALIGNED ? 0u : (kf + RING - 1) % RING;
  uint8_t* image = lds + (wave < 4 ? slot_a : slot_b) * kStage + (wave & 3u) * kImageBytes;
  // one 32 x 32 accumulator tile -> 16 bytes of this lane's row in the image (lane l: channels 0..15, lane l + 32: 16..31)
  auto stage_tile = [&](const v16i& a, uint8_t* dst) __attribute__((always_inline)) {
    uint32_t pk[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      pk[rg] = q31_requantize_pack4_clamp<SEQ, CLAMP>(a[rg * 4 + 0], a[rg * 4 + 1], a[rg * 4 + 2], a[rg * 4 + 3], p.rq);
    }
    const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
    *reinterpret_cast<uint4*>(dst) = make_uint4(s02[0], s02[1], s13[0], s13[1]);
  };
  const uint32_t m0 = m_tile * kBM + wm * (kTM * 32);
  const uint32_t n0 = (nb0 + wn * kTN) * 32;
  uint8_t* out0 = p.output + static_cast<uint64_t>(m0) * p.output_stride + static_cast<uint64_t>(g) * p.n + n0;
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      stage_tile(acc[tm][tn], image + (lane & 31u) * kImagePitch + tn * 32 + frag_khalf * 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
#pragma unroll
    for (int i = 0; i < (32 * kTN * 2) / 64; i++) {
      const uint32_t idx = i * 64 + lane;
      const uint32_t r = idx / (kTN * 2);
      const uint32_t c = idx % (kTN * 2);
      const uint4 v = *reinterpret_cast<const uint4*>(image + r * kImagePitch + c * 16);
      if (m0 + tm * 32 + r < p.rows && n0 + c * 16 < p.n && ((ABL & 64) == 0 || p.rows == 0xFFFFFFFFu)) {
        typedef int nt_v4i __attribute__((ext_vector_type(4)));     // whole lines, written once: streaming hint
        const nt_v4i x = {static_cast<int>(v.x), static_cast<int>(v.y), static_cast<int>(v.z), static_cast<int>(v.w)};
        nt_v4i* dst = reinterpret_cast<nt_v4i*>(out0 + static_cast<uint64_t>(tm * 32 + r) * p.output_stride + c * 16);
        if (p.stream_out) {                                    // ("streaming_stores", igemm_params.h)
          asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(x) : "memory");
        } else {
          *dst = x;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // read back before the other half overwrites it
    if (tm == 0) QNNP_C_STAMP(6);
  }
  QNNP_C_STAMP(7);
#ifdef QNNP_ENABLE_ABLATION
  if (p.trace != nullptr) {                      // when the stores have left the wave
    wait_vmcnt<0>();
    QNNP_C_STAMP_WALL(1);
  }
#endif
  };
  rest(S0{});
}
#undef QNNP_PIN

template <int OPT, bool ALIGNED>
int launch_c(const IgemmParams& p, const dim3& grid, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
#ifdef QNNP_ENABLE_ABLATION
  if constexpr (OPT == 0 && ALIGNED) {
    const char* env = getenv("QNNP_GFX950_ABLATE");
    const int abl = env != nullptr ? atoi(env) : 0;
#define QNNP_ABL_CASE(V) case V: hipLaunchKernelGGL((q8_gemm_mfma_256x256_c_kernel<kRqShift0Ofs, 1, true, 0, V>), grid, dim3(kThreads), 0, stream, p); \
        return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
    switch (abl) {
      QNNP_ABL_CASE(1) QNNP_ABL_CASE(2) QNNP_ABL_CASE(3) QNNP_ABL_CASE(4) QNNP_ABL_CASE(8) QNNP_ABL_CASE(16) QNNP_ABL_CASE(24)
      QNNP_ABL_CASE(27) QNNP_ABL_CASE(32) QNNP_ABL_CASE(59) QNNP_ABL_CASE(31) QNNP_ABL_CASE(64)
      default: break;
    }
#undef QNNP_ABL_CASE
  }
#endif
  if (p.rq.f.shift != 0 && p.rq.f.bounded && p.rq.f.ofs_kind == 2 && !p.rq.full_range) {
    // bounded accumulators, shift >= 1, a clamp other than [0, 255] (requant_dispatch_ofs knows the bounded form for the common
    // clamp only): the bounded sequence with the clamp class picked here
    if (p.rq.zp_late == 0) hipLaunchKernelGGL((q8_gemm_mfma_256x256_c_kernel<kRqBoundedOfs, 1, ALIGNED, OPT>), grid, dim3(kThreads), 0, stream, p);
    else hipLaunchKernelGGL((q8_gemm_mfma_256x256_c_kernel<kRqBoundedOfs, 2, ALIGNED, OPT>), grid, dim3(kThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  }
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    if constexpr (decltype(full)::value) {
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_c_kernel<kSeq, 0, ALIGNED, OPT>), grid, dim3(kThreads), 0, stream, p);
    } else if (p.rq.zp_late == 0) {             // zero point folded (or zero): no add behind the clamp
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_c_kernel<kSeq, 1, ALIGNED, OPT>), grid, dim3(kThreads), 0, stream, p);
    } else {
      hipLaunchKernelGGL((q8_gemm_mfma_256x256_c_kernel<kSeq, 2, ALIGNED, OPT>), grid, dim3(kThreads), 0, stream, p);
    }
    rc = hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  });
  return rc;
}

}  // namespace

/* p as the general kernels get it, with the centred image */
bool gemm256c_supported(const IgemmParams& p, uint32_t vec)
{
  if (p.offsets != nullptr) {                   // strided 1x1 convolution through the table: absolute 32-bit lane offsets
    if (p.offsets_dense == 0 || p.rows_per_image == 0) return false;
    const uint64_t images = (static_cast<uint64_t>(p.rows) + p.rows_per_image - 1) / p.rows_per_image;
    if (images * p.image_stride + p.k_pad >= (1ull << 32)) return false;
  }
  return vec == 16 && p.a_flip != 0 && p.bias2u != nullptr && p.store_mode == 2 &&
         p.k_total == p.k_pad && p.k_pad % kBK == 0 && p.k_pad / kBK >= 2 * kRing && p.n_pad % kBN == 0 &&
         p.k_pad <= (1u << 22) && static_cast<uint64_t>(p.input_stride) * 256u < (1ull << 32) &&
         p.residual == nullptr && p.rows >= 1;
}

/* `p` must carry the CENTRED weight image, its bias pair table and a_flip (q8igemm.hip).
 * opt: 0 = the product; 2 = the A/B structure (kernel comment) */
int gemm256c_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, uint32_t opt)
{
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = p.n_pad / kBN;
  const dim3 grid(tiles_m * tiles_n, groups, 1);
  IgemmParams pm = p;
  // x / tiles_n == hi32(x * magic) for x < 2^32 / tiles_n (the tile ids); 0 stands for tiles_n == 1
  pm.tiles_n_magic = tiles_n == 1 ? 0u : static_cast<uint32_t>((1ull << 32) / tiles_n) + 1u;
  const bool aligned = (p.k_pad / kBK) % kRing == 0;
  switch (opt) {
    case 0: *name = "q8_gemm_mfma_256x256_c"; return aligned ? launch_c<0, true>(pm, grid, stream) : launch_c<0, false>(pm, grid, stream);
#ifdef QNNP_ENABLE_ABLATION                     // (the A/B structure that lost: measurement builds only)
    case 2: *name = "q8_gemm_mfma_256x256_c_burst"; return aligned ? launch_c<2, true>(pm, grid, stream) : launch_c<2, false>(pm, grid, stream);
#endif
    default: return QNNP_HIP_EINVAL;
  }
}

}  // namespace qnnp
