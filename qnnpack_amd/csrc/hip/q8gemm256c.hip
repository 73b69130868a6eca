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
 * What else differs from the lean flavour there (each an A/B item of round 4):
 *   - the requantization sequence is a TEMPLATE argument chosen by the launcher (the kernel has no branch on it); with
 *     no row term the accumulators start from biasc + 2^31 and the offset forms of requant_math.h apply as they are;
 *   - RING = 5: the whole 160 KiB of LDS as five 32 KiB stages (the DMA runs one tile further ahead);
 *   - a barrier-free TAIL: the last RING - 1 K tiles are resident once the last LDS-DMA has landed, so one final
 *     vmcnt(0) + barrier releases the waves to run to the end on their own; with TAIL = 1 the older wave of each SIMD
 *     (waves 0-3) takes the matrix pipe first (s_setprio), finishes early and requantizes / stores its tile while the
 *     younger wave (4-7) multiplies -- the epilogue of one wave under the MFMAs of the other instead of both epilogues
 *     queueing on the SIMD's issue port behind an idle matrix pipe;
 *   - the folded bias arrives by inline-asm loads issued right behind the first tile's LDS-DMA (hipcc cannot see the
 *     LDS-DMA on the vmcnt queue: a visible load would be waited for with vmcnt(0), i.e. behind the whole ring);
 *   - the output tile is staged through the two ring slots that are free during the tail, half a wave tile at a time.
 *
 * Requirements (gemm256c_supported): plain GEMM, K % 64 == 0, K >= 128 * RING, N padded to 256, 16-byte aligned rows and
 * outputs (store_mode 2), a bias pair table.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

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

/* LDS-DMA, saddr form (q8gemm256.hip): 16 bytes per lane from base + lane_offset to m0 + lane * 16 */
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

/* 16 bytes per lane into registers, asynchronously: the result is valid behind bias_wait() only */
template <int OFFSET>
__device__ __forceinline__ v4i load16_async(const uint8_t* base, uint32_t lane_offset)
{
  v4i r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(r) : "v"(lane_offset), "s"(base), "n"(OFFSET));
  return r;
}

#define QNNP_PIN() __builtin_amdgcn_sched_barrier(0)

/*
 * SEQ / FULL: rounding sequence and clamp class of the requantization (requant.hip.h), chosen by the launcher.
 * RING: LDS stages (4 or 5). TAIL: 0 = barrier-free tail, both waves of a SIMD at equal priority; 1 = the older wave first.
 */
template <int SEQ, bool FULL, int RING, int TAIL>
__global__ __launch_bounds__(kThreads, 2)
void q8_gemm_mfma_256x256_c_kernel(const IgemmParams p)
{
  static_assert(RING == 4 || RING == 5, "ring of four or five 32 KiB stages");
  static_assert(SEQ == kRqShift0Ofs || SEQ == kRqBoundedOfs || SEQ == kRqGeneral, "offset forms, or the general one");
  constexpr int kGroups = (RING + 1) / 2;        // address registers per fragment: a ds_read immediate reaches 64 KiB = 2 stages

  __shared__ __attribute__((aligned(16))) uint8_t lds[RING * kStage];    // the ONE LDS object (guide 5, trap 4a)

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave >> 1;       // 64-row slice
  const uint32_t wn = wave & 1u;       // 128-channel half
  const uint32_t g = blockIdx.y;

  // Workgroup -> tile (q8gemm256.hip): contiguous logical ids per XCD, bands of four row tiles.
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = p.n_pad / kBN;
  uint32_t m_tile, n_tile;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    const uint32_t logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    constexpr uint32_t kBand = 4;
    const uint32_t per_band = kBand * tiles_n;
    const uint32_t band = logical / per_band;
    const uint32_t within = logical - band * per_band;
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
  const uint8_t* a_base = scalar_ptr(p.input + static_cast<uint64_t>(m_tile * kBM) * p.input_stride + static_cast<uint64_t>(g) * p.kc);
  uint32_t a_voff[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const uint32_t L = i * kThreads + tid;
    const uint32_t r = L >> 2;
    const uint32_t chunk = (L & 3u) ^ ((r >> 2) & 3u);
    uint32_t m = m_tile * kBM + r;
    if (m >= p.rows) m = p.rows - 1;             // clamp: results of those rows are never stored
    a_voff[i] = (m - m_tile * kBM) * p.input_stride + chunk * 16;
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

  // Accumulators start at the folded bias (+ 2^31 for the offset forms): lane l holds, in register r of tile tn,
  // channel (nb0 + wn * 4 + tn) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
  const int32_t* bias_tab = SEQ == kRqGeneral ? p.bias2 : p.bias2u;
  const uint8_t* bias_base = scalar_ptr(reinterpret_cast<const uint8_t*>(
      bias_tab + static_cast<uint64_t>(g) * p.n_pad + (nb0 + wn * kTN) * 32));
  const uint32_t bias_voff = (lane >> 5) * 16;
  v4i braw[kTN][4];
#define QNNP_BIAS_LOAD(TN, RG) braw[TN][RG] = load16_async<(TN) * 128 + (RG) * 32>(bias_base, bias_voff)
  QNNP_BIAS_LOAD(0, 0); QNNP_BIAS_LOAD(0, 1); QNNP_BIAS_LOAD(0, 2); QNNP_BIAS_LOAD(0, 3);
  QNNP_BIAS_LOAD(1, 0); QNNP_BIAS_LOAD(1, 1); QNNP_BIAS_LOAD(1, 2); QNNP_BIAS_LOAD(1, 3);
  QNNP_BIAS_LOAD(2, 0); QNNP_BIAS_LOAD(2, 1); QNNP_BIAS_LOAD(2, 2); QNNP_BIAS_LOAD(2, 3);
  QNNP_BIAS_LOAD(3, 0); QNNP_BIAS_LOAD(3, 1); QNNP_BIAS_LOAD(3, 2); QNNP_BIAS_LOAD(3, 3);
#undef QNNP_BIAS_LOAD

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
    if (h & 1) {
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
    acc[tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.w[tn], f.a[tm], acc[tm][tn], 0, 0, 0);
  };

  // ---- prologue, part 2: tile 0 and the bias have landed (loads complete in issue order) ----
  asm volatile("s_waitcnt vmcnt(%16)"
               : "+v"(braw[0][0]), "+v"(braw[0][1]), "+v"(braw[0][2]), "+v"(braw[0][3]),
                 "+v"(braw[1][0]), "+v"(braw[1][1]), "+v"(braw[1][2]), "+v"(braw[1][3]),
                 "+v"(braw[2][0]), "+v"(braw[2][1]), "+v"(braw[2][2]), "+v"(braw[2][3]),
                 "+v"(braw[3][0]), "+v"(braw[3][1]), "+v"(braw[3][2]), "+v"(braw[3][3])
               : "n"((RING - 1) * kDma) : "memory");
  __builtin_amdgcn_s_barrier();
  Frags fa, fb;
  read_frags_slot(0, 0, fa);
#pragma unroll
  for (int tn = 0; tn < kTN; tn++) {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
#pragma unroll
      for (int tm = 0; tm < kTM; tm++) {
        acc[tm][tn][rg * 4 + 0] = braw[tn][rg].x;
        acc[tm][tn][rg * 4 + 1] = braw[tn][rg].y;
        acc[tm][tn][rg * 4 + 2] = braw[tn][rg].z;
        acc[tm][tn][rg * 4 + 3] = braw[tn][rg].w;
      }
    }
  }
  flip_part(fa, 0); flip_part(fa, 1); flip_part(fa, 2); flip_part(fa, 3);
  settle_w(fa);

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
  auto iteration = [&](auto p1f_c, auto more_c, auto p2f_c, auto sync_c, auto known_c, uint32_t kt, uint32_t slot) __attribute__((always_inline)) {
    constexpr bool P1F = decltype(p1f_c)::value;
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool P2F = decltype(p2f_c)::value;
    constexpr int SYNC = decltype(sync_c)::value;
    constexpr bool KNOWN = decltype(known_c)::value;     // `slot` is a literal
    const uint32_t prev_slot = slot == 0 ? RING - 1 : slot - 1;
    const uint32_t next_slot = slot + 1 == RING ? 0 : slot + 1;

    QNNP_PIN();
    if constexpr (KNOWN) read_frags_slot(slot, 1, fb); else read_frags_rt(slot, 1, fb);
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      if constexpr (P1F && KNOWN) {
        if (i % 4 == 0) { dma16_set_m0(piece_dst(2 + i / 4, prev_slot)); QNNP_PIN(); }
      }
      mma(fa, i);
      QNNP_PIN();
      if constexpr (P1F && KNOWN) {
        if (i % 4 == 0) { dma16_saddr_m0_set(piece_src(kt + RING - 1, 2 + i / 4), piece_off(2 + i / 4)); QNNP_PIN(); }
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
      __builtin_amdgcn_s_barrier();
      if constexpr (TAIL == 1) {
        // from here on nothing synchronizes the waves: the older wave of each SIMD takes the matrix pipe first
        if (wave < 4) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
      }
    }
    QNNP_PIN();

    if constexpr (MORE) {
      if constexpr (KNOWN) read_frags_slot(next_slot, 0, fa); else read_frags_rt(next_slot, 0, fa);
    }
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      if constexpr (P2F && KNOWN) {
        if (i % 4 == 0) { dma16_set_m0(piece_dst(i / 4, slot)); QNNP_PIN(); }
      }
      mma(fb, i);
      QNNP_PIN();
      if constexpr (P2F && KNOWN) {
        if (i % 4 == 0) { dma16_saddr_m0_set(piece_src(kt + RING, i / 4), piece_off(i / 4)); QNNP_PIN(); }
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

  // (the launcher guarantees ktiles >= 2 * RING)
  iteration(F{}, T{}, T{}, Sync1{}, T{}, 0u, 0u);         // the prologue staged pieces 2, 3 of tile RING - 1 already
  uint32_t kt = 1;
  for (; kt + (RING - 1) + RING < ktiles; kt += RING) {   // steady state, ring slots as literals (kt % RING == 1 here)
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt, 1u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 1, 2u);
    iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 2, 3u);
    if constexpr (RING == 5) {
      iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 3, 4u);
      iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 4, 0u);
    } else {
      iteration(T{}, T{}, T{}, Sync1{}, T{}, kt + 3, 0u);
    }
  }
  uint32_t slot = 1;                                      // == kt % RING
  auto advance = [&]() __attribute__((always_inline)) { kt++; slot = slot + 1 == RING ? 0 : slot + 1; };
  while (kt + RING < ktiles) {                            // steady state, run-time slot
    iteration(T{}, T{}, T{}, Sync1{}, F{}, kt, slot);
    advance();
  }
  iteration(T{}, T{}, F{}, Sync1{}, F{}, kt, slot);       // kt == ktiles - RING: the last pieces of the last tile
  advance();
  iteration(F{}, T{}, F{}, Sync2{}, F{}, kt, slot);       // kt == ktiles - RING + 1: the final wait + barrier
  advance();
  while (kt + 1 < ktiles) {                               // tail: everything resident, no barriers
    iteration(F{}, T{}, F{}, Sync0{}, F{}, kt, slot);
    advance();
  }
  iteration(F{}, F{}, F{}, Sync0{}, F{}, kt, slot);       // last tile

  // ---- fused epilogue: Q31 requantize in registers -> half a wave tile at a time through LDS -> whole 128-byte lines ----
  if constexpr (TAIL == 1) __builtin_amdgcn_s_setprio(0);
  // The two ring slots nobody reads after the final barrier: those of tiles kf = ktiles - RING + 1 and kf - 1.
  const uint32_t kf = ktiles - RING + 1;
  const uint32_t slot_a = kf % RING;
  const uint32_t slot_b = (kf + RING - 1) % RING;
  uint8_t* image = lds + (wave < 4 ? slot_a : slot_b) * kStage + (wave & 3u) * kImageBytes;
  const int4 no_bias[4] = {};                    // (the bias is already in the accumulators)
  const uint32_t m0 = m_tile * kBM + wm * (kTM * 32);
  const uint32_t n0 = (nb0 + wn * kTN) * 32;
  uint8_t* out0 = p.output + static_cast<uint64_t>(m0) * p.output_stride + static_cast<uint64_t>(g) * p.n + n0;
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      igemm_stage_tile_rq<SEQ, FULL, false, 1>(acc[tm][tn], no_bias, 0, image + (lane & 31u) * kImagePitch, tn * 32, frag_khalf, p.rq);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
#pragma unroll
    for (int i = 0; i < (32 * kTN * 2) / 64; i++) {
      const uint32_t idx = i * 64 + lane;
      const uint32_t r = idx / (kTN * 2);
      const uint32_t c = idx % (kTN * 2);
      const uint4 v = *reinterpret_cast<const uint4*>(image + r * kImagePitch + c * 16);
      if (m0 + tm * 32 + r < p.rows && n0 + c * 16 < p.n) {
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
  }
}
#undef QNNP_PIN

template <int RING, int TAIL>
int launch_c(const IgemmParams& p, const dim3& grid, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    constexpr bool kFull = decltype(full)::value;
    hipLaunchKernelGGL((q8_gemm_mfma_256x256_c_kernel<kSeq, kFull, RING, TAIL>), grid, dim3(kThreads), 0, stream, p);
    rc = hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  });
  return rc;
}

}  // namespace

/* p as the general kernels get it; `ring` 4 or 5 */
bool gemm256c_supported(const IgemmParams& p, uint32_t vec, uint32_t ring)
{
  return vec == 16 && p.offsets == nullptr && p.a_flip != 0 && p.bias2u != nullptr && p.store_mode == 2 &&
         p.k_total == p.k_pad && p.k_pad % kBK == 0 && p.k_pad / kBK >= 2 * ring && p.n_pad % kBN == 0 &&
         p.k_pad <= (1u << 22) && static_cast<uint64_t>(p.input_stride) * 256u < (1ull << 32) &&
         p.residual == nullptr && p.rows >= 1;
}

/* `p` must carry the CENTRED weight image, its bias pair table and a_flip (q8igemm.hip) */
int gemm256c_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, uint32_t ring, uint32_t tail)
{
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = p.n_pad / kBN;
  const dim3 grid(tiles_m * tiles_n, groups, 1);
  if (ring == 5) {
    *name = tail ? "q8_gemm_mfma_256x256_c5_skew" : "q8_gemm_mfma_256x256_c5";
    return tail ? launch_c<5, 1>(p, grid, stream) : launch_c<5, 0>(p, grid, stream);
  }
  *name = tail ? "q8_gemm_mfma_256x256_c4_skew" : "q8_gemm_mfma_256x256_c4";
  return tail ? launch_c<4, 1>(p, grid, stream) : launch_c<4, 0>(p, grid, stream);
}

}  // namespace qnnp
