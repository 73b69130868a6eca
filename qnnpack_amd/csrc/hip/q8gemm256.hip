/*
 * q8gemm256.hip -- the large-problem uint8 GEMM / implicit-GEMM kernel:
 * 256x256 output tile per workgroup, operands staged by LDS-DMA, int8 MFMA.
 *
 * Same arithmetic and operand roles as q8igemm.hip (which see): it replaces the
 * same reference microkernels (src/q8gemm/4x4c2-sse2.c:14-318,
 * src/q8conv/4x4c2-sse2.c:14-273) for shapes big enough to be MFMA-bound, i.e.
 * BASELINE.json configs[1] (q8gemm M=N=K=4096).
 *
 * Structure (per workgroup, 8 waves as 4 (rows) x 2 (channels), 64 x 128 outputs per wave):
 *   - K advances 128 bytes per tile; two LDS stages of {activations 256x128 B,
 *     weights 256x128 B} = 128 KiB, filled with global_load_lds (16 B per lane, no
 *     VGPR round trip), one barrier per K tile, the next tile's DMA in flight while
 *     the current one is multiplied.
 *   - activations keep the caller's row-major image in full 128-byte lines; the
 *     16-byte chunk index is XOR-swizzled with (row >> 1) & 7 -- applied to the DMA
 *     SOURCE address and to the ds_read_b128 address (the LDS-DMA destination is
 *     lane-linear by construction), so fragment reads are bank-conflict free.
 *   - weights arrive already as MFMA fragments (pack.h), copied verbatim: fragment
 *     reads are linear.
 *   - uint8 -> int8 re-centring of activations is one v_xor per fragment dword after
 *     the LDS read; the per-row sum of a' needed for the kernel-zero-point term is
 *     taken with v_dot4 on those same registers, each of the 2 channel-waves doing
 *     one half of K (the wave-specific rotation of the K sub-step order makes that
 *     branch-free), combined through LDS at the end. Fragment reads of sub-step j+1
 *     are issued before the MFMAs of sub-step j.
 *   - convolution: each lane's DMA source comes from the device offset table
 *     (table entries for tile t+1 are fetched while tile t is multiplied); padding
 *     taps and K padding read constant 16-byte lines of the fill table.
 *   - epilogue fused in registers: + bias2 + row term -> Q31 requantize -> clamp ->
 *     4 channels per dword.
 *
 * Requirements (checked by gemm256_supported): 16-byte aligned activations with
 * group_input_channels % 16 == 0 and pixel stride % 16 == 0.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "igemm_epilogue.cuh"
#include "igemm_params.h"
#include "requant.cuh"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kBM = 256;
constexpr int kBN = 256;
constexpr int kBK = 64;                        // bytes of K per tile (two 32-deep MFMA sub-steps)
constexpr int kStages = 4;                     // LDS ring: tile t+3 is being fetched while tile t is multiplied
constexpr int kATile = kBM * kBK;              // 16 KiB
constexpr int kWTile = kBN * kBK;              // 16 KiB
constexpr int kStage = kATile + kWTile;        // 32 KiB
constexpr int kThreads = 512;
constexpr uint32_t kFlip = 0x80808080u;

__device__ __forceinline__ void dma16(const uint8_t* src, uint8_t* lds_wave_base)
{
  // 16 bytes per lane, LDS destination = wave-uniform base + lane * 16
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*) src,
      (__attribute__((address_space(3))) void*) lds_wave_base, 16, 0, 0);
}

constexpr int kWM = 4;                         // waves along rows
constexpr int kWN = 2;                         // waves along channels
constexpr int kTM = kBM / (kWM * 32);          // 2 MFMA tiles of 32 rows per wave
constexpr int kTN = kBN / (kWN * 32);          // 4 MFMA tiles of 32 channels per wave
static_assert(kWM * kWN * 64 == kThreads, "wave layout");
static_assert(kWN == 2 && kBK == 64, "row-sum split: each channel-wave owns one of the two K sub-steps");

// ABL: measurement-only ablation mask (builds with -DQNNP_ENABLE_ABLATION, env QNNP_GFX950_ABLATE);
// 0 in the product. 1 = no requantization in the epilogue, 2 = no recentring / row sums,
// 4 = no MFMA, 8 = no LDS-DMA after the prologue, 16 = no fragment reads after the first tile,
// 32 = no per-tile wait + barrier.
template <bool IS_CONV, int ABL = 0>
__global__ __launch_bounds__(kThreads, 2)
void q8_gemm_mfma_256x256_kernel(const IgemmParams p)
{
  // single LDS object: ring of {A, W} tiles, then kWN x 256 partial row sums
  __shared__ __attribute__((aligned(16))) uint8_t lds[kStages * kStage + kWN * kBM * 4];
  int32_t* lds_rowsum = reinterpret_cast<int32_t*>(lds + kStages * kStage);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave / kWN;      // 0..3: 64-row quarter
  const uint32_t wn = wave % kWN;      // 0..1: 128-channel half
  const uint32_t g = blockIdx.y;

  // Workgroup -> tile: hardware places block b on XCD b % 8, so give each XCD a contiguous run of
  // logical ids, and walk logical ids in bands of 4 row-tiles (channel-tile fastest inside a band)
  // so the ~32 co-resident tiles of an XCD form a compact patch that shares row panels and weight
  // panels in that XCD's L2.
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
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
    m_tile = band * kBand + within % rows_in_band;
    n_tile = within / rows_in_band;
  }

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t ktiles = p.k_pad / kBK;                                 // k_pad is a multiple of 64
  const uint8_t* pad_k = p.fill_table + 0x80 * 16;                       // a' == 0
  const uint8_t* pad_zp = p.fill_table + (p.izp_fill & 0xFFu) * 16;      // a == input zero point
  const uint8_t* pad_w = p.fill_table;                                   // w' == 0

  // ---- activation DMA assignment: 2 chunks per thread, LDS linear index L = i*512 + tid ----
  // LDS image of an activation tile: [256 rows][4 chunks of 16 B], chunk slot s of row r holds logical
  // chunk s ^ ((r >> 2) & 3): conflict-free ds_read_b128 fragment reads (4 rows share a 256-byte bank row).
  const uint8_t* a_row[2];      // gemm: row base (+ group); conv: image base (+ group)
  const int32_t* a_offs[2];     // conv: offset-table row of this pixel
  uint32_t a_chunk[2];          // logical 16-byte chunk (0..3) this lane fetches for its slot
#pragma unroll
  for (int i = 0; i < 2; i++) {
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

  // ---- weight DMA assignment: 16 fragments of 1 KiB per tile, 2 per wave ----
  const uint32_t nb0 = n_tile * (kBN / 32);
  const uint8_t* w_group = reinterpret_cast<const uint8_t*>(p.packed_w) +
      static_cast<uint64_t>(g) * nblocks * kblocks * 1024 + lane * 16;

  // One K tile = exactly 4 LDS-DMA instructions per thread (the vmcnt arithmetic depends on it):
  // pieces 0,1 = activation chunks, pieces 2,3 = weight fragments.
  auto stage_piece = [&](uint32_t kt, int piece) {
    uint8_t* a_dst = lds + (kt % kStages) * kStage;
    uint8_t* w_dst = a_dst + kATile;
    if (piece < 2) {
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
      const int i = piece - 2;
      const uint32_t F = i * 8 + wave;             // fragment slot: (32-channel block, 32-deep K block)
      const uint32_t nb = nb0 + (F >> 1);
      const uint32_t kb = kt * 2 + (F & 1u);
      const uint8_t* src = nb < nblocks ? w_group + (static_cast<uint64_t>(nb) * kblocks + kb) * 1024 : pad_w;
      dma16(src, w_dst + F * 1024);
    }
  };
  auto stage = [&](uint32_t kt) {
#pragma unroll
    for (int piece = 0; piece < 4; piece++) stage_piece(kt, piece);
  };

  v16i acc[kTM][kTN];
#pragma unroll
  for (int tm = 0; tm < kTM; tm++)
#pragma unroll
    for (int tn = 0; tn < kTN; tn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0;
  int32_t rs[kTM];
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
  auto read_frags = [&](const uint8_t* st, uint32_t ksub, Frags& f) {
#pragma unroll
    for (int tn = 0; tn < kTN; tn++) {
      f.w[tn] = *reinterpret_cast<const v4i*>(st + w_fbase + (tn * 2 + ksub) * 1024);
    }
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      f.a[tm] = *reinterpret_cast<const v4i*>(st + (a_fbase[tm] ^ (ksub << 5)));
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
  auto rowsum = [&](const Frags& f) {
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      int32_t s = rs[tm];
      s = __builtin_amdgcn_sdot4(f.a[tm].x, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4(f.a[tm].y, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4(f.a[tm].z, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4(f.a[tm].w, 0x01010101, s, false);
      rs[tm] = s;
    }
  };
  auto mma = [&](const Frags& f, int i) {       // i = 0..7 -> (tm, tn)
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
  auto rowsum_part = [&](const Frags& f, int h) {
    const int tm = h >> 1;
    if (h & 1) {
      rs[tm] = __builtin_amdgcn_sdot4(f.a[tm].z, 0x01010101, rs[tm], false);
      rs[tm] = __builtin_amdgcn_sdot4(f.a[tm].w, 0x01010101, rs[tm], false);
    } else {
      rs[tm] = __builtin_amdgcn_sdot4(f.a[tm].x, 0x01010101, rs[tm], false);
      rs[tm] = __builtin_amdgcn_sdot4(f.a[tm].y, 0x01010101, rs[tm], false);
    }
  };
#define QNNP_PIN() __builtin_amdgcn_sched_barrier(0)

  // Counted wait: tile `kt` has landed when at most the LDS-DMA groups of the tiles issued after it
  // (4 instructions each, completing in issue order) are still outstanding.
  auto wait_tile = [&](uint32_t later_tiles_in_flight) {
    if (later_tiles_in_flight >= 2) {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else if (later_tiles_in_flight == 1) {
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };

  // Each wave walks the two K sub-steps of a tile starting at sub-step wn (integer sums commute),
  // so the sub-step whose row sums it owns is always its FIRST one: the split costs no branch.
  const uint32_t s_first = wn;
  const uint32_t s_second = wn ^ 1u;

  // ---- prologue: tiles 0..2 in flight, tile 0 resident, its first fragments recentred ----
#pragma unroll
  for (int t = 0; t < kStages - 1; t++) {
    if (static_cast<uint32_t>(t) < ktiles) stage(t);
  }
  wait_tile(min(2u, ktiles - 1));
  __builtin_amdgcn_s_barrier();
  Frags fa, fb;
  read_frags(lds, s_first, fa);
  if (!(ABL & 2)) {
    flip(fa);
    rowsum(fa);
  }
  settle_w(fa);

  /*
   * Software pipeline, two phases per K tile, every phase = 8 MFMAs whose shadow hides the other work:
   *   phase 1: multiply fa = (tile kt, first sub-step)  | read fb = (tile kt, second sub-step), recentre fb
   *   -- counted vmcnt + raw barrier: tile kt+1 resident, ring slot of tile kt-1 free --
   *   phase 2: multiply fb                               | read fa = (tile kt+1, first sub-step), recentre +
   *                                                        row-sum fa, issue the LDS-DMA of tile kt+3
   * A __syncthreads() would drain the DMA queue (vmcnt(0)); the raw barrier keeps two tiles in flight.
   */
  // MORE: a tile kt+1 exists; FETCH: a tile kt+3 exists (compile-time so that the steady-state loop body
  // is one straight-line block: the wait-count pass is conservative at every control-flow join).
  auto iteration = [&](auto more_c, auto fetch_c, uint32_t kt, uint32_t later_in_flight) {
    constexpr bool MORE = decltype(more_c)::value;
    constexpr bool FETCH = decltype(fetch_c)::value;
    const uint8_t* st = lds + (kt % kStages) * kStage;

    // ---- phase 1: the order below is pinned (sched_barrier) so the reads, the recentring and the
    //      MFMAs interleave the way the pipeline needs instead of the way the scheduler clusters them
    QNNP_PIN();
    if (!(ABL & 16)) read_frags(st, s_second, fb);
    QNNP_PIN();
    if (!(ABL & 4)) { mma(fa, 0); mma(fa, 1); mma(fa, 2); mma(fa, 3); }
    QNNP_PIN();
#pragma unroll
    for (int h = 0; h < 4; h++) {
      if (!(ABL & 2)) flip_part(fb, h);
      QNNP_PIN();
      if (!(ABL & 4)) mma(fa, 4 + h);
      QNNP_PIN();
    }
    if (!(ABL & 16)) settle_w(fb);
    QNNP_PIN();

    if constexpr (MORE && !(ABL & 32)) {
      if constexpr (FETCH) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // tile kt+1 resident; tile kt+2 still flies
      } else {
        wait_tile(later_in_flight);
      }
      __builtin_amdgcn_s_barrier();
    }
    QNNP_PIN();

    // ---- phase 2 ----
    if constexpr (MORE) {
      if (!(ABL & 16)) read_frags(lds + ((kt + 1) % kStages) * kStage, s_first, fa);
    }
    QNNP_PIN();
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (!(ABL & 4)) mma(fb, i);
      QNNP_PIN();
      if constexpr (FETCH) {
        if (!(ABL & 8)) stage_piece(kt + kStages - 1, i);
      }
      QNNP_PIN();
    }
#pragma unroll
    for (int h = 0; h < 4; h++) {
      if (!(ABL & 4)) mma(fb, 4 + h);
      QNNP_PIN();
      if constexpr (MORE) {
        if (!(ABL & 2)) {
          flip_part(fa, h);
          rowsum_part(fa, h);
        }
      }
      QNNP_PIN();
    }
    if constexpr (MORE) {
      if (!(ABL & 16)) settle_w(fa);
    }
    QNNP_PIN();
    if (ABL & 4) {
      asm volatile("" :: "v"(fa.a[0]), "v"(fa.a[1]), "v"(fa.w[0]), "v"(fa.w[1]), "v"(fa.w[2]), "v"(fa.w[3]));
      asm volatile("" :: "v"(fb.a[0]), "v"(fb.a[1]), "v"(fb.w[0]), "v"(fb.w[1]), "v"(fb.w[2]), "v"(fb.w[3]));
    }
  };

  uint32_t kt = 0;
  for (; kt + kStages - 1 < ktiles; kt++) {                 // steady state
    iteration(std::true_type{}, std::true_type{}, kt, 1u);
  }
  for (; kt + 1 < ktiles; kt++) {                            // drain: no tile left to fetch
    iteration(std::true_type{}, std::false_type{}, kt, min(1u, ktiles - 2 - kt));
  }
  if (kt < ktiles) {
    iteration(std::false_type{}, std::false_type{}, kt, 0u);  // last tile
  }
#undef QNNP_PIN

  // ---- bias for this lane's 4-channel groups (issued before the barrier so the latency hides) ----
  int4 bias4[kTN][4];
#pragma unroll
  for (int tn = 0; tn < kTN; tn++) {
    uint32_t nb = nb0 + wn * kTN + tn;
    if (nb >= nblocks) nb = nblocks - 1;       // clamped blocks are never stored
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const uint32_t ncol = nb * 32 + rg * 8 + frag_khalf * 4;
      bias4[tn][rg] = *reinterpret_cast<const int4*>(p.bias2 + static_cast<uint64_t>(g) * p.n_pad + ncol);
    }
  }

  // ---- combine the row sums: 2 K halves (lane, lane+32) then the 2 channel-waves ----
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
    int32_t s = rs[tm];
    s += __shfl_xor(s, 32);
    if (lane < 32) lds_rowsum[wn * kBM + wm * (kTM * 32) + tm * 32 + lane] = s;
  }
  __syncthreads();

  // ---- fused epilogue (igemm_epilogue.cuh); the requantization flavour is chosen once ----
  requant_dispatch(p.rq, [&](auto shift0, auto full) {
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) {
      const uint32_t row = frag_row0 + tm * 32;
      const uint32_t m = m_tile * kBM + row;
      const int32_t rowterm = p.row_coeff * (lds_rowsum[row] + lds_rowsum[kBM + row]);
      uint8_t* out_row = p.output + static_cast<uint64_t>(m) * p.output_stride + static_cast<uint64_t>(g) * p.n;
#pragma unroll
      for (int tn = 0; tn < kTN; tn++) {
        const uint32_t nb = nb0 + wn * kTN + tn;
        if (nb >= nblocks) continue;       // wave-uniform
        igemm_store_tile<decltype(shift0)::value, decltype(full)::value, (ABL & 1) != 0>(
            acc[tm][tn], bias4[tn], rowterm, out_row, nb * 32, frag_khalf, m < p.rows, p);
      }
    }
  });
}

}  // namespace

bool gemm256_supported(const IgemmParams& p, uint32_t vec)
{
  return vec == 16 && p.fill_table != nullptr && p.rows >= 1 && p.k_total % 16 == 0;
}

int gemm256_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name)
{
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  const dim3 grid(tiles_m * tiles_n, groups, 1);
  const dim3 block(kThreads, 1, 1);
#ifdef QNNP_ENABLE_ABLATION
  if (p.offsets == nullptr) {
    const char* env = getenv("QNNP_GFX950_ABLATE");
    const int abl = env != nullptr ? atoi(env) : 0;
    *name = "q8_gemm_mfma_256x256";
    switch (abl) {
      case 0: break;
#define QNNP_ABL_CASE(V) case V: hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<false, V>), grid, block, 0, stream, p); \
        return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
      QNNP_ABL_CASE(1) QNNP_ABL_CASE(2) QNNP_ABL_CASE(3) QNNP_ABL_CASE(4) QNNP_ABL_CASE(8) QNNP_ABL_CASE(11)
      QNNP_ABL_CASE(12) QNNP_ABL_CASE(15) QNNP_ABL_CASE(27) QNNP_ABL_CASE(31) QNNP_ABL_CASE(7)
      QNNP_ABL_CASE(59) QNNP_ABL_CASE(63) QNNP_ABL_CASE(43) QNNP_ABL_CASE(35)
#undef QNNP_ABL_CASE
      default: break;
    }
  }
#endif
  if (p.offsets != nullptr) {
    *name = "q8_gemm_mfma_256x256_conv";
    hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<true>), grid, block, 0, stream, p);
  } else {
    *name = "q8_gemm_mfma_256x256";
    hipLaunchKernelGGL((q8_gemm_mfma_256x256_kernel<false>), grid, block, 0, stream, p);
  }
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

}  // namespace qnnp
