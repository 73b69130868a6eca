/*
 * q8gemm128x.hip -- the zero-point-centred uint8 GEMM on v_mfma_i32_16x16x64_i8 in 128 x 128 tiles of four waves (round 6).
 *
 * Role. The 1x1 convolutions and fully connected layers that are neither long-and-wide (q8gemm256x.hip: 256 x 256 tiles, one
 * workgroup per CU, >= 8 K tiles) nor output-streaming with a short reduction (q8pwconv.hip): MobileNetV2's late pointwise
 * layers (14 x 14 and 7 x 7 maps, K = 192 ... 960, N = 64 ... 1280: bench/convolution.cc:496-536) and ResNet-50's 7 x 7 / 14 x 14
 * bottleneck convolutions (:690-718). The reference runs them through the same q8gemm microkernel as everything else
 * (src/q8gemm/4x4c2-sse2.c:14-318 under compute_q8gemm, src/operator-run.c:39-70, 797-802). On this chip they are a few
 * GigaOP and a few MB each -- 1-3 us of matrix or memory time -- and a launch of the kernels above costs 6-12 us for them:
 * fixed cost (a cold prologue in front of 8 waves that all start and end together) is what this kernel removes:
 *   - 128 x 128 tiles, 256 threads, 64 KiB of LDS: two workgroups per CU, so one's prologue and epilogue run under the
 *     other's K loop, and 2-4 x the workgroups of the 256-wide tiling for the same problem;
 *   - the same operand path as q8gemm256x.hip: LDS-DMA ring of four 16-KiB stages (128 rows x 64 B of activations, row-major
 *     with the (row & 8) chunk swizzle; four 32-channel x 64-byte weight fragments of pack.h's image), three K tiles ahead;
 *   - 16x16x64 MFMAs (the shape that is cheapest in energy per MAC, tools/ubench_mfma2.hip), 4 x 4 tiles per wave: 64
 *     accumulator registers, 8 ds_read_b128 per 16 MFMAs;
 *   - every K tile count from 1 up: the ring's fill and drain are run-time (uniform) conditions, not code copies;
 *   - epilogue in registers: requantize -> 4 x 4 lane transpose -> one 16-byte store per lane and 16-row block.
 * Algebra, weight image, bias pair and re-centring mask: q8gemm256c.hip (kernel zero point 127 or 128).
 *
 * Requirements (gemm128x_supported): single GEMM (or a strided 1x1 convolution through the dense offset table), K % 64 == 0,
 * 16-byte aligned input rows, dword-aligned output rows (store_mode >= 1), a centred image and its bias pair table.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include <type_traits>

#include "igemm_params.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;
constexpr int kBK = 64;                        // bytes of K per tile = one 16x16x64 step
constexpr int kATile = kBM * kBK;              // 8 KiB
constexpr int kThreads = 256;                  // 4 waves: 2 (rows) x 2 (channels), 64 rows x 16 TN channels per wave
constexpr int kTM = 4;                         // 16-row MFMA tiles per wave
constexpr int kRing = 4;

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

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

/* chunk swizzle of the activation image (q8gemm256x.hip): rows 8..15 of every 16 keep their K chunks in slots c ^ 3 */
__device__ __forceinline__ uint32_t a_swizzle(uint32_t row) { return (row & 8u) != 0 ? 3u : 0u; }

#define QNNP_PIN() __builtin_amdgcn_sched_barrier(0)

/* TN: 16-channel MFMA tiles per wave: 4 = 128-channel workgroup tiles (64 KiB of LDS, two workgroups per CU), 2 = 64-channel ones
 * (48 KiB, three per CU) for channel counts that would leave most of a 128-wide tile empty (64, 96, 160, 320). */
template <int SEQ, int CLAMP, int TN>
__global__ __launch_bounds__(kThreads, 2)
void q8_gemm_mfma_128xN_c16_kernel(const IgemmParams p)
{
  static_assert(SEQ == kRqShift0Ofs || SEQ == kRqBoundedOfs || SEQ == kRqGeneral, "offset forms, or the general one");
  static_assert(TN == 4 || TN == 2, "128- or 64-channel tiles");
  constexpr uint32_t RING = kRing;
  constexpr int kTN = TN;
  constexpr int kBN = 32 * TN;                   // two waves side by side
  constexpr int kWTile = kBN * kBK;              // 8 / 4 KiB
  constexpr int kStage = kATile + kWTile;        // 16 / 12 KiB
  constexpr int kHalf = kTN / 2;                 // weight fragments per phase
  constexpr int kWPieces = kBN / 64;             // LDS-DMA instructions per thread for a weight tile
  constexpr int kDma = 2 + kWPieces;             // ... and for a K tile
  constexpr int kMma = kTM * kHalf;              // MFMAs per phase

  __shared__ __attribute__((aligned(16))) uint8_t lds[kRing * kStage + 4 * 256];    // the ONE LDS object: ring + the waves' bias lines

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave >> 1;       // 64-row half
  const uint32_t wn = wave & 1u;       // channel half
  const uint32_t g = blockIdx.y;

  // Workgroup -> tile: contiguous logical ids per XCD (blockIdx.x round-robins over the 8 XCDs), row tile major -- an XCD's
  // L2 sees a contiguous band of activation rows once and the (small) weight panel of every channel tile.
  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  uint32_t m_tile, n_tile;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    const uint32_t logical = xcd * q + min(xcd, r) + idx;
    m_tile = p.tiles_n_magic != 0 ? __umulhi(logical, p.tiles_n_magic) : logical;     // logical / tiles_n
    n_tile = logical - m_tile * tiles_n;
  }

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t ktiles = p.k_pad / kBK;
  const uint32_t nb0 = n_tile * (kBN / 32);

  // ---- LDS-DMA sources: wave-uniform bases + loop-invariant 32-bit lane offsets ----
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
  // weight piece i, wave w: fragment (channel block nb0 + 2 i + (w >> 1), K block (w & 1)) of the tile; blocks past the image's
  // last one re-read it (their results are never stored)
  const uint8_t* w_base[kWPieces];
#pragma unroll
  for (int i = 0; i < kWPieces; i++) {
    const uint32_t nb = min(nb0 + 2u * i + (wave >> 1), nblocks - 1u);
    w_base[i] = scalar_ptr(reinterpret_cast<const uint8_t*>(p.packed_w) + static_cast<uint64_t>(g) * nblocks * kblocks * 1024 +
        (static_cast<uint64_t>(nb) * kblocks + (wave & 1u)) * 1024);
  }
  const uint32_t w_voff = lane * 16;

  auto stage_a = [&](uint32_t kt, uint32_t slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      dma16_saddr(a_base + static_cast<uint64_t>(kt) * kBK, a_voff[i], lds + slot * kStage + (i * kThreads + wave * 64) * 16);
    }
  };
  auto stage_w = [&](uint32_t kt, uint32_t slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < kWPieces; i++) {
      dma16_saddr(w_base[i] + static_cast<uint64_t>(kt) * 2048, w_voff, lds + slot * kStage + kATile + (i * 4 + wave) * 1024);
    }
  };

  const uint32_t frow = lane & 15u;              // row / channel of a 16 x 16 tile
  const uint32_t fg = lane >> 4;                 // K chunk of the operand; channel quad of the result
  const uint32_t n0 = n_tile * kBN + wn * (kTN * 16);

  // ---- prologue: the first tile, the wave's 16 TN folded biases (+ 2^31 for the offset forms; ONE LDS-DMA instruction, lanes
  //      0..4 TN - 1; channels past the table's end re-read its last quad), the rest of the ring ----
  const uint32_t fill = min(ktiles, RING);
  stage_a(0u, 0u);
  stage_w(0u, 0u);
  uint8_t* bias_line = lds + kRing * kStage + wave * 256;
  if (lane < 4 * kTN) {
    const int32_t* bias_tab = (SEQ == kRqGeneral ? p.bias2 : p.bias2u) + static_cast<uint64_t>(g) * p.n_pad;
    dma16_saddr(scalar_ptr(reinterpret_cast<const uint8_t*>(bias_tab)), min(n0 + lane * 4u, p.n_pad - 4u) * 4u, bias_line);
  }
  for (uint32_t t = 1; t < fill; t++) {
    stage_a(t, t);
    stage_w(t, t);
  }

  // ---- fragment addresses ----
  const uint32_t a_off = (wm * 64 + frow) * kBK + ((fg ^ a_swizzle(frow)) << 4);                       // + tm * 1024
  const uint32_t w_off = kATile + (wn * kTN + (fg >> 1)) * 1024 + (frow + 32 * (fg & 1u)) * 16;        // + (tn >> 1) * 2048 + (tn & 1) * 256
  v4i fa[kTM];
  v4i wl[kHalf], wh[kHalf];
  auto read_a = [&](uint32_t slot, int tm) __attribute__((always_inline)) {
    fa[tm] = *reinterpret_cast<const v4i*>(lds + slot * kStage + a_off + tm * 1024);
  };
  auto read_w = [&](uint32_t slot, int tn, v4i& dst) __attribute__((always_inline)) {
    dst = *reinterpret_cast<const v4i*>(lds + slot * kStage + w_off + (tn >> 1) * 2048 + (tn & 1) * 256);
  };
  const uint32_t flip = p.a_flip;                // 0x80808080 (kzp 128) or 0x7F7F7F7F (kzp 127), scalar
  auto flip_a = [&](int tm) __attribute__((always_inline)) {
    fa[tm].x ^= static_cast<int>(flip);
    fa[tm].y ^= static_cast<int>(flip);
    fa[tm].z ^= static_cast<int>(flip);
    fa[tm].w ^= static_cast<int>(flip);
    asm volatile("" : "+v"(fa[tm]));
  };

  v4i acc[kTM][kTN];
  auto mma = [&](const v4i& w, int tm, int tn) __attribute__((always_inline)) {
    acc[tm][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, fa[tm], acc[tm][tn], 0, 0, 0);
  };

  // tile 0 and the bias line have landed when at most the later tiles' pieces are outstanding (loads complete in issue order)
  if (fill >= 4) wait_vmcnt<3 * kDma>();
  else if (fill == 3) wait_vmcnt<2 * kDma>();
  else if (fill == 2) wait_vmcnt<1 * kDma>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  QNNP_PIN();
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) read_a(0u, tm);
#pragma unroll
  for (int tn = 0; tn < kHalf; tn++) read_w(0u, tn, wl[tn]);
  // accumulators start from the bias: lane l holds, in register r of tile tn, channel n0 + 16 tn + 4 (l >> 4) + r
  // (its own wave's DMA: visible behind the vmcnt wait above)
#pragma unroll
  for (int tn = 0; tn < kTN; tn++) {
    const v4i b = *reinterpret_cast<const v4i*>(bias_line + tn * 64 + fg * 16);
#pragma unroll
    for (int tm = 0; tm < kTM; tm++) acc[tm][tn] = b;
  }
  // fa[0], fa[1] re-centred here, fa[2], fa[3] left raw: the state every tile's phase 1 starts in
  flip_a(0);
  flip_a(1);
  if constexpr (kHalf == 2) asm volatile("" : "+v"(wl[0]), "+v"(wl[1]));
  else asm volatile("" : "+v"(wl[0]));

  /*
   * One K tile: phase 1 = fa x wl while wh arrives; counted wait + barrier (tile kt + 1 resident, every read of tile kt
   * done); phase 2 = fa x wh while the next tile's wl and fa arrive, fa[tm] right behind the last MFMA that reads the old one.
   * Tile t's activation pieces are requested in phase 2 of iteration t - RING (its slot was freed by that iteration's
   * barrier), its weight pieces in phase 1 of iteration t - RING + 1.
   */
  for (uint32_t kt = 0; kt < ktiles; kt++) {
    const uint32_t slot = kt & (RING - 1u);
    const uint32_t next_slot = (kt + 1u) & (RING - 1u);
    const uint32_t prev_slot = (kt + RING - 1u) & (RING - 1u);
    const bool more = kt + 1u < ktiles;
    if (kt >= 1u && kt + RING - 1u < ktiles) stage_w(kt + RING - 1u, prev_slot);
    QNNP_PIN();
    // ---- phase 1 (the s_waitcnt lgkmcnt in front of every use of a fragment are the compiler's: LDS reads return in order
    //      and it counts them) ----
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      const int tm = i / kHalf, tn = (tm & 1) != 0 ? kHalf - 1 - i % kHalf : i % kHalf;   // snake: one operand changes per MFMA (q8gemm256x.hip)
      mma(wl[tn], tm, tn);
      QNNP_PIN();
      if (i == 0) { flip_a(2); QNNP_PIN(); }                       // read behind MFMA 3 kHalf - 1 of the previous phase 2; first use: MFMA 2 kHalf
      if (i < kHalf) { read_w(slot, kHalf + i, wh[i]); QNNP_PIN(); }
      if (i == kHalf) { flip_a(3); QNNP_PIN(); }                   // read behind the last MFMA of the previous phase 2; first use: MFMA 3 kHalf
    }
    if constexpr (kHalf == 2) asm volatile("" : "+v"(wh[0]), "+v"(wh[1]));
    else asm volatile("" : "+v"(wh[0]));
    QNNP_PIN();
    if (more) {
      // tile kt + 1 resident: behind it at most the tiles kt + 2 .. min(kt + RING, ktiles) - 1 are in flight
      const uint32_t ahead = min(kt + RING, ktiles) - (kt + 2u);
      if (ahead >= 2u) wait_vmcnt<2 * kDma>();
      else if (ahead == 1u) wait_vmcnt<1 * kDma>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
    }
    QNNP_PIN();
    if (kt + RING < ktiles) stage_a(kt + RING, slot);
    QNNP_PIN();
    // ---- phase 2 (the reads of a tile that does not exist fetch stale LDS into registers nobody multiplies) ----
#pragma unroll
    for (int i = 0; i < kMma; i++) {
      const int tm = i / kHalf, tn = (tm & 1) != 0 ? kHalf - 1 - i % kHalf : i % kHalf;   // snake: one operand changes per MFMA (q8gemm256x.hip)
      mma(wh[tn], tm, kHalf + tn);
      QNNP_PIN();
      if (tm == 0) { read_w(next_slot, tn, wl[tn]); QNNP_PIN(); }
      if (i % kHalf == kHalf - 1) {
        if (tm >= 2) { flip_a(tm - 2); QNNP_PIN(); }               // fa[0], fa[1]: read two row tiles ago
        read_a(next_slot, tm);
        QNNP_PIN();
      }
    }
    if constexpr (kHalf == 2) asm volatile("" : "+v"(wl[0]), "+v"(wl[1]));
    else asm volatile("" : "+v"(wl[0]));
    QNNP_PIN();
  }

  // ---- fused epilogue: Q31 requantize in registers -> 4 x 4 lane transpose -> one 16-byte store per lane and block ----
  const uint32_t m0 = m_tile * kBM + wm * 64;
  uint8_t* out0 = p.output + static_cast<uint64_t>(m0) * p.output_stride + static_cast<uint64_t>(g) * p.n + n0;
  const bool col_ok = n0 + fg * 16 < p.n;
  auto transpose4 = [&](uint32_t (&q)[4]) __attribute__((always_inline)) {
    const auto s02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    const auto lo = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    q[0] = lo[0]; q[1] = lo[1]; q[2] = hi[0]; q[3] = hi[1];
  };
#pragma unroll
  for (int tm = 0; tm < kTM; tm++) {
    uint32_t q[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < kTN; j++) {
      q[j] = q31_requantize_pack4_clamp<SEQ, CLAMP>(acc[tm][j][0], acc[tm][j][1], acc[tm][j][2], acc[tm][j][3], p.rq);
    }
    transpose4(q);         // lane (row, g): channels 16 g .. 16 g + 15 of its row within the wave's 16 TN (g < TN)
    // (round 6: rows that are whole dwords but not whole 16-byte pieces -- SqueezeNet's 512 -> 1000 classifier convolution: the store is a
    //  dword-aligned 16-byte one, the group's last piece may be 4 / 8 / 12 bytes)
    typedef int nt_v4i __attribute__((ext_vector_type(4), aligned(4)));
    typedef int nt_v2i __attribute__((ext_vector_type(2), aligned(4)));
    const nt_v4i x = {static_cast<int>(q[0]), static_cast<int>(q[1]), static_cast<int>(q[2]), static_cast<int>(q[3])};
    const uint32_t r = tm * 16 + frow;
    uint8_t* dst = out0 + static_cast<uint64_t>(r) * p.output_stride + fg * 16;
    if (m0 + r < p.rows && col_ok && fg < static_cast<uint32_t>(kTN)) {
      const uint32_t left = p.n - (n0 + fg * 16);                 // bytes of the group behind this piece's first one (> 0: col_ok)
      if (left >= 16u) *reinterpret_cast<nt_v4i*>(dst) = x;
      else {
        if (left >= 8u) *reinterpret_cast<nt_v2i*>(dst) = nt_v2i{x.x, x.y};
        if ((left & 4u) != 0u) *reinterpret_cast<int*>(dst + (left & 8u)) = left >= 8u ? x.z : x.x;
      }
    }
  }
}
#undef QNNP_PIN

}  // namespace

/* p as the general kernels get it, with the centred image */
bool gemm128x_supported(const IgemmParams& p, uint32_t vec)
{
  if (p.offsets != nullptr) {                   // strided 1x1 convolution through the table: absolute 32-bit lane offsets
    if (p.offsets_dense == 0 || p.rows_per_image == 0) return false;
    const uint64_t images = (static_cast<uint64_t>(p.rows) + p.rows_per_image - 1) / p.rows_per_image;
    if (images * p.image_stride + p.k_pad >= (1ull << 32)) return false;
  }
  return vec == 16 && p.a_flip != 0 && p.bias2u != nullptr && p.store_mode >= 1 &&
         p.k_total == p.k_pad && p.k_pad % kBK == 0 && p.k_pad >= kBK && p.n_pad % 32 == 0 && p.n % 4 == 0 &&
         p.k_pad <= (1u << 22) && static_cast<uint64_t>(p.input_stride) * kBM < (1ull << 32) &&
         p.residual == nullptr && p.rows >= 1;
}

namespace {
template <int TN>
int launch_mid(const IgemmParams& p, uint32_t groups, hipStream_t stream)
{
  constexpr uint32_t kBN = 32 * TN;
  const uint32_t tiles_m = (p.rows + kBM - 1) / kBM;
  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  if (static_cast<uint64_t>(tiles_m) * tiles_n * tiles_n >= (1ull << 32)) return QNNP_HIP_EINVAL;
  const dim3 grid(tiles_m * tiles_n, groups, 1);
  IgemmParams pm = p;
  pm.tiles_n_magic = tiles_n == 1 ? 0u : static_cast<uint32_t>((1ull << 32) / tiles_n) + 1u;
  int rc = QNNP_HIP_EINVAL;
  if (p.rq.f.shift != 0 && p.rq.f.bounded && p.rq.f.ofs_kind == 2 && !p.rq.full_range) {
    if (p.rq.zp_late == 0) hipLaunchKernelGGL((q8_gemm_mfma_128xN_c16_kernel<kRqBoundedOfs, 1, TN>), grid, dim3(kThreads), 0, stream, pm);
    else hipLaunchKernelGGL((q8_gemm_mfma_128xN_c16_kernel<kRqBoundedOfs, 2, TN>), grid, dim3(kThreads), 0, stream, pm);
    return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  }
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    if constexpr (decltype(full)::value) {
      hipLaunchKernelGGL((q8_gemm_mfma_128xN_c16_kernel<kSeq, 0, TN>), grid, dim3(kThreads), 0, stream, pm);
    } else if (p.rq.zp_late == 0) {
      hipLaunchKernelGGL((q8_gemm_mfma_128xN_c16_kernel<kSeq, 1, TN>), grid, dim3(kThreads), 0, stream, pm);
    } else {
      hipLaunchKernelGGL((q8_gemm_mfma_128xN_c16_kernel<kSeq, 2, TN>), grid, dim3(kThreads), 0, stream, pm);
    }
    rc = hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  });
  return rc;
}
}  // namespace

/* `p` must carry the CENTRED weight image, its bias pair table and a_flip (q8igemm.hip).
 * tile_n: 0 = by the channel count (64-wide tiles when they cover the channels with fewer padded columns than 128-wide ones:
 * N = 64, 160, 320 ... -- MobileNetV2's 7x7x960 -> 320 8.8 -> 8.0 us, 14x14x384 -> 64 6.3 -> 5.6; N = 96 pads to 128 either way and
 * keeps the wide tile, 6.2 against 6.9 us: profiles/r06/mid_gemm_by_forced_kernel_r06d.txt), 64 / 128 = forced (A/B). */
int gemm128x_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, uint32_t tile_n)
{
  const uint32_t cols128 = (p.n + 127u) / 128u * 128u, cols64 = (p.n + 63u) / 64u * 64u;
  const bool narrow = tile_n == 64u || (tile_n == 0u && cols64 < cols128);
  if (narrow) {
    *name = "q8_gemm_mfma_128x64_c16";
    return launch_mid<2>(p, groups, stream);
  }
  *name = "q8_gemm_mfma_128x128_c16";
  return launch_mid<4>(p, groups, stream);
}

}  // namespace qnnp
