/*
 * q8convws16s.hip -- dense 3x3 convolution (stride 1, dilation 1) with FEW input channels -- 16 / 32 / 48 / 64 -- and 16 ... 256 output
 * channels, weights in registers, on v_mfma_i32_16x16x64_i8 (round 6).
 *
 * Same operator and arithmetic as the other implicit-GEMM kernels (replaces q8conv_ukernel_4x4c2__sse2, src/q8conv/4x4c2-sse2.c:14-273,
 * + compute_q8conv, src/operator-run.c:183-217, 837-842, + the indirection buffer, src/indirection.c:18-79) for the fire modules of
 * SqueezeNet 1.0 / 1.1 (bench/convolution.cc:543-640: 55x55 16 -> 64, 55x55 / 27x27 32 -> 128, 27x27 / 13x13 48 -> 192, 64 -> 256), which
 * ran on the offset-table tile kernel and the LDS-tiled kernel at 0.05-0.2 of their bounds (profiles/r06/bench_full_r06w.json: 34 / 39 /
 * 40 us for 6-25 MB of traffic).
 *
 * It is q8_conv_wave_ws16_kernel (q8convwave.hip) with the channel count as a template argument:
 *   - unit = 4 output rows x 8 columns of one image; a wave stages the unit's 6 x 10 pixel patch into its own piece of LDS (buffer
 *     loads issued between the multiplies of the previous unit, re-centred a ^ flip on their way in, out-of-image pixels = the zero
 *     point) -- a pixel is CPP = C / 16 chunks of 16 bytes;
 *   - K = 9 taps x C bytes is walked in 64-byte steps = four 16-byte SLOTS per instruction: slot s = (tap s / CPP, chunk s % CPP), so with
 *     16 channels one instruction multiplies FOUR taps, with 48 channels a tap and a third; operand lane (position l & 15, slot 4 m + (l >> 4))
 *     reads its chunk of patch pixel (position + tap) -- an address that is a per-lane constant of the step plus the position's base;
 *     ceil(9 CPP / 4) = 3 / 5 / 7 / 9 instructions per 16 positions x 16 channels (the slots past the window meet zero weights);
 *   - weight fragments come from pack.h's standard 32 x 32 image (K order tap-major, channel-minor = slot order) with q8gemm256x.hip's
 *     lane addresses and stay in registers for the wave's life: M x TN16 x 4;
 *   - an N-tile of <= 64 output channels per workgroup (the tile index is the fastest-moving part of blockIdx.x: the tiles of a unit
 *     range run side by side and share its patches in L2); epilogue as the 64-channel kernel: requantize 4 -> 1 dword, one 4 x 4 lane
 *     transpose per position tile, one 16-byte store per lane (64 contiguous bytes per pixel).
 * Needs the zero-point-centred image (kernel zero point 127: convolution.c builds it for these shapes; 128: the standard image is the
 * centred one) -- other kernel zero points keep the kernels this one replaces.
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

constexpr int kSWaves = 8;
constexpr uint32_t kSLdsLimit = 160 * 1024;
constexpr uint32_t kSPatchRows = 6u, kSPatchCols = 10u;

struct SArgs {
  uint32_t tiles_x, tiles_y;  // units per image: ceil(OW / 8) x ceil(OH / 4)
  uint32_t units;             // batch * tiles_x * tiles_y
  uint32_t inv_tiles, inv_tiles_x;   // ceil(2^32 / d) (0: d == 1), exact while units * d < 2^32 (launcher)
  uint32_t n_tiles;           // 64-channel tiles of the output channels
  uint32_t inv_n_tiles;
  uint32_t ranges;            // unit ranges = gridDim.x / n_tiles
  uint32_t w_bytes;           // weight fragments of one N-tile
  uint32_t head_bytes;        // weights + bias, 1024-aligned: offset of the first wave's patch
  uint32_t patch_bytes;       // one wave's patch, 256-aligned
};

__device__ __forceinline__ uint32_t s_div(uint32_t n, uint32_t inv) { return inv != 0u ? __umulhi(n, inv) : n; }

__device__ __forceinline__ uint32_t s_lds_off(const void* p)
{
  return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*) p));
}

/* LDS-DMA, 16 bytes per lane (inline asm: q8convwave.hip dma16 says why); M0 restored */
__device__ __forceinline__ void s_dma16(const uint8_t* src, uint8_t* lds_wave_base)
{
  const uint32_t dst = __builtin_amdgcn_readfirstlane(s_lds_off(lds_wave_base));
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
}

__device__ __forceinline__ void s_ds_write16(uint32_t off, v4i x)
{
  asm volatile("ds_write_b128 %0, %1" :: "v"(off), "v"(x) : "memory");
}

template <int CPP, int TN16, int SEQ, bool FULL>
__global__ __launch_bounds__(kSWaves * 64, 2)
void q8_conv_ws16s_kernel(const IgemmParams p, const ConvGeom g, const SArgs a)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];   // [weights of the N-tile][bias][waves x patch]
  constexpr uint32_t cin = 16u * CPP;
  constexpr uint32_t pvec = kSPatchRows * kSPatchCols * CPP;      // 16-byte chunks of the patch
  constexpr int NP = (pvec + 63u) / 64u;                          // 1 KiB pieces of the patch
  constexpr int M = (9 * CPP + 3) / 4;                            // 64-byte K steps
  static_assert(M >= NP, "the next patch is requested between the steps");
  uint8_t* w_lds = lds;
  int32_t* bias_lds = reinterpret_cast<int32_t*>(lds + a.w_bytes);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint8_t* patch = lds + a.head_bytes + wave * a.patch_bytes;

  const uint32_t range = s_div(blockIdx.x, a.inv_n_tiles);
  const uint32_t tile = blockIdx.x - range * a.n_tiles;
  const uint32_t kblocks = p.k_pad / 32;

  // ---- prologue, first half: the tile's weights + bias by LDS-DMA, once per workgroup ----
  {
    const uint32_t pieces = a.w_bytes >> 10;
    const uint8_t* src = reinterpret_cast<const uint8_t*>(p.packed_w) + static_cast<size_t>(tile) * 2u * kblocks * 1024u + lane * 16u;
    for (uint32_t i = wave; i < pieces; i += kSWaves) s_dma16(src + i * 1024u, w_lds + i * 1024u);
    if (wave == kSWaves - 1 && lane < TN16 * 4u) {
      s_dma16(reinterpret_cast<const uint8_t*>((rq_is_lane<SEQ>() ? p.bias2u : p.bias2) + tile * 64u) + lane * 16u,
              reinterpret_cast<uint8_t*>(bias_lds));
    }
  }
  __builtin_amdgcn_sched_barrier(0);

  const uint32_t lo = static_cast<uint32_t>(static_cast<uint64_t>(range) * a.units / a.ranges);
  const uint32_t hi = static_cast<uint32_t>(static_cast<uint64_t>(range + 1) * a.units / a.ranges);
  const uint32_t tiles = a.tiles_x * a.tiles_y;
  const uint32_t fill4 = (p.izp_fill & 0xFFu) * 0x01010101u;
  const uint32_t fpos = lane & 15u;                // position inside a 16-position tile: row fpos >> 3, column fpos & 7
  const uint32_t fg = lane >> 4;                   // K slot of an operand; channel quad of a result

  // ---- the gather pattern of a patch: chunk v = lane + 64 u is chunk v % CPP of patch pixel v / CPP; only the origin moves ----
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(a.units / tiles * p.image_stride), 0x00020000);   // (launcher: < 2^31)
  uint32_t rel[NP], pyx[NP];
#pragma unroll
  for (int u = 0; u < NP; u++) {
    const uint32_t v = min(lane + u * 64u, pvec - 1u);
    const uint32_t q = v / CPP;
    const uint32_t sl = v - q * CPP;
    const uint32_t py = (q * 6554u) >> 16;                 // q / 10 for q < 100
    const uint32_t px = q - py * 10u;
    rel[u] = (py * g.W + px) * p.input_stride + (sl << 4);
    pyx[u] = (py << 16) | px;
  }
  struct Raw { v4i x[NP]; };
  struct Where { uint32_t origin; int32_t iy0, ix0; uint32_t out_img, oy0, ox0; bool border; };   // (wave-uniform)
  auto locate = [&](uint32_t unit) __attribute__((always_inline)) -> Where {
    const uint32_t img = s_div(unit, a.inv_tiles);
    const uint32_t rr = unit - img * tiles;
    const uint32_t tyi = s_div(rr, a.inv_tiles_x);
    const uint32_t txi = rr - tyi * a.tiles_x;
    Where w;
    w.oy0 = tyi * 4u;
    w.ox0 = txi * 8u;
    w.iy0 = static_cast<int32_t>(w.oy0) - static_cast<int32_t>(g.pad_top);
    w.ix0 = static_cast<int32_t>(w.ox0) - static_cast<int32_t>(g.pad_left);
    w.origin = img * static_cast<uint32_t>(p.image_stride) +
        static_cast<uint32_t>(w.iy0 * static_cast<int32_t>(g.W) + w.ix0) * p.input_stride;
    w.out_img = img * g.OH * g.OW * p.output_stride;
    w.border = w.iy0 < 0 || w.ix0 < 0 || w.iy0 + static_cast<int32_t>(kSPatchRows) > static_cast<int32_t>(g.H) ||
               w.ix0 + static_cast<int32_t>(kSPatchCols) > static_cast<int32_t>(g.W);
    return w;
  };
  auto fetch_piece = [&](const Where& w, int u, Raw& r) __attribute__((always_inline)) {
    r.x[u] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, rel[u] + w.origin, 0, 0));
  };
  // the fetched patch: (border units: pixels outside the image become the zero point,) re-centred into LDS
  auto fix_up = [&](Raw& r, const Where& w) __attribute__((always_inline)) {
    const uint32_t patch_off = s_lds_off(patch);
    if (w.border) {
#pragma unroll
      for (int u = 0; u < NP; u++) {
        const int32_t iy = w.iy0 + static_cast<int32_t>(pyx[u] >> 16);
        const int32_t ix = w.ix0 + static_cast<int32_t>(pyx[u] & 0xFFFFu);
        const bool inb = static_cast<uint32_t>(iy) < g.H && static_cast<uint32_t>(ix) < g.W;
        r.x[u].x = inb ? r.x[u].x : static_cast<int>(fill4);
        r.x[u].y = inb ? r.x[u].y : static_cast<int>(fill4);
        r.x[u].z = inb ? r.x[u].z : static_cast<int>(fill4);
        r.x[u].w = inb ? r.x[u].w : static_cast<int>(fill4);
      }
    }
    const int flip = static_cast<int>(p.a_flip);
#pragma unroll
    for (int u = 0; u < NP; u++) {
      const uint32_t v = lane + u * 64u;
      if ((u + 1) * 64u <= pvec || v < pvec) {             // (only the last piece is partly populated)
        const v4i x = r.x[u];
        s_ds_write16(patch_off + v * 16u, v4i{x.x ^ flip, x.y ^ flip, x.z ^ flip, x.w ^ flip});
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  };

  // ---- prologue, second half: the first patch (an HBM round trip) behind the weights
  uint32_t cur = lo + wave;
  Raw raw;
  Where here = locate(min(cur, a.units - 1u));
#pragma unroll
  for (int u = 0; u < NP; u++) fetch_piece(here, u, raw);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>((a.units / tiles * g.OH * g.OW - 1u) * p.output_stride + p.n), 0x00020000);   // (launcher: < 2^31)
  __builtin_amdgcn_sched_barrier(0);
  fix_up(raw, here);                                // needs the patch only
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---- every weight fragment of the tile into registers, for good: channel tile tn, step m -> slots 4 m + fg ----
  v4i wreg[M][TN16];
  {
    const uint8_t* w_lane = w_lds + (fpos + 32u * (fg & 1u)) * 16u;
#pragma unroll
    for (int m = 0; m < M; m++)
#pragma unroll
      for (int tn = 0; tn < TN16; tn++)
        wreg[m][tn] = *reinterpret_cast<const v4i*>(w_lane + ((tn >> 1) * kblocks + 2 * m + (fg >> 1)) * 1024u + (tn & 1) * 256u);
  }
  // this lane's slot of step m: tap (s / CPP), chunk (s % CPP) -> byte offset inside the patch relative to the position's pixel
  uint32_t soff[M];
#pragma unroll
  for (int m = 0; m < M; m++) {
    const uint32_t s = 4u * m + fg;
    const uint32_t t = s / CPP;
    const uint32_t c = s - t * CPP;
    const uint32_t ky = (t * 11u) >> 5;                    // t / 3 for t < 12
    const uint32_t kx = t - ky * 3u;
    soff[m] = s < 9u * CPP ? ((ky * kSPatchCols + kx) * CPP + c) * 16u : 0u;     // (past the window: zero weights, any chunk)
  }
  const uint32_t tyl = fpos >> 3;
  const uint32_t chan0 = tile * 64u;

  while (cur < hi) {
    const Where next = locate(min(cur + kSWaves, a.units - 1u));

    // accumulators start at the folded bias: register r of tile tn = channel 16 tn + 4 g + r
    v4i acc[2][TN16];
#pragma unroll
    for (int tn = 0; tn < TN16; tn++) {
      const v4i b = *reinterpret_cast<const v4i*>(bias_lds + tn * 16 + fg * 4);
      acc[0][tn] = b;
      acc[1][tn] = b;
    }
    {
      const uint8_t* abase[2];
#pragma unroll
      for (int tm = 0; tm < 2; tm++) abase[tm] = patch + ((tm * 2u + tyl) * kSPatchCols + (fpos & 7u)) * cin;
      struct AF { v4i a[2]; };
      auto read_a = [&](int m, AF& f) __attribute__((always_inline)) {
#pragma unroll
        for (int tm = 0; tm < 2; tm++) f.a[tm] = *reinterpret_cast<const v4i*>(abase[tm] + soff[m]);
      };
      auto mma = [&](int m, const AF& f) __attribute__((always_inline)) {
#pragma unroll
        for (int tm = 0; tm < 2; tm++)
#pragma unroll
          for (int j = 0; j < TN16; j++) {
            const int tn = tm == 0 ? j : TN16 - 1 - j;       // snake: one operand changes per MFMA (q8gemm256x.hip)
            acc[tm][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wreg[m][tn], f.a[tm], acc[tm][tn], 0, 0, 0);
          }
      };
      // the next unit's patch is requested piece by piece between the steps
      AF f[2];
      read_a(0, f[0]);
#pragma unroll
      for (int m = 0; m < M; m++) {
        if (m + 1 < M) read_a(m + 1, f[(m + 1) & 1]);
        mma(m, f[m & 1]);
        if (m < NP) fetch_piece(next, m, raw);
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- fused epilogue: requantization, one lane transpose per position tile, 16-byte stores ----
    {
      const int32_t rowterm = with_rq_offset<SEQ>(0);
      uint64_t row_addend = 0;
      if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
#pragma unroll
      for (int tm = 0; tm < 2; tm++) {
        uint32_t q[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int tn = 0; tn < TN16; tn++) {
          if constexpr (rq_is_lane<SEQ>()) {
            q[tn] = q31_requantize_pack4_lane<SEQ, FULL>(
                static_cast<uint32_t>(acc[tm][tn][0]), static_cast<uint32_t>(acc[tm][tn][1]),
                static_cast<uint32_t>(acc[tm][tn][2]), static_cast<uint32_t>(acc[tm][tn][3]), row_addend, p.lane, p.rq);
          } else {
            q[tn] = q31_requantize_pack4<SEQ, FULL, false>(
                add_wrap(acc[tm][tn][0], rowterm), add_wrap(acc[tm][tn][1], rowterm),
                add_wrap(acc[tm][tn][2], rowterm), add_wrap(acc[tm][tn][3], rowterm), p.rq);
          }
        }
        // 4 x 4 dword transpose over the four 16-lane rows: lane (position, g) then holds channels 16 g .. 16 g + 15 of the tile
        const auto s02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
        const auto tlo = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
        const auto thi = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
        const v4i outv = {static_cast<int>(tlo[0]), static_cast<int>(tlo[1]), static_cast<int>(thi[0]), static_cast<int>(thi[1])};
        const uint32_t oy = here.oy0 + tm * 2u + tyl;
        const uint32_t ox = here.ox0 + (fpos & 7u);
        const bool ok = oy < g.OH && ox < g.OW && fg < static_cast<uint32_t>(TN16);
        const uint32_t off = ok ? here.out_img + (oy * g.OW + ox) * p.output_stride + chan0 + fg * 16u : 0xFFFFFFF0u;
        const auto bits = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, outv);
        if (TN16 == 4 && p.stream_out != 0) __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, off, 0, 2);
        else __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, off, 0, 0);
      }
    }
    // ---- the next unit's patch (fetched between the steps) into the patch buffer ----
    fix_up(raw, next);
    here = next;
    cur += kSWaves;
  }
}

inline bool make_sargs(const IgemmParams& p, const ConvGeom& g, uint32_t batch, uint32_t tn16, SArgs* a, uint32_t* lds_bytes)
{
  a->tiles_x = (g.OW + 7u) / 8u;
  a->tiles_y = (g.OH + 3u) / 4u;
  const uint64_t tiles = static_cast<uint64_t>(a->tiles_x) * a->tiles_y;
  if (static_cast<uint64_t>(batch) * tiles * tiles >= (UINT64_C(1) << 32)) return false;
  a->units = batch * a->tiles_x * a->tiles_y;
  a->inv_tiles = tiles > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + tiles - 1) / tiles) : 0u;
  a->inv_tiles_x = a->tiles_x > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + a->tiles_x - 1) / a->tiles_x) : 0u;
  a->n_tiles = (p.n + 63u) / 64u;
  a->inv_n_tiles = a->n_tiles > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + a->n_tiles - 1) / a->n_tiles) : 0u;
  const uint32_t want = (a->units + kSWaves - 1) / kSWaves;
  // workgroups per CU the unit ranges are cut for, measured per channel count (profiles/r06/conv3x3_small_channels_per_cu_r06zd.txt):
  // 16 channels (<= 122 registers: two 8-wave workgroups share a CU) 2 -- 55x55 16 -> 64 10.8 -> 10.3 us; 48 channels 3 -- one
  // workgroup at a time by registers, but shorter ranges even out the tail: 27x27 48 -> 192 22.4 -> 19.3, 13x13 11.7 -> 10.4; 32 and 64
  // channels 1 (55x55 32 -> 128 22.3 against 24.7 / 25.8, 27x27 64 -> 256 18.3 against 20.7 / 24.1)
  const uint32_t cpp = p.kc / 16u;
  uint32_t per_cu = cpp == 1u ? 2u : (cpp == 3u ? 3u : 1u);
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_WS16S_PER_CU")) per_cu = static_cast<uint32_t>(atoi(env));
#endif
  uint32_t ranges = (p.cu_count * per_cu + a->n_tiles - 1) / a->n_tiles;
  if (ranges > want) ranges = want;
  if (ranges < 1) ranges = 1;
  a->ranges = ranges;
  a->w_bytes = ((tn16 + 1u) / 2u) * (p.k_pad / 32u) * 1024u;
  a->head_bytes = (a->w_bytes + 256u + 1023u) & ~1023u;
  a->patch_bytes = (kSPatchRows * kSPatchCols * p.kc + 255u) & ~255u;
  *lds_bytes = a->head_bytes + kSWaves * a->patch_bytes;
  return *lds_bytes <= kSLdsLimit;
}

template <int CPP, int TN16>
int launch_s(const IgemmParams& p, const ConvGeom& g, const SArgs& a, uint32_t lds_bytes, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    constexpr bool kFull = decltype(full)::value;
    static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
    if (auto once_scope = attr_once.begin()) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&q8_conv_ws16s_kernel<CPP, TN16, kSeq, kFull>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void) hipGetLastError();
      }
    }
    hipLaunchKernelGGL((q8_conv_ws16s_kernel<CPP, TN16, kSeq, kFull>), dim3(a.ranges * a.n_tiles), dim3(kSWaves * 64), lds_bytes, stream, p, g, a);
    rc = hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  });
  return rc;
}

template <int CPP>
int launch_s_n(const IgemmParams& p, const ConvGeom& g, const SArgs& a, uint32_t lds_bytes, uint32_t tn16, hipStream_t stream)
{
  switch (tn16) {
    case 1: return launch_s<CPP, 1>(p, g, a, lds_bytes, stream);
    case 2: return launch_s<CPP, 2>(p, g, a, lds_bytes, stream);
    case 3: return launch_s<CPP, 3>(p, g, a, lds_bytes, stream);
    default: return launch_s<CPP, 4>(p, g, a, lds_bytes, stream);
  }
}

}  // namespace

/* p: the operator's CENTRED parameters (a_flip != 0: the centred image, its bias pair table, no row term).
 * 3x3 / stride 1 / dilation 1, one group, 16 / 32 / 48 / 64 input channels in dense-enough 16-byte aligned pixels, output channels
 * 16 / 32 / 48 or a multiple of 64 up to 256, 16-byte aligned output pixels, tensors addressable with 31-bit offsets. */
bool convws16s_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch)
{
  if (groups != 1 || vec != 16 || p.a_flip == 0 || p.bias2u == nullptr || p.row_coeff != 0) return false;
  if (!(p.kc == 16 || p.kc == 32 || p.kc == 48 || p.kc == 64)) return false;
  if (g.KH != 3 || g.KW != 3 || g.sh != 1 || g.sw != 1 || g.dh != 1 || g.dw != 1) return false;
  if (p.k_total != 9u * p.kc || p.store_mode != 2) return false;
  if (!(p.n == 16 || p.n == 32 || p.n == 48 || (p.n % 64u == 0 && p.n <= 256u)) || p.n == 0) return false;
  if (batch == 0 || g.OH == 0 || g.OW == 0 || p.residual != nullptr) return false;
  const uint64_t in_bytes = static_cast<uint64_t>(batch) * p.image_stride;
  const uint64_t out_bytes = static_cast<uint64_t>(batch) * g.OH * g.OW * p.output_stride;
  if (in_bytes >= (UINT64_C(1) << 31) || out_bytes >= (UINT64_C(1) << 31)) return false;
  SArgs a;
  uint32_t lds_bytes = 0;
  return make_sargs(p, g, batch, p.n >= 64 ? 4u : p.n / 16u, &a, &lds_bytes);
}

int convws16s_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name)
{
  const uint32_t tn16 = p.n >= 64 ? 4u : p.n / 16u;
  SArgs a;
  uint32_t lds_bytes = 0;
  if (!make_sargs(p, g, batch, tn16, &a, &lds_bytes)) return QNNP_HIP_EINVAL;
  *name = "q8_conv_ws16s_mfma";
  switch (p.kc / 16u) {
    case 1: return launch_s_n<1>(p, g, a, lds_bytes, tn16, stream);
    case 2: return launch_s_n<2>(p, g, a, lds_bytes, tn16, stream);
    case 3: return launch_s_n<3>(p, g, a, lds_bytes, tn16, stream);
    default: return launch_s_n<4>(p, g, a, lds_bytes, tn16, stream);
  }
}

}  // namespace qnnp
