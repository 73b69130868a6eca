/*
 * q8convc3.hip -- first-layer convolution (3 input channels, small window) as ONE 16-byte fetch per kernel row.
 *
 * Same operator and arithmetic as the other implicit-GEMM kernels (replaces q8conv_ukernel_4x4c2__sse2,
 * src/q8conv/4x4c2-sse2.c:14-273, + compute_q8conv, src/operator-run.c:183-217, 837-842, + the indirection buffer,
 * src/indirection.c:18-79) for the network-entry layers of bench/convolution.cc (MobileNetV2 line 457: 224x224x3 ->
 * 112x112x32, 3x3 stride 2).
 *
 * Why a third kernel for this shape: the offset-table kernel of q8pwconv.hip (q8_conv_stream_c3s_kernel) gathers a
 * pixel's nine taps with eight unaligned dword loads per lane and assembles the MFMA operand from them -- 245 VALU + 140
 * scalar instructions per 32 output pixels for ONE useful MFMA, and a SIMD issues one instruction per four cycles
 * whatever its waves do: 22 us of the 35 were instruction issue with no memory operation at all.
 * With dense 3-byte pixels the kx * 3 + c bytes of ONE kernel row are contiguous in memory: KW * 3 <= 16 bytes starting at
 * pixel (oy*s + ky - pad, ox*s - pad). So K is laid out as [ky][16-byte row slot] -- K = 16 * KH <= 64, the slot's unused
 * bytes meet zero weights -- and a lane's MFMA operand half IS the 16 bytes it loads: two loads per lane and unit, no
 * assembly. What it costs: a second MFMA per 32 x 32 outputs (K 27 -> 48), which the matrix pipe does not notice.
 *   - unit = 32 consecutive flattened output pixels; lane (p, h) owns pixel p and row slots ky = h, h + 2;
 *   - pixels outside the image: the slot is fetched from wherever its address lands (the descriptor returns zeros out of
 *     range) and the bytes of out-of-image taps are REPLACED by the zero point -- in units that touch the border only
 *     (wave-uniform branch on a ballot);
 *   - row sums over the real K positions only: v_dot4_u32_u8 against a 0/1 byte mask (the slot's junk bytes do not count);
 *   - weights (two fragments per 32 channels) and folded bias live in registers: no LDS, no barrier;
 *   - epilogue: row term, Q31 requantization, v_permlane32_swap -> 16 contiguous channels per lane, one 16-byte store
 *     per lane and 32 channels: a wave writes its 32 pixels x 32 bytes as one contiguous kilobyte.
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
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kC3Waves = 4;
constexpr int kC3Threads = kC3Waves * 64;
constexpr uint32_t kFlip = 0x80808080u;

struct C3Geom {
  uint32_t H, W, OH, OW;
  uint32_t KH, KW;
  uint32_t sh, sw;
  uint32_t pad_top, pad_left;
  uint32_t segs, pairs;       // 16-column segments per output row, row pairs per image
  uint32_t inv_segs;          // ceil(2^32 / segs) (0: divisor 1); exact while units * segs < 2^32 (launcher)
  uint32_t inv_pairs;         // the same for pairs
  const int8_t* w_rows16;     // MFMA fragments of the [ky][16] K layout: [n_pad / 32][2][64 lanes][16 bytes]
  uint32_t abl;               // measurement builds only (QNNP_C3R_ABL): 1 = 16-byte aligned loads, 2 = no stores, 4 = no loads
};

__device__ __forceinline__ uint32_t div_magic(uint32_t n, uint32_t inv) { return inv != 0u ? __umulhi(n, inv) : n; }

// bytes [lo, hi) of a dword (lo, hi relative to the dword's first byte, any integers): 0xFF in each such byte
__device__ __forceinline__ uint32_t byte_range_mask(int32_t lo, int32_t hi)
{
  lo = lo < 0 ? 0 : (lo > 4 ? 4 : lo);
  hi = hi < 0 ? 0 : (hi > 4 ? 4 : hi);
  if (hi <= lo) return 0u;
  const uint64_t ones = ~UINT64_C(0);
  const uint32_t upto_hi = static_cast<uint32_t>(~(ones << (8 * hi)));      // hi in 0..4: the shift is < 64
  const uint32_t upto_lo = static_cast<uint32_t>(~(ones << (8 * lo)));
  return upto_hi & ~upto_lo;
}

/* The stores of one unit (2 output rows x 16 columns) of the row-slot kernels with 16-byte channel pieces. After the requantization lane (pixel p = 16 prow + pcol, half h) holds channels
 * 32 nb + 16 h .. + 15 of its pixel for every channel block nb: stored as they are, an instruction writes 32-byte pieces 64 (96) bytes
 * apart. With two blocks the 16-lane rows exchange instead (v_permlane16_swap: row 1 of block 0 <-> row 0 of block 1, row 3 <-> row 2):
 * the first instruction then writes the 16 pixels of the unit's FIRST output row whole -- 1 KiB of contiguous 128-byte lines -- and the
 * second one the second row's (lane row r = 2 h + prow writes channels 16 h + 32 prow of pixel pcol). */
template <int NB>
__device__ __forceinline__ void c3_store_unit(v4i (&v)[NB], const __amdgpu_buffer_rsrc_t out_rsrc, uint32_t unit_out0, uint32_t row_pitch,
                                              uint32_t ostride, uint32_t n, bool row0_ok, bool row1_ok, bool col_ok, uint32_t prow,
                                              uint32_t pcol, uint32_t h, bool stream_out)
{
  auto put = [&](const v4i& x, uint32_t off, bool ok) __attribute__((always_inline)) {
    const auto bits = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, x);
    if (stream_out) __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, ok ? off : 0xFFFFFFF0u, 0, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, ok ? off : 0xFFFFFFF0u, 0, 0);
  };
  const uint32_t pix = unit_out0 + pcol * ostride;
  if constexpr (NB >= 2) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
      const auto sw = __builtin_amdgcn_permlane16_swap(static_cast<uint32_t>(v[0][d]), static_cast<uint32_t>(v[1][d]), false, false);
      v[0][d] = static_cast<int>(sw[0]); v[1][d] = static_cast<int>(sw[1]);
    }
    const uint32_t ch = h * 16u + prow * 32u;
    put(v[0], pix + ch, row0_ok && col_ok && ch < n);
    put(v[1], pix + row_pitch + ch, row1_ok && col_ok && ch < n);
  } else {
    put(v[0], pix + prow * row_pitch + h * 16u, (prow != 0u ? row1_ok : row0_ok) && col_ok && h * 16u < n);
  }
  if constexpr (NB == 3) put(v[2], pix + prow * row_pitch + 64u + h * 16u, (prow != 0u ? row1_ok : row0_ok) && col_ok && 64u + h * 16u < n);
}

/* UNIT = 2 output rows x 16 output columns of one image (32 pixels; lane (p, h): row p >> 4, column p & 15 of the
 * unit, K half h). Everything that locates a unit is scalar arithmetic; a lane's byte offsets are constants of the
 * kernel plus one scalar per unit. Whether a unit touches the image border (some tap outside the image) or the tensor's
 * first / last bytes is a scalar test as well: the byte surgery below runs in those units only.
 *
 * The per-pixel term (128 - kzp) * sum_k a'(pixel, k) is made BY THE MATRIX CORE: a second pair of weight fragments holds
 * (128 - kzp) at every real K position of EVERY channel, and its product with the same activation operand accumulates
 * into the same registers -- two more MFMAs per 32 x 32 outputs on an idle pipe instead of 6 v_dot4 + a cross-half exchange
 * + a multiply + 16 adds per lane (an instruction is an instruction to the issue port: DESIGN section 4.2c).
 * (128 - kzp = 128, i.e. kernel zero point 0, does not fit int8: it is applied as 64 + 64 by two such pairs.) */
template <int NB, int ND, int RCP, int SEQ, bool FULL>
__global__ __launch_bounds__(kC3Threads, NB == 1 ? 4 : 3)
void q8_conv_c3rows_kernel(const IgemmParams p, const C3Geom cg)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t px = lane & 31u;
  const uint32_t h = lane >> 5;
  const uint32_t prow = px >> 4, pcol = px & 15u;

  const uint32_t kbytes = cg.KW * 3u;                       // real bytes of a slot (<= 16; launcher)
  // ---- weights, row-term fragments and bias (+ the requantization offset) of all NB channel blocks into registers ----
  v4i w[NB][2];
  v16i bias[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) {
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      w[nb][kb] = *reinterpret_cast<const v4i*>(cg.w_rows16 + ((nb * 2 + kb) * 64u + lane) * 16u);
    }
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const v4i b = *reinterpret_cast<const v4i*>(p.bias2 + nb * 32 + rg * 8 + h * 4);
      bias[nb][rg * 4 + 0] = with_rq_offset<SEQ>(b.x); bias[nb][rg * 4 + 1] = with_rq_offset<SEQ>(b.y);
      bias[nb][rg * 4 + 2] = with_rq_offset<SEQ>(b.z); bias[nb][rg * 4 + 3] = with_rq_offset<SEQ>(b.w);
    }
  }
  // row-term fragments: lane (n, h) of K block kb holds row slot ky = 2 kb + h of "channel" n -- the same 16 bytes for
  // every n: rc at the real positions of a real slot, 0 elsewhere
  v4i wrc[RCP][2];
  {
    const int32_t rc = p.row_coeff;                         // 128 - kernel zero point: 1 .. 128 (0: no term at all)
#pragma unroll
    for (int part = 0; part < RCP; part++) {
      const int32_t piece = RCP == 1 ? rc : (part == 0 ? rc / 2 : rc - rc / 2);
      const uint32_t b4 = (static_cast<uint32_t>(piece) & 0xFFu) * 0x01010101u;
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bool real = kb * 2u + h < cg.KH;
        int* d = reinterpret_cast<int*>(&wrc[part][kb]);
#pragma unroll
        for (int i = 0; i < 4; i++) {
          d[i] = real ? static_cast<int>(byte_range_mask(0 - 4 * i, static_cast<int32_t>(kbytes) - 4 * i) & b4) : 0;
        }
      }
    }
  }

  const uint32_t in_bytes = static_cast<uint32_t>(p.input_end - p.input);          // (launcher: < 2^31)
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>((p.rows - 1u) * p.output_stride + p.n), 0x00020000);   // (launcher: < 2^31)

  // ---- per-lane constants: byte offset of the lane's two row slots relative to the unit's first window, of its output
  //      pixel relative to the unit's first one ----
  const uint32_t row_bytes = cg.W * 3u;
  uint32_t lane_in[2];
  bool slot_real[2];
#pragma unroll
  for (int kb = 0; kb < 2; kb++) {
    const uint32_t ky = kb * 2u + h;
    slot_real[kb] = ky < cg.KH;
    lane_in[kb] = (prow * cg.sh + (slot_real[kb] ? ky : 0u)) * row_bytes + pcol * cg.sw * 3u;
  }
  const uint32_t lane_out = (prow * cg.OW + pcol) * p.output_stride + h * 16u;
  const uint32_t fill4 = (p.izp_fill & 0xFFu) * 0x01010101u;

  // x: sixteen bytes from the dword-aligned address at or below the slot's first byte; tail (ND == 4 only): the dword
  // behind them. A buffer load whose address is not a multiple of four costs the texture-address unit several passes
  // (measured on this kernel: 31.9 us with byte-aligned, 23.8 us with dword- or 16-byte-aligned addresses, same bytes), so
  // the slot is fetched from the dword below and moved into place with three v_alignbyte.
  struct Slots { v4i x[2]; uint32_t tail[2]; };
  // delta: bytes by which a slot's load address was moved to keep all 16 bytes inside the tensor (non-zero only in the
  // first rows of the first image and the last rows of the last one: a buffer load that starts before the tensor or
  // straddles its end returns zeros for every dword that is not wholly inside, real tap bytes included)
  struct Unit { uint32_t out0; int32_t iy0, ix0; uint32_t rows_left, cols_left; bool border, edge; int32_t delta[2]; };
  auto locate = [&](uint32_t unit) __attribute__((always_inline)) -> Unit {       // scalar
    const uint32_t t = div_magic(unit, cg.inv_segs);
    const uint32_t seg = unit - t * cg.segs;
    const uint32_t img = div_magic(t, cg.inv_pairs);
    const uint32_t pair = t - img * cg.pairs;
    const uint32_t oy = pair * 2u, ox = seg * 16u;
    Unit u;
    u.iy0 = static_cast<int32_t>(oy * cg.sh) - static_cast<int32_t>(cg.pad_top);
    u.ix0 = static_cast<int32_t>(ox * cg.sw) - static_cast<int32_t>(cg.pad_left);
    u.rows_left = cg.OH - oy;                                // output rows / columns of the image from this unit on
    u.cols_left = cg.OW - ox;
    u.out0 = ((img * cg.OH + oy) * cg.OW + ox) * p.output_stride;
    // first / last input row and column any lane of the unit touches
    const int32_t iy_last = u.iy0 + static_cast<int32_t>(cg.sh + cg.KH) - 1;
    const int32_t ix_last = u.ix0 + static_cast<int32_t>(15u * cg.sw + cg.KW) - 1;
    u.border = u.iy0 < 0 || u.ix0 < 0 || iy_last >= static_cast<int32_t>(cg.H) || ix_last >= static_cast<int32_t>(cg.W);
    u.edge = img == 0u || (img + 1u) * cg.OH * cg.OW >= p.rows;      // windows that may reach the tensor's first / last bytes
    u.delta[0] = u.delta[1] = 0;
    return u;
  };
  // byte offset of the unit's first window inside the tensor, modulo 2^32 (negative for the first row / column: out of
  // the descriptor's range -> zeros, or some other pixel; either way replaced below)
  auto origin = [&](uint32_t unit, const Unit& u) __attribute__((always_inline)) -> uint32_t {
    const uint32_t t = div_magic(unit, cg.inv_segs);
    const uint32_t img = div_magic(t, cg.inv_pairs);
    return img * static_cast<uint32_t>(p.image_stride) + static_cast<uint32_t>(u.iy0 * static_cast<int32_t>(cg.W) + u.ix0) * 3u;
  };
  auto fetch = [&](uint32_t unit, Unit& u, Slots& s, int32_t (&delta)[2]) __attribute__((always_inline)) {
    const uint32_t org = origin(unit, u);
    uint32_t addr[2] = {org + lane_in[0], org + lane_in[1]};
    if (u.edge) {                                          // (scalar branch; address arithmetic only: the loads are below)
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const int32_t want = static_cast<int32_t>(addr[kb]);
        const int32_t last = static_cast<int32_t>(in_bytes) - 16;
        const int32_t got = want < 0 ? 0 : (want > last ? last : want);   // (byte-aligned: the tensor's last bytes must stay in reach; 2 images of the batch pay for it)
        delta[kb] = got - want;
        addr[kb] = static_cast<uint32_t>(got);
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        delta[kb] = -static_cast<int32_t>(addr[kb] & 3u);      // 0 .. -3: loaded byte j + 3.. is slot byte j
        addr[kb] &= ~3u;
      }
    }
#ifdef QNNP_ENABLE_ABLATION
    if (cg.abl & 1u) { addr[0] &= ~15u; addr[1] &= ~15u; }
    if (cg.abl & 8u) { addr[0] &= ~3u; addr[1] &= ~3u; }
    if (cg.abl & 16u) { addr[0] &= ~7u; addr[1] &= ~7u; }
    if (cg.abl & 4u) { addr[0] = addr[1] = (lane & 3u) * 16u; }
#endif
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      s.x[kb] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, addr[kb], 0, 0));
      if constexpr (ND > 3) s.tail[kb] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, addr[kb] + 16u, 0, 0);
    }
  };

  const uint32_t units = (p.rows / (cg.OH * cg.OW)) * cg.pairs * cg.segs;
  const uint32_t unit_stride = gridDim.x * kC3Waves;
  uint32_t unit = blockIdx.x * kC3Waves + wave;
  if (unit >= units) return;

  // one unit: its slots were requested a trip ago
  auto process = [&](const Unit& u, Slots& s, const int32_t (&delta)[2]) __attribute__((always_inline)) {
    if (u.border || u.edge) {                              // (scalar) taps outside the image -> the zero point
      const int32_t ix0 = u.ix0 + static_cast<int32_t>(pcol * cg.sw);
      const int32_t iy0 = u.iy0 + static_cast<int32_t>(prow * cg.sh);
      const int32_t left = ix0 < 0 ? -ix0 : 0;                                              // pixels
      const int32_t right = ix0 + static_cast<int32_t>(cg.KW) - static_cast<int32_t>(cg.W);   // > 0: pixels past the row
      const int32_t lo = 3 * left;
      const int32_t hi = static_cast<int32_t>(kbytes) - 3 * (right > 0 ? right : 0);
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const bool row_out = slot_real[kb] && static_cast<uint32_t>(iy0 + static_cast<int32_t>(kb * 2u + h)) >= cg.H;
        if (delta[kb] != 0 && !row_out) {
          // the load was moved by delta bytes: slot byte j is loaded byte j - delta (a slot moved DOWN by 1..3 bytes whose
          // last real byte sits in the tail dword only exists in windows 5 pixels wide: the tail is shifted in below)
          const uint32_t tail_in = (ND > 3 && delta[kb] < 0 && delta[kb] >= -3)
              ? s.tail[kb] << (8u * static_cast<uint32_t>(4 + delta[kb])) : 0u;
          uint64_t l64 = static_cast<uint32_t>(s.x[kb].x) | (static_cast<uint64_t>(static_cast<uint32_t>(s.x[kb].y)) << 32);
          uint64_t h64 = static_cast<uint32_t>(s.x[kb].z) | (static_cast<uint64_t>(static_cast<uint32_t>(s.x[kb].w)) << 32);
          const int32_t dl = delta[kb] > 15 ? 15 : (delta[kb] < -15 ? -15 : delta[kb]);
          if (dl > 0) {
            const uint32_t sh = 8u * static_cast<uint32_t>(dl);
            if (sh >= 64u) { h64 = l64 << (sh - 64u); l64 = 0; }
            else { h64 = (h64 << sh) | (l64 >> (64u - sh)); l64 <<= sh; }
          } else {
            const uint32_t sh = 8u * static_cast<uint32_t>(-dl);
            if (sh >= 64u) { l64 = h64 >> (sh - 64u); h64 = 0; }
            else { l64 = (l64 >> sh) | (h64 << (64u - sh)); h64 >>= sh; }
          }
          s.x[kb].x = static_cast<int>(static_cast<uint32_t>(l64)); s.x[kb].y = static_cast<int>(static_cast<uint32_t>(l64 >> 32));
          s.x[kb].z = static_cast<int>(static_cast<uint32_t>(h64)); s.x[kb].w = static_cast<int>(static_cast<uint32_t>(h64 >> 32) | tail_in);
        }
        int* xs = reinterpret_cast<int*>(&s.x[kb]);
#pragma unroll
        for (int d = 0; d < ND; d++) {
          // bytes of the slot that ARE image pixels
          const uint32_t keep = row_out ? 0u : byte_range_mask(lo - 4 * d, hi - 4 * d);
          xs[d] = static_cast<int>((static_cast<uint32_t>(xs[d]) & keep) | (fill4 & ~keep));
        }
      }
    }
    else {
      // interior: the slot starts 0..3 bytes into the loaded dwords
#pragma unroll
      for (int kb = 0; kb < 2; kb++) {
        const uint32_t shb = static_cast<uint32_t>(-delta[kb]);
        const uint32_t d0 = static_cast<uint32_t>(s.x[kb].x), d1 = static_cast<uint32_t>(s.x[kb].y);
        const uint32_t d2 = static_cast<uint32_t>(s.x[kb].z), d3 = static_cast<uint32_t>(s.x[kb].w);
        s.x[kb].x = static_cast<int>(__builtin_amdgcn_alignbyte(d1, d0, shb));
        s.x[kb].y = static_cast<int>(__builtin_amdgcn_alignbyte(d2, d1, shb));
        s.x[kb].z = static_cast<int>(__builtin_amdgcn_alignbyte(d3, d2, shb));
        if constexpr (ND > 3) s.x[kb].w = static_cast<int>(__builtin_amdgcn_alignbyte(s.tail[kb], d3, shb));
      }
    }
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      // (0x7F7F7F7F: the image is centred on kernel zero point 127 -- pack.h qnnp_pack_conv_rows16_centred127; the row-term fragments
      //  are zero then)
      const int flip = static_cast<int>(p.a_flip != 0u ? p.a_flip : kFlip);
      s.x[kb].x ^= flip; s.x[kb].y ^= flip;
      s.x[kb].z ^= flip; s.x[kb].w ^= flip;
    }
    const bool pixel_ok = prow < u.rows_left && pcol < u.cols_left;
    const uint32_t out_off = u.out0 + lane_out;
    // (round 6) two whole channel blocks: the 16-lane rows trade pieces and each store instruction writes one output row's sixteen
    // pixels whole (c3_store_unit) -- 224x224 3x3 3 -> 64, VGG's first layer, wrote 32-byte pieces 64 bytes apart
    const bool whole_lines = NB == 2 && (p.n & 15u) == 0u && p.n > 32u;
    v4i outv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][0], s.x[0], bias[nb], 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][1], s.x[1], acc, 0, 0, 0);
#pragma unroll
      for (int part = 0; part < RCP; part++) {
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wrc[part][0], s.x[0], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(wrc[part][1], s.x[1], acc, 0, 0, 0);
      }
      uint32_t pk[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        pk[rg] = q31_requantize_pack4<SEQ, FULL>(acc[rg * 4 + 0], acc[rg * 4 + 1], acc[rg * 4 + 2], acc[rg * 4 + 3], p.rq);
      }
      const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
      const v4i v = {static_cast<int>(s02[0]), static_cast<int>(s02[1]), static_cast<int>(s13[0]), static_cast<int>(s13[1])};
      outv[nb] = v;
      if (whole_lines) continue;                                // (wave-uniform)
      if constexpr (NB == 1) {
        // (round 6) 24 output channels, dense pixels -- ShuffleNet's 3 -> 24 first layer, in nine of the reference's lists: a unit row's
        // sixteen pixels are 384 contiguous bytes that left as 16-byte and 8-byte pieces 24 bytes apart. Through the wave's own 768 bytes
        // of LDS they leave as 48 whole 16-byte chunks (lane = chunk: 24 per unit row), one store instruction for both rows.
        if (p.n == 24u && p.output_stride == 24u && u.cols_left >= 16u) {      // (wave-uniform)
          __shared__ __attribute__((aligned(16))) uint8_t c3flat[kC3Waves * 768];
          uint8_t* mine = c3flat + wave * 768u;
          const uint32_t o = prow * 384u + pcol * 24u + h * 16u;
          *reinterpret_cast<uint2*>(mine + o) = make_uint2(static_cast<uint32_t>(v.x), static_cast<uint32_t>(v.y));
          if (h == 0u) *reinterpret_cast<uint2*>(mine + o + 8u) = make_uint2(static_cast<uint32_t>(v.z), static_cast<uint32_t>(v.w));
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          __builtin_amdgcn_wave_barrier();
          const uint32_t lane48 = (h << 5) | px;                  // this lane's index in the wave
          const uint32_t crow = lane48 >= 24u ? 1u : 0u;
          const uint32_t cj = lane48 - crow * 24u;
          const v4i c = *reinterpret_cast<const v4i*>(mine + min(lane48, 47u) * 16u);
          const bool cok = lane48 < 48u && crow < u.rows_left;
#ifdef QNNP_ENABLE_ABLATION
          const bool cok2 = cok && !((cg.abl & 2u) && c.x != 0x12345678);
#else
          const bool cok2 = cok;
#endif
          const auto cbits = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, c);
          const uint32_t coff = cok2 ? u.out0 + crow * cg.OW * 24u + cj * 16u : 0xFFFFFFF0u;
          if (p.stream_out) __builtin_amdgcn_raw_buffer_store_b128(cbits, out_rsrc, coff, 0, 2);
          else __builtin_amdgcn_raw_buffer_store_b128(cbits, out_rsrc, coff, 0, 0);
          __builtin_amdgcn_wave_barrier();                        // (the next unit's pieces must not overtake this unit's reads)
          continue;
        }
      }
      bool ok = pixel_ok && nb * 32u + h * 16u + 16u <= p.n;   // (n % 8 == 0: launcher)
#ifdef QNNP_ENABLE_ABLATION
      if (cg.abl & 2u) ok = ok && v.x == 0x12345678;
#endif
      if ((p.n & 8u) != 0u) {
        // (round 6: channel counts that end on half a 16-byte piece -- ShuffleNet's 3 -> 24 first layer, bench/convolution.cc:112 --
        //  store that half with an 8-byte store; wave-uniform test, nothing changes for multiples of 16)
        const bool ok8 = pixel_ok && nb * 32u + h * 16u + 8u == p.n;
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 lo = {static_cast<unsigned int>(v.x), static_cast<unsigned int>(v.y)};
        __builtin_amdgcn_raw_buffer_store_b64(lo, out_rsrc, ok8 ? out_off + nb * 32u : 0xFFFFFFF0u, 0, 0);
      }
      // (a unit's sixteen pixels of a row x 32 channels are 512 contiguous bytes, written once: the streaming hint, under
      //  the operator's "streaming_stores" setting -- the two builtins differ in an immediate, so hipcc cannot merge them)
      if (p.stream_out) {
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), out_rsrc,
            ok ? out_off + nb * 32u : 0xFFFFFFF0u, 0, 2);
      } else {
        __builtin_amdgcn_raw_buffer_store_b128(
            __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, v), out_rsrc,
            ok ? out_off + nb * 32u : 0xFFFFFFF0u, 0, 0);
      }
    }
    if constexpr (NB == 2) {
      if (whole_lines) {
        bool col_ok = pcol < u.cols_left;
#ifdef QNNP_ENABLE_ABLATION
        if (cg.abl & 2u) col_ok = col_ok && outv[0].x == 0x12345678;
#endif
        c3_store_unit<2>(outv, out_rsrc, u.out0, cg.OW * p.output_stride, p.output_stride, p.n, u.rows_left > 0u, u.rows_left > 1u,
                         col_ok, prow, pcol, h, p.stream_out != 0);
      }
    }
  };

  Unit ua = locate(unit), ub;
  Slots sa, sb;
  int32_t da[2], db[2];
  fetch(unit, ua, sa, da);
  // (two copies of the body with the register sets swapping roles: a loop-carried copy of a just-loaded register is a
  //  wait for it, DESIGN section 4.3 rule 4)
  for (;;) {
    uint32_t next = min(unit + unit_stride, units - 1u);
    ub = locate(next);
    fetch(next, ub, sb, db);
    process(ua, sa, da);
    if (unit + unit_stride >= units) break;
    unit = next;
    next = min(unit + unit_stride, units - 1u);
    ua = locate(next);
    fetch(next, ua, sa, da);
    process(ub, sb, db);
    if (unit + unit_stride >= units) break;
    unit = next;
  }
}

template <int NB, int ND, int RCP>
int launch_c3rows(const IgemmParams& p, const C3Geom& cg, hipStream_t stream)
{
  const uint32_t units = (p.rows / (cg.OH * cg.OW)) * cg.pairs * cg.segs;
  uint32_t grid = p.cu_count * (NB == 1 ? 4u : 3u);        // four (three with 64 channels) 4-wave workgroups per CU
  const uint32_t needed = (units + kC3Waves - 1) / kC3Waves;
  if (grid > needed) grid = needed;
  requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    hipLaunchKernelGGL((q8_conv_c3rows_kernel<NB, ND, RCP, decltype(seq)::value, decltype(full)::value>), dim3(grid),
                       dim3(kC3Threads), 0, stream, p, cg);
  });
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/*
 * The same idea for LARGER windows (round 5): K = [ky][32-byte row slot] -- K block ky is kernel row ky, a lane's operand half
 * h is bytes 16 h .. 16 h + 15 of that row's slot: KR = kernel_height fetches of 16 bytes per lane and unit instead of two.
 * ResNet's entry layer (bench/convolution.cc:646, 224x224x3 -> 112x112x64, 7x7 stride 2: 21 bytes per window row) ran 307 us
 * on the offset-table tile kernel (profiles/r05/bench_r05a.json: 0.05 of its HBM bound); this is one tenth of that.
 * Differences from q8_conv_c3rows_kernel above, all of them simplifications:
 *   - every fetch is dword-aligned (address & ~3) and moved into place with v_alignbyte; the launcher requires the tensor's
 *     size to be a multiple of four bytes, so no dword straddles its end and the first / last image need no special path:
 *     what lies outside the tensor is padding, fetched as zeros and replaced like every out-of-image tap;
 *   - the kernel-zero-point row term comes from the matrix cores too: KR MFMAs per unit multiply the re-centred slots with a
 *     `ones` fragment (1 in every real byte of the slot, 0 in its padding bytes), i.e. an extra channel block that holds the row
 *     sums -- KR more MFMAs per UNIT, not per channel block (the first build did it with v_dot4_u32_u8 + v_permlane32_swap on
 *     the VALU, which this kernel has no slots left for);
 *   - two sets of slot registers with swapping roles, as above: the next unit's fetches fly under the whole current unit.
 */
template <int NB, int KR, int SEQ, bool FULL>
__global__ __launch_bounds__(kC3Threads, 2)
void q8_conv_c3rows32_kernel(const IgemmParams p, const C3Geom cg)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t px = lane & 31u;
  const uint32_t h = lane >> 5;
  const uint32_t prow = px >> 4, pcol = px & 15u;
  const uint32_t kbytes = cg.KW * 3u;                       // real bytes of a row slot (<= 32; launcher)
  const int32_t nreal = static_cast<int32_t>(kbytes) - 16 * static_cast<int32_t>(h);   // ... of this lane's half (may be <= 0)

  v4i w[NB][KR];
  v16i bias[NB];
  {
    const int32_t* bias_tab = rq_is_lane<SEQ>() ? p.bias2u : p.bias2;      // (lane forms: bias + 2^31, the second half of the pair table)
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
#pragma unroll
      for (int kb = 0; kb < KR; kb++) w[nb][kb] = *reinterpret_cast<const v4i*>(cg.w_rows16 + ((nb * KR + kb) * 64u + lane) * 16u);
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const v4i b = *reinterpret_cast<const v4i*>(bias_tab + nb * 32 + rg * 8 + h * 4);
        bias[nb][rg * 4 + 0] = b.x; bias[nb][rg * 4 + 1] = b.y; bias[nb][rg * 4 + 2] = b.z; bias[nb][rg * 4 + 3] = b.w;
      }
    }
  }
  // 0x01 in every REAL byte of this lane's half: as a weight fragment, the same for all 32 "channels", its product with the
  // activation operand is the row sum of a' over the real K positions -- in EVERY element of the accumulator, so each lane finds
  // its position's sum in its own registers: KR more MFMAs on a pipe that is a third busy instead of 4 KR v_dot4 and an exchange
  const v4i ones = {static_cast<int>(byte_range_mask(0, nreal) & 0x01010101u), static_cast<int>(byte_range_mask(-4, nreal - 4) & 0x01010101u),
                    static_cast<int>(byte_range_mask(-8, nreal - 8) & 0x01010101u), static_cast<int>(byte_range_mask(-12, nreal - 12) & 0x01010101u)};

  const uint32_t in_bytes = static_cast<uint32_t>(p.input_end - p.input);          // (launcher: < 2^31, a multiple of 4)
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>((p.rows - 1u) * p.output_stride + p.n), 0x00020000);   // (launcher: < 2^31)

  const uint32_t row_bytes = cg.W * 3u;
  const uint32_t lane_in = prow * cg.sh * row_bytes + pcol * cg.sw * 3u + h * 16u;   // this lane's half of row slot 0, relative to the unit's first window
  const uint32_t fill4 = (p.izp_fill & 0xFFu) * 0x01010101u;
  const uint32_t flip = p.a_flip != 0u ? p.a_flip : kFlip;     // 0x7F7F7F7F: the image is centred on kernel zero point 127 (pack.h)

  struct Where { uint32_t origin, out0; int32_t iy0, ix0; uint32_t rows_left, cols_left; bool border, slow; };   // (wave-uniform)
  auto locate = [&](uint32_t unit) __attribute__((always_inline)) -> Where {
    const uint32_t t = div_magic(unit, cg.inv_segs);
    const uint32_t seg = unit - t * cg.segs;
    const uint32_t img = div_magic(t, cg.inv_pairs);
    const uint32_t pair = t - img * cg.pairs;
    const uint32_t oy = pair * 2u, ox = seg * 16u;
    Where u;
    u.iy0 = static_cast<int32_t>(oy * cg.sh) - static_cast<int32_t>(cg.pad_top);
    u.ix0 = static_cast<int32_t>(ox * cg.sw) - static_cast<int32_t>(cg.pad_left);
    u.rows_left = cg.OH - oy;
    u.cols_left = cg.OW - ox;
    u.out0 = ((img * cg.OH + oy) * cg.OW + ox) * p.output_stride;
    // byte offset of the unit's first window inside the tensor, modulo 2^32 ("negative" = beyond the descriptor's range: zeros)
    u.origin = img * static_cast<uint32_t>(p.image_stride) + static_cast<uint32_t>(u.iy0 * static_cast<int32_t>(cg.W) + u.ix0) * 3u;
    const int32_t iy_last = u.iy0 + static_cast<int32_t>(cg.sh + cg.KH) - 1;
    const int32_t ix_last = u.ix0 + static_cast<int32_t>(15u * cg.sw + cg.KW) - 1;
    u.border = u.iy0 < 0 || u.ix0 < 0 || iy_last >= static_cast<int32_t>(cg.H) || ix_last >= static_cast<int32_t>(cg.W);
    // the first windows of the first image start BEFORE the tensor: a fetch from a "negative" offset returns zeros for all
    // sixteen bytes, the real pixels behind the tensor's first byte included -- those few units fetch from offset >= 0 and
    // shift (below)
    u.slow = static_cast<int32_t>(u.origin) < 0;
    return u;
  };
  struct Slots { v4i x[KR]; uint32_t tail[KR]; uint32_t shifts; };     // shifts: 2 bits per row, bytes by which the fetch address was rounded down
  auto fetch = [&](const Where& u, Slots& s) __attribute__((always_inline)) {
    // (W % 4 == 0 gives every kernel row of a lane the same alignment; a flavour that fetched all rows from ONE rounded-down vector
    //  offset, the row advancing as the instruction's scalar offset, measured 15-25 % SLOWER -- 73-86 against 60-68 us, same box:
    //  this kernel is bound by the vector-memory path, not by its address arithmetic)
    uint32_t shifts = 0;
#pragma unroll
    for (int kb = 0; kb < KR; kb++) {
      uint32_t addr = u.origin + lane_in + kb * row_bytes;
      if (u.slow) addr = static_cast<uint32_t>(max(static_cast<int32_t>(addr), 0));       // (scalar branch; byte-aligned fetch, no tail needed)
      else { shifts |= (addr & 3u) << (2 * kb); addr &= ~3u; }
      s.x[kb] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, addr, 0, 0));
      s.tail[kb] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, addr + 16u, 0, 0);
    }
    s.shifts = shifts;
  };

  const uint32_t units = (p.rows / (cg.OH * cg.OW)) * cg.pairs * cg.segs;
  const uint32_t unit_stride = gridDim.x * kC3Waves;
  uint32_t unit = blockIdx.x * kC3Waves + wave;
  if (unit >= units) return;

  Where here = locate(unit);
  Slots s;
  fetch(here, s);
  for (;;) {
    // ---- the slots into place; out-of-image taps -> the zero point; re-centre
    const int32_t ixl = here.ix0 + static_cast<int32_t>(pcol * cg.sw);
    const int32_t iyl = here.iy0 + static_cast<int32_t>(prow * cg.sh);
    const int32_t left = ixl < 0 ? -ixl : 0;                                                  // pixels
    const int32_t right = ixl + static_cast<int32_t>(cg.KW) - static_cast<int32_t>(cg.W);     // > 0: pixels past the row
    const int32_t lo = 3 * left - 16 * static_cast<int32_t>(h);
    const int32_t hi = static_cast<int32_t>(kbytes) - 3 * (right > 0 ? right : 0) - 16 * static_cast<int32_t>(h);
    uint32_t keep_cols[4] = {~0u, ~0u, ~0u, ~0u};          // bytes of this half that are pixels of the image row (the same for every kernel row)
    if (here.border) {
#pragma unroll
      for (int d = 0; d < 4; d++) keep_cols[d] = byte_range_mask(lo - 4 * d, hi - 4 * d);
    }
#pragma unroll
    for (int kb = 0; kb < KR; kb++) {
      const uint32_t shb = __builtin_amdgcn_ubfe(s.shifts, 2 * kb, 2);
      const uint32_t d0 = static_cast<uint32_t>(s.x[kb].x), d1 = static_cast<uint32_t>(s.x[kb].y);
      const uint32_t d2 = static_cast<uint32_t>(s.x[kb].z), d3 = static_cast<uint32_t>(s.x[kb].w);
      uint32_t x[4] = {__builtin_amdgcn_alignbyte(d1, d0, shb), __builtin_amdgcn_alignbyte(d2, d1, shb),
                       __builtin_amdgcn_alignbyte(d3, d2, shb), __builtin_amdgcn_alignbyte(s.tail[kb], d3, shb)};
      if (here.slow) {                                     // (scalar; a handful of units per launch) loaded byte i is slot byte i + delta
        const int32_t want = static_cast<int32_t>(here.origin + lane_in + kb * row_bytes);
        const uint32_t delta = want < 0 ? static_cast<uint32_t>(min(-want, 16)) : 0u;
        uint64_t l64 = d0 | (static_cast<uint64_t>(d1) << 32), h64 = d2 | (static_cast<uint64_t>(d3) << 32);
        const uint32_t sh = 8u * delta;
        if (sh >= 128u) { l64 = 0; h64 = 0; }
        else if (sh >= 64u) { h64 = l64 << (sh - 64u); l64 = 0; }
        else if (sh != 0u) { h64 = (h64 << sh) | (l64 >> (64u - sh)); l64 <<= sh; }
        x[0] = static_cast<uint32_t>(l64); x[1] = static_cast<uint32_t>(l64 >> 32);
        x[2] = static_cast<uint32_t>(h64); x[3] = static_cast<uint32_t>(h64 >> 32);
      }
      if (here.border) {                                   // (scalar)
        const bool row_out = static_cast<uint32_t>(iyl + kb) >= cg.H;
#pragma unroll
        for (int d = 0; d < 4; d++) {
          const uint32_t keep = row_out ? 0u : keep_cols[d];
          x[d] = (x[d] & keep) | (fill4 & ~keep);
        }
      }
      s.x[kb] = v4i{static_cast<int>(x[0] ^ flip), static_cast<int>(x[1] ^ flip), static_cast<int>(x[2] ^ flip), static_cast<int>(x[3] ^ flip)};
    }
    // ---- the multiplies (and the row sums, where the kernel zero point asks for them); then the next unit's fetches, into the
    //      registers they have just left
    v16i acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      acc[nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][0], s.x[0], bias[nb], 0, 0, 0);
#pragma unroll
      for (int kb = 1; kb < KR; kb++) acc[nb] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][kb], s.x[kb], acc[nb], 0, 0, 0);
    }
    int32_t rowterm = with_rq_offset<SEQ>(0);
    if (p.row_coeff != 0) {                                // (scalar)
      v16i racc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < KR; kb++) racc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ones, s.x[kb], racc, 0, 0, 0);
      rowterm = with_rq_offset<SEQ>(p.row_coeff * racc[0]);
    }
    const Where done = here;
    const bool more = unit + unit_stride < units;
    if (more) {
      unit += unit_stride;
      here = locate(unit);
      fetch(here, s);
    }
    // ---- epilogue of the unit just multiplied: row term (in the multiply-add's addend where the lane forms apply), Q31
    //      requantization, 16-byte stores
    uint64_t row_addend = 0;
    if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
    v4i outv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      uint32_t pk[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        if constexpr (rq_is_lane<SEQ>()) {
          pk[rg] = q31_requantize_pack4_lane<SEQ, FULL>(
              static_cast<uint32_t>(acc[nb][rg * 4 + 0]), static_cast<uint32_t>(acc[nb][rg * 4 + 1]),
              static_cast<uint32_t>(acc[nb][rg * 4 + 2]), static_cast<uint32_t>(acc[nb][rg * 4 + 3]), row_addend, p.lane, p.rq);
        } else {
          pk[rg] = q31_requantize_pack4<SEQ, FULL, false>(add_wrap(acc[nb][rg * 4 + 0], rowterm), add_wrap(acc[nb][rg * 4 + 1], rowterm),
                                                         add_wrap(acc[nb][rg * 4 + 2], rowterm), add_wrap(acc[nb][rg * 4 + 3], rowterm), p.rq);
        }
      }
      const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
      outv[nb] = v4i{static_cast<int>(s02[0]), static_cast<int>(s02[1]), static_cast<int>(s13[0]), static_cast<int>(s13[1])};
    }
    c3_store_unit<NB>(outv, out_rsrc, done.out0, cg.OW * p.output_stride, p.output_stride, p.n, done.rows_left > 0u, done.rows_left > 1u,
                      pcol < done.cols_left, prow, pcol, h, p.stream_out != 0);     // (n % 16 == 0: launcher)
    if (!more) break;
  }
}

template <int NB, int KR>
int launch_c3rows32(const IgemmParams& p, const C3Geom& cg, hipStream_t stream)
{
  const uint32_t units = (p.rows / (cg.OH * cg.OW)) * cg.pairs * cg.segs;
  uint32_t grid = p.cu_count * 2u;                          // two 4-wave workgroups per CU (two waves per SIMD: ~220 registers)
  const uint32_t needed = (units + kC3Waves - 1) / kC3Waves;
  if (grid > needed) grid = needed;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    hipLaunchKernelGGL((q8_conv_c3rows32_kernel<NB, KR, decltype(seq)::value, decltype(full)::value>), dim3(grid),
                       dim3(kC3Threads), 0, stream, p, cg);
  });
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}


/*
 * The 32-byte-slot kernel with the input rows of a BAND staged in LDS (round 6). q8_conv_c3rows32_kernel above fetches every input
 * byte ~12 times through the vector-memory path (a 7 x 7 stride-2 window: 18 KB of requests per unit for 1.1 KB of distinct bytes)
 * and is bound by those requests. Here a workgroup owns `ppb` output-row PAIRS of one image, all column segments:
 *   - its (2 ppb - 1) * stride + KH input rows go to LDS once, by coalesced 16-byte loads, already re-centred (a ^ 0x80) and WITH the
 *     padding materialised -- rows above / below the image and the pad columns left and right of a row hold the re-centred zero point --
 *     so a unit has no border path, no byte masks and no XOR;
 *   - a lane's operand half of kernel row ky is 16 bytes at (row, column) of that image in LDS: five dwords from the dword below it
 *     (ds_read2_b32 x 2 + ds_read_b32: LDS wants its reads aligned to their width) and four v_alignbyte;
 *   - multiplies, row sums by the matrix cores, requantization and stores are the kernel's above.
 * Needs 16-byte aligned image rows (W * 3 % 16 == 0, base and image stride with it); everything else stays on the kernel above.
 */
struct C3LdsGeom {
  uint32_t ppb;               // output-row pairs per band
  uint32_t bands, inv_bands;  // bands per image
  uint32_t nrows;             // input rows of a band: (2 ppb - 1) * sh + KH
  uint32_t pitch;             // bytes between rows in LDS (a multiple of 16, an odd number of 16-byte chunks)
  uint32_t cpr, inv_cpr;      // chunks per row = pitch / 16
  uint32_t c0, dchunks;       // first data chunk of a row (the chunks before it are left padding), data chunks = W * 3 / 16
};

template <int NB, int KR, int SEQ, bool FULL>
__global__ __launch_bounds__(kC3Threads, NB == 1 ? 4 : (NB == 2 ? 3 : 2))
void q8_conv_c3rows32_lds_kernel(const IgemmParams p, const C3Geom cg, const C3LdsGeom lg)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t c3lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t px = lane & 31u;
  const uint32_t h = lane >> 5;
  const uint32_t prow = px >> 4, pcol = px & 15u;
  const uint32_t kbytes = cg.KW * 3u;
  const int32_t nreal = static_cast<int32_t>(kbytes) - 16 * static_cast<int32_t>(h);

  // (measurement builds: cycle stamps per wave -- entry, staged + barrier, first unit done, all units done; tools/trace_c3lds.py)
  QNNP_TRACE_WAVE(p, blockIdx.x, wave, 0);
  const uint32_t img = div_magic(blockIdx.x, lg.inv_bands);
  const uint32_t band = blockIdx.x - img * lg.bands;
  const uint32_t pair0 = band * lg.ppb;

  // ---- staging requests first: chunk id = row * cpr + c, four per thread and trip, all loads of a trip in flight together
  const uint32_t total = lg.nrows * lg.cpr;
  const int32_t iy_first = static_cast<int32_t>(pair0 * 2u * cg.sh) - static_cast<int32_t>(cg.pad_top);
  const uint8_t* img_base = p.input + static_cast<uint64_t>(img) * p.image_stride;
  const uint32_t row_bytes = cg.W * 3u;
  const int flip = static_cast<int>(p.a_flip != 0u ? p.a_flip : kFlip);     // 0x7F7F7F7F: the image is centred on kernel zero point 127
  const int zp4 = static_cast<int>((p.izp_fill & 0xFFu) * 0x01010101u) ^ flip;
  constexpr int kTrip = 4;
  auto stage_trip = [&](uint32_t id0) __attribute__((always_inline)) {
    v4i v[kTrip];
    uint32_t dst[kTrip];
    bool in[kTrip];
#pragma unroll
    for (int i = 0; i < kTrip; i++) {
      const uint32_t id = id0 + static_cast<uint32_t>(i) * kC3Threads;
      const uint32_t r = div_magic(id, lg.inv_cpr);
      const uint32_t c = id - r * lg.cpr;
      const int32_t iy = iy_first + static_cast<int32_t>(r);
      in[i] = id < total && static_cast<uint32_t>(iy) < cg.H && c - lg.c0 < lg.dchunks;
      dst[i] = id < total ? id * 16u : 0xFFFFFFFFu;
      if (in[i]) v[i] = *reinterpret_cast<const v4i*>(img_base + static_cast<uint32_t>(iy) * row_bytes + (c - lg.c0) * 16u);
    }
#pragma unroll
    for (int i = 0; i < kTrip; i++) {
      v4i y = {zp4, zp4, zp4, zp4};
      if (in[i]) y = v4i{v[i].x ^ flip, v[i].y ^ flip, v[i].z ^ flip, v[i].w ^ flip};
      if (dst[i] != 0xFFFFFFFFu) *reinterpret_cast<v4i*>(c3lds + dst[i]) = y;
    }
  };

  // ---- weights and bias to registers (L2-resident; their round trip runs under the staging)
  v4i w[NB][KR];
  v16i bias[NB];
  {
    const int32_t* bias_tab = rq_is_lane<SEQ>() ? p.bias2u : p.bias2;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
#pragma unroll
      for (int kb = 0; kb < KR; kb++) w[nb][kb] = *reinterpret_cast<const v4i*>(cg.w_rows16 + ((nb * KR + kb) * 64u + lane) * 16u);
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const v4i b = *reinterpret_cast<const v4i*>(bias_tab + nb * 32 + rg * 8 + h * 4);
        bias[nb][rg * 4 + 0] = b.x; bias[nb][rg * 4 + 1] = b.y; bias[nb][rg * 4 + 2] = b.z; bias[nb][rg * 4 + 3] = b.w;
      }
    }
  }
  for (uint32_t id0 = tid; id0 < total; id0 += kTrip * kC3Threads) stage_trip(id0);
  QNNP_TRACE_WAVE(p, blockIdx.x, wave, 1);
  __syncthreads();
  QNNP_TRACE_WAVE(p, blockIdx.x, wave, 2);

  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>((p.rows - 1u) * p.output_stride + p.n), 0x00020000);
  // this lane's half of row slot 0 of the band's first unit, in LDS: row prow * sh, column pcol * sw - pad_left behind the data start
  const uint32_t lane_lds = prow * cg.sh * lg.pitch + lg.c0 * 16u + pcol * cg.sw * 3u + h * 16u - cg.pad_left * 3u;

  const uint32_t pairs_here = min(lg.ppb, cg.pairs - pair0);
  const uint32_t nunits = pairs_here * cg.segs;
  for (uint32_t unit = wave; unit < nunits; unit += kC3Waves) {
    const uint32_t pl = div_magic(unit, cg.inv_segs);
    const uint32_t seg = unit - pl * cg.segs;
    const uint32_t oy = (pair0 + pl) * 2u, ox = seg * 16u;
    const uint32_t off = lane_lds + pl * 2u * cg.sh * lg.pitch + ox * cg.sw * 3u;
    const uint32_t shb = off & 3u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(c3lds + (off & ~3u));
    v4i x[KR];
#pragma unroll
    for (int kb = 0; kb < KR; kb++) {
      const uint32_t* q = src + kb * (lg.pitch >> 2);
      const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
      x[kb] = v4i{static_cast<int>(__builtin_amdgcn_alignbyte(d1, d0, shb)), static_cast<int>(__builtin_amdgcn_alignbyte(d2, d1, shb)),
                  static_cast<int>(__builtin_amdgcn_alignbyte(d3, d2, shb)), static_cast<int>(__builtin_amdgcn_alignbyte(d4, d3, shb))};
    }
    // the row term first (kernel zero points other than 127 / 128), then channel block by channel block: multiply, requantize -- the
    // next block's multiplies run under this block's requantization, and one set of accumulator registers serves all blocks
    int32_t rowterm = with_rq_offset<SEQ>(0);
    if (p.row_coeff != 0) {                                // (scalar; the `ones` fragment of the kernel above, rebuilt per unit: four
      //  registers held across the loop cost this flavour its third wave per SIMD)
      int32_t nr = nreal;
      asm volatile("" : "+v"(nr));                         // (or hipcc hoists the fragment out of the loop again)
      const v4i ones = {static_cast<int>(byte_range_mask(0, nr) & 0x01010101u), static_cast<int>(byte_range_mask(-4, nr - 4) & 0x01010101u),
                        static_cast<int>(byte_range_mask(-8, nr - 8) & 0x01010101u), static_cast<int>(byte_range_mask(-12, nr - 12) & 0x01010101u)};
      v16i racc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int kb = 0; kb < KR; kb++) racc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ones, x[kb], racc, 0, 0, 0);
      rowterm = with_rq_offset<SEQ>(p.row_coeff * racc[0]);
    }
    uint64_t row_addend = 0;
    if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
    v4i outv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][0], x[0], bias[nb], 0, 0, 0);
#pragma unroll
      for (int kb = 1; kb < KR; kb++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][kb], x[kb], acc, 0, 0, 0);
      uint32_t pk[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        if constexpr (rq_is_lane<SEQ>()) {
          pk[rg] = q31_requantize_pack4_lane<SEQ, FULL>(
              static_cast<uint32_t>(acc[rg * 4 + 0]), static_cast<uint32_t>(acc[rg * 4 + 1]),
              static_cast<uint32_t>(acc[rg * 4 + 2]), static_cast<uint32_t>(acc[rg * 4 + 3]), row_addend, p.lane, p.rq);
        } else {
          pk[rg] = q31_requantize_pack4<SEQ, FULL, false>(add_wrap(acc[rg * 4 + 0], rowterm), add_wrap(acc[rg * 4 + 1], rowterm),
                                                         add_wrap(acc[rg * 4 + 2], rowterm), add_wrap(acc[rg * 4 + 3], rowterm), p.rq);
        }
      }
      const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
      outv[nb] = v4i{static_cast<int>(s02[0]), static_cast<int>(s02[1]), static_cast<int>(s13[0]), static_cast<int>(s13[1])};
    }
    c3_store_unit<NB>(outv, out_rsrc, ((img * cg.OH + oy) * cg.OW + ox) * p.output_stride, cg.OW * p.output_stride, p.output_stride, p.n,
                      oy < cg.OH, oy + 1u < cg.OH, ox + pcol < cg.OW, prow, pcol, h, p.stream_out != 0);
    if (unit == wave) QNNP_TRACE_WAVE(p, blockIdx.x, wave, 3);
  }
  QNNP_TRACE_WAVE(p, blockIdx.x, wave, 4);
}

/* the band plan of the LDS flavour; false: the shape stays on the register-path kernel */
bool c3lds_plan(const IgemmParams& p, const C3Geom& cg, C3LdsGeom* out)
{
  if ((cg.W * 3u) % 16u != 0 || (reinterpret_cast<uintptr_t>(p.input) & 15u) != 0 || p.image_stride % 16u != 0) return false;
  C3LdsGeom lg;
  lg.c0 = (cg.pad_left * 3u + 15u) / 16u;
  lg.dchunks = cg.W * 3u / 16u;
  // bytes a real tap can touch right of the data: the last window's end; + the 20 junk bytes a lane reads past its slot's real ones
  const uint32_t last_end = ((cg.OW - 1u) * cg.sw + cg.KW) * 3u;                       // relative to column -pad_left
  const uint32_t data_end = (cg.pad_left + cg.W) * 3u;
  const uint32_t right = last_end > data_end ? last_end - data_end : 0u;
  lg.cpr = lg.c0 + lg.dchunks + (right + 15u) / 16u + 1u;
  if ((lg.cpr & 1u) == 0u) lg.cpr++;
  lg.pitch = lg.cpr * 16u;
  lg.inv_cpr = static_cast<uint32_t>(((UINT64_C(1) << 32) + lg.cpr - 1) / lg.cpr);
  // pairs per band: as many as keep a band under 24 KiB (several workgroups per CU), units per band a multiple of the waves if possible
  uint32_t best = 0;
  for (uint32_t ppb = 1; ppb <= cg.pairs && ppb <= 16u; ppb++) {
    const uint32_t nrows = (2u * ppb - 1u) * cg.sh + cg.KH;
    if (nrows * lg.pitch > 24u * 1024u) break;
    if (best == 0 || (ppb * cg.segs) % kC3Waves == 0 || (best * cg.segs) % kC3Waves != 0) best = ppb;
  }
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_C3L_PPB")) {          // measurement builds: pairs per band by hand
    const uint32_t ppb = static_cast<uint32_t>(atoi(env));
    if (ppb >= 1 && ppb <= cg.pairs && ((2u * ppb - 1u) * cg.sh + cg.KH) * lg.pitch <= 60u * 1024u) best = ppb;
  }
#endif
  if (best == 0) return false;
  lg.ppb = best;
  lg.nrows = (2u * best - 1u) * cg.sh + cg.KH;
  lg.bands = (cg.pairs + best - 1u) / best;
  lg.inv_bands = lg.bands > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + lg.bands - 1) / lg.bands) : 0u;
  const uint64_t wgs = static_cast<uint64_t>(p.rows / (cg.OH * cg.OW)) * lg.bands;
  if (wgs * lg.bands >= (UINT64_C(1) << 32) || static_cast<uint64_t>(lg.nrows) * lg.cpr * lg.cpr >= (UINT64_C(1) << 32)) return false;
  *out = lg;
  return true;
}

template <int NB, int KR>
int launch_c3rows32_lds(const IgemmParams& p, const C3Geom& cg, const C3LdsGeom& lg, hipStream_t stream)
{
  const uint32_t grid = (p.rows / (cg.OH * cg.OW)) * lg.bands;
  // (slack behind the last row: lanes of positions past the image's edge read on, at most 16 columns and a slot further)
  const uint32_t lds_bytes = lg.nrows * lg.pitch + 16u * cg.sw * 3u + 64u;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    hipLaunchKernelGGL((q8_conv_c3rows32_lds_kernel<NB, KR, decltype(seq)::value, decltype(full)::value>), dim3(grid),
                       dim3(kC3Threads), lds_bytes, stream, p, cg, lg);
  });
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}


/*
 * The 16-byte-slot kernel with the band staged in LDS (round 6; q8_conv_c3rows32_lds_kernel's scheme for windows of <= 4 rows of <= 16
 * bytes: MobileNet's / ShuffleNet's 224 x 224 3x3 stride-2 first layers, in eleven of the reference's lists and in the sweep). The
 * register-path kernel above spends ~200 instructions per unit of 32 pixels -- address arithmetic, two unaligned fetches with their
 * tails, border surgery, re-centring -- around FOUR useful matrix instructions; here a unit is 2 x (ds_read2_b32 x 2 + ds_read_b32) + 8
 * v_alignbyte + 2 MFMAs per channel block + the requantization. Needs a row term of zero (kernel zero point 128, or 127 with the centred
 * image convolution.c builds) and image rows of whole 16-byte chunks; everything else stays on the kernel above.
 * Lane (pixel p, half h): K block kb holds window row ky = 2 kb + h (row 3 of a 3-row window meets zero weights).
 */
template <int NB, int SEQ, bool FULL>
__global__ __launch_bounds__(kC3Threads, NB == 1 ? 4 : 3)
void q8_conv_c3rows_lds_kernel(const IgemmParams p, const C3Geom cg, const C3LdsGeom lg)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t c3lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t px = lane & 31u;
  const uint32_t h = lane >> 5;
  const uint32_t prow = px >> 4, pcol = px & 15u;

  const uint32_t img = div_magic(blockIdx.x, lg.inv_bands);
  const uint32_t band = blockIdx.x - img * lg.bands;
  const uint32_t pair0 = band * lg.ppb;

  // ---- weights and bias to registers (L2-resident; their round trip runs under the staging) ----
  v4i w[NB][2];
  v16i bias[NB];
  {
    const int32_t* bias_tab = rq_is_lane<SEQ>() ? p.bias2u : p.bias2;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
#pragma unroll
      for (int kb = 0; kb < 2; kb++) w[nb][kb] = *reinterpret_cast<const v4i*>(cg.w_rows16 + ((nb * 2 + kb) * 64u + lane) * 16u);
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const v4i b = *reinterpret_cast<const v4i*>(bias_tab + nb * 32 + rg * 8 + h * 4);
        bias[nb][rg * 4 + 0] = b.x; bias[nb][rg * 4 + 1] = b.y; bias[nb][rg * 4 + 2] = b.z; bias[nb][rg * 4 + 3] = b.w;
      }
    }
  }

  // ---- staging: chunk id = row * cpr + c, four per thread and trip, all loads of a trip in flight together ----
  {
    const uint32_t total = lg.nrows * lg.cpr;
    const int32_t iy_first = static_cast<int32_t>(pair0 * 2u * cg.sh) - static_cast<int32_t>(cg.pad_top);
    const uint8_t* img_base = p.input + static_cast<uint64_t>(img) * p.image_stride;
    const uint32_t row_bytes = cg.W * 3u;
    const int flip = static_cast<int>(p.a_flip != 0u ? p.a_flip : kFlip);
    const int zp4 = static_cast<int>((p.izp_fill & 0xFFu) * 0x01010101u) ^ flip;
    constexpr int kTrip = 4;
    for (uint32_t id0 = tid; id0 < total; id0 += kTrip * kC3Threads) {
      v4i v[kTrip];
      uint32_t dst[kTrip];
      bool in[kTrip];
#pragma unroll
      for (int i = 0; i < kTrip; i++) {
        const uint32_t id = id0 + static_cast<uint32_t>(i) * kC3Threads;
        const uint32_t r = div_magic(id, lg.inv_cpr);
        const uint32_t c = id - r * lg.cpr;
        const int32_t iy = iy_first + static_cast<int32_t>(r);
        in[i] = id < total && static_cast<uint32_t>(iy) < cg.H && c - lg.c0 < lg.dchunks;
        dst[i] = id < total ? id * 16u : 0xFFFFFFFFu;
        if (in[i]) v[i] = *reinterpret_cast<const v4i*>(img_base + static_cast<uint32_t>(iy) * row_bytes + (c - lg.c0) * 16u);
      }
#pragma unroll
      for (int i = 0; i < kTrip; i++) {
        v4i y = {zp4, zp4, zp4, zp4};
        if (in[i]) y = v4i{v[i].x ^ flip, v[i].y ^ flip, v[i].z ^ flip, v[i].w ^ flip};
        if (dst[i] != 0xFFFFFFFFu) *reinterpret_cast<v4i*>(c3lds + dst[i]) = y;
      }
    }
  }
  __syncthreads();

  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>((p.rows - 1u) * p.output_stride + p.n), 0x00020000);
  // this lane's slot of window row h of the band's first unit: row prow * sh + h, column pcol * sw - pad_left behind the data start
  const uint32_t lane_lds = (prow * cg.sh + h) * lg.pitch + lg.c0 * 16u + pcol * cg.sw * 3u - cg.pad_left * 3u;

  const uint32_t pairs_here = min(lg.ppb, cg.pairs - pair0);
  const uint32_t nunits = pairs_here * cg.segs;
  for (uint32_t unit = wave; unit < nunits; unit += kC3Waves) {
    const uint32_t pl = div_magic(unit, cg.inv_segs);
    const uint32_t seg = unit - pl * cg.segs;
    const uint32_t oy = (pair0 + pl) * 2u, ox = seg * 16u;
    const uint32_t off = lane_lds + pl * 2u * cg.sh * lg.pitch + ox * cg.sw * 3u;
    const uint32_t shb = off & 3u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(c3lds + (off & ~3u));
    v4i x[2];
#pragma unroll
    for (int kb = 0; kb < 2; kb++) {
      const uint32_t* q = src + kb * 2 * (lg.pitch >> 2);
      const uint32_t d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
      x[kb] = v4i{static_cast<int>(__builtin_amdgcn_alignbyte(d1, d0, shb)), static_cast<int>(__builtin_amdgcn_alignbyte(d2, d1, shb)),
                  static_cast<int>(__builtin_amdgcn_alignbyte(d3, d2, shb)), static_cast<int>(__builtin_amdgcn_alignbyte(d4, d3, shb))};
    }
    const int32_t rowterm = with_rq_offset<SEQ>(0);
    uint64_t row_addend = 0;
    if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
    const uint32_t unit_out0 = ((img * cg.OH + oy) * cg.OW + ox) * p.output_stride;
    const bool row0_ok = oy < cg.OH, row1_ok = oy + 1u < cg.OH;
    const uint32_t cols_left = cg.OW - ox;
    v4i outv[NB];
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][0], x[0], bias[nb], 0, 0, 0);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[nb][1], x[1], acc, 0, 0, 0);
      uint32_t pk[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        if constexpr (rq_is_lane<SEQ>()) {
          pk[rg] = q31_requantize_pack4_lane<SEQ, FULL>(
              static_cast<uint32_t>(acc[rg * 4 + 0]), static_cast<uint32_t>(acc[rg * 4 + 1]),
              static_cast<uint32_t>(acc[rg * 4 + 2]), static_cast<uint32_t>(acc[rg * 4 + 3]), row_addend, p.lane, p.rq);
        } else {
          pk[rg] = q31_requantize_pack4<SEQ, FULL, false>(add_wrap(acc[rg * 4 + 0], rowterm), add_wrap(acc[rg * 4 + 1], rowterm),
                                                         add_wrap(acc[rg * 4 + 2], rowterm), add_wrap(acc[rg * 4 + 3], rowterm), p.rq);
        }
      }
      const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
      outv[nb] = v4i{static_cast<int>(s02[0]), static_cast<int>(s02[1]), static_cast<int>(s13[0]), static_cast<int>(s13[1])};
    }
    const bool pixel_ok = (prow != 0u ? row1_ok : row0_ok) && pcol < cols_left;
    if (NB == 2 && (p.n & 15u) == 0u && p.n > 32u) {            // (wave-uniform) two whole blocks: whole-line stores
      if constexpr (NB == 2) {
        c3_store_unit<2>(outv, out_rsrc, unit_out0, cg.OW * p.output_stride, p.output_stride, p.n, row0_ok, row1_ok, pcol < cols_left, prow,
                         pcol, h, p.stream_out != 0);
      }
      continue;
    }
    if (NB == 1 && p.n == 24u && p.output_stride == 24u && cols_left >= 16u) {   // (wave-uniform) 24 channels: flat rows of 384 bytes
      __shared__ __attribute__((aligned(16))) uint8_t c3flat_l[kC3Waves * 768];
      uint8_t* mine = c3flat_l + wave * 768u;
      const uint32_t o = prow * 384u + pcol * 24u + h * 16u;
      *reinterpret_cast<uint2*>(mine + o) = make_uint2(static_cast<uint32_t>(outv[0].x), static_cast<uint32_t>(outv[0].y));
      if (h == 0u) *reinterpret_cast<uint2*>(mine + o + 8u) = make_uint2(static_cast<uint32_t>(outv[0].z), static_cast<uint32_t>(outv[0].w));
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const uint32_t crow = lane >= 24u ? 1u : 0u;
      const v4i c = *reinterpret_cast<const v4i*>(mine + min(lane, 47u) * 16u);
      const bool cok = lane < 48u && (crow != 0u ? row1_ok : row0_ok);
      const auto cbits = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, c);
      const uint32_t coff = cok ? unit_out0 + crow * cg.OW * 24u + (lane - crow * 24u) * 16u : 0xFFFFFFF0u;
      if (p.stream_out) __builtin_amdgcn_raw_buffer_store_b128(cbits, out_rsrc, coff, 0, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(cbits, out_rsrc, coff, 0, 0);
      __builtin_amdgcn_wave_barrier();
      continue;
    }
    const uint32_t out_off = unit_out0 + (prow * cg.OW + pcol) * p.output_stride + h * 16u;
#pragma unroll
    for (int nb = 0; nb < NB; nb++) {
      const bool ok = pixel_ok && nb * 32u + h * 16u + 16u <= p.n;             // (n % 8 == 0: launcher)
      if ((p.n & 8u) != 0u) {
        const bool ok8 = pixel_ok && nb * 32u + h * 16u + 8u == p.n;
        typedef unsigned int u2 __attribute__((ext_vector_type(2)));
        const u2 lo = {static_cast<unsigned int>(outv[nb].x), static_cast<unsigned int>(outv[nb].y)};
        __builtin_amdgcn_raw_buffer_store_b64(lo, out_rsrc, ok8 ? out_off + nb * 32u : 0xFFFFFFF0u, 0, 0);
      }
      const auto bits = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, outv[nb]);
      if (p.stream_out) __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, ok ? out_off + nb * 32u : 0xFFFFFFF0u, 0, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, ok ? out_off + nb * 32u : 0xFFFFFFF0u, 0, 0);
    }
  }
}

template <int NB>
int launch_c3rows_lds(const IgemmParams& p, const C3Geom& cg, const C3LdsGeom& lg, hipStream_t stream)
{
  const uint32_t grid = (p.rows / (cg.OH * cg.OW)) * lg.bands;
  // (slack: lanes of positions past the image's edge read on -- at most 16 columns further -- and K block 1 of half 1 reads the row behind
  //  the window, which the band's last unit does not have)
  const uint32_t lds_bytes = (lg.nrows + 1u) * lg.pitch + 16u * cg.sw * 3u + 64u;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    hipLaunchKernelGGL((q8_conv_c3rows_lds_kernel<NB, decltype(seq)::value, decltype(full)::value>), dim3(grid),
                       dim3(kC3Threads), lds_bytes, stream, p, cg, lg);
  });
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

}  // namespace

/* dense 3-byte pixels, one group, window rows of <= 16 bytes and <= 4 rows, 32 or 64 output channels in whole 16-byte
 * pieces, tensors addressable with 32-bit offsets */
bool conv_c3rows_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, const int8_t* w_rows16, uint32_t real_kc)
{
  if (w_rows16 == nullptr || groups != 1 || real_kc != 3 || p.input_stride != 3) return false;
  if (g.KW * 3u > 16u || g.KH == 0 || g.KH > 4 || g.dh != 1 || g.dw != 1) return false;
  // (channel counts in multiples of 8, pixels 8-byte aligned: the last half piece leaves by an 8-byte store -- round 6)
  if (p.n_pad > 64 || p.n % 8u != 0 || p.output_stride % 8u != 0 || (reinterpret_cast<uintptr_t>(p.output) & 7u) != 0) return false;
  if (p.n % 16u == 0 && (p.output_stride % 16u != 0 || (reinterpret_cast<uintptr_t>(p.output) & 15u) != 0)) return false;
  if (p.rows == 0 || p.rows_per_image == 0 || g.OW == 0 || g.OH == 0 || p.rows_per_image != g.OH * g.OW) return false;
  if (p.rows % p.rows_per_image != 0) return false;
  const uint64_t in_bytes = static_cast<uint64_t>(p.input_end - p.input);
  const uint64_t out_bytes = static_cast<uint64_t>(p.rows) * p.output_stride;
  if (in_bytes < 16 || in_bytes >= (UINT64_C(1) << 31) || out_bytes >= (UINT64_C(1) << 31)) return false;
  // units = images x row pairs x 16-column segments; the reciprocal divisions of the unit index are exact while
  // dividend * divisor < 2^32
  const uint64_t segs = (g.OW + 15u) / 16u, pairs = (g.OH + 1u) / 2u;
  const uint64_t units = static_cast<uint64_t>(p.rows / p.rows_per_image) * pairs * segs;
  if (units * segs >= (UINT64_C(1) << 32) || units * pairs >= (UINT64_C(1) << 32)) return false;
  return p.row_coeff >= -127 && p.row_coeff <= 128;
}

/* flavour: 0 = auto (the LDS-staged kernel where there is no row term and its plan takes the shape), 1 = the register-path kernel,
 * 2 = the LDS-staged kernel or QNNP_HIP_EINVAL */
int conv_c3rows_launch(const IgemmParams& p, const ConvGeom& g, const int8_t* w_rows16, hipStream_t stream, const char** name, int flavour)
{
  C3Geom cg;
  cg.H = g.H; cg.W = g.W; cg.OH = g.OH; cg.OW = g.OW; cg.KH = g.KH; cg.KW = g.KW; cg.sh = g.sh; cg.sw = g.sw;
  cg.pad_top = g.pad_top; cg.pad_left = g.pad_left;
  cg.segs = (g.OW + 15u) / 16u;
  cg.pairs = (g.OH + 1u) / 2u;
  cg.inv_segs = cg.segs > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + cg.segs - 1) / cg.segs) : 0u;
  cg.inv_pairs = cg.pairs > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + cg.pairs - 1) / cg.pairs) : 0u;
  cg.w_rows16 = w_rows16;
  cg.abl = 0;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_C3R_ABL")) cg.abl = static_cast<uint32_t>(atoi(env));
#endif
  const bool two = p.n_pad > 32;
  {
    C3LdsGeom lg;
    const bool lds = flavour != 1 && p.row_coeff == 0 && c3lds_plan(p, cg, &lg);
    if (flavour == 2 && !lds) return QNNP_HIP_EINVAL;
    if (lds) {
      *name = "q8_conv_c3rows_lds_mfma";
      return two ? launch_c3rows_lds<2>(p, cg, lg, stream) : launch_c3rows_lds<1>(p, cg, lg, stream);
    }
  }
  *name = "q8_conv_c3rows_mfma";
  const bool wide = g.KW * 3u > 12u;
  if (p.row_coeff == 128) {         // kernel zero point 0: the row-term weight does not fit int8, applied as 64 + 64
    if (wide) return two ? launch_c3rows<2, 4, 2>(p, cg, stream) : launch_c3rows<1, 4, 2>(p, cg, stream);
    return two ? launch_c3rows<2, 3, 2>(p, cg, stream) : launch_c3rows<1, 3, 2>(p, cg, stream);
  }
  if (wide) return two ? launch_c3rows<2, 4, 1>(p, cg, stream) : launch_c3rows<1, 4, 1>(p, cg, stream);
  return two ? launch_c3rows<2, 3, 1>(p, cg, stream) : launch_c3rows<1, 3, 1>(p, cg, stream);
}

/* the 32-byte-slot flavour: dense 3-byte pixels, one group, 5 or 7 window rows of <= 32 bytes, a tensor of whole dwords */
bool conv_c3rows32_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, const int8_t* w_rows32, uint32_t real_kc)
{
  if (w_rows32 == nullptr || groups != 1 || real_kc != 3 || p.input_stride != 3) return false;
  if (g.KW * 3u > 32u || !(g.KH == 5 || g.KH == 7) || g.dh != 1 || g.dw != 1) return false;
  // (round 6: three channel blocks for the 7-row window -- SqueezeNet 1.0's 7x7 stride-2 3 -> 96 entry layer, bench/convolution.cc:541)
  if (p.n_pad > (g.KH == 7 ? 96u : 64u) || p.n % 16u != 0 || p.output_stride % 16u != 0 || (reinterpret_cast<uintptr_t>(p.output) & 15u) != 0) return false;
  if (p.rows == 0 || p.rows_per_image == 0 || g.OW == 0 || g.OH == 0 || p.rows_per_image != g.OH * g.OW) return false;
  if (p.rows % p.rows_per_image != 0) return false;
  const uint64_t in_bytes = static_cast<uint64_t>(p.input_end - p.input);
  const uint64_t out_bytes = static_cast<uint64_t>(p.rows) * p.output_stride;
  if (in_bytes < 16 || in_bytes % 4u != 0 || (reinterpret_cast<uintptr_t>(p.input) & 3u) != 0) return false;
  if (in_bytes >= (UINT64_C(1) << 31) || out_bytes >= (UINT64_C(1) << 31)) return false;
  if (p.image_stride != static_cast<uint64_t>(g.H) * g.W * 3u) return false;
  const uint64_t segs = (g.OW + 15u) / 16u, pairs = (g.OH + 1u) / 2u;
  const uint64_t units = static_cast<uint64_t>(p.rows / p.rows_per_image) * pairs * segs;
  if (units * segs >= (UINT64_C(1) << 32) || units * pairs >= (UINT64_C(1) << 32)) return false;
  return true;
}

/* flavour: 0 = auto (the LDS-staged kernel where its plan takes the shape), 1 = the register-path kernel, 2 = the LDS-staged kernel or
 * QNNP_HIP_EINVAL */
int conv_c3rows32_launch(const IgemmParams& p, const ConvGeom& g, const int8_t* w_rows32, hipStream_t stream, const char** name, int flavour)
{
  C3Geom cg;
  cg.H = g.H; cg.W = g.W; cg.OH = g.OH; cg.OW = g.OW; cg.KH = g.KH; cg.KW = g.KW; cg.sh = g.sh; cg.sw = g.sw;
  cg.pad_top = g.pad_top; cg.pad_left = g.pad_left;
  cg.segs = (g.OW + 15u) / 16u;
  cg.pairs = (g.OH + 1u) / 2u;
  cg.inv_segs = cg.segs > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + cg.segs - 1) / cg.segs) : 0u;
  cg.inv_pairs = cg.pairs > 1 ? static_cast<uint32_t>(((UINT64_C(1) << 32) + cg.pairs - 1) / cg.pairs) : 0u;
  cg.w_rows16 = w_rows32;
  cg.abl = 0;
  const bool two = p.n_pad > 32;
  C3LdsGeom lg;
  // (auto: three channel blocks stay on the register-path kernel -- 232 registers leave the LDS flavour two waves per SIMD and nothing
  //  to run under a workgroup's staging: 224x224 7x7 stride 2, 3 -> 96: 67 against 88 us, profiles/r06/conv7x7_lds_staged_ab_r06v.txt)
  const bool lds = flavour != 1 && !(flavour == 0 && p.n_pad > 64) && c3lds_plan(p, cg, &lg);
  if (flavour == 2 && !lds) return QNNP_HIP_EINVAL;
  if (lds) {
    *name = "q8_conv_c3rows32_lds_mfma";
    if (g.KH == 7 && p.n_pad > 64) return launch_c3rows32_lds<3, 7>(p, cg, lg, stream);
    if (g.KH == 7) return two ? launch_c3rows32_lds<2, 7>(p, cg, lg, stream) : launch_c3rows32_lds<1, 7>(p, cg, lg, stream);
    return two ? launch_c3rows32_lds<2, 5>(p, cg, lg, stream) : launch_c3rows32_lds<1, 5>(p, cg, lg, stream);
  }
  *name = "q8_conv_c3rows32_mfma";
  if (g.KH == 7 && p.n_pad > 64) return launch_c3rows32<3, 7>(p, cg, stream);
  if (g.KH == 7) return two ? launch_c3rows32<2, 7>(p, cg, stream) : launch_c3rows32<1, 7>(p, cg, stream);
  return two ? launch_c3rows32<2, 5>(p, cg, stream) : launch_c3rows32<1, 5>(p, cg, stream);
}

}  // namespace qnnp
