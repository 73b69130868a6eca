/*
 * q8dwconv.hip -- uint8 depthwise convolution, NHWC, fused Q31 requantization.
 *
 * Replaces, as whole-operator launches, the reference's per-output-row CPU
 * microkernels
 *   q8dwconv_ukernel_up8x9__sse2  (src/q8dwconv/up8x9-sse2.c:14-372, 3x3 unipass)
 *   q8dwconv_ukernel_mp8x25__sse2 (src/q8dwconv/mp8x25-sse2.c:14-742, 5x5 multipass)
 * their pthreadpool fan-out compute_dwconv_unipass / _multiipass
 * (src/operator-run.c:238-284, 675-679) and the pointer indirection buffer of
 * qnnp_indirection_init_dwconv2d (src/indirection.c:81-132), which is not needed:
 * tap coordinates are computed in-kernel.
 *
 * Arithmetic (exact int32, same folding as pack_q8dw_w, src/qnnpack/pack.h:146-159):
 *   out[c] = requant( bias1[c] + sum_taps a(tap, c) * (w(tap, c) - kzp) ),
 *   bias1 = bias + taps*izp*kzp - izp*sum_taps w,   padding taps read a = izp.
 *
 * This is not a dense contraction (one input channel per output channel), so it
 * does not go to the matrix cores; it is an HBM-streaming kernel whose work is
 * the coalesced NHWC traffic plus the per-element requantization.
 *
 * Kernel A  q8_dwconv_lds_kernel<KH, KW, VECL>   (3x3 and 5x5, C % 4 == 0)
 *   workgroup = one image x a band of output rows x all output columns x a channel
 *   slab. The input band (with halo, padding materialised as the zero point) is
 *   staged into LDS with 16-byte (or 4-byte) coalesced loads along C, laid out
 *   [row][4-channel group][column] so an output's taps are immediate offsets apart; each thread
 *   owns one 4-channel group (its tap weights live in registers as int16 pairs)
 *   and walks output positions, reading dwords from LDS, pairing taps with
 *   v_perm_b32 and accumulating two taps per v_dot2_i32_i16; one coalesced dword
 *   store of 4 requantized channels per position.
 *
 * Kernel B  q8_dwconv_direct_kernel   (any kernel size / channel count / stride)
 *   one thread per output element, channels fastest; byte loads through L1/L2.
 *   Correctness fallback for shapes kernel A does not take.
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include "igemm_epilogue.hip.h"
#include "igemm_params.h"
#include "qnnp_hip.h"
#include "requant.hip.h"

extern "C" void* qnnp_hip_get_stream(void);

namespace {

typedef short v2s __attribute__((ext_vector_type(2)));

struct DwParams {
  const uint8_t* input;
  uint8_t* output;
  const int16_t* wadj;
  const int32_t* bias1;
  uint32_t batch;
  uint32_t H, W, OH, OW;
  uint32_t C, c_pad;
  uint32_t KH, KW;
  uint32_t sh, sw, dh, dw;
  uint32_t pad_top, pad_left;
  uint32_t in_stride, out_stride;
  uint32_t izp;
  // LDS-tiled kernel geometry
  uint32_t CS;        // channels per slab (multiple of 4, divides C)
  uint32_t TOH;       // output rows per band
  uint32_t IR, IC;    // staged input rows / columns per band
  uint32_t PP;        // LDS dwords per (row, 4-channel group) line: IC padded for bank spread
  uint32_t bands;     // ceil(OH / TOH)
  uint32_t slabs;     // C / CS
  uint32_t cu_count;  // compute units of the bound device
  // matrix-core kernel
  const int8_t* dwm_x;
  const int32_t* dwm_bias;
  uint32_t dwm_parts, c_pad32;
  uint32_t wrange;       // qnnp_dwconv_weight_range of wadj (pack.h): 1 / 2 = the int8 dot-product flavour of kernel G applies
  const uint32_t* dot4;  // [4][c_pad] register image of that flavour (pack.h qnnp_pack_dwconv_dot4), or null
  uint32_t xcd_ranges;   // kernel G: 1 = contiguous ranges of the work per XCD (launch_col)
  uint32_t inv_bands, inv_slabs, inv_q4;   // kernel G: ceil(2^32 / d) of its three divisors (0 when d == 1), launch_col
  uint32_t inv_dh;       // kernel G, dilated flavour: the same for the row dilation
  uint32_t store_mode;   // as igemm_epilogue.hip.h: 2 = 16-byte stores, 1 = dword stores, 0 = byte stores
  uint32_t abl;          // measurement builds only: bit 0 = no stores, bit 1 = no global loads (kernel F)
  unsigned long long* trace;   // measurement builds only (QNNP_ENABLE_ABLATION + env QNNP_GFX950_TRACE)
  qnnp::RequantDev rq;
  uint32_t stream_out;       // 1: the column-walk kernels mark their output stores as streaming ("streaming_stores")
};

#ifdef QNNP_ENABLE_ABLATION
#define QNNP_DW_TRACE(p, slot)                                                                      \
  do {                                                                                              \
    if ((p).trace != nullptr && threadIdx.x == 0 && blockIdx.x < 4096)                              \
      (p).trace[(blockIdx.x * 4) * 8 + (slot)] = __builtin_readcyclecounter();                     \
  } while (0)
#else
#define QNNP_DW_TRACE(p, slot) do { } while (0)
#endif

// --------------------------------------------------------------------------
// Kernel B: generic direct
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void q8_dwconv_direct_kernel(const DwParams p)
{
  const uint64_t total = static_cast<uint64_t>(p.batch) * p.OH * p.OW * p.C;
  for (uint64_t idx = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; idx < total;
       idx += static_cast<uint64_t>(gridDim.x) * blockDim.x) {
    const uint32_t c = static_cast<uint32_t>(idx % p.C);
    const uint64_t pix = idx / p.C;
    const uint32_t ox = static_cast<uint32_t>(pix % p.OW);
    const uint64_t t = pix / p.OW;
    const uint32_t oy = static_cast<uint32_t>(t % p.OH);
    const uint32_t n = static_cast<uint32_t>(t / p.OH);
    int32_t acc = p.bias1[c];
    for (uint32_t ky = 0; ky < p.KH; ky++) {
      const uint32_t iy = oy * p.sh + ky * p.dh - p.pad_top;   // unsigned wrap = out of range
      for (uint32_t kx = 0; kx < p.KW; kx++) {
        const uint32_t ix = ox * p.sw + kx * p.dw - p.pad_left;
        int32_t a = static_cast<int32_t>(p.izp);
        if (iy < p.H && ix < p.W) {
          a = p.input[((static_cast<uint64_t>(n) * p.H + iy) * p.W + ix) * p.in_stride + c];
        }
        acc += a * static_cast<int32_t>(p.wadj[(ky * p.KW + kx) * p.c_pad + c]);
      }
    }
    p.output[pix * p.out_stride + c] = static_cast<uint8_t>(qnnp::q31_requantize(acc, p.rq));
  }
}

// --------------------------------------------------------------------------
// Kernel B4 (round 6): generic direct, four channels per thread
// --------------------------------------------------------------------------
/*
 * What the shapes nothing else takes ran on until round 6 was kernel B: one output BYTE per thread, nine byte loads each -- ShuffleNet
 * v2's depthwise layers with 58 / 122 channels (bench/convolution.cc:335-426) at 0.03-0.07 of their bounds (28 x 28 x 122: 102 us).
 * Same arithmetic (reference: q8dwconv_ukernel_up8x9__sse2 / mp8x25, src/q8dwconv/up8x9-sse2.c:14-372), any window, stride, dilation,
 * channel count and pixel stride; a thread owns four consecutive channels of one output pixel: per tap two DWORD-ALIGNED loads around
 * its four bytes (a pixel of 58 bytes starts anywhere) joined by v_alignbyte, one 8-byte load of the four int16 tap weights. Bytes
 * past the tensor read 0 (buffer descriptor) and channels past C are computed and not stored.
 */
__global__ __launch_bounds__(256)
void q8_dwconv_direct4_kernel(const DwParams p)
{
  const uint32_t q4 = (p.C + 3u) / 4u;
  const uint64_t total = static_cast<uint64_t>(p.batch) * p.OH * p.OW * q4;
  // (the extent rounded up to whole dwords: the dword holding the tensor's last bytes must not read as out of range)
  const uint64_t in_bytes = ((static_cast<uint64_t>(p.batch) * p.H * p.W - 1u) * p.in_stride + p.C + 3u) & ~static_cast<uint64_t>(3);
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(in_bytes), 0x00020000);         // (make_plan: < 2^31 bytes)
  // (32-bit index arithmetic -- make_plan: fewer than 2^32 channel groups in all -- a 64-bit division costs ~100 instructions)
  const uint32_t total32 = static_cast<uint32_t>(total);
  for (uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total32; idx += gridDim.x * blockDim.x) {
    const uint32_t pix = idx / q4;
    const uint32_t cg = idx - pix * q4;
    const uint32_t t = pix / p.OW;
    const uint32_t ox = pix - t * p.OW;
    const uint32_t n = t / p.OH;
    const uint32_t oy = t - n * p.OH;
    const uint32_t c0 = cg * 4u;
    const int4 bv = *reinterpret_cast<const int4*>(p.bias1 + c0);                        // (c_pad is a multiple of four: pack.h)
    int32_t acc[4] = {bv.x, bv.y, bv.z, bv.w};
    for (uint32_t ky = 0; ky < p.KH; ky++) {
      const uint32_t iy = oy * p.sh + ky * p.dh - p.pad_top;   // unsigned wrap = out of range
      for (uint32_t kx = 0; kx < p.KW; kx++) {
        const uint32_t ix = ox * p.sw + kx * p.dw - p.pad_left;
        uint32_t a4 = p.izp * 0x01010101u;
        if (iy < p.H && ix < p.W) {
          const uint32_t off = ((n * p.H + iy) * p.W + ix) * p.in_stride + c0;
          const uint32_t lo = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, off & ~3u, 0, 0);
          const uint32_t hi = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, (off & ~3u) + 4u, 0, 0);
          a4 = __builtin_amdgcn_alignbyte(hi, lo, off & 3u);
        }
        const uint2 wv = *reinterpret_cast<const uint2*>(p.wadj + (ky * p.KW + kx) * p.c_pad + c0);
        acc[0] += static_cast<int32_t>(a4 & 0xFFu) * static_cast<int16_t>(wv.x & 0xFFFFu);
        acc[1] += static_cast<int32_t>((a4 >> 8) & 0xFFu) * static_cast<int16_t>(wv.x >> 16);
        acc[2] += static_cast<int32_t>((a4 >> 16) & 0xFFu) * static_cast<int16_t>(wv.y & 0xFFFFu);
        acc[3] += static_cast<int32_t>(a4 >> 24) * static_cast<int16_t>(wv.y >> 16);
      }
    }
    uint8_t* out = p.output + static_cast<uint64_t>(pix) * p.out_stride + c0;
    const uint32_t q = static_cast<uint32_t>(qnnp::q31_requantize(acc[0], p.rq)) | (static_cast<uint32_t>(qnnp::q31_requantize(acc[1], p.rq)) << 8) |
                       (static_cast<uint32_t>(qnnp::q31_requantize(acc[2], p.rq)) << 16) | (static_cast<uint32_t>(qnnp::q31_requantize(acc[3], p.rq)) << 24);
    const uint32_t nv = min(4u, p.C - c0);
    const uintptr_t addr = reinterpret_cast<uintptr_t>(out);
    if (nv == 4u && (addr & 3u) == 0u) {
      *reinterpret_cast<uint32_t*>(out) = q;
    } else if (nv == 4u && (addr & 1u) == 0u) {
      *reinterpret_cast<uint16_t*>(out) = static_cast<uint16_t>(q);
      *reinterpret_cast<uint16_t*>(out + 2) = static_cast<uint16_t>(q >> 16);
    } else {
      for (uint32_t b = 0; b < nv; b++) out[b] = static_cast<uint8_t>(q >> (8 * b));
    }
  }
}

// --------------------------------------------------------------------------
// Kernel A: LDS-tiled
// --------------------------------------------------------------------------
#ifndef QNNP_DW_THREADS
#define QNNP_DW_THREADS 512
#endif
constexpr int kDwThreads = QNNP_DW_THREADS;

/*
 * LDS image of the staged band: [input row][4-channel group][column] dwords, i.e. for one row and one
 * 4-channel group the columns are consecutive dwords. The KW taps of an output are then 4*dw bytes apart
 * (immediate ds_read offsets when dw == 1) and only one address per kernel row is computed.
 * `p.PP` = dwords per (row, group) line, padded so that consecutive groups start on spread-out banks.
 */
template <int KH, int KW, int VECL, bool DW1>
__global__ __launch_bounds__(kDwThreads)
void q8_dwconv_lds_kernel(const DwParams p)
{
  constexpr int TAPS = KH * KW;
  constexpr int PAIRS = (TAPS + 1) / 2;
  extern __shared__ __attribute__((aligned(16))) uint8_t tile[];   // [IR][CS/4][PP] dwords

  const uint32_t tid = threadIdx.x;
  QNNP_DW_TRACE(p, 0);
  // block -> (image, band, slab); slab fastest so that the blocks sharing input
  // cache lines (same pixels, neighbouring channel slabs) are dispatched together
  // XCD-aware bijective remap (hardware: block b -> XCD b % 8): consecutive logical ids -- the channel slabs
  // of one band, which read the same cache lines -- land on the same XCD / L2
  uint32_t b;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const uint32_t slab = b % p.slabs; b /= p.slabs;
  const uint32_t band = b % p.bands;
  const uint32_t n = b / p.bands;
  const uint32_t c0 = slab * p.CS;
  const uint32_t oy0 = band * p.TOH;
  const uint32_t toh = min(p.TOH, p.OH - oy0);
  const uint32_t ir = (toh - 1) * p.sh + (KH - 1) * p.dh + 1;   // rows actually needed
  const uint32_t q4 = p.CS / 4;                 // 4-channel groups in the slab
  const uint32_t line_bytes = p.PP * 4;         // one (row, group) line
  const uint32_t row_bytes = q4 * line_bytes;

  // ---- stage the input band: coalesced VECL-byte vectors along C, scattered into the group planes ----
  // Batches of kStageBatch vectors per thread: every global load of a batch is issued before the first
  // LDS write, so a thread keeps several loads in flight (a load -> write -> load loop is latency-bound).
  {
    constexpr int kStageBatch = (VECL == 16) ? 4 : 8;
    const uint32_t vpp = p.CS / VECL;                 // vectors per pixel
    const uint32_t nvec = ir * p.IC * vpp;
    const int32_t iy_base = static_cast<int32_t>(oy0 * p.sh) - static_cast<int32_t>(p.pad_top);
    const int32_t ix_base = -static_cast<int32_t>(p.pad_left);
    const uint8_t* img = p.input + static_cast<uint64_t>(n) * p.H * p.W * p.in_stride + c0;
    const uint32_t fill = p.izp * 0x01010101u;
    // incremental (vector-in-pixel, column, row) walk: no divisions in the loop
    uint32_t cv = tid % vpp;
    uint32_t px = tid / vpp;
    uint32_t ixl = px % p.IC;
    uint32_t iyl = px / p.IC;
    const uint32_t d_cv = kDwThreads % vpp;
    const uint32_t d_px = kDwThreads / vpp;
    const uint32_t d_ix = d_px % p.IC;
    const uint32_t d_iy = d_px / p.IC;
    for (uint32_t v0 = tid; v0 < nvec; v0 += kDwThreads * kStageBatch) {
      uint32_t dst[kStageBatch];
      bool live[kStageBatch];
      uint4 val16[VECL == 16 ? kStageBatch : 1];
      uint32_t val4[VECL == 16 ? 1 : kStageBatch];
#pragma unroll
      for (int u = 0; u < kStageBatch; u++) {
        live[u] = v0 + u * kDwThreads < nvec;
        const int32_t iy = iy_base + static_cast<int32_t>(iyl);
        const int32_t ix = ix_base + static_cast<int32_t>(ixl);
        const bool inb = live[u] && iy >= 0 && iy < static_cast<int32_t>(p.H) && ix >= 0 && ix < static_cast<int32_t>(p.W);
        dst[u] = iyl * row_bytes + (cv * (VECL / 4)) * line_bytes + ixl * 4;
        const uint8_t* src =
            img + (static_cast<uint64_t>(static_cast<uint32_t>(iy)) * p.W + static_cast<uint32_t>(ix)) * p.in_stride + cv * VECL;
        if constexpr (VECL == 16) {
          val16[u] = make_uint4(fill, fill, fill, fill);
          if (inb) val16[u] = *reinterpret_cast<const uint4*>(src);
        } else {
          val4[u] = fill;
          if (inb) val4[u] = *reinterpret_cast<const uint32_t*>(src);
        }
        // advance by kDwThreads vectors
        cv += d_cv;
        uint32_t carry = 0;
        if (cv >= vpp) { cv -= vpp; carry = 1; }
        ixl += d_ix + carry;
        iyl += d_iy;
        if (ixl >= p.IC) { ixl -= p.IC; iyl += 1; }
      }
#pragma unroll
      for (int u = 0; u < kStageBatch; u++) {
        if (live[u]) {
          uint8_t* d = tile + dst[u];
          if constexpr (VECL == 16) {
            *reinterpret_cast<uint32_t*>(d) = val16[u].x;
            *reinterpret_cast<uint32_t*>(d + line_bytes) = val16[u].y;
            *reinterpret_cast<uint32_t*>(d + 2 * line_bytes) = val16[u].z;
            *reinterpret_cast<uint32_t*>(d + 3 * line_bytes) = val16[u].w;
          } else {
            *reinterpret_cast<uint32_t*>(d) = val4[u];
          }
        }
      }
    }
  }

  // ---- per-thread channel group: weights as (tap 2i, tap 2i+1) int16 pairs ----
  const uint32_t nslots = kDwThreads / q4;      // positions processed concurrently
  const uint32_t c4 = tid % q4;
  const uint32_t slot = tid / q4;
  const bool active = slot < nslots;
  const uint32_t cg = c0 + c4 * 4;              // first global channel of this thread

  uint32_t wpair[PAIRS][4];
  int32_t bias[4];
  if (active) {
#pragma unroll
    for (int i = 0; i < PAIRS; i++) {
      const uint2 lo = *reinterpret_cast<const uint2*>(p.wadj + (2 * i) * p.c_pad + cg);   // 4 x int16
      uint2 hi = make_uint2(0u, 0u);
      if (2 * i + 1 < TAPS) hi = *reinterpret_cast<const uint2*>(p.wadj + (2 * i + 1) * p.c_pad + cg);
      wpair[i][0] = (lo.x & 0xFFFFu) | (hi.x << 16);
      wpair[i][1] = (lo.x >> 16) | (hi.x & 0xFFFF0000u);
      wpair[i][2] = (lo.y & 0xFFFFu) | (hi.y << 16);
      wpair[i][3] = (lo.y >> 16) | (hi.y & 0xFFFF0000u);
    }
    const int4 bv = *reinterpret_cast<const int4*>(p.bias1 + cg);
    bias[0] = bv.x; bias[1] = bv.y; bias[2] = bv.z; bias[3] = bv.w;
  }

  QNNP_DW_TRACE(p, 1);
  __syncthreads();
  QNNP_DW_TRACE(p, 2);
  if (!active) return;

  const uint32_t npos = toh * p.OW;
  uint8_t* out_ptr = p.output + (static_cast<uint64_t>(n) * p.OH + oy0) * p.OW * p.out_stride + cg +
                     static_cast<uint64_t>(slot) * p.out_stride;
  const uint64_t out_step = static_cast<uint64_t>(nslots) * p.out_stride;
  // incremental (row, column) walk over this thread's positions slot, slot + nslots, ...
  uint32_t ox = slot % p.OW;
  uint32_t oyl = slot / p.OW;
  const uint32_t d_ox = nslots % p.OW;
  const uint32_t d_oy = nslots / p.OW;
  const uint32_t tap_row_bytes = p.dh * row_bytes;
  const uint32_t tap_col_bytes = p.dw * 4;
  // the requantization flavour (shift == 0? clamp == [0,255]?) is chosen once, outside the position loop
  // the LDS address of a position's first tap walks along with (ox, oyl): no multiplies inside the loop
  uint32_t base_off = (oyl * p.sh) * row_bytes + c4 * line_bytes + (ox * p.sw) * 4;
  const uint32_t d_base = (d_oy * p.sh) * row_bytes + (d_ox * p.sw) * 4;
  const uint32_t wrap_base = p.sh * row_bytes - (p.OW * p.sw) * 4;      // one row down, OW columns back
  qnnp::requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
    // the offset rounding sequences (requant.hip.h) take accumulator + 2^31: folded into the bias, once per thread
#pragma unroll
    for (int c = 0; c < 4; c++) bias[c] = qnnp::with_rq_offset<decltype(shift0)::value>(bias[c]);
    for (uint32_t pos = slot; pos < npos; pos += nslots) {
      const uint8_t* base = tile + base_off;
      uint32_t in[TAPS];
#pragma unroll
      for (int ky = 0; ky < KH; ky++) {
        const uint8_t* row = base + ky * tap_row_bytes;
#pragma unroll
        for (int kx = 0; kx < KW; kx++) {
          if constexpr (DW1) {
            in[ky * KW + kx] = *reinterpret_cast<const uint32_t*>(row + kx * 4);          // immediate offset
          } else {
            in[ky * KW + kx] = *reinterpret_cast<const uint32_t*>(row + kx * tap_col_bytes);
          }
        }
      }
      int32_t acc0 = bias[0], acc1 = bias[1], acc2 = bias[2], acc3 = bias[3];
#pragma unroll
      for (int i = 0; i < PAIRS; i++) {
        const uint32_t in0 = in[2 * i];
        const uint32_t in1 = (2 * i + 1 < TAPS) ? in[2 * i + 1] : 0u;
        // v_perm_b32: result bytes {in0.c, 0, in1.c, 0} = the two taps of channel c as int16 x2
        const uint32_t p0 = __builtin_amdgcn_perm(in1, in0, 0x0c040c00u);
        const uint32_t p1 = __builtin_amdgcn_perm(in1, in0, 0x0c050c01u);
        const uint32_t p2 = __builtin_amdgcn_perm(in1, in0, 0x0c060c02u);
        const uint32_t p3 = __builtin_amdgcn_perm(in1, in0, 0x0c070c03u);
        acc0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p0), __builtin_bit_cast(v2s, wpair[i][0]), acc0, false);
        acc1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p1), __builtin_bit_cast(v2s, wpair[i][1]), acc1, false);
        acc2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p2), __builtin_bit_cast(v2s, wpair[i][2]), acc2, false);
        acc3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p3), __builtin_bit_cast(v2s, wpair[i][3]), acc3, false);
      }
      const uint32_t packed = qnnp::q31_requantize_pack4<decltype(shift0)::value, decltype(full)::value>(
          acc0, acc1, acc2, acc3, p.rq);
      *reinterpret_cast<uint32_t*>(out_ptr) = packed;
      out_ptr += out_step;
      ox += d_ox;
      base_off += d_base;
      if (ox >= p.OW) { ox -= p.OW; base_off += wrap_base; }
    }
  });
  QNNP_DW_TRACE(p, 3);
  QNNP_DW_TRACE(p, 4);
  QNNP_DW_TRACE(p, 5);
}

// --------------------------------------------------------------------------
// Kernel C: register sliding window, 3x3, dilation 1, stride 1 or 2, C % 4 == 0
// --------------------------------------------------------------------------
/*
 * No LDS, no barrier: a thread owns one 4-channel group of one output row segment and slides a 3x3 window
 * of input dwords (4 channels each) along it in registers -- per output only the SW new window columns are
 * loaded (3 or 6 dwords, coalesced along C across the lanes: consecutive lanes = consecutive channel
 * groups, then consecutive output rows, so the three kernel rows of neighbouring lanes hit the same lines
 * in L1). The vertical reuse (an input row feeds up to three output rows) is left to L1/L2.
 * Why: the LDS kernel spends ~70 % of a workgroup's life staging its band behind a barrier (in-kernel cycle
 * stamps); here loads, the VALU tap arithmetic and the requantization of different waves overlap freely at
 * 8+ waves per SIMD.
 * Padding: rows / columns outside the image are clamped to a valid address and replaced by the input zero
 * point after the load (checked path, taken only by waves that touch a border).
 */
constexpr int kRowThreads = 256;

/* UNAL (round 6): any channel count >= 4 and pixels / tensors of any alignment -- ShuffleNet v2's 58 / 122 / 244 / 488-channel depthwise
 * layers (bench/convolution.cc:338-426), which ran on the generic q8_dwconv_direct4 at 0.05-0.13 of their bound. A pixel's channels are
 * cut into (C + 3) / 4 groups of four, the last one starting at C - 4 (it recomputes up to three channels of its neighbour: the same
 * bytes, written twice); every load and store is a dword at whatever address it falls on (the hardware takes unaligned dwords from
 * global memory; the weight and bias tables are read the same way), everything else is the kernel's. */
template <typename T>
struct __attribute__((packed)) DwAnyAligned { T v; };
template <typename T>
__device__ __forceinline__ T dw_load_any(const void* ptr) { return reinterpret_cast<const DwAnyAligned<T>*>(ptr)->v; }
// (HIP's vector types are not POD for the packed attribute: their unaligned loads are spelled out dword by dword)
template <>
__device__ __forceinline__ uint2 dw_load_any<uint2>(const void* ptr)
{
  const uint8_t* b = static_cast<const uint8_t*>(ptr);
  return make_uint2(dw_load_any<uint32_t>(b), dw_load_any<uint32_t>(b + 4));
}
template <>
__device__ __forceinline__ int4 dw_load_any<int4>(const void* ptr)
{
  const uint8_t* b = static_cast<const uint8_t*>(ptr);
  return make_int4(dw_load_any<int32_t>(b), dw_load_any<int32_t>(b + 4), dw_load_any<int32_t>(b + 8), dw_load_any<int32_t>(b + 12));
}
template <typename T>
__device__ __forceinline__ void dw_store_any(void* ptr, T v) { reinterpret_cast<DwAnyAligned<T>*>(ptr)->v = v; }

template <int SW, bool UNAL = false>
__global__ __launch_bounds__(kRowThreads)
void q8_dwconv_row3x3_kernel(const DwParams p)
{
  constexpr int PAIRS = 5;
  QNNP_DW_TRACE(p, 0);
  const uint32_t q4 = UNAL ? (p.C + 3u) / 4u : p.C / 4;
  const uint32_t t = blockIdx.x * kRowThreads + threadIdx.x;       // host checked: fits 32 bits
  const uint32_t c4 = t % q4;
  uint32_t r = t / q4;
  const uint32_t oy = r % p.OH; r /= p.OH;
  const uint32_t seg = r % p.slabs;                                 // `slabs` = column segments per row here
  const uint32_t n = r / p.slabs;
  const bool live = n < p.batch;
  const uint32_t nn = live ? n : 0u;
  const uint32_t cg = UNAL ? min(c4 * 4u, p.C - 4u) : c4 * 4;

  // tap weights as (tap 2i, tap 2i+1) int16 pairs, bias
  uint32_t wpair[PAIRS][4];
  int32_t bias[4];
#pragma unroll
  for (int i = 0; i < PAIRS; i++) {
    const uint2 lo = dw_load_any<uint2>(p.wadj + (2 * i) * p.c_pad + cg);   // 4 x int16
    uint2 hi = make_uint2(0u, 0u);
    if (2 * i + 1 < 9) hi = dw_load_any<uint2>(p.wadj + (2 * i + 1) * p.c_pad + cg);
    wpair[i][0] = (lo.x & 0xFFFFu) | (hi.x << 16);
    wpair[i][1] = (lo.x >> 16) | (hi.x & 0xFFFF0000u);
    wpair[i][2] = (lo.y & 0xFFFFu) | (hi.y << 16);
    wpair[i][3] = (lo.y >> 16) | (hi.y & 0xFFFF0000u);
  }
  {
    const int4 bv = dw_load_any<int4>(p.bias1 + cg);
    bias[0] = bv.x; bias[1] = bv.y; bias[2] = bv.z; bias[3] = bv.w;
  }

  const uint32_t fill = p.izp * 0x01010101u;
  // the three input rows of this output row: 32-bit byte offsets from p.input (host checked < 4 GiB)
  uint32_t voff[3];
  bool row_ok[3];
  bool rows_need_check = false;
#pragma unroll
  for (int ky = 0; ky < 3; ky++) {
    const int32_t iy = static_cast<int32_t>(oy * p.sh) - static_cast<int32_t>(p.pad_top) + ky;
    row_ok[ky] = iy >= 0 && iy < static_cast<int32_t>(p.H);
    rows_need_check |= !row_ok[ky];
    const uint32_t iyc = row_ok[ky] ? static_cast<uint32_t>(iy) : 0u;
    voff[ky] = ((nn * p.H + iyc) * p.W) * p.in_stride + cg;
  }
  const bool wave_rows_check = __builtin_amdgcn_ballot_w64(rows_need_check) != 0;

  const uint32_t ox0 = seg * p.TOH;                                  // `TOH` = outputs per column segment here
  const uint32_t ox1 = live ? min(p.OW, ox0 + p.TOH) : ox0;
  uint8_t* out_ptr = p.output + (static_cast<uint64_t>(nn * p.OH + oy) * p.OW + ox0) * p.out_stride + cg;

  // one window column (three kernel rows) at input column ix
  auto load_col = [&](int32_t ix, uint32_t (&col)[3], bool check) __attribute__((always_inline)) {
    if (check) {
      const bool col_ok = ix >= 0 && ix < static_cast<int32_t>(p.W);
      const uint32_t coff = (col_ok ? static_cast<uint32_t>(ix) : 0u) * p.in_stride;
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        const uint32_t v = dw_load_any<uint32_t>(p.input + (voff[ky] + coff));
        col[ky] = (col_ok && row_ok[ky]) ? v : fill;
      }
    } else {
      const uint32_t coff = static_cast<uint32_t>(ix) * p.in_stride;
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        col[ky] = dw_load_any<uint32_t>(p.input + (voff[ky] + coff));
      }
    }
  };

  qnnp::requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
#pragma unroll
    for (int c = 0; c < 4; c++) bias[c] = qnnp::with_rq_offset<decltype(shift0)::value>(bias[c]);
    // window columns kx = 0, 1, 2 of the current output plus the SW columns the NEXT output adds: those are
    // loaded one step ahead, so a load has a whole step of VALU work (and the other waves) to land
    uint32_t w0[3], w1[3], w2[3], n0[3], n1[3];
    int32_t ix = static_cast<int32_t>(ox0 * SW) - static_cast<int32_t>(p.pad_left);   // leftmost window column
    QNNP_DW_TRACE(p, 1);
    load_col(ix, w0, true);
    load_col(ix + 1, w1, true);
    load_col(ix + 2, w2, true);
#ifdef QNNP_ENABLE_ABLATION
    if (p.trace != nullptr) { asm volatile("" :: "v"(w0[0]), "v"(w1[1]), "v"(w2[2])); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
#endif
    QNNP_DW_TRACE(p, 2);
    for (uint32_t ox = ox0; ox < ox1; ox++) {
      // next step's new columns; a wave takes the checked path if any of its lanes touches a border
      const int32_t nx = ix + 3;
      const bool lane_check = nx + (SW - 1) >= static_cast<int32_t>(p.W);
      const bool check = wave_rows_check || lane_check;     // (no ballot here: it would block the unrolling)
      if (ox + 1 < ox1) {
        if (check) {
          load_col(nx, n0, true);
          if (SW == 2) load_col(nx + 1, n1, true);
        } else {
          load_col(nx, n0, false);
          if (SW == 2) load_col(nx + 1, n1, false);
        }
      }
      const uint32_t in[9] = {w0[0], w1[0], w2[0], w0[1], w1[1], w2[1], w0[2], w1[2], w2[2]};
      int32_t acc0 = bias[0], acc1 = bias[1], acc2 = bias[2], acc3 = bias[3];
#pragma unroll
      for (int i = 0; i < PAIRS; i++) {
        const uint32_t in0 = in[2 * i];
        const uint32_t in1 = (2 * i + 1 < 9) ? in[2 * i + 1] : 0u;
        // v_perm_b32: result bytes {in0.c, 0, in1.c, 0} = the two taps of channel c as int16 x2
        const uint32_t p0 = __builtin_amdgcn_perm(in1, in0, 0x0c040c00u);
        const uint32_t p1 = __builtin_amdgcn_perm(in1, in0, 0x0c050c01u);
        const uint32_t p2 = __builtin_amdgcn_perm(in1, in0, 0x0c060c02u);
        const uint32_t p3 = __builtin_amdgcn_perm(in1, in0, 0x0c070c03u);
        acc0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p0), __builtin_bit_cast(v2s, wpair[i][0]), acc0, false);
        acc1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p1), __builtin_bit_cast(v2s, wpair[i][1]), acc1, false);
        acc2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p2), __builtin_bit_cast(v2s, wpair[i][2]), acc2, false);
        acc3 = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, p3), __builtin_bit_cast(v2s, wpair[i][3]), acc3, false);
      }
      const uint32_t packed = qnnp::q31_requantize_pack4<decltype(shift0)::value, decltype(full)::value>(
          acc0, acc1, acc2, acc3, p.rq);
      dw_store_any<uint32_t>(out_ptr, packed);
      out_ptr += p.out_stride;
      // slide the window by SW columns (register renaming once the loop is unrolled)
#pragma unroll
      for (int ky = 0; ky < 3; ky++) {
        if (SW == 1) { w0[ky] = w1[ky]; w1[ky] = w2[ky]; w2[ky] = n0[ky]; }
        else { w0[ky] = w2[ky]; w1[ky] = n0[ky]; w2[ky] = n1[ky]; }
      }
      ix += SW;
    }
    QNNP_DW_TRACE(p, 3);
    QNNP_DW_TRACE(p, 4);
    QNNP_DW_TRACE(p, 5);
  });
}

// geometry of kernel C: `slabs` = column segments per output row, `TOH` = outputs per segment
bool plan_row(DwParams& p, bool unaligned = false)
{
  if ((unaligned ? p.C < 4 : p.C % 4 != 0) || p.KH != 3 || p.KW != 3 || p.dh != 1 || p.dw != 1) return false;
  if (p.sh != p.sw || (p.sw != 1 && p.sw != 2)) return false;
  if (p.pad_left > 2 || p.pad_top > 2) return false;
  // 32-bit input offsets
  const uint64_t in_bytes = static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride;
  if (in_bytes >= (UINT64_C(1) << 32)) return false;
  // Column segments: enough threads for `waves` waves per SIMD over the whole chip (oversubscribing the 7
  // resident ones measured faster than exactly one resident round), but segments of at least 8 outputs
  // (each re-loads two halo columns and pays the start-up latency again).
  uint32_t waves = 8;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_DW_ROW_WAVES")) waves = static_cast<uint32_t>(atoi(env));
#endif
  const uint64_t base_threads = static_cast<uint64_t>(p.batch) * p.OH * ((p.C + 3u) / 4u);
  const uint64_t target = static_cast<uint64_t>(p.cu_count) * 4u * waves * 64u;
  uint32_t segs = static_cast<uint32_t>((target + base_threads - 1) / base_threads);
  const uint32_t max_segs = p.OW >= 8 ? p.OW / 8 : 1u;
  if (segs > max_segs) segs = max_segs;
  if (segs < 1) segs = 1;
  p.TOH = (p.OW + segs - 1) / segs;
  p.slabs = (p.OW + p.TOH - 1) / p.TOH;
  return true;
}

int launch_row(const DwParams& p, hipStream_t stream, bool unaligned = false)
{
  const uint64_t threads = static_cast<uint64_t>(p.batch) * p.slabs * p.OH * ((p.C + 3u) / 4u);
  const uint64_t blocks = (threads + kRowThreads - 1) / kRowThreads;
  if (blocks * kRowThreads > 0xFFFFFFFFull) return QNNP_HIP_EINVAL;
  if (unaligned) {
    if (p.sw == 1) hipLaunchKernelGGL((q8_dwconv_row3x3_kernel<1, true>), dim3(static_cast<uint32_t>(blocks)), dim3(kRowThreads), 0, stream, p);
    else hipLaunchKernelGGL((q8_dwconv_row3x3_kernel<2, true>), dim3(static_cast<uint32_t>(blocks)), dim3(kRowThreads), 0, stream, p);
    return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  }
  if (p.sw == 1) {
    hipLaunchKernelGGL(q8_dwconv_row3x3_kernel<1>, dim3(static_cast<uint32_t>(blocks)), dim3(kRowThreads), 0, stream, p);
  } else {
    hipLaunchKernelGGL(q8_dwconv_row3x3_kernel<2>, dim3(static_cast<uint32_t>(blocks)), dim3(kRowThreads), 0, stream, p);
  }
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

// --------------------------------------------------------------------------
// Kernel G: column-sliding register window, 3x3, dilation 1, stride 1 or 2 (both axes), C % 4 == 0
// --------------------------------------------------------------------------
/*
 * What bounds the depthwise kernels is VALU issue (PMC: 24.5 VALU instructions per output, 78 % VALU-active at a third
 * of the HBM rate), and ten of those instructions are the tap arithmetic: five v_perm_b32 that pair two taps' bytes as
 * int16 x 2 and five v_dot2_i32_i16. tools/ubench_valu.hip prices both at 3.2 cycles per wave-instruction (every
 * 8-byte-encoded VALU op; the 4-byte VOP2 ones take 2.0), so the lever is the COUNT. This kernel shares the pairing:
 *
 *   a thread owns one 4-channel dword column j = ox * (C/4) + c/4 of the flattened output row and walks DOWN the output
 *   rows of a segment. Per step it loads one new input row (stride 2: two) -- the three dwords at columns
 *   ix0..ix0+2 -- and pairs
 *       H[r] = (col0, col1) of input row r        (per channel: v_perm -> two int16)   used by the THREE output rows
 *       Q[r] = (col2 @ r, col2 @ r+1)                                                  whose windows contain row r
 *   so an output costs 2 x 4 new v_perm (stride 2: 3 x 4) instead of 5 x 4, and 5 x 4 v_dot2:
 *       out(t) = sum_r H[t+r] . (w_r0, w_r1)  +  Q[t-1] . (0, w_02)  +  Q[t+1] . (w_12, w_22)        (stride 1)
 *   Everything the window re-uses vertically stays in registers; the horizontal overlap (a dword is col0 / col1 / col2
 *   of three neighbouring threads) is served by L1, and because consecutive lanes are consecutive dwords of the NHWC
 *   row, every load and store instruction of a wave covers one contiguous run of memory (256 bytes for dense tensors).
 *   No LDS, no barrier; the rows two steps ahead are in flight while a step computes.
 * Padding: rows outside the image are wave-uniform (a wave = one image, one row segment) and replaced by the zero
 * point without touching memory; columns outside the image exist only in the waves at the ends of a row (FIX flavour).
 */
constexpr int kColThreads = 256;

/* DIL (the int8 walk at stride 1 only): dilated windows. Columns: the three taps sit p.dw pixels apart -- other lane
 * offsets, nothing else. Rows: output rows of one residue class `rho` mod p.dh form an UNDILATED walk over every
 * p.dh-th input row, so a wave takes (image, residue, segment of that residue's rows, chunk) and walks with a row
 * stride of p.dh rows: `oy0` / `oy1` then count the residue's rows (output row rho + p.dh * t), a row index `iy` of the
 * walk is input row rho - pad_top + p.dh * iy, valid for v_lo <= iy < v_hi. */
template <int S, bool FIX, int SEQ, bool FULL, bool DEEP, bool QUAD, bool DIL = false>
__device__ __forceinline__ void dwconv_col3x3_body(
    const DwParams& p, const uint32_t n, const uint32_t oy0, const uint32_t oy1, const uint32_t ox, const uint32_t cg,
    const bool ok0, const bool ok1, const bool ok2, const uint32_t rho = 0)
{
  static_assert(!DIL || (S == 1 && QUAD), "the dilated walk exists for the int8 flavour at stride 1");
  // kLate: the walks of the default path (int8 dot-product walk at stride 1, pair walk at stride 2) replace padding
  // where a row is CONSUMED, not where it is requested; the int16 pair walks at stride 1 (weights outside the int8
  // classes) keep the round-2 scheme
  constexpr bool kLate = QUAD || S == 2;
  // tap weights: W01[r] = (w_r0, w_r1) pairs, WQA = (0, w_02), WQB = (w_12, w_22); per channel of the group
  uint32_t w01[3][4], wqa[4], wqb[4];
  // QUAD flavour: W4[r] = (x_r0, x_r1, x_r2, 0) as int8, x = +-(w - kzp) (see the step below)
  uint32_t w4[3][4];
  const uint32_t kx = p.wrange == 2u ? 0x7f7f7f7fu : 0x80808080u;    // wave-uniform
  int32_t bias[4];
  // The tap weights and the bias are REQUESTED here; the int8 walk moves them to their registers only after the first
  // input rows have been requested too (unpack_weights()): traced on MobileNetV2 layer 8, a wave spent 5.3k cycles
  // until its weights were in registers and another 2.9k until its first rows had arrived -- one round trip behind
  // the other.
  uint2 tw[9];
  int4 bv;
  uint4 wv[3];
  if constexpr (QUAD) {
    // host-made register image: W4[r] for the thread's four channels, and the bias that goes with a ^ kx
#pragma unroll
    for (int r = 0; r < 3; r++) wv[r] = *reinterpret_cast<const uint4*>(p.dot4 + r * p.c_pad + cg);
    bv = *reinterpret_cast<const int4*>(p.dot4 + 3u * p.c_pad + cg);
  } else {
#pragma unroll
    for (int i = 0; i < 9; i++) tw[i] = *reinterpret_cast<const uint2*>(p.wadj + i * p.c_pad + cg);   // 4 x int16
    bv = *reinterpret_cast<const int4*>(p.bias1 + cg);
  }
  auto unpack_weights = [&]() __attribute__((always_inline)) {
    if constexpr (QUAD) __builtin_amdgcn_sched_barrier(0);             // behind the row loads issued so far
    auto lo16 = [](uint2 v, int c) -> uint32_t { return ((c < 2 ? v.x : v.y) >> ((c & 1) * 16)) & 0xFFFFu; };
    bias[0] = bv.x; bias[1] = bv.y; bias[2] = bv.z; bias[3] = bv.w;
    if constexpr (QUAD) {
#pragma unroll
      for (int r = 0; r < 3; r++) { w4[r][0] = wv[r].x; w4[r][1] = wv[r].y; w4[r][2] = wv[r].z; w4[r][3] = wv[r].w; }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      if constexpr (!QUAD) {
#pragma unroll
        for (int r = 0; r < 3; r++) w01[r][c] = lo16(tw[r * 3 + 0], c) | (lo16(tw[r * 3 + 1], c) << 16);
        wqa[c] = lo16(tw[2], c) << 16;
        wqb[c] = lo16(tw[5], c) | (lo16(tw[8], c) << 16);
      }
      // the offset rounding sequences (requant.hip.h) take accumulator + 2^31: folded into the bias, once per thread
      bias[c] = qnnp::with_rq_offset<SEQ>(bias[c]);
    }
    if constexpr (kLate && FIX) {
      // Padding COLUMNS of this lane are never loaded (out-of-range offsets: the buffer returns 0) and never selected:
      // reading 0 where the input zero point belongs is a per-lane constant -- izp * (the column's weights) -- that goes
      // into the bias here, once, instead of three v_cndmask per step on registers just requested (which made every step
      // of a border wave wait for its newest row: s_waitcnt vmcnt(3) / vmcnt(0) in the loop, ISA of the round-2 build).
      const bool okc[3] = {ok0, ok1, ok2};
#pragma unroll
      for (int c = 0; c < 4; c++) {
        int32_t sum = 0;
#pragma unroll
        for (int r = 0; r < 3; r++) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            int32_t x;
            if constexpr (QUAD) x = static_cast<int8_t>(w4[r][c] >> (8 * k));            // +-(w - kzp), as multiplied
            else x = static_cast<int16_t>(lo16(tw[r * 3 + k], c));                       // w - kzp
            sum += okc[k] ? 0 : x;
          }
        }
        // QUAD multiplies a' = a ^ kx: a' (izp) - a'(0) = izp (kx = 0x80) or -izp (0x7f); the pair walks multiply a itself
        const int32_t step = (QUAD && p.wrange == 2u) ? -static_cast<int32_t>(p.izp) : static_cast<int32_t>(p.izp);
        bias[c] = qnnp::add_wrap(bias[c], step * sum);
      }
    }
  };
  // (the pair walks unpack at once: with the raw taps live across the first row requests they need 93-98 VGPRs
  //  instead of 79-82, a wave per SIMD less, and the stride-2 layers measured no better for the overlap)
  if constexpr (!QUAD) unpack_weights();
  QNNP_DW_TRACE(p, 1);
  const uint32_t fill = p.izp * 0x01010101u;
  const uint32_t row_bytes = p.W * p.in_stride;
#ifdef QNNP_ENABLE_ABLATION
  // measurement knobs that leave the instruction stream alone: bit 2 = every step re-reads the segment's first row
  // (loads served by L1 / L2), bit 3 = every step stores to the segment's first output row (writes combine in L2)
  const uint32_t row_adv = (p.abl & 4u) ? 0u : row_bytes;
#else
  const uint32_t row_adv = row_bytes;
#endif
  // Buffer addressing: a descriptor over the whole tensor (built from kernel arguments only, so it stays in SGPRs),
  // per lane a constant 32-bit byte offset inside a row, per row a SCALAR offset -- no vector address arithmetic in
  // the loop. Invalid columns are clamped to column 0 (their value is replaced below).
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(p.batch * p.H * row_bytes), 0x00020000);
  const int32_t ix0 = static_cast<int32_t>(ox * S) - static_cast<int32_t>(p.pad_left);
  uint32_t coff[3];
  coff[0] = cg + (ok0 ? static_cast<uint32_t>(ix0) : 0u) * p.in_stride;
  const int32_t dcol = DIL ? static_cast<int32_t>(p.dw) : 1;
  coff[1] = cg + (ok1 ? static_cast<uint32_t>(ix0 + dcol) : 0u) * p.in_stride;
  coff[2] = cg + (ok2 ? static_cast<uint32_t>(ix0 + 2 * dcol) : 0u) * p.in_stride;
  // what a padding ROW reads at this lane's columns: the input zero point -- or 0 at a padding column of a kLate walk
  uint32_t fillk[3] = {fill, fill, fill};
  if constexpr (kLate && FIX) {
    // (beyond the descriptor's extent -- plan_col keeps the tensor below 2^31 bytes: the load returns 0)
    if (!ok0) { coff[0] = 0x80000000u; fillk[0] = 0u; }
    if (!ok1) { coff[1] = 0x80000000u; fillk[1] = 0u; }
    if (!ok2) { coff[2] = 0x80000000u; fillk[2] = 0u; }
  }
  const uint32_t img_off = n * p.H * row_bytes;                     // wave-uniform
  int32_t iy_first = static_cast<int32_t>(oy0 * S) - static_cast<int32_t>(p.pad_top);
  // DIL: the walk's row index -> input row base + dh * iy (scalars)
  int32_t v_lo = 0, v_hi = static_cast<int32_t>(p.H);
  uint32_t row_org = img_off, row_adv_w = row_adv, out_rows = 1u;
  if constexpr (DIL) {
    const int32_t d = static_cast<int32_t>(p.dh);
    const int32_t base = static_cast<int32_t>(rho) - static_cast<int32_t>(p.pad_top);
    v_lo = base >= 0 ? 0 : (-base + d - 1) / d;
    v_hi = static_cast<int32_t>(p.H) - 1 - base >= 0 ? (static_cast<int32_t>(p.H) - 1 - base) / d + 1 : 0;
    row_org = img_off + static_cast<uint32_t>(base) * row_bytes;    // (wraps for base < 0: rows iy >= v_lo bring it back)
    row_adv_w = row_adv * p.dh;
    out_rows = p.dh;
    iy_first = static_cast<int32_t>(oy0);
  }

  struct Row { uint32_t c[3]; bool ok; };
  // One input row: three dwords. The loads are ALWAYS issued (a row outside the image is clamped to a valid one and
  // its values replaced afterwards, wave-uniformly): a branch around the loads makes the number of outstanding
  // operations path-dependent, and hipcc then falls back to s_waitcnt vmcnt(<3) right behind the newest loads --
  // the rows "in flight" were waited for at once (measured: 17 of 45 us on MobileNetV2 layer 8).
  // CHECK = false: the caller knows the row is inside the image (the steady state of a segment).
  auto load_row = [&](auto check, int32_t iy) __attribute__((always_inline)) -> Row {
    constexpr bool CHECK = decltype(check)::value;
    Row r;
    bool row_ok = true;
    // (kLate walks: always -- the clamp is scalar work, and a steady-state step may request a row below the image for the
    //  checked steps behind it; what makes a step "steady" there is that the row it CONSUMES is inside)
    uint32_t ro;
    if constexpr (DIL) {
      row_ok = iy >= v_lo && iy < v_hi;
      iy = iy >= v_hi ? v_hi - 1 : iy;
      iy = iy < v_lo ? v_lo : iy;                 // (no valid row at all: a row below the image, or beyond the tensor -- the buffer answers 0)
      ro = row_org + static_cast<uint32_t>(iy) * row_adv_w;
    } else {
    if constexpr (CHECK || kLate) {
      row_ok = iy >= 0 && iy < static_cast<int32_t>(p.H);
      iy = iy < 0 ? 0 : (iy >= static_cast<int32_t>(p.H) ? static_cast<int32_t>(p.H) - 1 : iy);
    }
    ro = img_off + static_cast<uint32_t>(iy) * row_adv;        // scalar
    }
    r.ok = row_ok;
    r.c[0] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, coff[0], ro, 0);
    r.c[1] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, coff[1], ro, 0);
    r.c[2] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, coff[2], ro, 0);
    if constexpr (!kLate) {
      // (a select on a register just requested is a wait for it: the kLate walks do this in settle(), at the use)
      if constexpr (FIX) {
        r.c[0] = ok0 ? r.c[0] : fill;
        r.c[1] = ok1 ? r.c[1] : fill;
        r.c[2] = ok2 ? r.c[2] : fill;
      }
      if constexpr (CHECK) {
        r.c[0] = row_ok ? r.c[0] : fill;
        r.c[1] = row_ok ? r.c[1] : fill;
        r.c[2] = row_ok ? r.c[2] : fill;
      }
    }
    return r;
  };
  // kLate walks: a padding ROW becomes the zero point where it is consumed (CHECK = false: the caller knows the row is
  // inside the image -- the steady-state steps, whose rows lie between the first trip's and the ones they request)
  auto settle = [&](auto check, Row& r) __attribute__((always_inline)) {
    if constexpr (kLate && decltype(check)::value) {
      r.c[0] = r.ok ? r.c[0] : fillk[0];
      r.c[1] = r.ok ? r.c[1] : fillk[1];
      r.c[2] = r.ok ? r.c[2] : fillk[2];
    }
  };
  constexpr std::true_type kChecked{};
  constexpr std::false_type kInside{};
  struct Pair { uint32_t v[4]; };
  // per channel c: (lo.byte c, hi.byte c) as two zero-extended int16
  auto pair = [](uint32_t lo, uint32_t hi) __attribute__((always_inline)) -> Pair {
    Pair q;
    q.v[0] = __builtin_amdgcn_perm(hi, lo, 0x0c040c00u);
    q.v[1] = __builtin_amdgcn_perm(hi, lo, 0x0c050c01u);
    q.v[2] = __builtin_amdgcn_perm(hi, lo, 0x0c060c02u);
    q.v[3] = __builtin_amdgcn_perm(hi, lo, 0x0c070c03u);
    return q;
  };
  auto dot = [](const Pair& a, const uint32_t (&w)[4], int32_t (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      acc[c] = __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a.v[c]), __builtin_bit_cast(v2s, w[c]), acc[c], false);
    }
  };
  // QUAD flavour: per channel c the bytes (col0, col1, col2, 0) of one input row, re-centred to int8
  struct Quad { uint32_t v[4]; };
  auto quad = [kx](const Row& r) __attribute__((always_inline)) -> Quad {
    const uint32_t d0 = r.c[0] ^ kx, d1 = r.c[1] ^ kx, d2 = r.c[2] ^ kx;
    const uint32_t lo = __builtin_amdgcn_perm(d1, d0, 0x05010400u);      // (d0.b0, d1.b0, d0.b1, d1.b1)
    const uint32_t hi = __builtin_amdgcn_perm(d1, d0, 0x07030602u);      // (d0.b2, d1.b2, d0.b3, d1.b3)
    Quad q;
    q.v[0] = __builtin_amdgcn_perm(d2, lo, 0x0c040100u);
    q.v[1] = __builtin_amdgcn_perm(d2, lo, 0x0c050302u);
    q.v[2] = __builtin_amdgcn_perm(d2, hi, 0x0c060100u);
    q.v[3] = __builtin_amdgcn_perm(d2, hi, 0x0c070302u);
    return q;
  };
  auto dot4 = [](const Quad& a, const uint32_t (&w)[4], int32_t (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      acc[c] = __builtin_amdgcn_sdot4(static_cast<int32_t>(a.v[c]), static_cast<int32_t>(w[c]), acc[c], false);
    }
  };
  auto dot4_first = [](const Quad& a, const uint32_t (&w)[4], const int32_t (&b)[4], int32_t (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(acc[c]) : "v"(a.v[c]), "v"(w[c]), "v"(b[c]));
    }
  };
  // first product of an output: the three-operand form takes the bias as its addend (the two-operand accumulate
  // form the compiler prefers would need a copy of the bias per output first)
  auto dot_first = [](const Pair& a, const uint32_t (&w)[4], const int32_t (&b)[4], int32_t (&acc)[4]) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
      asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(acc[c]) : "v"(a.v[c]), "v"(w[c]), "v"(b[c]));
    }
  };

  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>(p.batch * p.OH * p.OW * p.out_stride), 0x00020000);
  const uint32_t out_voff = ox * p.out_stride + cg;
  uint32_t out_soff = (n * p.OH + (DIL ? rho + out_rows * oy0 : oy0)) * p.OW * p.out_stride;   // scalar, advances one output row (DIL: p.dh rows) per step
#ifdef QNNP_ENABLE_ABLATION
  const uint32_t out_step = (p.abl & 8u) ? 0u : p.OW * p.out_stride * out_rows;
#else
  const uint32_t out_step = p.OW * p.out_stride * out_rows;
#endif

  {
    auto finish = [&](int32_t (&acc)[4]) __attribute__((always_inline)) {
      const uint32_t packed = qnnp::q31_requantize_pack4<SEQ, FULL>(
          acc[0], acc[1], acc[2], acc[3], p.rq);
      // (a wave's 64 dwords of an output row are contiguous: whole lines, written once -- the streaming hint, under the
      //  "streaming_stores" option; the two builtins differ in an immediate, so hipcc cannot merge them)
      if (p.stream_out) __builtin_amdgcn_raw_buffer_store_b32(packed, out_rsrc, out_voff, out_soff, 2);
      else __builtin_amdgcn_raw_buffer_store_b32(packed, out_rsrc, out_voff, out_soff, 0);
      out_soff += out_step;
    };
    const uint32_t steps = oy1 - oy0;
    if constexpr (S == 1 && QUAD) {
      // The int8 dot-product flavour (weights whose x = w - kzp, or -x, fit int8: DwParams::wrange). An input row is
      // transposed ONCE into per-channel quads T[r] = (col0, col1, col2, 0) -- two v_perm over (col0, col1), four that
      // append col2 -- and serves the three output rows whose windows contain it:
      //     out(t) = T[t] . W4[0] + T[t+1] . W4[1] + T[t+2] . W4[2]               (v_dot4_i32_i8, 3 per channel)
      // i.e. 6 v_perm + 3 v_xor + 12 dot products per output dword against 8 v_perm + 20 v_dot2 of the pair walk.
      // a' = a ^ 0x80 = a - 128 pairs with x, a' = a ^ 0x7f = 127 - a with -x; the difference to sum a * x is a
      // constant per channel, folded into the bias above. A row buffer is dead as soon as its quad is built, so
      // NBUF buffers keep NBUF rows in flight: row r lives in buffer r % NBUF, T[.] rotate with period three.
      constexpr int NBUF = DEEP ? 6 : 3;
      Row r0 = load_row(kChecked, iy_first);
      Row r1 = load_row(kChecked, iy_first + 1);
      Row r2 = load_row(kChecked, iy_first + 2);
      Row r3, r4, r5;
      if constexpr (DEEP) {
        r3 = load_row(kChecked, iy_first + 3);
        r4 = load_row(kChecked, iy_first + 4);
        r5 = load_row(kChecked, iy_first + 5);
      }
      unpack_weights();
      settle(kChecked, r0);
      Quad q0 = quad(r0);
      r0 = load_row(kChecked, iy_first + NBUF);
      settle(kChecked, r1);
      Quad q1 = quad(r1);
      r1 = load_row(kChecked, iy_first + NBUF + 1);
      Quad q2;
      uint32_t t = 0;
      QNNP_DW_TRACE(p, 2);
      // steps whose newest window row (iy_first + t + 2) is inside the image: t < t_inside
      const int32_t inside = (DIL ? v_hi : static_cast<int32_t>(p.H)) - 2 - iy_first;
      const uint32_t t_inside = inside <= 0 ? 0u : (static_cast<uint32_t>(inside) < steps ? static_cast<uint32_t>(inside) : steps);
#define QNNP_DW_COL_STEPQ(CHECK, TA, TB, TC, RC)                                      \
      {                                                                             \
        settle(CHECK, RC);                                                          \
        TC = quad(RC);                                      /* T[t+2] */            \
        RC = load_row(CHECK, iy_first + static_cast<int32_t>(t) + 2 + NBUF);        \
        __builtin_amdgcn_sched_barrier(0);                                          \
        int32_t acc[4];                                                             \
        dot4_first(TA, w4[0], bias, acc);                                           \
        dot4(TB, w4[1], acc);                                                       \
        dot4(TC, w4[2], acc);                                                       \
        finish(acc);                                                                \
        t++;                                                                        \
      }
      // (a steady-state step consumes row iy_first + t + 2 >= 0 -- plan_col: pad_top <= 2, the rows above the image
      //  were settled above -- and requests a row inside the image: what it consumes lies between the two)
      if constexpr (DEEP) {
        while (t + 6 <= t_inside) {                // steady state: straight-line body, counted waits
          QNNP_DW_COL_STEPQ(kInside, q0, q1, q2, r2)
          QNNP_DW_COL_STEPQ(kInside, q1, q2, q0, r3)
          QNNP_DW_COL_STEPQ(kInside, q2, q0, q1, r4)
          QNNP_DW_COL_STEPQ(kInside, q0, q1, q2, r5)
          QNNP_DW_COL_STEPQ(kInside, q1, q2, q0, r0)
          QNNP_DW_COL_STEPQ(kInside, q2, q0, q1, r1)
        }
        QNNP_DW_TRACE(p, 3);
        while (t < steps) {                        // the last steps of a segment (and of the image: padding rows)
          QNNP_DW_COL_STEPQ(kChecked, q0, q1, q2, r2)
          if (t >= steps) break;
          QNNP_DW_COL_STEPQ(kChecked, q1, q2, q0, r3)
          if (t >= steps) break;
          QNNP_DW_COL_STEPQ(kChecked, q2, q0, q1, r4)
          if (t >= steps) break;
          QNNP_DW_COL_STEPQ(kChecked, q0, q1, q2, r5)
          if (t >= steps) break;
          QNNP_DW_COL_STEPQ(kChecked, q1, q2, q0, r0)
          if (t >= steps) break;
          QNNP_DW_COL_STEPQ(kChecked, q2, q0, q1, r1)
        }
      } else {
        while (t + 3 <= t_inside) {
          QNNP_DW_COL_STEPQ(kInside, q0, q1, q2, r2)
          QNNP_DW_COL_STEPQ(kInside, q1, q2, q0, r0)
          QNNP_DW_COL_STEPQ(kInside, q2, q0, q1, r1)
        }
        while (t < steps) {
          QNNP_DW_COL_STEPQ(kChecked, q0, q1, q2, r2)
          if (t >= steps) break;
          QNNP_DW_COL_STEPQ(kChecked, q1, q2, q0, r0)
          if (t >= steps) break;
          QNNP_DW_COL_STEPQ(kChecked, q2, q0, q1, r1)
        }
      }
      QNNP_DW_TRACE(p, 4);
#undef QNNP_DW_COL_STEPQ
    } else if constexpr (S == 1 && DEEP) {
      // The same walk with FOUR rows in flight instead of two: six row buffers (row r lives in buffer r % 6), the pair
      // registers keep their period of three, so six steps are written out per trip. PMC on the two-rows-ahead loop
      // (MobileNetV2 layers 8 / 13, batch 128): waves parked on s_waitcnt for 60 / 50 % of their lifetime, VALU busy
      // 59 / 50 % -- a step of 4-5 waves sharing a SIMD takes ~850 cycles, two of them are less than an HBM round trip.
      // State entering step t: HA = H[t], HB = H[t+1], QA = Q[t-1], QB = Q[t], RP = row t+1, RC = row t+2,
      // RL = the buffer of row t (dead: both its pairs are built), reloaded with row t+6.
      Row r0 = load_row(kChecked, iy_first);
      Row r1 = load_row(kChecked, iy_first + 1);
      Row r2 = load_row(kChecked, iy_first + 2);
      Row r3 = load_row(kChecked, iy_first + 3);
      Row r4 = load_row(kChecked, iy_first + 4);
      Row r5 = load_row(kChecked, iy_first + 5);
      Pair h0 = pair(r0.c[0], r0.c[1]);
      Pair h1 = pair(r1.c[0], r1.c[1]);
      Pair qa = pair(r0.c[2], r0.c[2]);            // Q[-1] = (don't care, col2 @ 0)
      Pair qb = pair(r0.c[2], r1.c[2]);            // Q[0]
      Pair h2, qc;
      uint32_t t = 0;
      // steps whose prefetched row (iy_first + t + 6) is inside the image: t < t_inside
      const int32_t inside = static_cast<int32_t>(p.H) - 6 - iy_first;
      const uint32_t t_inside = inside <= 0 ? 0u : (static_cast<uint32_t>(inside) < steps ? static_cast<uint32_t>(inside) : steps);
#define QNNP_DW_COL_STEP6(CHECK, HA, HB, HC, QA, QB, QC, RP, RC, RL)                 \
      {                                                                             \
        HC = pair(RC.c[0], RC.c[1]);                        /* H[t+2] */            \
        QC = pair(RP.c[2], RC.c[2]);                        /* Q[t+1] */            \
        RL = load_row(CHECK, iy_first + static_cast<int32_t>(t) + 6);               \
        __builtin_amdgcn_sched_barrier(0);   /* (left alone the scheduler sinks the loads towards their uses) */ \
        int32_t acc[4];                                                             \
        dot_first(HA, w01[0], bias, acc);                                           \
        dot(HB, w01[1], acc);                                                       \
        dot(HC, w01[2], acc);                                                       \
        dot(QA, wqa, acc);                                                          \
        dot(QC, wqb, acc);                                                          \
        finish(acc);                                                                \
        t++;                                                                        \
      }
      while (t + 6 <= t_inside) {                  // steady state: straight-line body, counted waits
        QNNP_DW_COL_STEP6(kInside, h0, h1, h2, qa, qb, qc, r1, r2, r0)
        QNNP_DW_COL_STEP6(kInside, h1, h2, h0, qb, qc, qa, r2, r3, r1)
        QNNP_DW_COL_STEP6(kInside, h2, h0, h1, qc, qa, qb, r3, r4, r2)
        QNNP_DW_COL_STEP6(kInside, h0, h1, h2, qa, qb, qc, r4, r5, r3)
        QNNP_DW_COL_STEP6(kInside, h1, h2, h0, qb, qc, qa, r5, r0, r4)
        QNNP_DW_COL_STEP6(kInside, h2, h0, h1, qc, qa, qb, r0, r1, r5)
      }
      while (t < steps) {                          // the last steps of a segment (and of the image: padding rows)
        QNNP_DW_COL_STEP6(kChecked, h0, h1, h2, qa, qb, qc, r1, r2, r0)
        if (t >= steps) break;
        QNNP_DW_COL_STEP6(kChecked, h1, h2, h0, qb, qc, qa, r2, r3, r1)
        if (t >= steps) break;
        QNNP_DW_COL_STEP6(kChecked, h2, h0, h1, qc, qa, qb, r3, r4, r2)
        if (t >= steps) break;
        QNNP_DW_COL_STEP6(kChecked, h0, h1, h2, qa, qb, qc, r4, r5, r3)
        if (t >= steps) break;
        QNNP_DW_COL_STEP6(kChecked, h1, h2, h0, qb, qc, qa, r5, r0, r4)
        if (t >= steps) break;
        QNNP_DW_COL_STEP6(kChecked, h2, h0, h1, qc, qa, qb, r0, r1, r5)
      }
#undef QNNP_DW_COL_STEP6
    } else if constexpr (S == 1) {
      // window rows of output t: t, t+1, t+2 (relative to iy_first). State entering step t:
      //   HA = H[t], HB = H[t+1], QA = Q[t-1] (only its high half matters), QB = Q[t],
      //   RP = row t+1 (its col2 is still needed), RC = row t+2, third row buffer = row t+3 (in flight).
      // The step pairs row t+2, then re-uses RP for row t+4: two rows are always in flight. Roles rotate through
      // the same registers with period three, so three steps are written out per trip and nothing is copied.
      const Row r0 = load_row(kChecked, iy_first);
      Row b1 = load_row(kChecked, iy_first + 1);
      Row b2 = load_row(kChecked, iy_first + 2);
      Row b3 = load_row(kChecked, iy_first + 3);
      Pair h0 = pair(r0.c[0], r0.c[1]);
      Pair h1 = pair(b1.c[0], b1.c[1]);
      Pair qa = pair(r0.c[2], r0.c[2]);            // Q[-1] = (don't care, col2 @ 0)
      Pair qb = pair(r0.c[2], b1.c[2]);            // Q[0]
      Pair h2, qc;
      uint32_t t = 0;
      // steps whose prefetched row (iy_first + t + 4) is inside the image: t < t_inside
      const int32_t inside = static_cast<int32_t>(p.H) - 4 - iy_first;
      const uint32_t t_inside = inside <= 0 ? 0u : (static_cast<uint32_t>(inside) < steps ? static_cast<uint32_t>(inside) : steps);
#define QNNP_DW_COL_STEP(CHECK, HA, HB, HC, QA, QB, QC, RP, RC)                      \
      {                                                                             \
        HC = pair(RC.c[0], RC.c[1]);                        /* H[t+2] */            \
        QC = pair(RP.c[2], RC.c[2]);                        /* Q[t+1] */            \
        RP = load_row(CHECK, iy_first + static_cast<int32_t>(t) + 4);               \
        int32_t acc[4];                                                             \
        dot_first(HA, w01[0], bias, acc);                                           \
        dot(HB, w01[1], acc);                                                       \
        dot(HC, w01[2], acc);                                                       \
        dot(QA, wqa, acc);                                                          \
        dot(QC, wqb, acc);                                                          \
        finish(acc);                                                                \
        t++;                                                                        \
      }
      while (t + 3 <= t_inside) {                  // steady state: straight-line body, counted waits
        QNNP_DW_COL_STEP(kInside, h0, h1, h2, qa, qb, qc, b1, b2)
        QNNP_DW_COL_STEP(kInside, h1, h2, h0, qb, qc, qa, b2, b3)
        QNNP_DW_COL_STEP(kInside, h2, h0, h1, qc, qa, qb, b3, b1)
      }
      while (t < steps) {                          // the last steps of a segment (and of the image: padding rows)
        QNNP_DW_COL_STEP(kChecked, h0, h1, h2, qa, qb, qc, b1, b2)
        if (t < steps) {
          QNNP_DW_COL_STEP(kChecked, h1, h2, h0, qb, qc, qa, b2, b3)
          if (t < steps) {
            QNNP_DW_COL_STEP(kChecked, h2, h0, h1, qc, qa, qb, b3, b1)
          }
        }
      }
#undef QNNP_DW_COL_STEP
    } else {
      // stride 2: window rows of output t: 2t, 2t+1, 2t+2. State entering step t:
      //   HA = H[2t], QA = (don't care, col2 @ 2t), (RA, RB) = rows 2t+1 / 2t+2, the other row pair = rows 2t+3 / 2t+4
      //   (in flight). Period two: two steps per trip.
      const Row r0 = load_row(kChecked, iy_first);
      Row a1 = load_row(kChecked, iy_first + 1);
      Row a2 = load_row(kChecked, iy_first + 2);
      Row c1 = load_row(kChecked, iy_first + 3);
      Row c2 = load_row(kChecked, iy_first + 4);
      // steps whose window rows (iy_first + 2t + 1, + 2) are inside the image: t < t_inside
      const int32_t inside = (static_cast<int32_t>(p.H) - 1 - iy_first) / 2;
      const uint32_t t_inside = (static_cast<int32_t>(p.H) - 1 - iy_first) <= 0 ? 0u :
          (static_cast<uint32_t>(inside) < steps ? static_cast<uint32_t>(inside) : steps);
      Row r0s = r0;
      settle(kChecked, r0s);
      Pair ha = pair(r0s.c[0], r0s.c[1]);
      Pair qa = pair(r0s.c[2], r0s.c[2]);
      Pair hb, qb;
      uint32_t t = 0;
#define QNNP_DW_COL_STEP2(CHECK, HA, HC, QA, QC, RA, RB)                            \
      {                                                                             \
        settle(CHECK, RA);                                                          \
        settle(CHECK, RB);                                                          \
        const Pair hmid = pair(RA.c[0], RA.c[1]);           /* H[2t+1] */           \
        HC = pair(RB.c[0], RB.c[1]);                        /* H[2t+2] = H[2(t+1)] */ \
        QC = pair(RA.c[2], RB.c[2]);                        /* (col2 @ 2t+1, col2 @ 2t+2) */ \
        RA = load_row(CHECK, iy_first + 2 * static_cast<int32_t>(t) + 5);           \
        RB = load_row(CHECK, iy_first + 2 * static_cast<int32_t>(t) + 6);           \
        __builtin_amdgcn_sched_barrier(0);   /* (left alone the scheduler sinks the loads to the end of the trip) */ \
        int32_t acc[4];                                                             \
        dot_first(HA, w01[0], bias, acc);                                           \
        dot(hmid, w01[1], acc);                                                     \
        dot(HC, w01[2], acc);                                                       \
        dot(QA, wqa, acc);                                                          \
        dot(QC, wqb, acc);                                                          \
        finish(acc);                                                                \
        t++;                                                                        \
      }
      if (iy_first + 1 < 0) {                      // padding 2: row 1 of the walk is above the image -- one checked trip
        QNNP_DW_COL_STEP2(kChecked, ha, hb, qa, qb, a1, a2)
        if (t >= steps) return;
        QNNP_DW_COL_STEP2(kChecked, hb, ha, qb, qa, c1, c2)
      }
      if (t + 2 <= t_inside) {
        // (hipcc sizes the waits at a loop header for the fewest operations in flight over all ways in; entered from the
        //  checked first trip that came out as vmcnt(1) on every trip. With the queue drained once on the way in, the back
        //  edge alone decides)
        __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
        do {
          QNNP_DW_COL_STEP2(kInside, ha, hb, qa, qb, a1, a2)
          QNNP_DW_COL_STEP2(kInside, hb, ha, qb, qa, c1, c2)
        } while (t + 2 <= t_inside);
      }
      while (t < steps) {
        QNNP_DW_COL_STEP2(kChecked, ha, hb, qa, qb, a1, a2)
        if (t < steps) {
          QNNP_DW_COL_STEP2(kChecked, hb, ha, qb, qa, c1, c2)
        }
      }
#undef QNNP_DW_COL_STEP2
    }
  }
}

/* SEQ / FULL: the requantization flavour (requant.hip.h), chosen on the host -- one kernel per flavour, so that the
 * common one is not charged the registers of the rare ones (83 against 77 VGPRs: 5 instead of 6 waves per SIMD) */
template <int S, int SEQ, bool FULL, bool DEEP, bool QUAD, bool DIL = false>
__global__ __launch_bounds__(kColThreads)
void q8_dwconv_col3x3_kernel(const DwParams p)
{
  // wave -> (image, row segment, 64-dword chunk of the flattened output row); `slabs` = row segments, `TOH` = rows
  // per segment, `bands` = chunks per row here
  const uint32_t lane = threadIdx.x & 63u;
  QNNP_DW_TRACE(p, 0);
#ifdef QNNP_ENABLE_ABLATION
  if (p.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 4096) {
    p.trace[(blockIdx.x * 4 + 3) * 8 + 0] = wall_clock64();
    // where the workgroup's wave 0 runs: HW_REG_HW_ID (wave 3:0, SIMD 5:4, CU 11:8, SH 12, SE 15:13) and HW_REG_XCC_ID
    p.trace[(blockIdx.x * 4 + 2) * 8 + 0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
    p.trace[(blockIdx.x * 4 + 2) * 8 + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);
  }
#endif
  // Workgroup b runs on XCD b % 8 (observed, MI355X_MICROARCH.md "Workgroup dispatch"; a speed assumption only). For
  // tensors that more or less live in the caches (launch_col: <= 96 MiB in + out) every XCD takes a CONTIGUOUS range of
  // the flattened (image, segment, chunk) space, so that the halo pixels either side of a chunk are found in that XCD's
  // L2 by the neighbouring workgroups: same box, batch 128, 28x28x192 s2 8.8 -> 8.2 us, 14x14x576 s2 7.1 -> 6.9,
  // 56x56x144 s2 19.3 -> 18.9. The big HBM-streaming layers keep the round-robin order, where the eight XCDs sweep
  // neighbouring addresses at the same time: 112x112x96 s2 41.9 -> 43.8 us with the ranges, 112x112x32 +0.3.
  uint32_t b = blockIdx.x;
  if (p.xcd_ranges != 0u) {
    const uint32_t q = gridDim.x >> 3, r = gridDim.x & 7u, xcd = b & 7u;
    b = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (b >> 3);
  }
  uint32_t w = __builtin_amdgcn_readfirstlane(b * (kColThreads / 64) + (threadIdx.x >> 6));
  // (divisions by run-time values through host-made reciprocals: ~40 instructions each otherwise, on every wave's way
  //  to its first load; exact while dividend * divisor < 2^32, which plan_col checks)
  auto div_by = [](uint32_t x, uint32_t inv) __attribute__((always_inline)) { return inv != 0u ? __umulhi(x, inv) : x; };
  const uint32_t wb = div_by(w, p.inv_bands);
  const uint32_t chunk = w - wb * p.bands;
  const uint32_t n = div_by(wb, p.inv_slabs);
  const uint32_t seg = wb - n * p.slabs;
  if (n >= p.batch) return;
  const uint32_t q4 = p.C >> 2;
  const uint32_t cols = p.OW * q4;
  // lanes past the end of the row repeat its last dword column: same loads, same results, same stores -- harmless
  uint32_t j = chunk * 64u + lane;
  if (j >= cols) j = cols - 1u;
  const uint32_t ox = div_by(j, p.inv_q4);
  const uint32_t cg = (j - ox * q4) * 4u;
  // DIL: segment index -> (segment of a residue class of the output rows, residue); `slabs` counts both
  uint32_t rho = 0, rows = p.OH, sidx = seg;
  if constexpr (DIL) {
    sidx = div_by(seg, p.inv_dh);
    rho = seg - sidx * p.dh;
    rows = rho < p.OH ? div_by(p.OH - rho + p.dh - 1u, p.inv_dh) : 0u;       // output rows rho, rho + dh, ...
  }
  const uint32_t oy0 = sidx * p.TOH;
  if (DIL && oy0 >= rows) return;
  const uint32_t oy1 = min(rows, oy0 + p.TOH);
  const int32_t dcol = DIL ? static_cast<int32_t>(p.dw) : 1;
  const int32_t ix0 = static_cast<int32_t>(ox * S) - static_cast<int32_t>(p.pad_left);
  const bool ok0 = ix0 >= 0 && ix0 < static_cast<int32_t>(p.W);
  const bool ok1 = ix0 + dcol >= 0 && ix0 + dcol < static_cast<int32_t>(p.W);
  const bool ok2 = ix0 + 2 * dcol >= 0 && ix0 + 2 * dcol < static_cast<int32_t>(p.W);
  if (__builtin_amdgcn_ballot_w64(!(ok0 && ok1 && ok2)) != 0) {
    dwconv_col3x3_body<S, true, SEQ, FULL, DEEP, QUAD, DIL>(p, n, oy0, oy1, ox, cg, ok0, ok1, ok2, rho);
  } else {
    dwconv_col3x3_body<S, false, SEQ, FULL, DEEP, QUAD, DIL>(p, n, oy0, oy1, ox, cg, true, true, true, rho);
  }
  QNNP_DW_TRACE(p, 5);
#ifdef QNNP_ENABLE_ABLATION
  if (p.trace != nullptr && threadIdx.x == 0 && blockIdx.x < 4096) p.trace[(blockIdx.x * 4 + 3) * 8 + 1] = wall_clock64();
#endif
}

// measurement knob, read once: output rows per segment of kernel G (0 = automatic)
uint32_t col_rows_override()
{
  static const uint32_t rows = [] {
    if (const char* env = getenv("QNNP_GFX950_DW_COL_ROWS")) {
      const int v = atoi(env);
      if (v >= 1 && v <= 4096) return static_cast<uint32_t>(v);
    }
    return 0u;
  }();
  return rows;
}

bool col_uses_dot4(const DwParams& p);

// geometry of kernel G: `bands` = 64-dword chunks per flattened output row, `slabs` = row segments, `TOH` = rows each
bool plan_col(DwParams& p)
{
  if (p.C % 4 != 0 || p.KH != 3 || p.KW != 3) return false;
  if (p.sh != p.sw || (p.sw != 1 && p.sw != 2)) return false;
  // dilated windows: the int8 walk at stride 1 only (its residue-class form, see the body); up to 64 rows apart
  const bool dilated = p.dh != 1 || p.dw != 1;
  if (dilated && !(p.sw == 1 && col_uses_dot4(p) && p.dh <= 64 && p.dw <= 64)) return false;
  // The walk's steady-state steps fetch row iy_first + t + 2 + NBUF with no lower bound (only the start-up steps
  // check it): any API-legal padding beyond the window's own reach (top / left > 2 [x the dilation], which no 3x3 layer
  // of a real network has) would index rows before the image. Those shapes take the generic kernels.
  if (p.pad_top > 2 * p.dh || p.pad_left > 2 * p.dw) return false;
  // 32-bit byte offsets into both tensors
  const uint64_t in_bytes = static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride;
  const uint64_t out_bytes = static_cast<uint64_t>(p.batch) * p.OH * p.OW * p.out_stride;
  if (in_bytes >= (UINT64_C(1) << 31) || out_bytes >= (UINT64_C(1) << 32)) return false;
  const uint32_t cols = p.OW * (p.C / 4);
  const uint32_t chunks = (cols + 63u) / 64u;
  // Row segments: each re-loads its halo (two rows at stride 1) and builds the first pairs again, so as few as
  // still give every SIMD several waves' worth of work (the tail of the last round is what it buys back).
  // (dilated: a "segment" here is one segment of EVERY residue class of the output rows, rows_v rows each at most)
  const uint32_t rows_v = dilated ? (p.OH + p.dh - 1) / p.dh : p.OH;
  const uint32_t classes = dilated ? p.dh : 1u;
  const uint64_t waves_per_seg = static_cast<uint64_t>(p.batch) * chunks * classes;
  // (measured on the MobileNetV2 layers, batch 128: 1.3-2 rounds of waves beat 3-4 -- 35.6 against 39.6 us on
  //  layer 8 -- now that the rows in flight are really in flight; shorter segments only add start-ups and halo rows)
  // wave slots the segment count is sized for: 6 per SIMD at stride 2, 5 at stride 1 (what the 79- and 82-VGPR kernels
  // of the time allowed; today's 70-72 VGPRs would admit 7, but a rows-per-segment sweep of the present kernels --
  // QNNP_GFX950_DW_COL_ROWS, same box -- found this choice at or within 3 % of the best on all ten MobileNetV2 layers:
  // more, shorter segments pay the ~8k-cycle start-up of a wave again)
  const uint64_t slots = static_cast<uint64_t>(p.cu_count) * 4u * (p.sw == 1 ? 5u : 6u);
  const uint64_t target = slots * 3u / 2u;                                         // ~1.5 rounds
  uint32_t segs = static_cast<uint32_t>((target + waves_per_seg - 1) / waves_per_seg);
  // Stride 1 with enough columns to give every SIMD a few waves: at most ONE round (28x28x192, batch 128: two
  // segments 15.2 us, four 16.6; 14x14x576: one 12.5, two 14.3 -- a second, partly filled round costs more than it
  // hides; the stride-2 layers are HBM-bound and keep the deeper queue)
  if (p.sw == 1 && waves_per_seg * 5u >= slots * 2u) {
    const uint32_t one_round = static_cast<uint32_t>(slots / waves_per_seg);
    segs = one_round < 1u ? 1u : one_round;
  }
  const uint32_t min_rows = 7;
  uint32_t max_segs = rows_v / min_rows;
  if (max_segs < 1) max_segs = 1;
  if (segs > max_segs) segs = max_segs;
  if (segs < 1) segs = 1;
  uint32_t toh = (rows_v + segs - 1) / segs;
  if (const uint32_t forced = col_rows_override()) toh = forced < rows_v ? forced : rows_v;
  p.TOH = toh;
  p.slabs = ((rows_v + toh - 1) / toh) * classes;
  p.bands = chunks;
  const uint64_t waves = static_cast<uint64_t>(p.batch) * p.slabs * chunks;
  // (the kernel divides wave and column indices through 32-bit reciprocals: exact while dividend * divisor < 2^32)
  const uint64_t dmax = chunks > p.slabs ? chunks : p.slabs;
  if ((waves + 8u * (kColThreads / 64)) * dmax >= (UINT64_C(1) << 32)) return false;
  if (static_cast<uint64_t>(cols) * (p.C / 4) >= (UINT64_C(1) << 32)) return false;
  return waves < (UINT64_C(1) << 31);
}

// stride 1 with weights in int8 range (DwParams::wrange): the v_dot4_i32_i8 flavour of kernel G
bool col_uses_dot4(const DwParams& p)
{
  bool quad = p.sw == 1u && (p.wrange == 1u || p.wrange == 2u) && p.dot4 != nullptr;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_GFX950_DW_COL_QUAD")) quad = quad && atoi(env) != 0;
#endif
  return quad;
}

int launch_col(const DwParams& geometry, hipStream_t stream)
{
  DwParams p = geometry;
  auto reciprocal = [](uint32_t d) { return d > 1u ? static_cast<uint32_t>(((UINT64_C(1) << 32) + d - 1u) / d) : 0u; };
  p.inv_dh = reciprocal(p.dh);
  p.inv_bands = reciprocal(p.bands);
  p.inv_slabs = reciprocal(p.slabs);
  p.inv_q4 = reciprocal(p.C / 4u);
  p.xcd_ranges = (static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride +
                  static_cast<uint64_t>(p.batch) * p.OH * p.OW * p.out_stride) <= (UINT64_C(96) << 20) ? 1u : 0u;
  const uint64_t waves = static_cast<uint64_t>(p.batch) * p.slabs * p.bands;
  const uint32_t blocks = static_cast<uint32_t>((waves + (kColThreads / 64) - 1) / (kColThreads / 64));
  qnnp::requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    constexpr bool kFull = decltype(full)::value;
    if (p.dh != 1 || p.dw != 1) {                  // (plan_col: stride 1, the int8 flavour)
      if (p.TOH >= 12u) {
        hipLaunchKernelGGL((q8_dwconv_col3x3_kernel<1, kSeq, kFull, true, true, true>), dim3(blocks), dim3(kColThreads), 0, stream, p);
      } else {
        hipLaunchKernelGGL((q8_dwconv_col3x3_kernel<1, kSeq, kFull, false, true, true>), dim3(blocks), dim3(kColThreads), 0, stream, p);
      }
    } else if (p.sw == 1) {
      // four rows in flight (see the kernel) when a segment is long enough to use them: 112x112x32 30.7 -> 29.1 us,
      // 56x56x144 33.4 -> 32.3, 14-row segments level, 7x7x960 8.25 -> 8.55 with its six-row start-up
      bool deep = p.TOH >= 12u;
#ifdef QNNP_ENABLE_ABLATION
      if (const char* env = getenv("QNNP_GFX950_DW_COL_DEEP")) deep = atoi(env) != 0;
#endif
      const bool quad = col_uses_dot4(p);
      if (quad && deep) {
        hipLaunchKernelGGL((q8_dwconv_col3x3_kernel<1, kSeq, kFull, true, true>), dim3(blocks), dim3(kColThreads), 0, stream, p);
      } else if (quad) {
        hipLaunchKernelGGL((q8_dwconv_col3x3_kernel<1, kSeq, kFull, false, true>), dim3(blocks), dim3(kColThreads), 0, stream, p);
      } else if (deep) {
        hipLaunchKernelGGL((q8_dwconv_col3x3_kernel<1, kSeq, kFull, true, false>), dim3(blocks), dim3(kColThreads), 0, stream, p);
      } else {
        hipLaunchKernelGGL((q8_dwconv_col3x3_kernel<1, kSeq, kFull, false, false>), dim3(blocks), dim3(kColThreads), 0, stream, p);
      }
    } else {
      hipLaunchKernelGGL((q8_dwconv_col3x3_kernel<2, kSeq, kFull, false, false>), dim3(blocks), dim3(kColThreads), 0, stream, p);
    }
  });
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

// --------------------------------------------------------------------------
// Kernel H: column-sliding register window, 5x5, stride 1 | 2, weights in int8 range (v_dot4_i32_i8)
// --------------------------------------------------------------------------
/*
 * Kernel G's int8 dot-product walk for 5x5 windows (replaces q8dwconv_ukernel_mp8x25__sse2, reference
 * src/q8dwconv/mp8x25-sse2.c:14-742, for these shapes). A lane owns a 4-channel group of one output column and walks
 * down the output rows of its segment; an input row is five dwords (window columns 0..4, four channels each):
 *   - columns 0..3 are transposed ONCE into per-channel quads T[r][c] = (col0, col1, col2, col3) -- eight v_perm -- and
 *     serve the five output rows whose windows contain the row: five v_dot4_i32_i8 per channel against
 *     WQ[ky][c] = (x_ky0, x_ky1, x_ky2, x_ky3);
 *   - column 4 needs no transpose of its own: V[c] holds channel c's column-4 bytes of the FOUR rows before the newest
 *     one, slid along by one v_perm per channel and row, and meets WV[c] = (x_04, x_14, x_24, x_34) in one dot product;
 *     the newest row's column-4 dword meets the one-hot W5[c] = x_44 at byte c.
 * Per output dword at stride 1: 5 loads, 5 v_xor, 12 v_perm, 28 dot products, the requantization, one store -- ~65
 * instructions against the ~150 of the LDS-tiled kernel's int16 pair walk (25 LDS reads, 52 v_perm, 52 v_dot2).
 * x = w - kzp with a' = a ^ 0x80, or x = kzp - w with a' = a ^ 0x7f (DwParams::wrange, as kernel G); the host packs
 * the register image (pack.h qnnp_pack_dwconv_dot4_5x5: rows 0..4 WQ, 5 WV, 6 W5, 7 the bias that goes with a').
 * Rows: row r of the walk lives in raw buffer r % 5 until it is transposed into T[r % 5]; the buffer is re-loaded at
 * once with row r + 5, so five rows are always in flight and no register is ever copied (five steps per trip).
 */
template <int S, bool FIX, int SEQ, bool FULL>
__device__ __forceinline__ void dwconv_col5x5_body(
    const DwParams& p, const uint32_t n, const uint32_t oy0, const uint32_t oy1, const uint32_t ox, const uint32_t cg,
    const bool (&okc)[5])
{
  const uint32_t kx = p.wrange == 2u ? 0x7f7f7f7fu : 0x80808080u;    // wave-uniform
  uint4 wqv[5];
#pragma unroll
  for (int ky = 0; ky < 5; ky++) wqv[ky] = *reinterpret_cast<const uint4*>(p.dot4 + ky * p.c_pad + cg);
  const uint4 wvv = *reinterpret_cast<const uint4*>(p.dot4 + 5u * p.c_pad + cg);
  const uint4 w5v = *reinterpret_cast<const uint4*>(p.dot4 + 6u * p.c_pad + cg);
  const int4 bv = *reinterpret_cast<const int4*>(p.dot4 + 7u * p.c_pad + cg);

  const uint32_t fill = p.izp * 0x01010101u;
  const uint32_t row_bytes = p.W * p.in_stride;
  // buffer addressing as kernel G: per lane constant byte offsets inside a row, per row a SCALAR offset
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(p.batch * p.H * row_bytes), 0x00020000);
  const int32_t ix0 = static_cast<int32_t>(ox * S) - static_cast<int32_t>(p.pad_left);
  uint32_t coff[5];
#pragma unroll
  for (int k = 0; k < 5; k++) coff[k] = cg + (okc[k] ? static_cast<uint32_t>(ix0 + k) : 0u) * p.in_stride;
  const uint32_t img_off = n * p.H * row_bytes;                     // wave-uniform
  const int32_t iy_first = static_cast<int32_t>(oy0 * S) - static_cast<int32_t>(p.pad_top);

  struct Row { uint32_t c[5]; bool ok; };
  // (the loads are ALWAYS issued: a row outside the image is clamped to a valid one and replaced afterwards, see kernel
  //  G. "Afterwards" is where the row is CONSUMED -- settle() -- not here: a select on a register just requested is a
  //  wait for it, and the rows are requested five steps ahead.)
  auto load_row = [&](auto, int32_t iy) __attribute__((always_inline)) -> Row {
    Row r;
    // (always clamped -- scalar work -- whatever the step knows about the rows it CONSUMES: a steady-state step may
    //  well request a row below the image, for the checked steps behind it)
    r.ok = iy >= 0 && iy < static_cast<int32_t>(p.H);
    iy = iy < 0 ? 0 : (iy >= static_cast<int32_t>(p.H) ? static_cast<int32_t>(p.H) - 1 : iy);
    const uint32_t ro = img_off + static_cast<uint32_t>(iy) * row_bytes;      // scalar
#pragma unroll
    for (int k = 0; k < 5; k++) r.c[k] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, coff[k], ro, 0);
    return r;
  };
  constexpr std::true_type kChecked{};
  constexpr std::false_type kInside{};
  // padding columns (per lane) and padding rows (per wave; CHECK = false: the caller knows the rows it consumes are
  // inside the image -- the steady-state steps)
  auto settle = [&](auto check, Row& r) __attribute__((always_inline)) {
    constexpr bool CHECK = decltype(check)::value;
    if constexpr (FIX) {
#pragma unroll
      for (int k = 0; k < 5; k++) r.c[k] = okc[k] ? r.c[k] : fill;
    }
    if constexpr (CHECK) {
#pragma unroll
      for (int k = 0; k < 5; k++) r.c[k] = r.ok ? r.c[k] : fill;
    }
  };

  struct Quads { uint32_t q[4]; };
  // columns 0..3 of a row -> per channel c the bytes (col0, col1, col2, col3), re-centred; e = column 4, re-centred
  auto transpose = [kx](const Row& r, Quads& t, uint32_t& e) __attribute__((always_inline)) {
    const uint32_t d0 = r.c[0] ^ kx, d1 = r.c[1] ^ kx, d2 = r.c[2] ^ kx, d3 = r.c[3] ^ kx;
    e = r.c[4] ^ kx;
    // (pinned here: left to itself hipcc keeps the RAW dword alive past the re-load of its buffer -- a register copy per
    //  step, and the copies of freshly loaded registers at the top of the trip are s_waitcnt vmcnt(6): every row of
    //  the previous trip had to land before the next began, 49 % of the wave cycles waiting)
    asm volatile("" : "+v"(e));
    const uint32_t lo01 = __builtin_amdgcn_perm(d1, d0, 0x05010400u);    // (d0.b0, d1.b0, d0.b1, d1.b1)
    const uint32_t hi01 = __builtin_amdgcn_perm(d1, d0, 0x07030602u);    // (d0.b2, d1.b2, d0.b3, d1.b3)
    const uint32_t lo23 = __builtin_amdgcn_perm(d3, d2, 0x05010400u);
    const uint32_t hi23 = __builtin_amdgcn_perm(d3, d2, 0x07030602u);
    t.q[0] = __builtin_amdgcn_perm(lo23, lo01, 0x05040100u);
    t.q[1] = __builtin_amdgcn_perm(lo23, lo01, 0x07060302u);
    t.q[2] = __builtin_amdgcn_perm(hi23, hi01, 0x05040100u);
    t.q[3] = __builtin_amdgcn_perm(hi23, hi01, 0x07060302u);
  };
  // V[c] = (V[c].b1, V[c].b2, V[c].b3, e.bc): one row further down column 4
  auto slide = [](uint32_t (&v)[4], uint32_t e) __attribute__((always_inline)) {
    v[0] = __builtin_amdgcn_perm(e, v[0], 0x04030201u);
    v[1] = __builtin_amdgcn_perm(e, v[1], 0x05030201u);
    v[2] = __builtin_amdgcn_perm(e, v[2], 0x06030201u);
    v[3] = __builtin_amdgcn_perm(e, v[3], 0x07030201u);
  };

  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>(p.batch * p.OH * p.OW * p.out_stride), 0x00020000);
  const uint32_t out_voff = ox * p.out_stride + cg;
  uint32_t out_soff = (n * p.OH + oy0) * p.OW * p.out_stride;       // scalar, advances one output row per step
  const uint32_t out_step = p.OW * p.out_stride;

  // The first ten rows less S are requested at once, before the weights are moved to their registers: rows 0..4 into
  // their buffers, rows 5..9-S -- whose buffers still hold rows 0..4-S -- into the start-up buffers X, which the first
  // trip consumes in their place. Without X a wave's second batch of rows was requested only after the first had
  // arrived and been transposed: two memory round trips before step 1, 43 % of a 14-step wave's lifetime in s_waitcnt
  // (PMC, 28x28x240 batch 128: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES).
  Row rb[5], X[5 - S];
#pragma unroll
  for (int r = 0; r < 5; r++) rb[r] = load_row(kChecked, iy_first + r);
#pragma unroll
  for (int r = 0; r < 5 - S; r++) X[r] = load_row(kChecked, iy_first + 5 + r);
  __builtin_amdgcn_sched_barrier(0);
  uint32_t wq[5][4], wv[4], w5[4];
  int32_t bias[4];
#pragma unroll
  for (int ky = 0; ky < 5; ky++) { wq[ky][0] = wqv[ky].x; wq[ky][1] = wqv[ky].y; wq[ky][2] = wqv[ky].z; wq[ky][3] = wqv[ky].w; }
  wv[0] = wvv.x; wv[1] = wvv.y; wv[2] = wvv.z; wv[3] = wvv.w;
  w5[0] = w5v.x; w5[1] = w5v.y; w5[2] = w5v.z; w5[3] = w5v.w;
  // the offset rounding sequences (requant.hip.h) take accumulator + 2^31: folded into the bias, once per thread
  bias[0] = qnnp::with_rq_offset<SEQ>(bv.x); bias[1] = qnnp::with_rq_offset<SEQ>(bv.y);
  bias[2] = qnnp::with_rq_offset<SEQ>(bv.z); bias[3] = qnnp::with_rq_offset<SEQ>(bv.w);

  Quads tq[5];
  uint32_t V[4] = {0u, 0u, 0u, 0u};
  // rows 0 .. 4 - S of the walk are part of the first window only: transposed here, their buffers re-loaded
#pragma unroll
  for (int r = 0; r < 5 - S; r++) {
    uint32_t e;
    settle(kChecked, rb[r]);
    transpose(rb[r], tq[r], e);
    slide(V, e);
    rb[r] = load_row(kChecked, iy_first + r + 10);                  // (row r + 5 is on its way to X[r])
  }

  const uint32_t steps = oy1 - oy0;
  uint32_t t = 0;
  // steps whose window (rows iy_first + S*t .. + 4) is inside the image: t < t_inside (from the second trip on the
  // rows are below the top padding: pad_top <= 4)
  const int32_t inside = static_cast<int32_t>(p.H) - 5 - iy_first;
  const uint32_t t_inside = inside < 0 ? 0u : min(steps, static_cast<uint32_t>(inside) / S + 1u);

  // FIRST: the first trip (t = PH): rows 5..9-S come from X and their buffers are not re-loaded (they already hold
  // the row after next)
  auto step = [&](auto phase, auto check, auto first) __attribute__((always_inline)) {
    constexpr int PH = decltype(phase)::value;
    constexpr bool FIRST = decltype(first)::value;
    uint32_t e_new = 0;
#pragma unroll
    for (int k = 0; k < S; k++) {
      const int rel = S * PH + 5 - S + k;                                // row of the walk, first trip (a constant after unrolling)
      const int slot = rel % 5;
      uint32_t e;
      // (a steady-state step consumes rows that are inside the image: its own re-loads lie further down)
      if (FIRST && rel >= 5 && rel < 10 - S) {
        settle(check, X[rel - 5]);
        transpose(X[rel - 5], tq[slot], e);
      } else {
        settle(check, rb[slot]);
        transpose(rb[slot], tq[slot], e);
        rb[slot] = load_row(check, iy_first + static_cast<int32_t>(S * t) + (5 - S) + k + 5);
      }
      if (k < S - 1) slide(V, e); else e_new = e;
    }
    __builtin_amdgcn_sched_barrier(0);
    int32_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
      asm("v_dot4_i32_i8 %0, %1, %2, %3" : "=v"(acc[c]) : "v"(tq[(S * PH) % 5].q[c]), "v"(wq[0][c]), "v"(bias[c]));
    }
#pragma unroll
    for (int ky = 1; ky < 5; ky++) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        acc[c] = __builtin_amdgcn_sdot4(static_cast<int32_t>(tq[(S * PH + ky) % 5].q[c]), static_cast<int32_t>(wq[ky][c]), acc[c], false);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {
      acc[c] = __builtin_amdgcn_sdot4(static_cast<int32_t>(V[c]), static_cast<int32_t>(wv[c]), acc[c], false);
      acc[c] = __builtin_amdgcn_sdot4(static_cast<int32_t>(e_new), static_cast<int32_t>(w5[c]), acc[c], false);
    }
    const uint32_t packed = qnnp::q31_requantize_pack4<SEQ, FULL>(acc[0], acc[1], acc[2], acc[3], p.rq);
    if (p.stream_out) __builtin_amdgcn_raw_buffer_store_b32(packed, out_rsrc, out_voff, out_soff, 2);   // ("streaming_stores", as kernel G)
    else __builtin_amdgcn_raw_buffer_store_b32(packed, out_rsrc, out_voff, out_soff, 0);
    out_soff += out_step;
    slide(V, e_new);
    t++;
  };
  using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>; using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  constexpr std::true_type kFirst{};
  constexpr std::false_type kLater{};
  do {                                             // the first trip
    step(P0{}, kChecked, kFirst);
    if (t >= steps) return;
    step(P1{}, kChecked, kFirst);
    if (t >= steps) return;
    step(P2{}, kChecked, kFirst);
    if (t >= steps) return;
    step(P3{}, kChecked, kFirst);
    if (t >= steps) return;
    step(P4{}, kChecked, kFirst);
  } while (false);
  if (t + 5 <= t_inside) {
    // The first trip issues fewer loads than a steady one (X), and hipcc's wait for the oldest row at the top of the
    // steady loop is the minimum over both ways in: entered from the first trip it came out as vmcnt(5) -- everything but
    // the newest row -- on EVERY trip. With nothing in flight on the way in (rows 9..13 were requested five steps ago)
    // the back edge alone decides: vmcnt(25).
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0)
    do {                                           // steady state: straight-line body, counted waits
      step(P0{}, kInside, kLater); step(P1{}, kInside, kLater); step(P2{}, kInside, kLater); step(P3{}, kInside, kLater);
      step(P4{}, kInside, kLater);
    } while (t + 5 <= t_inside);
  }
  while (t < steps) {                              // the last steps of a segment (and of the image: padding rows)
    step(P0{}, kChecked, kLater);
    if (t >= steps) break;
    step(P1{}, kChecked, kLater);
    if (t >= steps) break;
    step(P2{}, kChecked, kLater);
    if (t >= steps) break;
    step(P3{}, kChecked, kLater);
    if (t >= steps) break;
    step(P4{}, kChecked, kLater);
  }
}

template <int S, int SEQ, bool FULL>
__global__ __launch_bounds__(kColThreads)
void q8_dwconv_col5x5_kernel(const DwParams p)
{
  // wave -> (image, row segment, 64-dword chunk of the flattened output row), as kernel G
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t b = blockIdx.x;
  if (p.xcd_ranges != 0u) {
    const uint32_t q = gridDim.x >> 3, r = gridDim.x & 7u, xcd = b & 7u;
    b = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + (b >> 3);
  }
  const uint32_t w = __builtin_amdgcn_readfirstlane(b * (kColThreads / 64) + (threadIdx.x >> 6));
  auto div_by = [](uint32_t x, uint32_t inv) __attribute__((always_inline)) { return inv != 0u ? __umulhi(x, inv) : x; };
  const uint32_t wb = div_by(w, p.inv_bands);
  const uint32_t chunk = w - wb * p.bands;
  const uint32_t n = div_by(wb, p.inv_slabs);
  const uint32_t seg = wb - n * p.slabs;
  if (n >= p.batch) return;
  const uint32_t q4 = p.C >> 2;
  const uint32_t cols = p.OW * q4;
  uint32_t j = chunk * 64u + lane;
  if (j >= cols) j = cols - 1u;                    // lanes past the end repeat the last dword column: harmless
  const uint32_t ox = div_by(j, p.inv_q4);
  const uint32_t cg = (j - ox * q4) * 4u;
  const uint32_t oy0 = seg * p.TOH;
  const uint32_t oy1 = min(p.OH, oy0 + p.TOH);
  const int32_t ix0 = static_cast<int32_t>(ox * S) - static_cast<int32_t>(p.pad_left);
  bool okc[5];
  bool all_ok = true;
#pragma unroll
  for (int k = 0; k < 5; k++) {
    okc[k] = ix0 + k >= 0 && ix0 + k < static_cast<int32_t>(p.W);
    all_ok = all_ok && okc[k];
  }
  if (__builtin_amdgcn_ballot_w64(!all_ok) != 0) {
    dwconv_col5x5_body<S, true, SEQ, FULL>(p, n, oy0, oy1, ox, cg, okc);
  } else {
    const bool yes[5] = {true, true, true, true, true};
    dwconv_col5x5_body<S, false, SEQ, FULL>(p, n, oy0, oy1, ox, cg, yes);
  }
}

// geometry of kernel H: as plan_col (`bands` = 64-dword chunks per flattened output row, `slabs` = row segments)
bool plan_col5(DwParams& p)
{
  if (p.C % 4 != 0 || p.KH != 5 || p.KW != 5 || p.dh != 1 || p.dw != 1) return false;
  if (p.sh != p.sw || (p.sw != 1 && p.sw != 2)) return false;
  if (p.wrange != 1u && p.wrange != 2u) return false;
  if (p.dot4 == nullptr) return false;
  // (the steady-state steps fetch row iy_first + S*t + 10 - S + k with no lower bound: >= 0 while pad_top <= 5)
  if (p.pad_top > 4 || p.pad_left > 4) return false;
  const uint64_t in_bytes = static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride;
  const uint64_t out_bytes = static_cast<uint64_t>(p.batch) * p.OH * p.OW * p.out_stride;
  if (in_bytes >= (UINT64_C(1) << 31) || out_bytes >= (UINT64_C(1) << 32)) return false;
  const uint32_t cols = p.OW * (p.C / 4);
  const uint32_t chunks = (cols + 63u) / 64u;
  // Row segments. A wave's start-up (weights, ten row requests, four transposes, two memory round trips) is worth about
  // as much as fourteen steps, so segments stay LONG -- 28 output rows or the whole image -- and there are only as many
  // as bring the launch to about six waves per SIMD. Measured with the waits of the walk fixed (batch 128, us per launch,
  // segments of 7 / 14 / 28 / 56 / 112 rows): 56x56x72 s2 17.2 / 15.2 / 14.6; 28x28x240 - / 25.1 / 23.0; 112x112x32
  // - / 41.9 / 35.8 / 36.5 / 42.4.
  const uint64_t waves_per_seg = static_cast<uint64_t>(p.batch) * chunks;
  const uint64_t target = static_cast<uint64_t>(p.cu_count) * 4u * 6u;
  uint32_t segs = static_cast<uint32_t>((target + waves_per_seg - 1) / waves_per_seg);
  const uint32_t min_rows = 28;
  uint32_t max_segs = p.OH / min_rows;
  if (max_segs < 1) max_segs = 1;
  if (segs > max_segs) segs = max_segs;
  if (segs < 1) segs = 1;
  uint32_t toh = (p.OH + segs - 1) / segs;
  if (const uint32_t forced = col_rows_override()) toh = forced < p.OH ? forced : p.OH;
  p.TOH = toh;
  p.slabs = (p.OH + toh - 1) / toh;
  p.bands = chunks;
  const uint64_t waves = static_cast<uint64_t>(p.batch) * p.slabs * chunks;
  const uint64_t dmax = chunks > p.slabs ? chunks : p.slabs;
  if ((waves + 8u * (kColThreads / 64)) * dmax >= (UINT64_C(1) << 32)) return false;
  if (static_cast<uint64_t>(cols) * (p.C / 4) >= (UINT64_C(1) << 32)) return false;
  return waves < (UINT64_C(1) << 31);
}

int launch_col5(const DwParams& geometry, hipStream_t stream)
{
  DwParams p = geometry;
  auto reciprocal = [](uint32_t d) { return d > 1u ? static_cast<uint32_t>(((UINT64_C(1) << 32) + d - 1u) / d) : 0u; };
  p.inv_bands = reciprocal(p.bands);
  p.inv_slabs = reciprocal(p.slabs);
  p.inv_q4 = reciprocal(p.C / 4u);
  p.xcd_ranges = (static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride +
                  static_cast<uint64_t>(p.batch) * p.OH * p.OW * p.out_stride) <= (UINT64_C(96) << 20) ? 1u : 0u;
  const uint64_t waves = static_cast<uint64_t>(p.batch) * p.slabs * p.bands;
  const uint32_t blocks = static_cast<uint32_t>((waves + (kColThreads / 64) - 1) / (kColThreads / 64));
  qnnp::requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    constexpr bool kFull = decltype(full)::value;
    if (p.sw == 1) {
      hipLaunchKernelGGL((q8_dwconv_col5x5_kernel<1, kSeq, kFull>), dim3(blocks), dim3(kColThreads), 0, stream, p);
    } else {
      hipLaunchKernelGGL((q8_dwconv_col5x5_kernel<2, kSeq, kFull>), dim3(blocks), dim3(kColThreads), 0, stream, p);
    }
  });
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

// --------------------------------------------------------------------------
// Kernel D: matrix cores, 3x3, any stride / dilation, C % 16 == 0
// --------------------------------------------------------------------------
/*
 * The VALU formulations above spend ~23 instructions per output (PMC: the kernel is 60-70 % VALU-busy and
 * still 3x off the HBM floor) -- ten of them to unpack bytes and multiply-accumulate the nine taps -- while
 * the matrix cores idle. Here the tap arithmetic goes to v_mfma_i32_32x32x32_i8 with a DIAGONAL weight
 * operand: for a tile of 32 output pixels x 32 channels and one tap,
 *     acc[c][m] += sum_k Wd[c][k] * A[k][m],   Wd[c][k] = x_tap[c] if k == c else 0,  A[k][m] = a'(pixel m, channel k)
 * 31/32 of the multiplies are by zero, but one instruction retires 1024 useful MACs, and the activation
 * operand is exactly a coalesced NHWC load: lane l = pixel l % 32, 16 consecutive channels (l / 32).
 * This is not a reshaping of the problem into a GEMM -- data movement is unchanged (each input byte is
 * still fetched once per tap through L1, outputs stored once); only the multiply unit changes.
 * x = w - kzp needs 9 bits: it is split into int8 parts (pack.h qnnp_pack_dwconv_mfma), one MFMA per tap
 * and part. a' = a ^ 0x80; padding taps read a = izp. The accumulator starts at the folded bias.
 * A workgroup keeps one 32-channel block (its diagonal operands live in LDS) and its four waves walk tiles
 * of 32 consecutive output pixels of the flattened (image, row, column) index space.
 */
constexpr int kMfThreads = 256;
typedef int dw_v4i __attribute__((ext_vector_type(4)));
typedef int dw_v16i __attribute__((ext_vector_type(16)));

template <int KH, int KW, int PARTS>
__global__ __launch_bounds__(kMfThreads, 4)
void q8_dwconv_mfma_kernel(const DwParams p)
{
  constexpr int TAPS = KH * KW;
  __shared__ __attribute__((aligned(16))) uint8_t lds_w[TAPS * PARTS * 1024];
  __shared__ __attribute__((aligned(16))) int32_t lds_bias[32];

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t m = lane & 31u;                 // pixel of the tile (operand B column / accumulator column)
  const uint32_t khalf = lane >> 5;              // 16-channel half of the block
  const uint32_t cblocks = p.c_pad32 / 32;
  const uint32_t cb = blockIdx.x % cblocks;      // this workgroup's channel block
  const uint32_t walker = blockIdx.x / cblocks;
  const uint32_t walkers = gridDim.x / cblocks;

  // ---- diagonal weight operands into LDS: lane (n = l % 32, k half = l / 32) holds bytes k = 16*half + j,
  //      non-zero only at k == n ----
  for (uint32_t f = wave; f < static_cast<uint32_t>(TAPS * PARTS); f += kMfThreads / 64) {
    const uint32_t t = f / PARTS;
    const uint32_t part = f - t * PARTS;
    const uint32_t x = static_cast<uint8_t>(p.dwm_x[(static_cast<size_t>(part) * TAPS + t) * p.c_pad32 + cb * 32 + m]);
    const bool mine = (m >> 4) == khalf;
    const uint32_t val = mine ? x << ((m & 3u) * 8u) : 0u;
    const uint32_t dw = (m & 15u) >> 2;
    dw_v4i v;
    v.x = static_cast<int>(dw == 0 ? val : 0u);
    v.y = static_cast<int>(dw == 1 ? val : 0u);
    v.z = static_cast<int>(dw == 2 ? val : 0u);
    v.w = static_cast<int>(dw == 3 ? val : 0u);
    *reinterpret_cast<dw_v4i*>(lds_w + f * 1024 + lane * 16) = v;
  }
  if (tid < 32) lds_bias[tid] = p.dwm_bias[cb * 32 + tid];
  __syncthreads();

  const uint32_t ohw = p.OH * p.OW;
  const uint32_t total = p.batch * ohw;                             // flattened output pixels (host checked < 2^32)
  const uint32_t tiles = (total + 31u) / 32u;
  const uint32_t fill = p.izp * 0x01010101u;
  const uint32_t c_first = cb * 32 + khalf * 16;                    // first channel of this lane's 16
  const bool chan_ok = c_first < p.C;                               // C % 16 == 0: all sixteen or none
  const uint8_t* in_c = p.input + (chan_ok ? c_first : 0u);

  qnnp::IgemmParams ep;                                             // only what the shared epilogue reads
  ep.n = p.C;
  ep.store_mode = p.store_mode;
  ep.rq = p.rq;

  qnnp::requant_dispatch(p.rq, [&](auto shift0, auto full) {
    // each workgroup owns a CONTIGUOUS run of tiles (its waves interleave inside it): the three input rows of
    // a tile are then re-used from L1 / the same L2 by the tiles one output row further on (PMC: with a
    // strided assignment the fabric read 2.8x the input)
    const uint32_t per_walker = (tiles + walkers - 1) / walkers;
    const uint32_t tile_end = min(tiles, (walker + 1) * per_walker);
    for (uint32_t tile = walker * per_walker + wave; tile < tile_end; tile += kMfThreads / 64) {
      const uint32_t gp = tile * 32u + m;
      const bool valid = gp < total;
      const uint32_t gpc = valid ? gp : total - 1;
      const uint32_t n = gpc / ohw;
      const uint32_t rem = gpc - n * ohw;
      const uint32_t oy = rem / p.OW;
      const uint32_t ox = rem - oy * p.OW;
      const uint32_t img_off = n * p.H * p.W;                       // in pixels (host checked: bytes < 2^32)
      const int32_t iy0 = static_cast<int32_t>(oy * p.sh) - static_cast<int32_t>(p.pad_top);
      const int32_t ix0 = static_cast<int32_t>(ox * p.sw) - static_cast<int32_t>(p.pad_left);

      // all tap operands first (independent loads in flight together), then the multiplies
      dw_v4i a[TAPS];
      bool ok[TAPS];
#pragma unroll
      for (int ky = 0; ky < KH; ky++) {
        const int32_t iy = iy0 + ky * static_cast<int32_t>(p.dh);
        const bool row_ok = valid && chan_ok && iy >= 0 && iy < static_cast<int32_t>(p.H);
#pragma unroll
        for (int kx = 0; kx < KW; kx++) {
          const int32_t ix = ix0 + kx * static_cast<int32_t>(p.dw);
          const bool o = row_ok && ix >= 0 && ix < static_cast<int32_t>(p.W);
          const uint32_t pix = o ? img_off + static_cast<uint32_t>(iy) * p.W + static_cast<uint32_t>(ix) : 0u;
          ok[ky * KW + kx] = o;
          a[ky * KW + kx] = *reinterpret_cast<const dw_v4i*>(in_c + static_cast<uint64_t>(pix) * p.in_stride);
        }
      }
      // the diagonal operands are re-read from LDS for every tile ON PURPOSE: left to itself the compiler
      // hoists the 18 loop-invariant reads (72 VGPRs) out of the tile loop and spills
      uint32_t w_off = lane * 16;
      asm volatile("" : "+v"(w_off));
      dw_v16i acc;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const int4 b = *reinterpret_cast<const int4*>(&lds_bias[rg * 8 + khalf * 4]);
        acc[rg * 4 + 0] = b.x; acc[rg * 4 + 1] = b.y; acc[rg * 4 + 2] = b.z; acc[rg * 4 + 3] = b.w;
      }
#pragma unroll
      for (int t = 0; t < TAPS; t++) {
        dw_v4i v = a[t];
        v.x = static_cast<int>((ok[t] ? static_cast<uint32_t>(v.x) : fill) ^ 0x80808080u);
        v.y = static_cast<int>((ok[t] ? static_cast<uint32_t>(v.y) : fill) ^ 0x80808080u);
        v.z = static_cast<int>((ok[t] ? static_cast<uint32_t>(v.z) : fill) ^ 0x80808080u);
        v.w = static_cast<int>((ok[t] ? static_cast<uint32_t>(v.w) : fill) ^ 0x80808080u);
#pragma unroll
        for (int part = 0; part < PARTS; part++) {
          const dw_v4i w = *reinterpret_cast<const dw_v4i*>(lds_w + (t * PARTS + part) * 1024 + w_off);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, v, acc, 0, 0, 0);
        }
      }
      uint8_t* out_row = p.output + static_cast<uint64_t>(gp) * p.out_stride;
      const int4 unused[4] = {};
      qnnp::igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, true>(
          acc, unused, 0, out_row, cb * 32, khalf, valid, ep);
    }
  });
}

bool plan_mfma(const DwParams& p, const struct qnnp_hip_dwconv_args* a)
{
  if (a->dwm_x == nullptr || a->dwm_bias == nullptr || a->dwm_parts < 1 || a->dwm_parts > 3) return false;
  if (!(p.KH == 3 && p.KW == 3)) return false;     // (25 taps x 4 operand registers do not fit beside the accumulators)
  if (p.C % 16 != 0 || p.in_stride % 16 != 0 || reinterpret_cast<uintptr_t>(p.input) % 16 != 0) return false;
  const uint64_t in_bytes = static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride;
  const uint64_t out_px = static_cast<uint64_t>(p.batch) * p.OH * p.OW;
  return in_bytes < (UINT64_C(1) << 32) && out_px < (UINT64_C(1) << 31);
}

template <int KH, int KW>
int launch_mfma(const DwParams& p, hipStream_t stream)
{
  const uint32_t cblocks = p.c_pad32 / 32;
  const uint64_t tiles = (static_cast<uint64_t>(p.batch) * p.OH * p.OW + 31) / 32;
  // persistent: up to 8 workgroups per CU (LDS: 9 or 18 KiB each), a multiple of the channel blocks
  uint64_t walkers = (static_cast<uint64_t>(p.cu_count) * 8u + cblocks - 1) / cblocks;
  const uint64_t max_walkers = (tiles + 3) / 4;
  if (walkers > max_walkers) walkers = max_walkers;
  if (walkers < 1) walkers = 1;
  const dim3 grid(static_cast<uint32_t>(walkers * cblocks));
  switch (p.dwm_parts) {
    case 1: hipLaunchKernelGGL((q8_dwconv_mfma_kernel<KH, KW, 1>), grid, dim3(kMfThreads), 0, stream, p); break;
    case 2: hipLaunchKernelGGL((q8_dwconv_mfma_kernel<KH, KW, 2>), grid, dim3(kMfThreads), 0, stream, p); break;
    default: hipLaunchKernelGGL((q8_dwconv_mfma_kernel<KH, KW, 3>), grid, dim3(kMfThreads), 0, stream, p); break;
  }
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

// --------------------------------------------------------------------------
// Kernel F: matrix cores + LDS-staged band (3x3, any stride / dilation, C % 16 == 0)
// --------------------------------------------------------------------------
/*
 * Kernel D showed that the diagonal-MFMA formulation needs only ~9-18 MFMAs per 1024 outputs (never the
 * bottleneck) but drowned in per-tap address arithmetic and 16-byte gathers. Here the two good halves are
 * joined: the band of ONE 32-channel block is staged into LDS like kernel A -- once per input byte, coalesced
 * 16-byte vectors, padding materialised, bytes already recentred (a' = a ^ 0x80), natural [row][column][32
 * channels] layout -- and every tap operand of a 32-pixel tile is then ONE ds_read_b128 at a constant offset
 * from the lane's base (lane l = pixel l % 32, channels 16*(l / 32)..+15): no select, no xor, no address math in
 * the tap loop. The diagonal weight operands of the workgroup's channel block live in REGISTERS (part 0; the
 * rare taps whose w - kzp needs a second / third int8 part fetch that operand from LDS, guarded by a per-tap
 * bit mask found at start-up). Per 1024 outputs: 9 LDS reads, 9+ MFMAs, ~100 VALU (requantization + pack).
 * Workgroup (4 waves) = image x band of output rows x one 32-channel block; a wave takes tiles of 32 consecutive
 * output positions of the band.
 */
#ifndef QNNP_ML_THREADS
#define QNNP_ML_THREADS 256
#endif
constexpr int kMlThreads = QNNP_ML_THREADS;
constexpr int kMlMaxVec = 6;          // 16-byte staging vectors a thread holds for the NEXT band (24 VGPRs)

/*
 * PERSISTENT and software-pipelined: with one band per workgroup every workgroup of the grid is in the same phase
 * at the same time (all load, then all compute, then all store: HBM idles while the VALUs work and vice versa,
 * and the last partial round of workgroups runs on a nearly empty chip -- three structurally different kernels
 * all measured the same 2 TB/s). Here a workgroup keeps its channel block and walks (image, band) pairs; the
 * global loads of the NEXT band are issued into registers before the current band is multiplied out of LDS.
 */
template <int PARTS>
__global__ __launch_bounds__(kMlThreads, 4)
void q8_dwconv_mfma_lds_kernel(const DwParams p)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t tile[];   // [IR][IC][32] a' bytes, then extra weight parts
  const uint32_t tid = threadIdx.x;
  QNNP_DW_TRACE(p, 0);
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t m = lane & 31u;
  const uint32_t khalf = lane >> 5;
  uint32_t b;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;   // channel blocks of a band share an XCD / L2
  }
  // the grid is a multiple of the channel-block count: workgroup b owns block b % cblocks for its whole life
  const uint32_t cblocks = p.slabs;
  const uint32_t cb = b % cblocks;
  const uint32_t pair0 = b / cblocks;
  const uint32_t pair_stride = gridDim.x / cblocks;
  const uint32_t segs = p.PP;                                      // column segments per output row
  const uint32_t npairs = p.batch * p.bands * segs;               // work items of this channel block: (image, band, segment)
  const uint32_t band_bytes = p.IR * p.IC * 32u;
  uint8_t* w_extra = tile + band_bytes;                            // [(part - 1) * 9 + tap] fragments of 1 KiB
  const uint32_t fill = p.izp * 0x01010101u;
  const bool my_chan_ok = cb * 32u + (tid & 1u) * 16u < p.C;     // C % 16 == 0: this thread's 16 channels exist or not
  // this thread's first staging pixel (tid / 2) inside a band and the advance of 128 pixels, both item-independent
  const uint32_t px0_row = (tid >> 1) / p.IC;
  const uint32_t px0_col = (tid >> 1) - px0_row * p.IC;
  const uint32_t step_row = (kMlThreads / 2) / p.IC;
  const uint32_t step_col = (kMlThreads / 2) - step_row * p.IC;

  // ---- staging, phase 1: global -> registers for one (image, band) pair ----
  uint4 st_val[kMlMaxVec];
  uint32_t st_nvec = 0;
  auto stage_load = [&](uint32_t pair) __attribute__((always_inline)) {
    const uint32_t seg = pair % segs;
    const uint32_t nb = pair / segs;
    const uint32_t n = nb / p.bands;
    const uint32_t band = nb - n * p.bands;
    const uint32_t oy0 = band * p.TOH;
    const uint32_t toh = min(p.TOH, p.OH - oy0);
    const uint32_t ir = (toh - 1) * p.sh + 2 * p.dh + 1;          // rows actually needed
    st_nvec = ir * p.IC * 2u;                                     // (a short last segment stages a few unused columns)
    const int32_t iy_base = static_cast<int32_t>(oy0 * p.sh) - static_cast<int32_t>(p.pad_top);
    const int32_t ix_base = static_cast<int32_t>(seg * p.CS * p.sw) - static_cast<int32_t>(p.pad_left);   // CS = outputs per segment
    const uint8_t* img = p.input + static_cast<uint64_t>(n) * p.H * p.W * p.in_stride + cb * 32u + (tid & 1u) * 16u;
    // vector v = tid + u * 256 is pixel (tid / 2 + u * 128), half tid % 2: its (row, column) inside the band is
    // walked incrementally from the thread's first pixel -- no division per vector (the address arithmetic of
    // this phase was as long as the multiply phase when it divided)
    uint32_t iyl = px0_row, ixl = px0_col;
#pragma unroll
    for (int u = 0; u < kMlMaxVec; u++) {
      const uint32_t v = tid + u * kMlThreads;
      const int32_t iy = iy_base + static_cast<int32_t>(iyl);
      const int32_t ix = ix_base + static_cast<int32_t>(ixl);
      const bool inb = v < st_nvec && my_chan_ok && !(p.abl & 2u) &&
          iy >= 0 && iy < static_cast<int32_t>(p.H) && ix >= 0 && ix < static_cast<int32_t>(p.W);
      st_val[u] = make_uint4(fill, fill, fill, fill);
      if (inb) {
        st_val[u] = *reinterpret_cast<const uint4*>(
            img + (static_cast<uint32_t>(iy) * p.W + static_cast<uint32_t>(ix)) * p.in_stride);
      }
      ixl += step_col;
      iyl += step_row;
      if (ixl >= p.IC) { ixl -= p.IC; iyl += 1; }
    }
  };
  // ---- staging, phase 2: registers -> LDS, recentred; [pixel][half] is exactly the vector order ----
  auto stage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kMlMaxVec; u++) {
      const uint32_t v = tid + u * kMlThreads;
      if (v < st_nvec) {
        uint4 x = st_val[u];
        x.x ^= 0x80808080u; x.y ^= 0x80808080u; x.z ^= 0x80808080u; x.w ^= 0x80808080u;
        *reinterpret_cast<uint4*>(tile + v * 16u) = x;
      }
    }
  };

  uint32_t pair = pair0;
  if (pair < npairs) stage_load(pair);

  // ---- diagonal weight operands: lane (c = l % 32, k half = l / 32) holds bytes k = 16*half + j, non-zero only
  //      at k == c. Part 0 in registers, further parts in LDS with a mask of the taps that need them ----
  // (all weight bytes are fetched first, back to back: one exposed memory latency instead of 9 * PARTS)
  uint32_t xb[PARTS * 9];
#pragma unroll
  for (int i = 0; i < PARTS * 9; i++) {
    xb[i] = static_cast<uint8_t>(p.dwm_x[static_cast<size_t>(i) * p.c_pad32 + cb * 32 + m]);   // [part][tap][channel]
  }
  auto diag = [&](uint32_t x) __attribute__((always_inline)) {
    const bool mine = (m >> 4) == khalf;
    const uint32_t val = mine ? x << ((m & 3u) * 8u) : 0u;
    const uint32_t dw = (m & 15u) >> 2;
    dw_v4i v;
    v.x = static_cast<int>(dw == 0 ? val : 0u);
    v.y = static_cast<int>(dw == 1 ? val : 0u);
    v.z = static_cast<int>(dw == 2 ? val : 0u);
    v.w = static_cast<int>(dw == 3 ? val : 0u);
    return v;
  };
  // all parts live in LDS (fragment (part * 9 + tap) of 1 KiB): with the next band's vectors held in registers
  // there is no room for nine operand quads; the reads ride beside the MFMAs
  uint32_t extra_mask = 0;                                         // bit (part - 1) * 9 + tap: that operand is not all zero
#pragma unroll
  for (int part = 0; part < PARTS; part++) {
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const dw_v4i v = diag(xb[part * 9 + t]);
      if (part > 0) {
        const bool nz = (v.x | v.y | v.z | v.w) != 0;
        if (__builtin_amdgcn_ballot_w64(nz) != 0) extra_mask |= 1u << ((part - 1) * 9 + t);
      }
      if (part == 0) {
        if (wave == 0) *reinterpret_cast<dw_v4i*>(w_extra + t * 1024 + lane * 16) = v;
      } else {
        // parts beyond the first are needed by few taps: kept as 32 bytes per (part, tap), the operand is
        // rebuilt from its byte on demand
        if (tid < 32) w_extra[9 * 1024 + ((part - 1) * 9 + t) * 32 + tid] = static_cast<uint8_t>(xb[part * 9 + t]);
      }
    }
  }
  // folded bias of the block: 32 int32 in LDS behind the weight parts (re-read per tile: registers are scarce)
  int32_t* lds_bias = reinterpret_cast<int32_t*>(w_extra + 9 * 1024 + (PARTS - 1) * 9 * 32);
  if (tid < 32) lds_bias[tid] = p.dwm_bias[cb * 32 + tid];

  qnnp::IgemmParams ep;                                             // only what the shared epilogue reads
  ep.n = p.C;
  ep.store_mode = p.store_mode;
  ep.rq = p.rq;
  const uint32_t row_pitch = p.IC * 32u;

  QNNP_DW_TRACE(p, 1);
  qnnp::requant_dispatch(p.rq, [&](auto shift0, auto full) {
    uint32_t trace_item = 0;
    for (bool first = true; pair < npairs; pair += pair_stride, first = false, trace_item++) {
      if (!first) __syncthreads();              // every reader of the previous band is done
      stage_store();
      __syncthreads();
      if (pair + pair_stride < npairs) stage_load(pair + pair_stride);   // flies while this band is multiplied

      const uint32_t seg = pair % segs;
      const uint32_t nb = pair / segs;
      const uint32_t n = nb / p.bands;
      const uint32_t band = nb - n * p.bands;
      const uint32_t oy0 = band * p.TOH;
      const uint32_t toh = min(p.TOH, p.OH - oy0);
      const uint32_t ox0 = seg * p.CS;
      const uint32_t ow = min(p.CS, p.OW - ox0);                     // output columns of this segment
      const uint32_t npos = toh * ow;
      const uint32_t ntiles = (npos + 31u) / 32u;
      uint8_t* out_band = p.output + ((static_cast<uint64_t>(n) * p.OH + oy0) * p.OW + ox0) * p.out_stride;
      for (uint32_t t = wave; t < ntiles; t += kMlThreads / 64) {
        const uint32_t pos = t * 32u + m;
        const bool valid = pos < npos && !(p.abl & 1u);
        const uint32_t pc = pos < npos ? pos : npos - 1;
        const uint32_t oyl = pc / ow;
        const uint32_t ox = pc - oyl * ow;
        const uint8_t* base = tile + (oyl * p.sh) * row_pitch + (ox * p.sw) * 32u + khalf * 16u;
        uint32_t w_off = lane * 16;                // opaque per tile: keeps the compiler from hoisting the nine
        asm volatile("" : "+v"(w_off));            // loop-invariant operand reads (36 VGPRs) out of the loop
        dw_v16i acc;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          const int4 bv = *reinterpret_cast<const int4*>(&lds_bias[rg * 8 + khalf * 4]);
          acc[rg * 4 + 0] = bv.x; acc[rg * 4 + 1] = bv.y; acc[rg * 4 + 2] = bv.z; acc[rg * 4 + 3] = bv.w;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
#pragma unroll
          for (int kx = 0; kx < 3; kx++) {
            const dw_v4i a = *reinterpret_cast<const dw_v4i*>(base + (ky * p.dh) * row_pitch + (kx * p.dw) * 32u);
            const dw_v4i w0 = *reinterpret_cast<const dw_v4i*>(w_extra + (ky * 3 + kx) * 1024 + w_off);
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w0, a, acc, 0, 0, 0);
            if constexpr (PARTS > 1) {
#pragma unroll
              for (int part = 1; part < PARTS; part++) {
                if ((extra_mask >> ((part - 1) * 9 + ky * 3 + kx)) & 1u) {      // wave-uniform, rarely taken
                  const dw_v4i w = diag(w_extra[9 * 1024 + ((part - 1) * 9 + ky * 3 + kx) * 32 + m]);
                  acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a, acc, 0, 0, 0);
                }
              }
            }
          }
        }
        uint8_t* out_row = out_band + (static_cast<uint64_t>(oyl) * p.OW + ox) * p.out_stride;
        const int4 unused[4] = {};
        qnnp::igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 1>(
            acc, unused, 0, out_row, cb * 32, khalf, valid, ep);
      }
      if (trace_item < 3) QNNP_DW_TRACE(p, 2 + trace_item);
    }
  });
  QNNP_DW_TRACE(p, 5);
}

// geometry of kernel F: `slabs` = 32-channel blocks; work item = (image, band of TOH output rows, segment of CS
// output columns, `PP` segments per row); the staged band is IR x IC pixels of 32 bytes
bool plan_mfma_lds(DwParams& p, const struct qnnp_hip_dwconv_args* a)
{
  if (a->dwm_x == nullptr || a->dwm_bias == nullptr || a->dwm_parts < 1 || a->dwm_parts > 3) return false;
  if (!(p.KH == 3 && p.KW == 3)) return false;
  if (p.C % 16 != 0 || p.in_stride % 16 != 0 || reinterpret_cast<uintptr_t>(p.input) % 16 != 0) return false;
  const uint32_t halo_r = 2 * p.dh + 1, halo_c = 2 * p.dw + 1;
  // the next band is held in registers: at most kMlMaxVec vectors of 16 bytes per thread = `cap` pixels; pick the
  // 2-D band (rows x column segment) that re-reads the least halo
  const uint32_t cap = static_cast<uint32_t>(kMlMaxVec * kMlThreads) / 2u;
  uint32_t best_segs = 0, best_toh = 0, best_ow = 0, best_ic = 0;
  double best_amp = 1e30;
  for (uint32_t segs = 1; segs <= 8 && segs <= p.OW; segs++) {
    const uint32_t ow = (p.OW + segs - 1) / segs;
    const uint32_t ic = (ow - 1) * p.sw + halo_c;
    if (cap / ic < halo_r) continue;
    uint32_t toh = (cap / ic - halo_r) / p.sh + 1;
    if (toh > p.OH) toh = p.OH;
    const double amp = (static_cast<double>((toh - 1) * p.sh + halo_r) / (toh * p.sh)) * (static_cast<double>(ic) / (ow * p.sw));
    if (amp < best_amp * 0.97) { best_amp = amp; best_segs = segs; best_toh = toh; best_ow = ow; best_ic = ic; }
  }
  if (best_segs == 0) return false;
  p.slabs = p.c_pad32 / 32;
  uint32_t toh = best_toh;
  // a few items per (persistent) workgroup, so that the pipeline has something to overlap and the shares are even
  while (toh > 1 && static_cast<uint64_t>(p.batch) * ((p.OH + toh - 1) / toh) * best_segs * p.slabs < 3u * 4u * p.cu_count) toh = (toh + 1) / 2;
  p.TOH = toh;
  p.CS = best_ow;
  p.PP = (p.OW + best_ow - 1) / best_ow;
  p.IC = best_ic;
  p.bands = (p.OH + toh - 1) / toh;
  p.IR = (toh - 1) * p.sh + halo_r;
  const uint64_t items = static_cast<uint64_t>(p.batch) * p.bands * p.PP * p.slabs;
  const uint64_t image_bytes = static_cast<uint64_t>(p.H) * p.W * p.in_stride;      // 32-bit offsets inside an image
  return items < (UINT64_C(1) << 31) && image_bytes < (UINT64_C(1) << 32);
}

int launch_mfma_lds(const DwParams& p, hipStream_t stream)
{
  const size_t lds_bytes = static_cast<size_t>(p.IR) * p.IC * 32u + 9u * 1024u + (p.dwm_parts - 1) * 9u * 32u + 128u;
  // persistent: four workgroups per CU (<= 128 VGPRs), a multiple of the channel-block count
  const uint32_t items = p.batch * p.bands * p.PP * p.slabs;
  uint32_t per_cu = static_cast<uint32_t>((160u * 1024u) / (lds_bytes > 0 ? lds_bytes : 1));
  if (per_cu > 4) per_cu = 4;
  if (per_cu < 1) per_cu = 1;
  uint32_t blocks = p.cu_count * per_cu;
  if (blocks > items) blocks = items;
  blocks = (blocks / p.slabs) * p.slabs;
  if (blocks == 0) blocks = p.slabs;
  switch (p.dwm_parts) {
    case 1: hipLaunchKernelGGL((q8_dwconv_mfma_lds_kernel<1>), dim3(blocks), dim3(kMlThreads), lds_bytes, stream, p); break;
    case 2: hipLaunchKernelGGL((q8_dwconv_mfma_lds_kernel<2>), dim3(blocks), dim3(kMlThreads), lds_bytes, stream, p); break;
    default: hipLaunchKernelGGL((q8_dwconv_mfma_lds_kernel<3>), dim3(blocks), dim3(kMlThreads), lds_bytes, stream, p); break;
  }
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

constexpr uint32_t kDwLdsBudgetDefault = 48 * 1024;   // bytes per workgroup (3 workgroups per CU)

// Pick slab width / band height for kernel A. Returns false if the shape does not fit.
bool plan_lds(DwParams& p, uint32_t budget)
{
  if (p.C % 4 != 0) return false;
  p.IC = (p.OW - 1) * p.sw + (p.KW - 1) * p.dw + 1;
  const uint32_t halo = (p.KH - 1) * p.dh + 1;
  const uint32_t want = p.OH < 4 ? p.OH : 4;
  uint32_t best_cs = 0, best_toh = 0, best_pp = 0;
  // candidate slabs: divisors of C, multiples of 4 (prefer 16), at most 4*kDwThreads channels
  for (uint32_t parts = 1; parts <= p.C / 4; parts++) {
    if (p.C % parts != 0) continue;
    const uint32_t cs = p.C / parts;
    if (cs % 4 != 0 || cs / 4 > kDwThreads) continue;
    if (p.C % 16 == 0 && cs % 16 != 0) continue;      // keep 16-byte staging loads when the tensor allows them
    const uint32_t q4 = cs / 4;
    // line pitch (dwords): >= IC, and == spread (mod 32) so that the q4 group lines of a row start on
    // different banks: spread = 32/q4 for few groups, 1 (odd) for many
    const uint32_t spread = q4 >= 32 ? 1u : (32u + q4 - 1) / q4;
    uint32_t pp = p.IC;
    while (pp % 32 != spread % 32) pp++;
    const uint32_t row_bytes = q4 * pp * 4;
    const uint32_t max_rows = budget / row_bytes;
    if (max_rows < halo) continue;
    uint32_t toh = (max_rows - halo) / p.sh + 1;
    if (toh > p.OH) toh = p.OH;
    if (best_cs == 0) { best_cs = cs; best_toh = toh; best_pp = pp; }   // widest slab that fits at all
    if (toh >= want) { best_cs = cs; best_toh = toh; best_pp = pp; break; }
  }
  if (best_cs == 0) return false;
  p.CS = best_cs;
  p.PP = best_pp;
  p.slabs = p.C / best_cs;
  // keep the machine busy: at least ~4 workgroups per CU when the batch is small
  uint32_t toh = best_toh;
  while (toh > 1 && static_cast<uint64_t>(p.batch) * ((p.OH + toh - 1) / toh) * p.slabs < 1024) {
    toh = (toh + 1) / 2;
  }
  p.TOH = toh;
  p.bands = (p.OH + toh - 1) / toh;
  p.IR = (toh - 1) * p.sh + halo;
  return true;
}

template <int KH, int KW>
int launch_lds(const DwParams& p, bool vec16, hipStream_t stream)
{
  const uint32_t blocks = p.batch * p.bands * p.slabs;
  const size_t lds_bytes = static_cast<size_t>(p.IR) * (p.CS / 4) * p.PP * 4;
  const bool dw1 = p.dw == 1;
#define QNNP_DW_LAUNCH(V, D) \
  hipLaunchKernelGGL((q8_dwconv_lds_kernel<KH, KW, V, D>), dim3(blocks), dim3(kDwThreads), lds_bytes, stream, p)
  if (vec16) {
    if (dw1) QNNP_DW_LAUNCH(16, true); else QNNP_DW_LAUNCH(16, false);
  } else {
    if (dw1) QNNP_DW_LAUNCH(4, true); else QNNP_DW_LAUNCH(4, false);
  }
#undef QNNP_DW_LAUNCH
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

// --------------------------------------------------------------------------
// Kernel M16 (round 6): 3x3, stride 1, dilation 1, channels % 16 == 0, weights in int8 range -- the window on v_mfma_i32_16x16x64_i8
// --------------------------------------------------------------------------
/*
 * Replaces q8dwconv_ukernel_up8x9__sse2 (reference src/q8dwconv/up8x9-sse2.c:14-372) for the stride-1 layers whose column walk
 * (kernel G) is bound by instruction issue: 33 VALU per output dword, of which 18 are the multiplies (6 v_perm + 12 v_dot4).
 * Here the multiplies go to the matrix cores with NO transposition of the data:
 *   D[c][p] = sum_k A[c][k] B[k][p],   K = 64 = 4 tap slots x 16 channels,
 *   B[(g, c')][p] = a'(pixel of position p shifted by tap g)[c0 + c']   -- the 16 contiguous channel bytes of an NHWC pixel:
 *                   operand lane (p = l & 15, g = l >> 4) LOADS its fragment with one 16-byte load, re-centres it, done;
 *   A[c][(g, c')] = x[ky][kx = g][c] if c' == c else 0                   -- a diagonal: lane (c, g) holds one non-zero byte.
 * One instruction = one kernel ROW (kx = 0, 1, 2 in slots g = 0, 1, 2; slot 3 multiplies by zero) of 16 positions x 16 channels.
 * A wave owns a strip of 16 output columns x TN channel blocks and walks DOWN a segment of rows like kernel G: an input row is
 * loaded ONCE (TN fragments) and feeds three MFMAs per block -- kernel row 0 of output row t (its first product: the bias is the
 * C operand), row 1 of t - 1, row 2 of t - 2, which completes it: requantize 4 accumulators -> one dword, a 4 x 4 lane transpose,
 * one 16-byte store per lane (a pixel's 16 TN contiguous channels). Three accumulator sets rotate; a trip of the loop is three
 * rows, so the rotation is compile-time. Per output dword: 4 v_xor + the requantization + 0.4 other VALU against 33 in kernel G.
 * Padding as kernel G's late scheme: out-of-image COLUMNS are never loaded (offset beyond the descriptor: 0) and their constant
 * izp * x goes into the lane's bias once; out-of-image ROWS are replaced where they are consumed (checked trips only).
 * Weights and bias come from the dot-product image of pack.h (qnnp_pack_dwconv_dot4: x = +-(w - kzp), activations ^ 0x80 | 0x7f).
 *
 * STATUS (round 6): parity-green (tests/test_gpu_dwmfma16.py) and SLOWER than kernel G on every stride-1 MobileNetV2 layer -- 26-32
 * against 25 us (layer 2), 59 against 28 (layer 8), 15-19 against 11 (layer 13), 10-13 against 7-9.5 (layers 18 / 22 / 27): kept as
 * "dwconv_kernel" 7 for the record, never chosen automatically. Four builds (profiles/r06/dw_mfma16_walk_negative_result_r06n.txt):
 * fragments loaded in the operand pattern (16 pixels at the pixel stride per 16-lane group: bound by address processing), one
 * coalesced 16-byte load per lane + a wave-private LDS transposition, six rows in flight + stores in pixel order, and the finish of a
 * row deferred by one step. Counters on layer 8 (pmc_dw_mfma16_vs_col_walk_layer8_r06o.txt): 5.66 M VALU + MFMA instructions per launch
 * against kernel G's 7.52 M -- a quarter fewer, as planned -- but 174-190 registers leave two waves per SIMD where kernel G runs six
 * or seven, and a step is one chain (row -> LDS -> fragments -> MFMAs -> requantize -> LDS -> store): waves parked on s_waitcnt 45 % of
 * their cycles and stalled at issue another 35 %. The diagonal weight fragments (one non-zero byte in sixteen per lane) are what
 * the registers go to; a form with <= 96 registers would be needed to compete.
 */
constexpr int kM16Threads = 256;

template <int TN, int SEQ, bool FULL>
__global__ __launch_bounds__(kM16Threads, TN >= 3 ? 2 : 3)
void q8_dwconv_mfma16_3x3_kernel(const DwParams p)
{
  typedef int m16_v4i __attribute__((ext_vector_type(4)));
  // wave-private row buffers: a row of the wave's 18 pixels x TN chunks is fetched with ONE 16-byte load per lane -- lane i takes
  // chunk i % TN of pixel i / TN: a pixel's 16 TN bytes are contiguous, so the load touches ~6 lines per 16-lane group where the
  // fragment pattern itself (16 pixels at the pixel stride per group) touched 16-18 and bound the first build by address processing
  // (layer 8: 64 us) -- re-centred, written to LDS lane-linearly and read back in the fragment pattern. Pixel pitch in LDS: 3 chunks
  // (an odd number: the sixteen pixels of a ds_read_b128 lane group fall into sixteen different bank quads).
  constexpr uint32_t kPitch = 3u;                  // 16-byte chunks per pixel in LDS
  // (a row buffer: 18 pixels x kPitch chunks = 864 bytes of its 1 KiB slot)
  __shared__ __attribute__((aligned(16))) uint8_t lds[(kM16Threads / 64) * 4 * 1024];      // per wave: 3 row buffers + the output row image
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t fp = lane & 15u;                 // operand: position; result: position
  const uint32_t fg = lane >> 4;                  // operand: tap slot (kernel column); result: channel quad
  auto div_by = [](uint32_t x, uint32_t inv) __attribute__((always_inline)) { return inv != 0u ? __umulhi(x, inv) : x; };
  // wave -> (image, row segment, strip of 16 columns, channel group): channel groups fastest, so that the waves of a workgroup
  // read the same pixels' lines
  const uint32_t wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t w = __builtin_amdgcn_readfirstlane(blockIdx.x * (kM16Threads / 64) + wave_in_wg);
  const uint32_t cgroups = p.IR, strips = p.bands;
  const uint32_t w1 = div_by(w, p.inv_q4);        // / cgroups
  const uint32_t cgrp = w - w1 * cgroups;
  const uint32_t w2 = div_by(w1, p.inv_bands);    // / strips
  const uint32_t strip = w1 - w2 * strips;
  const uint32_t n = div_by(w2, p.inv_slabs);     // / segments
  const uint32_t seg = w2 - n * p.slabs;
  if (n >= p.batch) return;
  const uint32_t c0 = cgrp * (16u * TN);
  const uint32_t x0 = strip * 16u;
  const uint32_t oy0 = seg * p.TOH;
  const uint32_t oy1 = min(p.OH, oy0 + p.TOH);
  const uint32_t flip = p.wrange == 2u ? 0x7f7f7f7fu : 0x80808080u;    // wave-uniform
  uint8_t* rowbuf = lds + wave_in_wg * (4u * 1024u);
  uint8_t* outbuf = rowbuf + 3u * 1024u;

  // ---- loader role: lane i < 18 TN fetches chunk i % TN of strip pixel i / TN (input column x0 - pad_left + i / TN) ----
  const uint32_t lpix = lane / TN, lchunk = lane - lpix * TN;
  const bool loader = lane < 18u * TN;
  const int32_t lix = static_cast<int32_t>(x0 + lpix) - static_cast<int32_t>(p.pad_left);
  const bool lcol_ok = loader && lix >= 0 && lix < static_cast<int32_t>(p.W);
  const uint32_t row_bytes = p.W * p.in_stride;
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(p.batch * p.H * row_bytes), 0x00020000);
  // (an offset beyond the descriptor's extent: the load returns 0 -- plan_m16 keeps the tensor below 2^31 bytes)
  const uint32_t coff = lcol_ok ? static_cast<uint32_t>(lix) * p.in_stride + c0 + 16u * lchunk : 0x80000000u;
  const uint32_t fillraw = lcol_ok ? p.izp * 0x01010101u : 0u;                     // what a padding ROW holds at this lane's column
  const uint32_t img_off = n * p.H * row_bytes;
  const uint32_t wr_off = (lpix * kPitch + lchunk) * 16u;                          // this lane's chunk inside a row buffer
  // ---- operand role: position fp, tap slot fg (slot 3 repeats slot 2: its weights are zero) ----
  const uint32_t rd_off = ((fp + min(fg, 2u)) * kPitch) * 16u;                     // + tn * 16

  // ---- weights: the diagonal fragments, from the dot-product image ((x_r0, x_r1, x_r2, 0) per channel) ----
  m16_v4i wf[3][TN];
#pragma unroll
  for (int ky = 0; ky < 3; ky++) {
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      const uint32_t word = p.dot4[ky * p.c_pad + c0 + 16u * tn + fp];
      const uint32_t b = ((word >> (8u * fg)) & 0xFFu) << (8u * (fp & 3u));        // (slot 3: the image's fourth byte is 0)
      const uint32_t q = fp >> 2;
      wf[ky][tn] = m16_v4i{static_cast<int>(q == 0u ? b : 0u), static_cast<int>(q == 1u ? b : 0u),
                           static_cast<int>(q == 2u ? b : 0u), static_cast<int>(q == 3u ? b : 0u)};
    }
  }
  // ---- bias of this lane's result channels 16 tn + 4 fg + r (+ 2^31 for the offset rounding forms) ----
  const uint32_t ox = x0 + fp;
  const bool pos_ok = ox < p.OW;
  m16_v4i bias[TN];
  {
    // taps of this lane's POSITION that read outside the image: 0 was multiplied where the zero point belongs
    const int32_t ixb = static_cast<int32_t>(ox) - static_cast<int32_t>(p.pad_left);
    bool tap_out[3];
#pragma unroll
    for (int k = 0; k < 3; k++) tap_out[k] = (ixb + k) < 0 || (ixb + k) >= static_cast<int32_t>(p.W);
    const bool any_out = __builtin_amdgcn_ballot_w64(tap_out[0] || tap_out[1] || tap_out[2]) != 0;
    const int32_t step = p.wrange == 2u ? -static_cast<int32_t>(p.izp) : static_cast<int32_t>(p.izp);
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      const uint32_t cq = c0 + 16u * tn + 4u * fg;
      const int4 bv = *reinterpret_cast<const int4*>(p.dot4 + 3u * p.c_pad + cq);
      int32_t b[4] = {bv.x, bv.y, bv.z, bv.w};
      if (any_out) {                                                               // (border strips only: wave-uniform)
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
          const uint4 xv = *reinterpret_cast<const uint4*>(p.dot4 + ky * p.c_pad + cq);
          const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
          for (int r = 0; r < 4; r++) {
            int32_t sum = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) sum += tap_out[k] ? static_cast<int8_t>(xw[r] >> (8 * k)) : 0;
            b[r] = qnnp::add_wrap(b[r], step * sum);
          }
        }
      }
      bias[tn] = m16_v4i{qnnp::with_rq_offset<SEQ>(b[0]), qnnp::with_rq_offset<SEQ>(b[1]),
                         qnnp::with_rq_offset<SEQ>(b[2]), qnnp::with_rq_offset<SEQ>(b[3])};
    }
  }

  const __amdgpu_buffer_rsrc_t out_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      p.output, 0, static_cast<int>(p.batch * p.OH * p.OW * p.out_stride), 0x00020000);
  // after the lane transpose lane (position, g) holds channel block g: 16 bytes (lanes with g >= TN hold nothing). Stored like that,
  // a pixel's 16 TN bytes would leave in TN separate 16-byte pieces from lanes 16 apart (the first build: layer 8 at 65 us); they go
  // through the wave's LDS image once more and leave in the LOADER's lane order -- lane i = chunk i % TN of position i / TN, a
  // pixel's bytes from adjacent lanes.
  const uint32_t st_pos = lane / TN, st_chunk = lane - st_pos * TN;
  const bool st_ok = lane < 16u * TN && x0 + st_pos < p.OW;
  const uint32_t out_voff = st_ok ? (x0 + st_pos) * p.out_stride + c0 + 16u * st_chunk : 0x80000000u;
  const uint32_t ow_off = (fp * TN + min(fg, static_cast<uint32_t>(TN - 1))) * 16u;   // where lane (position, g) puts its block
  (void) pos_ok;
  const uint32_t out_row = p.OW * p.out_stride;
  const uint32_t out_img = n * p.OH * out_row;

  // ---- the walk: t = input row + pad_top; row t feeds kernel row 0 of output row t, row 1 of t - 1, row 2 of t - 2 ----
  m16_v4i raw[6];                                 // SIX rows in flight (this lane's chunk): row t lives in raw[t % 6]
  m16_v4i acc[3][TN];                             // output row oy accumulates in set oy % 3
  auto request = [&](auto ph_c, int32_t t) __attribute__((always_inline)) {
    constexpr int PH = decltype(ph_c)::value;                                        // 0 .. 5
    int32_t iy = t - static_cast<int32_t>(p.pad_top);
    iy = iy < 0 ? 0 : (iy >= static_cast<int32_t>(p.H) ? static_cast<int32_t>(p.H) - 1 : iy);      // (scalar; replaced where consumed)
    const uint32_t ro = img_off + static_cast<uint32_t>(iy) * row_bytes;
    raw[PH] = __builtin_bit_cast(m16_v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, coff, ro, 0));
  };
  // The row an earlier step completed is finished one step LATER, in two halves around the next row's LDS round trip and MFMAs, so
  // that nothing in a step waits for what the step itself started (the first build ran write -> read -> MFMA -> requantize -> write
  // -> read -> store as one chain per step: waves parked 45 % of their cycles, two per SIMD).
  auto finish_a = [&](auto slot_c) __attribute__((always_inline)) {                // requantize set S, lane transpose, into the image
    constexpr int S = decltype(slot_c)::value;
    uint32_t q[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      q[tn] = qnnp::q31_requantize_pack4<SEQ, FULL>(acc[S][tn][0], acc[S][tn][1], acc[S][tn][2], acc[S][tn][3], p.rq);
    }
    const auto s02 = __builtin_amdgcn_permlane32_swap(q[0], q[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(q[1], q[3], false, false);
    const auto tlo = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    const auto thi = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    const m16_v4i blk = {static_cast<int>(tlo[0]), static_cast<int>(tlo[1]), static_cast<int>(thi[0]), static_cast<int>(thi[1])};
    if (fg < static_cast<uint32_t>(TN)) *reinterpret_cast<m16_v4i*>(outbuf + ow_off) = blk;
  };
  auto finish_b = [&](const m16_v4i& outv, uint32_t oy) __attribute__((always_inline)) {
    const auto bits = __builtin_bit_cast(__attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int, outv);
    const uint32_t soff = out_img + oy * out_row;
    if (p.stream_out) __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, out_voff, soff, 2);
    else __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, out_voff, soff, 0);
  };
  // one row: PH6 = t % 6 (the raw row register), accumulator phase PH = t % 3. CHECK: the row may lie outside the image (then it
  // holds the zero point / 0 at this lane's column). Output row t - 3 was completed by the previous step in set PH -- the set this
  // step's first product overwrites: it is requantized first.
  auto step = [&](auto ph_c, auto check_c, int32_t t, int32_t t_next) __attribute__((always_inline)) {
    constexpr int PH6 = decltype(ph_c)::value;
    constexpr int PH = PH6 % 3;
    constexpr bool CHECK = decltype(check_c)::value;
    m16_v4i x = raw[PH6];
    if constexpr (CHECK) {
      const int32_t iy = t - static_cast<int32_t>(p.pad_top);
      if (iy < 0 || iy >= static_cast<int32_t>(p.H)) {                             // (wave-uniform)
        x = m16_v4i{static_cast<int>(fillraw), static_cast<int>(fillraw), static_cast<int>(fillraw), static_cast<int>(fillraw)};
      }
    }
    x.x ^= static_cast<int>(flip); x.y ^= static_cast<int>(flip); x.z ^= static_cast<int>(flip); x.w ^= static_cast<int>(flip);
    uint8_t* buf = rowbuf + PH * 1024u;
    if (loader) *reinterpret_cast<m16_v4i*>(buf + wr_off) = x;
    request(ph_c, t_next);                                                          // the row a trip ahead, into the register just consumed
    const int32_t oy_done = t - 3;
    const bool done = oy_done >= static_cast<int32_t>(oy0) && oy_done < static_cast<int32_t>(oy1);   // (wave-uniform)
    if (done) finish_a(std::integral_constant<int, PH>{});
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                         // the wave's own LDS writes before its reads
    m16_v4i xf[TN];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) xf[tn] = *reinterpret_cast<const m16_v4i*>(buf + rd_off + tn * 16);
    const m16_v4i outv = *reinterpret_cast<const m16_v4i*>(outbuf + lane * 16u);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      acc[PH][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[0][tn], xf[tn], bias[tn], 0, 0, 0);
      acc[(PH + 2) % 3][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[1][tn], xf[tn], acc[(PH + 2) % 3][tn], 0, 0, 0);
      acc[(PH + 1) % 3][tn] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[2][tn], xf[tn], acc[(PH + 1) % 3][tn], 0, 0, 0);
    }
    if (done) finish_b(outv, static_cast<uint32_t>(oy_done));
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  using P2 = std::integral_constant<int, 2>;
  using P3 = std::integral_constant<int, 3>;
  using P4 = std::integral_constant<int, 4>;
  using P5 = std::integral_constant<int, 5>;
  // (segments start at multiples of SIX rows: plan_m16)
  int32_t t = static_cast<int32_t>(oy0);
  const int32_t t_end = static_cast<int32_t>(oy1) + 2;                              // rows oy0 .. oy1 + 1 are consumed
  request(P0{}, t);
  request(P1{}, t + 1);
  request(P2{}, t + 2);
  request(P3{}, t + 3);
  request(P4{}, t + 4);
  request(P5{}, t + 5);
  // sets 1 and 2 are accumulated into before their first bias product on the first two rows (output rows before the segment,
  // never stored): give them a defined value
#pragma unroll
  for (int tn = 0; tn < TN; tn++) { acc[1][tn] = bias[tn]; acc[2][tn] = bias[tn]; }
  for (; t < t_end; t += 6) {
    const int32_t iy_lo = t - static_cast<int32_t>(p.pad_top);
    if (iy_lo >= 0 && iy_lo + 5 < static_cast<int32_t>(p.H)) {
      step(P0{}, std::false_type{}, t, t + 6);
      step(P1{}, std::false_type{}, t + 1, t + 7);
      step(P2{}, std::false_type{}, t + 2, t + 8);
      if (t + 3 < t_end) {
        step(P3{}, std::false_type{}, t + 3, t + 9);
        step(P4{}, std::false_type{}, t + 4, t + 10);
        step(P5{}, std::false_type{}, t + 5, t + 11);
      }
    } else {
      step(P0{}, std::true_type{}, t, t + 6);
      step(P1{}, std::true_type{}, t + 1, t + 7);
      step(P2{}, std::true_type{}, t + 2, t + 8);
      if (t + 3 < t_end) {
        step(P3{}, std::true_type{}, t + 3, t + 9);
        step(P4{}, std::true_type{}, t + 4, t + 10);
        step(P5{}, std::true_type{}, t + 5, t + 11);
      }
    }
  }
  // the segment's last row is completed by step t_end - 1 (in set t_end % 3) and finished by step t_end -- which exists only when
  // the last half trip does not end at t_end - 1: then it is finished here
  if ((t_end - static_cast<int32_t>(oy0)) % 3 == 0) {
    const uint32_t last = static_cast<uint32_t>(t_end) % 3u;
    if (last == 0u) finish_a(P0{}); else if (last == 1u) finish_a(P1{}); else finish_a(P2{});
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const m16_v4i outv = *reinterpret_cast<const m16_v4i*>(outbuf + lane * 16u);
    finish_b(outv, oy1 - 1u);
  }
}

// geometry of kernel M16: `bands` = strips of 16 output columns, `slabs` = row segments (TOH rows each),
// `CS` = channel blocks per wave (TN), `IR` = channel groups (C / (16 TN)); TOH a multiple of six (the walk's trip)
bool plan_m16(DwParams& p, uintptr_t in_addr, uintptr_t out_addr)
{
  if (p.KH != 3 || p.KW != 3 || p.sh != 1 || p.sw != 1 || p.dh != 1 || p.dw != 1) return false;
  if (p.C % 16 != 0 || p.in_stride % 16 != 0 || p.out_stride % 16 != 0 || in_addr % 16 != 0 || out_addr % 16 != 0) return false;
  if (!(p.wrange == 1u || p.wrange == 2u) || p.dot4 == nullptr) return false;
  if (p.pad_top > 2 || p.pad_left > 2) return false;
  const uint64_t in_bytes = static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride;
  const uint64_t out_bytes = static_cast<uint64_t>(p.batch) * p.OH * p.OW * p.out_stride;
  if (in_bytes >= (UINT64_C(1) << 31) || out_bytes >= (UINT64_C(1) << 31)) return false;
  const uint32_t blocks = p.C / 16u;
  // (three blocks per wave where they divide the channels: 140 registers, three waves per SIMD; four would leave one)
  uint32_t tn = blocks % 3u == 0 ? 3u : (blocks % 2u == 0 ? 2u : 1u);
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_DW_M16_TN")) {            // measurement builds: blocks per wave
    const uint32_t v = static_cast<uint32_t>(atoi(env));
    if (v >= 1u && v <= 3u && blocks % v == 0u) tn = v;
  }
#endif
  const uint32_t cgroups = blocks / tn;
  const uint32_t strips = (p.OW + 15u) / 16u;
  const uint64_t per_seg = static_cast<uint64_t>(p.batch) * strips * cgroups;
  // ~3 waves per SIMD in all; segments of at least 9 rows, in multiples of three
  const uint64_t target = static_cast<uint64_t>(p.cu_count) * 4u * 3u;
  uint32_t segs = static_cast<uint32_t>((target + per_seg - 1) / per_seg);
  uint32_t max_segs = p.OH / 9u;
  if (max_segs < 1u) max_segs = 1u;
  if (segs > max_segs) segs = max_segs;
  if (segs < 1u) segs = 1u;
  uint32_t toh = (p.OH + segs - 1u) / segs;
  toh = (toh + 5u) / 6u * 6u;
  if (const uint32_t forced = col_rows_override()) toh = (forced + 5u) / 6u * 6u;
  p.TOH = toh;
  p.slabs = (p.OH + toh - 1u) / toh;
  p.bands = strips;
  p.CS = tn;
  p.IR = cgroups;
  const uint64_t waves = per_seg * p.slabs;
  const uint64_t dmax = strips > p.slabs ? (strips > cgroups ? strips : cgroups) : (p.slabs > cgroups ? p.slabs : cgroups);
  return (waves + 8u) * dmax < (UINT64_C(1) << 32);
}

int launch_m16(const DwParams& geometry, hipStream_t stream)
{
  DwParams p = geometry;
  auto reciprocal = [](uint32_t d) { return d > 1u ? static_cast<uint32_t>(((UINT64_C(1) << 32) + d - 1u) / d) : 0u; };
  p.inv_q4 = reciprocal(p.IR);
  p.inv_bands = reciprocal(p.bands);
  p.inv_slabs = reciprocal(p.slabs);
  const uint64_t waves = static_cast<uint64_t>(p.batch) * p.slabs * p.bands * p.IR;
  const uint32_t blocks = static_cast<uint32_t>((waves + (kM16Threads / 64) - 1) / (kM16Threads / 64));
  qnnp::requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
    constexpr int kSeq = decltype(seq)::value;
    constexpr bool kFull = decltype(full)::value;
    switch (p.CS) {
      case 3: hipLaunchKernelGGL((q8_dwconv_mfma16_3x3_kernel<3, kSeq, kFull>), dim3(blocks), dim3(kM16Threads), 0, stream, p); break;
      case 2: hipLaunchKernelGGL((q8_dwconv_mfma16_3x3_kernel<2, kSeq, kFull>), dim3(blocks), dim3(kM16Threads), 0, stream, p); break;
      default: hipLaunchKernelGGL((q8_dwconv_mfma16_3x3_kernel<1, kSeq, kFull>), dim3(blocks), dim3(kM16Threads), 0, stream, p); break;
    }
  });
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

enum : uint32_t { kPlanDirect = 1, kPlanLds33, kPlanLds55, kPlanRow, kPlanMfma, kPlanMfmaLds, kPlanCol, kPlanCol5, kPlanM16, kPlanDirect4, kPlanRowAny };

// measurement knob, read once: LDS budget per workgroup of the LDS-tiled kernel in KiB
uint32_t lds_budget()
{
  static const uint32_t budget = [] {
    if (const char* env = getenv("QNNP_GFX950_DW_LDS_KB")) {
      const int kb = atoi(env);
      if (kb >= 4 && kb <= 64) return static_cast<uint32_t>(kb) * 1024u;
    }
    return kDwLdsBudgetDefault;
  }();
  return budget;
}

// Kernel choice and launch geometry for one (shape, variant, alignment); fills `p` and `plan`.
int make_plan(DwParams& p, const struct qnnp_hip_dwconv_args* a, uintptr_t in_addr, uintptr_t out_addr,
              struct qnnp_hip_dwconv_plan* plan)
{
  const bool k33 = p.KH == 3 && p.KW == 3;
  const bool k55 = p.KH == 5 && p.KW == 5;
  const bool aligned4 = p.in_stride % 4 == 0 && p.out_stride % 4 == 0 && in_addr % 4 == 0 && out_addr % 4 == 0;
  p.store_mode = 0;
  if (p.C % 16 == 0 && p.out_stride % 16 == 0 && out_addr % 16 == 0) p.store_mode = 2;
  else if (p.C % 4 == 0 && p.out_stride % 4 == 0 && out_addr % 4 == 0) p.store_mode = 1;
  plan->vec16 = 0;
  plan->kernel = 0;
  if (a->variant == 8) {
    // (round 6) the sliding-window kernel on unaligned dwords, forced (auto: below, where nothing aligned takes the shape)
    if (!k33 || a->c_pad % 4 != 0 || !plan_row(p, true)) return QNNP_HIP_EINVAL;
    plan->kernel = kPlanRowAny;
  } else if (a->variant == 7) {
    // (round 6) the 16x16x64 matrix-core walk, forced
    if (!k33 || !plan_m16(p, in_addr, out_addr)) return QNNP_HIP_EINVAL;
    plan->kernel = kPlanM16;
  } else if (a->variant == 6 && k55) {
    if (!aligned4 || !plan_col5(p)) return QNNP_HIP_EINVAL;
    plan->kernel = kPlanCol5;
  } else if (a->variant == 6) {
    if (!aligned4 || !plan_col(p)) return QNNP_HIP_EINVAL;
    plan->kernel = kPlanCol;
  } else if (a->variant == 5) {
    if (!plan_mfma_lds(p, a)) return QNNP_HIP_EINVAL;
    plan->kernel = kPlanMfmaLds;
  } else if (a->variant == 4) {
    if (!plan_mfma(p, a)) return QNNP_HIP_EINVAL;
    plan->kernel = kPlanMfma;
  } else if (a->variant == 0 && k33 && aligned4 && plan_col(p)) {
    // 3x3, dilation 1, stride 1 | 2: the column-sliding window (kernel G) -- 20-35 % ahead of the LDS-tiled and the
    // matrix-core kernels on every one of the ten MobileNetV2 depthwise layers at batch 128 (same-box A/B,
    // scripts/gpu_dwab.sh, profiles/r02/dwconv_kernel_ab_*.txt).
    plan->kernel = kPlanCol;
  } else if (a->variant == 0 && k55 && aligned4 && plan_col5(p)) {
    // 5x5, dilation 1, stride 1 | 2, weights in int8 range: the same walk with five-row windows (kernel H)
    plan->kernel = kPlanCol5;
  } else if (a->variant == 0 && k33 && p.OW >= 56 && p.C <= 96 && plan_mfma_lds(p, a)) {
    // (shapes kernel G declines, e.g. tensors beyond its 32-bit offsets) large images with few channels: the
    // matrix-core kernel with the LDS-staged band
    plan->kernel = kPlanMfmaLds;
  } else {
    // Order of preference (same-box A/B over the MobileNetV2 layers: the LDS-tiled and the sliding-window
    // kernel are within +-10 % of each other, LDS ahead on the small late layers): LDS-tiled, then the
    // register sliding window (3x3 shapes whose band does not fit the LDS budget), then the direct kernel.
    const bool use_lds = a->variant != 1 && a->variant != 3 && (k33 || k55) && aligned4 && plan_lds(p, lds_budget());
    if (a->variant == 2 && !use_lds) return QNNP_HIP_EINVAL;
    if (!use_lds && (a->variant == 0 || a->variant == 3) && k33 && aligned4 && plan_row(p)) {
      plan->kernel = kPlanRow;
    } else if (a->variant == 3) {
      return QNNP_HIP_EINVAL;
    } else if (use_lds) {
      plan->vec16 = (p.CS % 16 == 0 && p.in_stride % 16 == 0 && in_addr % 16 == 0) ? 1u : 0u;
      plan->kernel = k33 ? kPlanLds33 : kPlanLds55;
    } else {
      // (round 6) 3x3 windows of any channel count >= 4 and any alignment: the sliding-window kernel on unaligned dwords
      // ("dwconv_kernel" 8 forces it -- on aligned tensors too --, 1 / 9 keep the generic kernels below)
      if (a->variant == 0 && k33 && a->c_pad % 4 == 0 && plan_row(p, true)) {
        plan->kernel = kPlanRowAny;
        plan->CS = p.CS; plan->TOH = p.TOH; plan->IR = p.IR; plan->IC = p.IC; plan->PP = p.PP;
        plan->bands = p.bands; plan->slabs = p.slabs; plan->store_mode = p.store_mode;
        return QNNP_HIP_OK;
      }
      // (round 6) four channels per thread where the tensors allow 32-bit offsets; "dwconv_kernel" 1 keeps the byte-per-thread kernel
      const uint64_t ib = static_cast<uint64_t>(p.batch) * p.H * p.W * p.in_stride;
      const uint64_t groups4 = static_cast<uint64_t>(p.batch) * p.OH * p.OW * ((p.C + 3u) / 4u);
      plan->kernel = (a->variant != 1 && p.C >= 4 && ib + 8 < (UINT64_C(1) << 31) && groups4 + 256u * 8192u < (UINT64_C(1) << 32) && a->c_pad % 4 == 0)
          ? kPlanDirect4 : kPlanDirect;
    }
  }
  plan->CS = p.CS; plan->TOH = p.TOH; plan->IR = p.IR; plan->IC = p.IC; plan->PP = p.PP;
  plan->bands = p.bands; plan->slabs = p.slabs; plan->store_mode = p.store_mode;
  return QNNP_HIP_OK;
}

}  // namespace

extern "C" int qnnp_hip_dwconv_run(const struct qnnp_hip_dwconv_args* a, const char** kernel_name)
{
  if (a == nullptr || a->batch == 0 || a->channels == 0) return QNNP_HIP_EINVAL;
  DwParams p;
  p.input = a->input;
  p.output = a->output;
  p.wadj = a->wadj;
  p.bias1 = a->bias1;
  p.batch = a->batch;
  p.H = a->input_height; p.W = a->input_width;
  p.OH = a->output_height; p.OW = a->output_width;
  p.C = a->channels; p.c_pad = a->c_pad;
  p.KH = a->kernel_height; p.KW = a->kernel_width;
  p.sh = a->stride_height; p.sw = a->stride_width;
  p.dh = a->dilation_height; p.dw = a->dilation_width;
  p.pad_top = a->pad_top; p.pad_left = a->pad_left;
  p.in_stride = a->input_stride; p.out_stride = a->output_stride;
  p.izp = a->input_zero_point & 0xFFu;
  p.CS = p.TOH = p.IR = p.IC = p.PP = p.bands = p.slabs = 0;
  p.rq = qnnp::make_requant_dev(a->rq);
  p.stream_out = a->streaming_mode == 0 ? (qnnp_hip_streaming_stores() != 0 ? 1u : 0u) : (a->streaming_mode == 2 ? 1u : 0u);
  p.trace = nullptr;
  p.abl = 0;
#ifdef QNNP_ENABLE_ABLATION
  p.trace = static_cast<unsigned long long*>(qnnp_hip_trace_buffer());
  if (const char* env = getenv("QNNP_DW_ABL")) p.abl = static_cast<uint32_t>(atoi(env));
#endif
  {
    const int cus = qnnp_hip_compute_units();
    p.cu_count = cus > 0 ? static_cast<uint32_t>(cus) : 256u;
  }

  hipStream_t stream = reinterpret_cast<hipStream_t>(qnnp_hip_get_stream());
  const uintptr_t in_addr = reinterpret_cast<uintptr_t>(a->input);
  const uintptr_t out_addr = reinterpret_cast<uintptr_t>(a->output);
  p.dwm_x = a->dwm_x; p.dwm_bias = a->dwm_bias; p.dwm_parts = a->dwm_parts; p.c_pad32 = a->c_pad32;
  p.wrange = a->w_range;
  p.dot4 = a->dot4;
  p.inv_bands = p.inv_slabs = p.inv_q4 = p.inv_dh = 0;
  p.xcd_ranges = 0;

  // The plan (kernel choice + band / slab geometry) depends on the shapes fixed at setup, the variant and the
  // pointers' alignment only: it is computed at the first run after a setup and kept with the operator.
  const uint32_t key = 0x80000000u | (static_cast<uint32_t>(in_addr & 15u)) | (static_cast<uint32_t>(out_addr & 15u) << 4) |
                       (static_cast<uint32_t>(a->variant & 0xFF) << 8) | ((a->batch & 0x7FFFu) << 16);
  struct qnnp_hip_dwconv_plan local_plan;
  struct qnnp_hip_dwconv_plan* plan = a->plan != nullptr ? a->plan : &local_plan;
  if (a->plan == nullptr || plan->key != key) {
    plan->key = 0;
    const int rc = make_plan(p, a, in_addr, out_addr, plan);
    if (rc != QNNP_HIP_OK) return rc;
    plan->key = key;
  } else {
    p.CS = plan->CS; p.TOH = plan->TOH; p.IR = plan->IR; p.IC = plan->IC; p.PP = plan->PP;
    p.bands = plan->bands; p.slabs = plan->slabs; p.store_mode = plan->store_mode;
  }
  switch (plan->kernel) {
    case kPlanMfmaLds:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_mfma_lds_3x3";
      return launch_mfma_lds(p, stream);
    case kPlanMfma:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_mfma_3x3";
      return launch_mfma<3, 3>(p, stream);
    case kPlanRow:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_row_3x3";
      return launch_row(p, stream);
    case kPlanRowAny:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_row_3x3_any";
      return launch_row(p, stream, true);
    case kPlanCol:
      if (kernel_name != nullptr) {
        *kernel_name = (p.dh != 1 || p.dw != 1) ? "q8_dwconv_col_3x3_dot4_dilated" : (col_uses_dot4(p) ? "q8_dwconv_col_3x3_dot4" : "q8_dwconv_col_3x3");
      }
      return launch_col(p, stream);
    case kPlanCol5:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_col_5x5_dot4";
      return launch_col5(p, stream);
    case kPlanM16:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_mfma16_3x3";
      return launch_m16(p, stream);
    case kPlanDirect4: {
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_direct4";
      const uint64_t total4 = static_cast<uint64_t>(p.batch) * p.OH * p.OW * ((p.C + 3u) / 4u);
      uint64_t blocks4 = (total4 + 255) / 256;
      if (blocks4 > 256u * 32u) blocks4 = 256u * 32u;
      hipLaunchKernelGGL(q8_dwconv_direct4_kernel, dim3(static_cast<uint32_t>(blocks4)), dim3(256), 0, stream, p);
      return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
    }
    case kPlanLds33:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_lds_3x3";
      return launch_lds<3, 3>(p, plan->vec16 != 0, stream);
    case kPlanLds55:
      if (kernel_name != nullptr) *kernel_name = "q8_dwconv_lds_5x5";
      return launch_lds<5, 5>(p, plan->vec16 != 0, stream);
    default:
      break;
  }
  if (kernel_name != nullptr) *kernel_name = "q8_dwconv_direct";
  const uint64_t total = static_cast<uint64_t>(p.batch) * p.OH * p.OW * p.C;
  uint64_t blocks = (total + 255) / 256;
  if (blocks > 256u * 16u) blocks = 256u * 16u;   // grid-stride beyond 16 workgroups per CU
  hipLaunchKernelGGL(q8_dwconv_direct_kernel, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, p);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}
