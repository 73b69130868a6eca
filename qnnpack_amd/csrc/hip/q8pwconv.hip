/*
 * q8pwconv.hip -- streaming MFMA kernel for pointwise (1x1, stride 1) convolutions and
 * fully-connected layers with a SHORT reduction (K <= 256) over MANY rows: the
 * MobileNet-style expand / project layers. Same arithmetic as q8igemm.hip (which see);
 * replaces the same reference path: qnnp_ukernel_type_gemm -> q8gemm 4x4c2 / 8x8
 * (src/operator-run.c:770-804, src/q8gemm/4x4c2-sse2.c:14-318).
 *
 * Why a separate kernel: these layers move 10-100x more bytes than they multiply
 * (K = 16..192 against N = 16..576), so they are bound by HBM and by the VALU work
 * of the requantization, not by the matrix cores. The tiled kernels stage
 * activations through LDS behind workgroup barriers, which serialises
 * load -> barrier -> multiply -> barrier -> requantize -> store per tile. Here
 *   - the whole weight matrix (<= 64 KiB of MFMA fragments) and the folded bias are
 *     copied into LDS ONCE per (persistent) workgroup;
 *   - each WAVE then walks 32-row blocks on its own, no barrier in the loop: a
 *     lane loads its 16 bytes of the activation row straight from global memory in
 *     MFMA B-operand layout (lane l = row l%32, K half l/32), the loads of the next
 *     block being in flight while the current one is multiplied and requantized;
 *   - per 32-channel block: K/32 MFMAs against fragments read from LDS, the fused
 *     Q31 epilogue of igemm_epilogue.hip.h, one 16-byte store per lane.
 * Rows are independent, so there is no exchange between waves at all.
 *
 * Zero-point algebra as in pack.h; row sums over the RAW bytes with v_sad_u8 (K
 * padding is loaded as 0x80 = a' of 0 from the fill table, so
 * sum(a') = sum(a) - 128 * 32 * KB).
 */
#include <hip/hip_runtime.h>

#include <stdint.h>
#include <stdlib.h>

#include "igemm_epilogue.hip.h"
#include "igemm_params.h"
#include "per_device.h"
#include "requant.hip.h"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr uint32_t kFlip = 0x80808080u;
constexpr uint32_t kMaxLds = 64 * 1024;     // weights + bias: two workgroups per CU

/* KB = 32-deep K blocks (k_total <= 32 * KB); VEC = bytes per activation load (16, or 8 when rows are
 * only 8-byte aligned, e.g. 24 channels) */
/* D2S: depth-to-space stores (igemm_params.h) -- a deconvolution whose kernel equals its stride is this pointwise
 * GEMM with stride_h*stride_w times the channels, each phase's block landing on its own output pixel. */
template <int KB, int VEC, bool STAGED, bool D2S = false>
__global__ __launch_bounds__(kThreads, (KB <= 5) ? 4 : 2)
void q8_pw_stream_mfma_kernel(const IgemmParams p)
{
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t row_in_block = lane & 31u;
  const uint32_t khalf = lane >> 5;

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;          // fragment blocks per channel block in the packed image

  // ---- once per workgroup: weight fragments (only the KB non-empty K blocks) and bias2 into LDS ----
  // LDS image: [nb][kb < KB] fragments of 1 KiB, then n_pad int32 of bias2
  {
    // LDS-DMA (no VGPR round trip, all of a wave's fragments in flight at once)
    const uint32_t frags = nblocks * KB;
    for (uint32_t f = wave; f < frags; f += kWaves) {
      const uint32_t nb = f / KB;
      const uint32_t kb = f - nb * KB;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (p.packed_w + (static_cast<uint64_t>(nb) * kblocks + kb) * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*) (lds + f * 1024), 16, 0, 0);
    }
    uint8_t* lds_bias = lds + frags * 1024;
    const uint32_t bias_chunks = p.n_pad / 4;             // 16-byte pieces; n_pad is a multiple of 32
    for (uint32_t c0 = wave * 64; c0 < bias_chunks; c0 += kThreads) {
      const uint32_t c = min(c0 + lane, bias_chunks - 1); // the tail lanes repeat the last piece (same bytes)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (reinterpret_cast<const uint8_t*>(p.bias2) + c * 16),
          (__attribute__((address_space(3))) void*) (lds_bias + c0 * 16), 16, 0, 0);
    }
    // (no wait here: the first row block's loads are issued first, so both latencies overlap)
  }
  const uint8_t* lds_w = lds + lane * 16;
  const int4* lds_bias4 = reinterpret_cast<const int4*>(lds + nblocks * KB * 1024);
  // per-wave image of a unit's 32 x n output block (store_mode 3 only; the launcher sizes it)
  uint8_t* stage = lds + nblocks * KB * 1024 + ((p.n_pad * 4u + 1023u) & ~1023u) + wave * (32u * p.n);

  // ---- which of this lane's 16-byte K pieces exist (only the last block can be short) ----
  const uint8_t* pad16 = p.fill_table + 0x80 * 16;        // 16 bytes of a' == 0
  const uint32_t k_last = (KB - 1) * 32 + khalf * 16;     // first K position of the lane's last piece
  // VEC 16: the piece is whole or absent. VEC 8: each 8-byte half is whole or absent.
  const bool last_lo_ok = k_last < p.k_total;
  const bool last_hi_ok = k_last + 8 < p.k_total;

  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t unit_stride = gridDim.x * kWaves;

  auto load_rows = [&](uint32_t unit, v4i (&a)[KB]) __attribute__((always_inline)) {
    uint32_t m = unit * 32u + row_in_block;
    if (m >= p.rows) m = p.rows - 1;                      // clamped rows are never stored
    const uint8_t* row = p.input + static_cast<uint64_t>(m) * p.input_stride + khalf * 16;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const uint8_t* src = row + kb * 32;
      if constexpr (VEC == 16) {
        if (kb == KB - 1) src = last_lo_ok ? src : pad16;
        a[kb] = *reinterpret_cast<const v4i*>(src);
      } else {
        const uint8_t* lo = src;
        const uint8_t* hi = src + 8;
        if (kb == KB - 1) {
          lo = last_lo_ok ? lo : pad16;
          hi = last_hi_ok ? hi : pad16;
        }
        const int2 vlo = *reinterpret_cast<const int2*>(lo);
        const int2 vhi = *reinterpret_cast<const int2*>(hi);
        a[kb] = v4i{vlo.x, vlo.y, vhi.x, vhi.y};
      }
    }
  };

  const uint32_t raw_to_centred = 128u * 32u * KB;

  uint32_t unit = blockIdx.x * kWaves + wave;
  v4i a_next[KB];
  if (unit < units) load_rows(unit, a_next);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // weights + bias are in LDS (and the first rows landed)
  __syncthreads();

  requant_dispatch(p.rq, [&](auto shift0, auto full) {
    for (; unit < units; unit += unit_stride) {
      v4i a[KB];
#pragma unroll
      for (int kb = 0; kb < KB; kb++) a[kb] = a_next[kb];
      if (unit + unit_stride < units) load_rows(unit + unit_stride, a_next);

      // row sum over the raw bytes, then recentre at 128
      uint32_t rs = 0;
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        rs = __builtin_amdgcn_sad_u8(a[kb].x, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].y, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].z, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].w, 0u, rs);
        a[kb].x ^= static_cast<int>(kFlip);
        a[kb].y ^= static_cast<int>(kFlip);
        a[kb].z ^= static_cast<int>(kFlip);
        a[kb].w ^= static_cast<int>(kFlip);
      }
      rs += __shfl_xor(rs, 32);                           // the other K half of the same row
      const int32_t rowterm = p.row_coeff * static_cast<int32_t>(rs - raw_to_centred);

      const uint32_t m = unit * 32u + row_in_block;
      uint8_t* out_row = p.output + static_cast<uint64_t>(m) * p.output_stride;
      bool row_ok = m < p.rows;
      if constexpr (D2S) {
        // input pixel (img, iy, ix) -> output pixel (img, iy*sh, ix*sw); the phase offset is added per channel block
        const uint32_t mm = row_ok ? m : 0u;
        const uint32_t rowi = mm / p.d2s_in_w;             // img * in_h + iy
        const uint32_t ix = mm - rowi * p.d2s_in_w;
        const uint64_t out_pixel = (static_cast<uint64_t>(rowi) * p.d2s_sh * p.d2s_in_w + ix) * p.d2s_sw;
        out_row = p.output + out_pixel * p.output_stride;
      }
#ifdef QNNP_ENABLE_ABLATION
      if (p.izp_fill & 1u) row_ok = false;                // measurement: no stores
#endif

      // accumulators start at bias + row term (the MFMA adds into them): two VALU adds per value saved
      int4 bias4[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[rg * 2 + khalf];
      auto multiply = [&](uint32_t nb, v16i& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          acc[rg * 4 + 0] = bias4[rg].x + rowterm;
          acc[rg * 4 + 1] = bias4[rg].y + rowterm;
          acc[rg * 4 + 2] = bias4[rg].z + rowterm;
          acc[rg * 4 + 3] = bias4[rg].w + rowterm;
        }
        if (nb + 1 < nblocks) {
#pragma unroll
          for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[(nb + 1) * 8 + rg * 2 + khalf];
        }
        const uint8_t* wf = lds_w + nb * (KB * 1024);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a[kb], acc, 0, 0, 0);
        }
      };

      // Stores. A direct 16-byte store per lane writes 32 rows x 32 bytes per instruction: the L2 sees
      // 32-byte partial-line writes, and that stream alone runs at 2.8 TB/s (55 us on the 16 -> 96 layer against
      // 34 us for the same bytes stored contiguously -- ablation). With DENSE rows (stride == channels) the
      // 32 x n block of a unit is one contiguous run of memory, so the wave drops its requantized 16-byte
      // pieces into a private LDS image of it (ds_write_b128 in place of the global store) and copies the image
      // out LINEARLY, 1 KiB per instruction: one ds_read_b128 + one global store per KiB extra.
      if constexpr (STAGED) {
        uint8_t* img = stage + row_in_block * p.n;
        for (uint32_t nb = 0; nb < nblocks; nb++) {
          v16i acc;
          multiply(nb, acc);
          // (every lane takes part in the half-wave exchange inside; n % 16 == 0: a lane's 16 channels exist or not)
          igemm_stage_tile<decltype(shift0)::value, decltype(full)::value, false, true>(
              acc, bias4, 0, img, nb * 32, khalf, p, nb * 32 + khalf * 16 < p.n);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
        const uint32_t rows_here = min(32u, p.rows - unit * 32u);
        const uint32_t bytes = rows_here * p.n;
        uint8_t* blk = p.output + static_cast<uint64_t>(unit) * 32u * p.n;
        for (uint32_t o = lane * 16; o < bytes; o += 1024) {
          *reinterpret_cast<uint4*>(blk + o) = *reinterpret_cast<const uint4*>(stage + o);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // read back before the next unit overwrites it
        continue;
      }
#ifdef QNNP_ENABLE_ABLATION
      if (p.izp_fill & 4u) {                              // measurement: CONTIGUOUS stores only (dense rows assumed)
        uint8_t* blk = p.output + static_cast<uint64_t>(unit) * 32u * p.output_stride;
        const uint32_t bytes = 32u * p.output_stride;
        for (uint32_t o = lane * 16; o + 16 <= bytes; o += 1024) {
          if (unit * 32u + 32u <= p.rows) *reinterpret_cast<uint4*>(blk + o) = make_uint4(a[0].x, a[0].y, a[0].z, a[0].w);
        }
        continue;
      }
#endif
      for (uint32_t nb = 0; nb < nblocks; nb++) {
        v16i acc;
#ifdef QNNP_ENABLE_ABLATION
        if (p.izp_fill & 2u) {                            // measurement: stores only (no multiply, no requantization)
          const uint32_t c = nb * 32 + khalf * 16;
          if (row_ok && c < p.n) *reinterpret_cast<uint4*>(out_row + c) = make_uint4(a[0].x, a[0].y, a[0].z, a[0].w);
          continue;
        }
#endif
        multiply(nb, acc);
        if constexpr (D2S) {
          const uint32_t phase = nb / p.d2s_nbpp;
          const uint32_t py = phase / p.d2s_sw;
          const uint32_t px = phase - py * p.d2s_sw;
          uint8_t* phase_row = out_row + (static_cast<uint64_t>(py) * (p.d2s_in_w * p.d2s_sw) + px) * p.output_stride;
          igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, true>(
              acc, bias4, 0, phase_row, (nb - phase * p.d2s_nbpp) * 32, khalf, row_ok, p);
          continue;
        }
        igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, true>(
            acc, bias4, 0, out_row, nb * 32, khalf, row_ok, p);
      }
    }
  });
}

/*
 * Second flavour for pointwise / fully-connected layers whose weights do NOT fit LDS (K up to ~1000 over few
 * rows: the late MobileNet layers, classifier heads): one WAVE = one 32-row x 32-channel output block, both
 * MFMA operands straight from global memory / L2 (activations: 16 B per lane in B-operand layout; weights: one
 * coalesced 1 KiB fragment), four K blocks of loads in flight ahead of the multiplies, no LDS, no barrier.
 * The tiled kernels launch 49-196 workgroups for these shapes on a 256-CU chip; here every 32x32 block is its
 * own wave (980-7840 waves). Operand traffic is L2-resident by construction (the whole problem is a few MB).
 */
constexpr int kGwUnroll = 4;

__global__ __launch_bounds__(kThreads, 4)
void q8_pw_stream_gw_kernel(const IgemmParams p)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t row_in_block = lane & 31u;
  const uint32_t khalf = lane >> 5;
  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t units = ((p.rows + 31u) / 32u) * nblocks;
  const uint32_t unit = blockIdx.x * kWaves + wave;        // channel block fastest: neighbours share the rows
  if (unit >= units) return;
  const uint32_t rb = unit / nblocks;
  const uint32_t nb = unit - rb * nblocks;

  uint32_t m = rb * 32u + row_in_block;
  const bool row_ok = m < p.rows;
  if (!row_ok) m = p.rows - 1;
  const uint8_t* row = p.input + static_cast<uint64_t>(m) * p.input_stride + khalf * 16;
  const uint8_t* pad16 = p.fill_table + 0x80 * 16;          // 16 bytes of a' == 0
  const int8_t* wf = p.packed_w + static_cast<uint64_t>(nb) * kblocks * 1024 + lane * 16;
  const uint32_t kbt = (p.k_total + 31u) / 32u;

  int4 bias4[4];
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    bias4[rg] = *reinterpret_cast<const int4*>(p.bias2 + nb * 32 + rg * 8 + khalf * 4);
  }

  v16i acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0;
  uint32_t rs = 0;
  for (uint32_t kb0 = 0; kb0 < kbt; kb0 += kGwUnroll) {
    v4i a[kGwUnroll], w[kGwUnroll];
#pragma unroll
    for (int u = 0; u < kGwUnroll; u++) {
      const uint32_t kb = kb0 + u;
      const bool have = kb < kbt && kb * 32 + khalf * 16 < p.k_total;     // k_total % 16 == 0: whole piece or none
      const uint8_t* src = have ? row + kb * 32 : pad16;
      a[u] = *reinterpret_cast<const v4i*>(src);
      const uint32_t kbc = kb < kbt ? kb : kbt - 1;                        // (clamped fragments meet a' == 0)
      w[u] = *reinterpret_cast<const v4i*>(wf + static_cast<uint64_t>(kbc) * 1024);
    }
#pragma unroll
    for (int u = 0; u < kGwUnroll; u++) {
      rs = __builtin_amdgcn_sad_u8(a[u].x, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].y, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].z, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].w, 0u, rs);
      a[u].x ^= static_cast<int>(kFlip);
      a[u].y ^= static_cast<int>(kFlip);
      a[u].z ^= static_cast<int>(kFlip);
      a[u].w ^= static_cast<int>(kFlip);
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[u], a[u], acc, 0, 0, 0);
    }
  }
  rs += __shfl_xor(rs, 32);                                   // the other K half of the same row
  // every piece a lane read is either data or 0x80 padding: kGwUnroll-rounded blocks, 16 bytes each
  const uint32_t pieces = ((kbt + kGwUnroll - 1) / kGwUnroll) * kGwUnroll * 2;
  const int32_t rowterm = p.row_coeff * static_cast<int32_t>(rs - 128u * 16u * pieces);
  uint8_t* out_row = p.output + static_cast<uint64_t>(rb * 32u + row_in_block) * p.output_stride;
  requant_dispatch(p.rq, [&](auto shift0, auto full) {
    igemm_store_tile<decltype(shift0)::value, decltype(full)::value>(
        acc, bias4, rowterm, out_row, nb * 32, khalf, row_ok, p);
  });
}

/*
 * Third flavour: convolutions over 3-channel images (network first layers, e.g. 3x3 stride 2, 3 -> 32). The
 * reduction is tiny (taps x 4-byte slots <= 64 bytes, pack.h "channel slots"), so the same barrier-free
 * scheme applies with an in-register gather: a lane fetches the (at most 8) taps of its K half with one
 * unaligned dword each -- addresses from the operator's offset table, padding taps and the slot's 4th byte
 * replaced in registers -- and multiplies against the LDS-resident weights. One wave = 32 consecutive
 * output pixels of the flattened (image, row, column) space.
 */
typedef uint32_t __attribute__((aligned(1))) pw_u32_unaligned;

__global__ __launch_bounds__(kThreads, 4)
void q8_conv_stream_c3_kernel(const IgemmParams p)
{
  constexpr int KB = 2;                                    // k_pad == 64: up to 16 taps of 4-byte slots
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t row_in_block = lane & 31u;
  const uint32_t khalf = lane >> 5;
  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  {
    const uint32_t frags = nblocks * KB;
    for (uint32_t f = wave; f < frags; f += kWaves) {
      const uint32_t nb = f / KB;
      const uint32_t kb = f - nb * KB;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (p.packed_w + (static_cast<uint64_t>(nb) * kblocks + kb) * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*) (lds + f * 1024), 16, 0, 0);
    }
    uint8_t* lds_bias = lds + frags * 1024;
    const uint32_t bias_chunks = p.n_pad / 4;
    for (uint32_t c0 = wave * 64; c0 < bias_chunks; c0 += kThreads) {
      const uint32_t c = min(c0 + lane, bias_chunks - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (reinterpret_cast<const uint8_t*>(p.bias2) + c * 16),
          (__attribute__((address_space(3))) void*) (lds_bias + c0 * 16), 16, 0, 0);
    }
    // (no wait here: the first unit's gather is issued first, so the latencies overlap)
  }
  const uint8_t* lds_w = lds + lane * 16;
  const int4* lds_bias4 = reinterpret_cast<const int4*>(lds + nblocks * KB * 1024);

  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t unit_stride = gridDim.x * kWaves;
  const uint32_t fillpix = p.izp_fill;                     // {izp, izp, izp, 0x80}: padding tap, 4th byte = K padding
  const uint32_t raw_to_centred = 128u * 32u * KB;

  // The gather is two dependent loads (offset-table entry, then the pixel). They are split so that the table
  // entries of unit u+2 and the pixels of unit u+1 are in flight while unit u is multiplied: no load waits on
  // another one issued in the same iteration.
  struct Taps { int32_t off[KB * 4]; const uint8_t* base; };
  auto load_offsets = [&](uint32_t unit, Taps& t) __attribute__((always_inline)) {
    uint32_t m = unit * 32u + row_in_block;
    if (m >= p.rows) m = p.rows - 1;                       // clamped rows are never stored
    const uint32_t img = m / p.rows_per_image;
    const uint32_t pix = m - img * p.rows_per_image;
    t.base = p.input + static_cast<uint64_t>(img) * p.image_stride;
    const int32_t* offs = p.offsets + static_cast<uint64_t>(pix) * p.ks;
#pragma unroll
    for (int i = 0; i < KB * 4; i++) {
      const uint32_t tap = (i >> 2) * 8 + khalf * 4 + (i & 3);     // kb*8 + khalf*4 + j
      t.off[i] = tap < p.ks ? offs[tap] : -2;                      // -2: beyond the kernel (K padding), -1: padding tap
    }
  };
  auto load_pixels = [&](const Taps& t, v4i (&a)[KB]) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int32_t off = t.off[kb * 4 + j];
        const bool inside = off >= 0;
        const uint8_t* src = t.base + (inside ? off : 0);
        uint32_t x;
        if (src + 4 <= p.input_end) {
          x = *reinterpret_cast<const pw_u32_unaligned*>(src);
        } else {                                           // the very last pixel of the tensor: bytewise
          x = static_cast<uint32_t>(src[0]) | (static_cast<uint32_t>(src[1]) << 8) | (static_cast<uint32_t>(src[2]) << 16);
        }
        x = (x & 0x00FFFFFFu) | 0x80000000u;               // the slot's 4th byte is K padding: a' == 0
        v[j] = inside ? x : (off == -1 ? fillpix : 0x80808080u);
      }
      a[kb] = v4i{static_cast<int>(v[0]), static_cast<int>(v[1]), static_cast<int>(v[2]), static_cast<int>(v[3])};
    }
  };

  uint32_t unit = blockIdx.x * kWaves + wave;
  v4i a_next[KB];
  Taps t_next;
  if (unit < units) {
    load_offsets(unit, t_next);
    load_pixels(t_next, a_next);
    if (unit + unit_stride < units) load_offsets(unit + unit_stride, t_next);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // weights + bias are in LDS
  __syncthreads();

  requant_dispatch(p.rq, [&](auto shift0, auto full) {
    for (; unit < units; unit += unit_stride) {
      v4i a[KB];
#pragma unroll
      for (int kb = 0; kb < KB; kb++) a[kb] = a_next[kb];
      if (unit + unit_stride < units) {
        load_pixels(t_next, a_next);                       // offsets landed one iteration ago
        if (unit + 2 * unit_stride < units) load_offsets(unit + 2 * unit_stride, t_next);
      }

      uint32_t rs = 0;
#pragma unroll
      for (int kb = 0; kb < KB; kb++) {
        rs = __builtin_amdgcn_sad_u8(a[kb].x, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].y, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].z, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].w, 0u, rs);
        a[kb].x ^= static_cast<int>(kFlip);
        a[kb].y ^= static_cast<int>(kFlip);
        a[kb].z ^= static_cast<int>(kFlip);
        a[kb].w ^= static_cast<int>(kFlip);
      }
      rs += __shfl_xor(rs, 32);
      const int32_t rowterm = p.row_coeff * static_cast<int32_t>(rs - raw_to_centred);

      const uint32_t m = unit * 32u + row_in_block;
      uint8_t* out_row = p.output + static_cast<uint64_t>(m) * p.output_stride;
      const bool row_ok = m < p.rows;
      int4 bias4[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[rg * 2 + khalf];
      for (uint32_t nb = 0; nb < nblocks; nb++) {
        v16i acc;
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          acc[rg * 4 + 0] = bias4[rg].x + rowterm;
          acc[rg * 4 + 1] = bias4[rg].y + rowterm;
          acc[rg * 4 + 2] = bias4[rg].z + rowterm;
          acc[rg * 4 + 3] = bias4[rg].w + rowterm;
        }
        if (nb + 1 < nblocks) {
#pragma unroll
          for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[(nb + 1) * 8 + rg * 2 + khalf];
        }
        const uint8_t* wf = lds_w + nb * (KB * 1024);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a[kb], acc, 0, 0, 0);
        }
        igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, true>(
            acc, bias4, 0, out_row, nb * 32, khalf, row_ok, p);
      }
    }
  });
}

template <int KB, int VEC, bool STAGED, bool D2S = false>
int launch_pw(const IgemmParams& p, uint32_t lds_bytes, hipStream_t stream)
{
  static int blocks_per_cu = 0;       // per instantiation; benign race (same value)
  auto kernel = q8_pw_stream_mfma_kernel<KB, VEC, STAGED, D2S>;
  if (blocks_per_cu == 0) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    blocks_per_cu = (KB <= 5) ? 4 : 2;
  }
  // LDS bounds the residency: kMaxLds -> 2 per CU, half of that -> 4
  uint32_t per_cu = static_cast<uint32_t>(blocks_per_cu);
  const uint32_t by_lds = lds_bytes > 0 ? (160u * 1024u) / lds_bytes : per_cu;
  if (by_lds < per_cu) per_cu = by_lds > 0 ? by_lds : 1u;
  const uint32_t units = (p.rows + 31u) / 32u;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_PW_BLOCKS")) per_cu = static_cast<uint32_t>(atoi(env));
#endif
  uint32_t grid = p.cu_count * per_cu;
  const uint32_t needed = (units + kWaves - 1) / kWaves;
  if (grid > needed) grid = needed;
#ifdef QNNP_ENABLE_ABLATION
  IgemmParams pa = p;
  pa.izp_fill = 0;
  if (const char* env = getenv("QNNP_PW_ABL")) pa.izp_fill = static_cast<uint32_t>(atoi(env));
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds_bytes, stream, pa);
#else
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), lds_bytes, stream, p);
#endif
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int VEC, bool STAGED>
int dispatch_kb(const IgemmParams& p, uint32_t kb, uint32_t lds_bytes, hipStream_t stream)
{
  switch (kb) {
    case 1: return launch_pw<1, VEC, STAGED>(p, lds_bytes, stream);
    case 2: return launch_pw<2, VEC, STAGED>(p, lds_bytes, stream);
    case 3: return launch_pw<3, VEC, STAGED>(p, lds_bytes, stream);
    case 4: return launch_pw<4, VEC, STAGED>(p, lds_bytes, stream);
    case 5: return launch_pw<5, VEC, STAGED>(p, lds_bytes, stream);
    case 6: return launch_pw<6, VEC, STAGED>(p, lds_bytes, stream);
    case 7: return launch_pw<7, VEC, STAGED>(p, lds_bytes, stream);
    default: return launch_pw<8, VEC, STAGED>(p, lds_bytes, stream);
  }
}

int dispatch_kb_d2s(const IgemmParams& p, uint32_t kb, uint32_t lds_bytes, hipStream_t stream)
{
  switch (kb) {
    case 1: return launch_pw<1, 16, false, true>(p, lds_bytes, stream);
    case 2: return launch_pw<2, 16, false, true>(p, lds_bytes, stream);
    case 3: return launch_pw<3, 16, false, true>(p, lds_bytes, stream);
    case 4: return launch_pw<4, 16, false, true>(p, lds_bytes, stream);
    case 5: return launch_pw<5, 16, false, true>(p, lds_bytes, stream);
    case 6: return launch_pw<6, 16, false, true>(p, lds_bytes, stream);
    case 7: return launch_pw<7, 16, false, true>(p, lds_bytes, stream);
    default: return launch_pw<8, 16, false, true>(p, lds_bytes, stream);
  }
}

uint32_t pw_lds_bytes(const IgemmParams& p)
{
  const uint32_t kb = (p.k_total + 31u) / 32u;
  // the bias region is rounded up to whole 1 KiB LDS-DMA rows (a wave-instruction always writes 64 x 16 B)
  return (p.n_pad / 32u) * kb * 1024u + ((p.n_pad * 4u + 1023u) & ~1023u);
}

}  // namespace

/* pointwise / fully-connected form only (no offset table), one group, K <= 256, weights + bias <= 64 KiB */
bool pwstream_supported(const IgemmParams& p, uint32_t groups, uint32_t vec)
{
  if (p.offsets != nullptr || groups != 1 || (vec != 16 && vec != 8)) return false;
  if (p.fill_table == nullptr || p.rows == 0 || p.k_total == 0 || p.k_total > 256u) return false;
  if (p.k_total % vec != 0) return false;
  return pw_lds_bytes(p) <= kMaxLds;
}

/* 3-channel-image flavour: offset-table convolution in 4-byte tap slots, at most 16 taps, one group */
bool convstream_c3_supported(const IgemmParams& p, uint32_t groups)
{
  if (p.offsets == nullptr || groups != 1 || p.kc != 4 || p.ks == 0 || p.ks > 16 || p.k_pad != 64) return false;
  if (p.rows == 0 || p.rows_per_image == 0) return false;
  return (p.n_pad / 32u) * 2u * 1024u + ((p.n_pad * 4u + 1023u) & ~1023u) <= kMaxLds;
}

int convstream_c3_launch(const IgemmParams& p, hipStream_t stream, const char** name)
{
  const uint32_t lds_bytes = (p.n_pad / 32u) * 2u * 1024u + ((p.n_pad * 4u + 1023u) & ~1023u);
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (attr_once.first()) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(q8_conv_stream_c3_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
  }
  const uint32_t units = (p.rows + 31u) / 32u;
  uint32_t grid = p.cu_count * 4u;
  const uint32_t needed = (units + kWaves - 1) / kWaves;
  if (grid > needed) grid = needed;
  *name = "q8_conv_stream_c3_mfma";
  hipLaunchKernelGGL(q8_conv_stream_c3_kernel, dim3(grid), dim3(kThreads), lds_bytes, stream, p);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

/* global-weights flavour: pointwise / fully-connected form, one group, 16-byte aligned rows, any K and N */
bool pwstream_gw_supported(const IgemmParams& p, uint32_t groups, uint32_t vec)
{
  if (p.offsets != nullptr || groups != 1 || vec != 16) return false;
  if (p.fill_table == nullptr || p.rows == 0 || p.k_total == 0 || p.k_total % 16 != 0) return false;
  const uint64_t units = static_cast<uint64_t>((p.rows + 31u) / 32u) * (p.n_pad / 32u);
  return units < (UINT64_C(1) << 31);
}

int pwstream_gw_launch(const IgemmParams& p, hipStream_t stream, const char** name)
{
  const uint32_t units = ((p.rows + 31u) / 32u) * (p.n_pad / 32u);
  *name = "q8_pw_stream_gw_mfma";
  hipLaunchKernelGGL(q8_pw_stream_gw_kernel, dim3((units + kWaves - 1) / kWaves), dim3(kThreads), 0, stream, p);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int pwstream_launch(const IgemmParams& p0, uint32_t vec, hipStream_t stream, const char** name)
{
  IgemmParams p = p0;
  const uint32_t kb = (p.k_total + 31u) / 32u;
  uint32_t lds_bytes = pw_lds_bytes(p);
  // dense 16-byte-aligned rows wider than one 32-channel block: stage the unit's output block per wave and
  // store it contiguously ("store_mode 3", private to this kernel), if the images fit beside the weights
  const uint32_t stage_bytes = kWaves * 32u * p.n;
  if (p.store_mode == 2 && p.output_stride == p.n && p.n > 32 && p.n <= 256 && lds_bytes + stage_bytes <= kMaxLds) {
    p.store_mode = 3;
    lds_bytes += stage_bytes;
  }
  if (p.d2s_sh != 0) {
    if (vec != 16) return QNNP_HIP_EINVAL;
    if (p.store_mode == 3) {
      p.store_mode = 2;
      lds_bytes -= stage_bytes;
    }
    *name = "q8_pw_stream_d2s_mfma";
    return dispatch_kb_d2s(p, kb, lds_bytes, stream);
  }
  *name = "q8_pw_stream_mfma";
  if (p.store_mode == 3) {
    return vec == 16 ? dispatch_kb<16, true>(p, kb, lds_bytes, stream) : dispatch_kb<8, true>(p, kb, lds_bytes, stream);
  }
  return vec == 16 ? dispatch_kb<16, false>(p, kb, lds_bytes, stream) : dispatch_kb<8, false>(p, kb, lds_bytes, stream);
}

}  // namespace qnnp
