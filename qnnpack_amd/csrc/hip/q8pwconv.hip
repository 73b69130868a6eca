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

/* A store of whole lines that are written exactly once (a wave's contiguous run of an output block) carries the
 * streaming hint: layer 4 (112x112x16 -> 96, 154 MB of output) 39 -> 30.5 us, layer 7 17.6 -> 13.9, same box. The hint
 * on anything that is touched again -- activation loads with column overlaps or re-read per channel column, 4-byte
 * stores that complete a line over several instructions -- costs 25-120 %.
 * `streaming` = IgemmParams::stream_out (wave-uniform): callers that chain operators turn the hint off -- the consumer of
 * a streamed tensor finds none of it in the last-level cache (whole network 214 k -> 212 k images/s with it, the
 * per-layer sweep 300 k -> 313 k). */
__device__ __forceinline__ void store16_once(uint8_t* dst, const uint4& v, uint32_t streaming)
{
  typedef int nt_v4i __attribute__((ext_vector_type(4)));
  const nt_v4i x = {static_cast<int>(v.x), static_cast<int>(v.y), static_cast<int>(v.z), static_cast<int>(v.w)};
  if (streaming) {
    // (as an instruction: with the builtin, hipcc hoists / sinks the two stores of this branch into one and drops the hint)
    asm volatile("global_store_dwordx4 %0, %1, off nt" : : "v"(dst), "v"(x) : "memory");
  } else {
    *reinterpret_cast<nt_v4i*>(dst) = x;
  }
}

constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr uint32_t kFlip = 0x80808080u;
constexpr uint32_t kMaxLds = 64 * 1024;     // weights + bias: two workgroups per CU

/* KB = 32-deep K blocks (k_total <= 32 * KB); VEC = bytes per activation load (16, or 8 when rows are
 * only 8-byte aligned, e.g. 24 channels) */
/* D2S: depth-to-space stores (igemm_params.h) -- a deconvolution whose kernel equals its stride is this pointwise
 * GEMM with stride_h*stride_w times the channels, each phase's block landing on its own output pixel. */
/* RES: the residual add of igemm_params.h rides in the epilogue -- a kernel of its own, so that layers without one
 * run exactly the code they ran before it existed (as a wave-uniform run-time branch it cost them 2-6 %, same box:
 * 28x28x144 -> 32 6.7 -> 7.1 us, 56x56x144 -> 24 15.8 -> 16.4) */
template <int KB, int VEC, bool D2S = false, bool RES = false>
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
    // (where the lane forms of the requantization apply -- requant_dispatch_lane below -- the accumulators start from
    //  bias + 2^31, the second half of the pair table)
    const bool lane_rq = p.bias2u != nullptr && (p.lane.kind == 1 || (p.lane.kind == 2 && p.rq.full_range != 0));
    const int32_t* bias_src = lane_rq ? p.bias2u : p.bias2;
    for (uint32_t c0 = wave * 64; c0 < bias_chunks; c0 += kThreads) {
      const uint32_t c = min(c0 + lane, bias_chunks - 1); // the tail lanes repeat the last piece (same bytes)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (reinterpret_cast<const uint8_t*>(bias_src) + c * 16),
          (__attribute__((address_space(3))) void*) (lds_bias + c0 * 16), 16, 0, 0);
    }
    // (no wait here: the first row block's loads are issued first, so both latencies overlap)
  }
  const uint8_t* lds_w = lds + lane * 16;
  const int4* lds_bias4 = reinterpret_cast<const int4*>(lds + nblocks * KB * 1024);

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

  requant_dispatch_lane(p.rq, p.lane, [&](auto shift0, auto full) {
    constexpr int kSeq = decltype(shift0)::value;
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
      // (+ 2^31 for the offset rounding sequences, requant.hip.h: the accumulators start from bias + this)
      const int32_t rowterm = with_rq_offset<decltype(shift0)::value>(p.row_coeff * static_cast<int32_t>(rs - raw_to_centred));
      uint64_t row_addend = 0;                            // lane forms: the row term as the multiply-add's addend
      if constexpr (rq_is_lane<kSeq>()) row_addend = lane_addend(rowterm, p.lane);

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
      auto multiply = [&](uint32_t nb, v16i& acc) __attribute__((always_inline)) {
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          acc[rg * 4 + 0] = bias4[rg].x; acc[rg * 4 + 1] = bias4[rg].y;
          acc[rg * 4 + 2] = bias4[rg].z; acc[rg * 4 + 3] = bias4[rg].w;
        }
        const uint8_t* wf = lds_w + nb * (KB * 1024);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a[kb], acc, 0, 0, 0);
        }
      };

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
        // (read per block, beside the weight fragment: carried over from the previous trip it cost 16 v_mov_b64 per block)
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[nb * 8 + rg * 2 + khalf];
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
          if constexpr (rq_is_lane<kSeq>()) {
            igemm_store_tile_lane<kSeq, decltype(full)::value>(
                acc, row_addend, phase_row, (nb - phase * p.d2s_nbpp) * 32, khalf, row_ok, p);
          } else {
            igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 2>(
                acc, bias4, rowterm, phase_row, (nb - phase * p.d2s_nbpp) * 32, khalf, row_ok, p);
          }
          continue;
        }
        if constexpr (RES) {                              // project layer with its residual add folded in
          const uint8_t* res_row = p.residual + static_cast<uint64_t>(row_ok ? m : 0u) * p.residual_stride;
          if constexpr (rq_is_lane<kSeq>()) {
            igemm_store_tile_lane<kSeq, decltype(full)::value, true>(acc, row_addend, out_row, nb * 32, khalf, row_ok, p, res_row, &p.add);
          } else {
            igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 2, true>(
                acc, bias4, rowterm, out_row, nb * 32, khalf, row_ok, p, res_row, &p.add);
          }
          continue;
        }
        if constexpr (rq_is_lane<kSeq>()) {
          igemm_store_tile_lane<kSeq, decltype(full)::value>(acc, row_addend, out_row, nb * 32, khalf, row_ok, p);
        } else {
          igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 2>(
              acc, bias4, rowterm, out_row, nb * 32, khalf, row_ok, p);
        }
      }
    }
  });
}

/* A wave's staged 32-row output block (LDS image `stage`) -> global memory. Dense rows and the whole N: the block
 * is one contiguous run, 1 KiB per store instruction. Otherwise row chunks: the image has 2^log_cpr 16-byte pieces
 * per row (pitch 16 << log_cpr), of which the first cw bytes exist; consecutive lanes take consecutive pieces of a
 * row. c0 = first channel of the chunk. */
/* dense8 (round 4): dense rows whose length is a multiple of 8 but not of 16 bytes (24-channel MobileNet layers): the
 * block is still ONE contiguous, 16-byte aligned run in memory (32 rows x n bytes), only the image keeps its 16-byte
 * pitch; a lane gathers its 16 output bytes as two 8-byte halves, which may sit in two image rows. Those layers took the
 * direct-store kernel before: 4-byte stores, 3.9-4.3 TB/s where their 16-channel neighbours stream 4.9. */
template <bool RES = false>
__device__ __forceinline__ void stream_copy_out(
    const uint8_t* stage, bool whole_dense, uint32_t log_cpr, uint32_t unit, uint32_t c0, uint32_t cw,
    const IgemmParams& p, uint32_t lane, bool dense8 = false)
{
  const uint32_t rows_here = min(32u, p.rows - unit * 32u);
  if (dense8) {
    const uint32_t bytes = rows_here * p.n;                  // a multiple of 8
    const uint64_t block_ofs = static_cast<uint64_t>(unit) * 32u * p.n;
    uint8_t* blk = p.output + block_ofs;
    const uint32_t pitch = 16u << log_cpr;
    for (uint32_t o = lane * 16; o < bytes; o += 1024) {
      const uint32_t r = __umulhi(o, p.tiles_n_magic);       // o / n (launcher: magic32(n); o < 2^13)
      const uint32_t c = o - r * p.n;                        // multiple of 8
      const uint2 lo = *reinterpret_cast<const uint2*>(stage + r * pitch + c);
      const bool wrap = c + 8 >= p.n;                        // the second half starts the next row
      const uint2 hi = *reinterpret_cast<const uint2*>(stage + (wrap ? (r + 1) * pitch : r * pitch + c + 8));
      uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
      if constexpr (RES) {
        const uint8_t* res = p.residual + block_ofs + o;
        if (o + 16 <= bytes) {
          const uint4 q = *reinterpret_cast<const uint4*>(res);
          v = make_uint4(add_quantize4(q.x, v.x, p.add), add_quantize4(q.y, v.y, p.add), add_quantize4(q.z, v.z, p.add), add_quantize4(q.w, v.w, p.add));
        } else {
          const uint2 q = *reinterpret_cast<const uint2*>(res);
          v.x = add_quantize4(q.x, v.x, p.add);
          v.y = add_quantize4(q.y, v.y, p.add);
        }
      }
      if (o + 16 <= bytes) store16_once(blk + o, v, p.stream_out);
      else *reinterpret_cast<uint2*>(blk + o) = make_uint2(v.x, v.y);       // (an odd number of rows: the last 8 bytes)
    }
    return;
  }
  if constexpr (RES) {
    // fused residual add: the same walk, each 16-byte piece summed with the residual's bytes of the same pixel and
    // channels (launcher: residual rows are laid out like the output rows, 16-byte aligned)
    const uint64_t block_ofs = static_cast<uint64_t>(unit) * 32u * p.output_stride + c0;
    const uint8_t* res = p.residual + block_ofs;
    uint8_t* blk = p.output + block_ofs;
    if (whole_dense) {
      const uint32_t bytes = rows_here * p.n;
      for (uint32_t o = lane * 16; o < bytes; o += 1024) {
        const uint4 r = *reinterpret_cast<const uint4*>(res + o);
        const uint4 v = *reinterpret_cast<const uint4*>(stage + o);
        store16_once(blk + o, make_uint4(add_quantize4(r.x, v.x, p.add), add_quantize4(r.y, v.y, p.add),
                                         add_quantize4(r.z, v.z, p.add), add_quantize4(r.w, v.w, p.add)), p.stream_out);
      }
    } else {
      const uint32_t pieces = 32u << log_cpr;
      for (uint32_t q = lane; q < pieces; q += 64) {
        const uint32_t rr = q >> log_cpr;
        const uint32_t cb = (q - (rr << log_cpr)) * 16u;
        if (rr < rows_here && cb < cw) {
          const uint64_t o = static_cast<uint64_t>(rr) * p.output_stride + cb;
          const uint4 r = *reinterpret_cast<const uint4*>(res + o);
          const uint4 v = *reinterpret_cast<const uint4*>(stage + q * 16u);
          store16_once(blk + o, make_uint4(add_quantize4(r.x, v.x, p.add), add_quantize4(r.y, v.y, p.add),
                                           add_quantize4(r.z, v.z, p.add), add_quantize4(r.w, v.w, p.add)), p.stream_out);
        }
      }
    }
    return;
  }
  if (whole_dense) {
    const uint32_t bytes = rows_here * p.n;
    uint8_t* blk = p.output + static_cast<uint64_t>(unit) * 32u * p.n;
    for (uint32_t o = lane * 16; o < bytes; o += 2048) {
      const uint4 v0 = *reinterpret_cast<const uint4*>(stage + o);
      const bool two = o + 1024 < bytes;
      const uint4 v1 = *reinterpret_cast<const uint4*>(stage + (two ? o + 1024 : o));
      store16_once(blk + o, v0, p.stream_out);
      if (two) store16_once(blk + o + 1024, v1, p.stream_out);
    }
  } else {
    const uint32_t pieces = 32u << log_cpr;
    uint8_t* blk = p.output + static_cast<uint64_t>(unit) * 32u * p.output_stride + c0;
    for (uint32_t q = lane; q < pieces; q += 64) {
      const uint32_t r = q >> log_cpr;
      const uint32_t cb = (q - (r << log_cpr)) * 16u;
      const uint4 v = *reinterpret_cast<const uint4*>(stage + q * 16u);
      if (r < rows_here && cb < cw) {
        store16_once(blk + static_cast<uint64_t>(r) * p.output_stride + cb, v, p.stream_out);
      }
    }
  }
}

/*
 * Staged flavour: 16-byte aligned rows, n % 16 == 0 -- every MobileNet-style expand / project layer. Differences to
 * the kernel above:
 *   - N split: gridDim.y workgroup columns each own `nbp` of the 32-channel blocks (their weights only in LDS), so a
 *     layer with few row blocks and many channels (14x14x96 -> 576: 784 row blocks) still fills the chip, and any N
 *     fits (the whole-matrix kernel stops at 64 KiB of weights);
 *   - a unit's requantized 32 x (nbp*32) block goes to a per-wave LDS image and leaves in row chunks of >= 64
 *     contiguous bytes (whole contiguous 32-row blocks when rows are dense and N is not split): direct 16-byte
 *     stores put 32-byte partial-line writes on the L2 (2.8 TB/s against 5+ for whole lines, ablation);
 *   - the wave's instruction order is  loads(u+1) -> multiply / requantize(u) -> wait -> re-centre(u+1) ->
 *     stores(u): the only vmcnt wait of the loop sits BEFORE the unit's stores, where everything outstanding
 *     (the next rows, the previous unit's stores) has had a whole multiply phase to complete. hipcc counts loads
 *     and stores in one counter and cannot count a run-time number of stores, so with the stores first it waited
 *     for the stores it had just issued (vmcnt(0) at the loop end) -- a full store round trip per unit per wave.
 *   - the loads are always issued (clamped unit index): a branch around them makes the outstanding count
 *     path-dependent and costs the same vmcnt(0).
 */
/* resident workgroups per CU (= waves per SIMD) the staged kernel is compiled for, i.e. what its register need
 * allows (54 / 66 / 74 / 79 / 88 / 98 / 107 / 118 VGPRs for 1..8 K blocks, per requantization flavour, since the bias
 * is no longer carried from block to block; 72 ... 134 before -- one more resident wave per SIMD, which the
 * registers would now allow, measured level or slightly behind: layer 3 17.2 -> 17.7 us, layer 21 9.9 -> 10.7). A wave of
 * this kernel spends most of a unit's time waiting -- the next rows, the previous unit's store acknowledgements --
 * and residency is what hides that. */
constexpr int staged_waves(int kb) { return kb <= 1 ? 7 : (kb == 2 ? 6 : (kb <= 4 ? 5 : (kb <= 7 ? 4 : 3))); }

/* SEQ / FULL: the requantization flavour (requant.hip.h), chosen on the host -- one kernel per flavour, so that the
 * common ones are not charged the registers of the rare ones */
template <int KB, int VEC, int SEQ, bool FULL, bool RES = false>
__global__ __launch_bounds__(kThreads, staged_waves(KB))
void q8_pw_stream_staged_kernel(const IgemmParams p, const uint32_t nbp, const uint32_t log_cpr_arg)
{
  const uint32_t log_cpr = log_cpr_arg & 31u;
  const bool dense8 = (log_cpr_arg >> 31) != 0u;           // stream_copy_out's third mode
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t row_in_block = lane & 31u;
  const uint32_t khalf = lane >> 5;

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t nb0 = blockIdx.y * nbp;                  // this workgroup column's channel blocks
  const uint32_t nbn = min(nbp, nblocks - nb0);
  const uint32_t c0 = nb0 * 32u;                          // first channel
  const uint32_t cw = min(p.n - c0, nbn * 32u);           // valid bytes per row of the chunk (n % 16 == 0, or dense8)
  const bool whole_dense = !dense8 && gridDim.y == 1 && p.output_stride == p.n;
  const uint32_t pitch = whole_dense ? p.n : (16u << log_cpr);   // row pitch of the image

  // ---- once per workgroup: the column's weight fragments and bias2 into LDS (LDS-DMA) ----
  {
    const uint32_t frags = nbn * KB;
    for (uint32_t f = wave; f < frags; f += kWaves) {
      const uint32_t nb = f / KB;
      const uint32_t kb = f - nb * KB;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (p.packed_w + (static_cast<uint64_t>(nb0 + nb) * kblocks + kb) * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*) (lds + f * 1024), 16, 0, 0);
    }
    uint8_t* lds_bias = lds + nbp * KB * 1024;
    const uint32_t bias_chunks = nbn * 8u;                // 16-byte pieces
    // (lane forms of the requantization: the accumulators start from bias + 2^31, the second half of the pair table)
    const int32_t* bias_src = rq_is_lane<SEQ>() ? p.bias2u : p.bias2;
    for (uint32_t q0 = wave * 64; q0 < bias_chunks; q0 += kThreads) {
      const uint32_t q = min(q0 + lane, bias_chunks - 1); // the tail lanes repeat the last piece (same bytes)
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (reinterpret_cast<const uint8_t*>(bias_src + c0) + q * 16),
          (__attribute__((address_space(3))) void*) (lds_bias + q0 * 16), 16, 0, 0);
    }
  }
  const uint8_t* lds_w = lds + lane * 16;
  const uint32_t bias_bytes = (nbp * 128u + 1023u) & ~1023u;
  const int4* lds_bias4 = reinterpret_cast<const int4*>(lds + nbp * KB * 1024);
  uint8_t* stage = lds + nbp * KB * 1024 + bias_bytes + wave * (32u * pitch);

  const uint8_t* pad16 = p.fill_table + 0x80 * 16;        // 16 bytes of a' == 0
  const uint32_t k_last = (KB - 1) * 32 + khalf * 16;
  const bool last_lo_ok = k_last < p.k_total;
  const bool last_hi_ok = k_last + 8 < p.k_total;

  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t unit_stride = gridDim.x * kWaves;

  auto load_rows = [&](uint32_t unit, v4i (&a)[KB]) __attribute__((always_inline)) {
    uint32_t m = unit * 32u + row_in_block;
    if (m >= p.rows) m = p.rows - 1;                      // clamped rows are never stored
    const uint8_t* row = p.input + static_cast<uint64_t>(m) * p.input_stride + khalf * 16;
    if (p.offsets_dense != 0) {                           // strided 1x1 convolution (wave-uniform): the row's table entry
      const uint32_t img = p.rpi_magic != 0 ? __umulhi(m, p.rpi_magic) : m / p.rows_per_image;
      const uint32_t pix = m - img * p.rows_per_image;
      row = p.input + static_cast<uint64_t>(img) * p.image_stride + static_cast<uint32_t>(p.offsets[pix]) + khalf * 16;
    }
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
  // raw bytes -> row sum, then re-centred at 128 in place
  auto recentre = [&](v4i (&a)[KB]) __attribute__((always_inline)) -> uint32_t {
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
    rs += __shfl_xor(rs, 32);                             // the other K half of the same row
    return rs;
  };
  const uint32_t raw_to_centred = 128u * 32u * KB;

  uint32_t unit = blockIdx.x * kWaves + wave;
  v4i a[KB];
  load_rows(min(unit, units - 1u), a);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // weights + bias are in LDS (and the first rows landed)
  __syncthreads();
  if (unit >= units) return;

  {
    using shift0_t = std::integral_constant<int, SEQ>;
    using full_t = std::integral_constant<bool, FULL>;
    const shift0_t shift0{};
    const full_t full{};
    (void) shift0; (void) full;
    uint32_t rs = recentre(a);
    for (;;) {
      const uint32_t next = unit + unit_stride;
      v4i raw[KB];
      load_rows(min(next, units - 1u), raw);              // always issued (the last one of a wave is wasted)
      // (+ 2^31 for the offset rounding sequences, requant.hip.h: the accumulators start from bias + this)
      const int32_t rowterm = with_rq_offset<decltype(shift0)::value>(p.row_coeff * static_cast<int32_t>(rs - raw_to_centred));
      uint64_t row_addend = 0;                              // lane forms: the row term as the multiply-add's addend
      if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
      uint8_t* img = stage + row_in_block * pitch;
      for (uint32_t nb = 0; nb < nbn; nb++) {
        v16i acc;
        // (read per block, beside the weight fragment: carried over from the previous trip it cost 16 v_mov_b64 per block)
        int4 bias4[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[nb * 8 + rg * 2 + khalf];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          acc[rg * 4 + 0] = bias4[rg].x; acc[rg * 4 + 1] = bias4[rg].y;
          acc[rg * 4 + 2] = bias4[rg].z; acc[rg * 4 + 3] = bias4[rg].w;
        }
        const uint8_t* wf = lds_w + nb * (KB * 1024);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a[kb], acc, 0, 0, 0);
        }
        // (every lane takes part in the half-wave exchange inside; a lane's 16 channels exist or not)
        if constexpr (rq_is_lane<SEQ>()) {
          igemm_stage_tile_lane<SEQ, FULL>(acc, row_addend, img, nb * 32, khalf, p, nb * 32 + khalf * 16 < cw);
        } else {
          igemm_stage_tile<decltype(shift0)::value, decltype(full)::value, false, 2>(
              acc, bias4, rowterm, img, nb * 32, khalf, p, nb * 32 + khalf * 16 < cw);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
      // the next unit's rows: first use here, so the wait for them lands before this unit's stores
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kb = 0; kb < KB; kb++) a[kb] = raw[kb];
      rs = recentre(a);
      __builtin_amdgcn_sched_barrier(0);
#ifdef QNNP_ENABLE_ABLATION
      if (p.izp_fill & 1u) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); if (next >= units) break; unit = next; continue; }
#endif
      stream_copy_out<RES>(stage, whole_dense, log_cpr, unit, c0, cw, p, lane, dense8);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // read back before the next unit overwrites it
      if (next >= units) break;
      unit = next;
    }
  }
}

/*
 * Long reductions (256 < K <= 1024) on the same scheme: the late MobileNet project layers (14x14x384 -> 96,
 * 7x7x960 -> 320 ...). A wave has about one unit there, so there is no next unit to prefetch; instead ALL the K
 * blocks of the unit's rows are requested at once (up to 32 x 16 bytes per lane in flight) and the column's weights sit
 * in LDS (up to 96 KiB per workgroup). The one-wave-per-32x32-block kernel below fetches both operands from L2 in
 * groups of four K blocks: a chain of dependent round trips as long as the reduction.
 * KBMAX: register budget in K blocks (12 / 20 / 32); the real count is a run-time value, blocks beyond it are skipped.
 */
constexpr uint32_t kMaxLdsLongK = 96 * 1024;
constexpr int longk_waves(int kbmax) { return kbmax <= 12 ? 4 : (kbmax <= 20 ? 3 : 2); }

/* PF (round 5; KBMAX <= 20): many rows after all -- ResNet-50's 28x28 512 -> 128 is 3136 units, two or more per wave. The rows of
 * a wave's NEXT unit are requested before the current one is multiplied (a second register set, the loop unrolled twice so
 * that the sets swap roles without copies); without it every unit is a full memory round trip in front of its first MFMA. */
template <int KBMAX, int SEQ, bool FULL, bool RES = false, bool PF = false>
__global__ __launch_bounds__(kThreads, PF ? 2 : longk_waves(KBMAX))
void q8_pw_stream_longk_kernel(const IgemmParams p, const uint32_t nbp, const uint32_t log_cpr)
{
  static_assert(!PF || KBMAX <= 20, "two row sets of 32 K blocks do not fit the register file");
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = tid >> 6;
  const uint32_t row_in_block = lane & 31u;
  const uint32_t khalf = lane >> 5;

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t kbn = (p.k_total + 31u) / 32u;           // K blocks with data (<= KBMAX)
  const uint32_t nb0 = blockIdx.y * nbp;
  const uint32_t nbn = min(nbp, nblocks - nb0);
  const uint32_t c0 = nb0 * 32u;
  const uint32_t cw = min(p.n - c0, nbn * 32u);
  const bool whole_dense = gridDim.y == 1 && p.output_stride == p.n;
  const uint32_t pitch = whole_dense ? p.n : (16u << log_cpr);
  {
    const uint32_t frags = nbn * kbn;
    for (uint32_t f = wave; f < frags; f += kWaves) {
      const uint32_t nb = f / kbn;
      const uint32_t kb = f - nb * kbn;
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (p.packed_w + (static_cast<uint64_t>(nb0 + nb) * kblocks + kb) * 1024 + lane * 16),
          (__attribute__((address_space(3))) void*) (lds + f * 1024), 16, 0, 0);
    }
    uint8_t* lds_bias = lds + nbp * kbn * 1024;
    const uint32_t bias_chunks = nbn * 8u;
    const int32_t* bias_src = rq_is_lane<SEQ>() ? p.bias2u : p.bias2;   // (lane forms: bias + 2^31, the pair table's second half)
    for (uint32_t q0 = wave * 64; q0 < bias_chunks; q0 += kThreads) {
      const uint32_t q = min(q0 + lane, bias_chunks - 1);
      __builtin_amdgcn_global_load_lds(
          (const __attribute__((address_space(1))) void*) (reinterpret_cast<const uint8_t*>(bias_src + c0) + q * 16),
          (__attribute__((address_space(3))) void*) (lds_bias + q0 * 16), 16, 0, 0);
    }
  }
  const uint8_t* lds_w = lds + lane * 16;
  const uint32_t bias_bytes = (nbp * 128u + 1023u) & ~1023u;
  const int4* lds_bias4 = reinterpret_cast<const int4*>(lds + nbp * kbn * 1024);
  uint8_t* stage = lds + nbp * kbn * 1024 + bias_bytes + wave * (32u * pitch);
  const uint8_t* pad16 = p.fill_table + 0x80 * 16;        // 16 bytes of a' == 0
  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t unit_stride = gridDim.x * kWaves;
  const uint32_t raw_to_centred = 128u * 32u * kbn;

  // every K block of the unit's rows at once; a piece beyond K (only in the last block) reads the a' == 0 line
  auto load_rows = [&](uint32_t unit, v4i (&a)[KBMAX]) __attribute__((always_inline)) {
    uint32_t m = unit * 32u + row_in_block;
    if (m >= p.rows) m = p.rows - 1;
    const uint8_t* row = p.input + static_cast<uint64_t>(m) * p.input_stride + khalf * 16;
#pragma unroll
    for (int kb = 0; kb < KBMAX; kb++) {
      if (static_cast<uint32_t>(kb) < kbn) {
        const uint8_t* src = (kb * 32u + khalf * 16u < p.k_total) ? row + kb * 32 : pad16;
        a[kb] = *reinterpret_cast<const v4i*>(src);
      }
    }
  };
  auto recentre = [&](v4i (&a)[KBMAX]) __attribute__((always_inline)) -> uint32_t {
    uint32_t rs = 0;
#pragma unroll
    for (int kb = 0; kb < KBMAX; kb++) {
      if (static_cast<uint32_t>(kb) < kbn) {
        rs = __builtin_amdgcn_sad_u8(a[kb].x, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].y, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].z, 0u, rs);
        rs = __builtin_amdgcn_sad_u8(a[kb].w, 0u, rs);
        a[kb].x ^= static_cast<int>(kFlip);
        a[kb].y ^= static_cast<int>(kFlip);
        a[kb].z ^= static_cast<int>(kFlip);
        a[kb].w ^= static_cast<int>(kFlip);
      }
    }
    rs += __shfl_xor(rs, 32);
    return rs;
  };

  uint32_t unit = blockIdx.x * kWaves + wave;
  v4i a[KBMAX];
  load_rows(min(unit, units - 1u), a);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // weights + bias are in LDS (and the first rows landed)
  __syncthreads();
  if (unit >= units) return;

  const std::integral_constant<int, SEQ> shift0{};
  const std::integral_constant<bool, FULL> full{};
  (void) shift0; (void) full;
  auto process = [&](uint32_t unit_now, v4i (&x)[KBMAX]) __attribute__((always_inline)) {
    const uint32_t rs = recentre(x);
    const int32_t rowterm = with_rq_offset<SEQ>(p.row_coeff * static_cast<int32_t>(rs - raw_to_centred));
    uint64_t row_addend = 0;
    if constexpr (rq_is_lane<SEQ>()) row_addend = lane_addend(rowterm, p.lane);
    uint8_t* img = stage + row_in_block * pitch;
    for (uint32_t nb = 0; nb < nbn; nb++) {
      v16i acc;
      // (read per block, beside the weight fragment: carried over from the previous trip it cost 16 v_mov_b64 per block)
      int4 bias4[4];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[nb * 8 + rg * 2 + khalf];
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        acc[rg * 4 + 0] = bias4[rg].x; acc[rg * 4 + 1] = bias4[rg].y;
        acc[rg * 4 + 2] = bias4[rg].z; acc[rg * 4 + 3] = bias4[rg].w;
      }
      const uint8_t* wf = lds_w + nb * (kbn * 1024);
#pragma unroll
      for (int kb = 0; kb < KBMAX; kb++) {
        if (static_cast<uint32_t>(kb) < kbn) {
          const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, x[kb], acc, 0, 0, 0);
        }
      }
      if constexpr (rq_is_lane<SEQ>()) {
        igemm_stage_tile_lane<SEQ, FULL>(acc, row_addend, img, nb * 32, khalf, p, nb * 32 + khalf * 16 < cw);
      } else {
        igemm_stage_tile<SEQ, FULL, false, 2>(acc, bias4, rowterm, img, nb * 32, khalf, p, nb * 32 + khalf * 16 < cw);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
    stream_copy_out<RES>(stage, whole_dense, log_cpr, unit_now, c0, cw, p, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // read back before the next unit overwrites it
  };
  if constexpr (PF) {
    v4i b[KBMAX];
    for (;;) {
      uint32_t next = unit + unit_stride;
      if (next < units) load_rows(next, b);
      process(unit, a);
      if (next >= units) break;
      unit = next;
      next = unit + unit_stride;
      if (next < units) load_rows(next, a);
      process(unit, b);
      if (next >= units) break;
      unit = next;
    }
  } else {
    for (;;) {
      process(unit, a);
      const uint32_t next = unit + unit_stride;
      if (next >= units) break;
      unit = next;
      load_rows(unit, a);
    }
  }
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

template <bool RES>
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
  // (launcher: the input tensor is addressable with 32-bit offsets)
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>((p.rows - 1u) * p.input_stride + p.k_total), 0x00020000);
  const uint32_t row_off = m * p.input_stride;
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
  // Two groups of kGwUnroll K blocks are in flight: group g+1 is issued before group g is multiplied (every load is
  // unconditional -- clamped block index, padding source -- so hipcc counts the waits instead of draining). One
  // group per round made the launch a chain of K/128 dependent L2 round trips.
  // Activations come through a buffer descriptor: a piece beyond the row's K gets an out-of-range offset and reads
  // as zeros -- nothing for the row sum, and its weights are zero -- without any branch or pointer select (a
  // uniform "is this block there" branch around a load made hipcc copy the prefetched registers at the loop end,
  // which waits for them).
  auto load_group = [&](uint32_t kb0, v4i (&a)[kGwUnroll], v4i (&w)[kGwUnroll]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kGwUnroll; u++) {
      const uint32_t kb = kb0 + u;
      const uint32_t koff = kb * 32 + khalf * 16;                          // k_total % 16 == 0: whole piece or none
      a[u] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, koff < p.k_total ? row_off + koff : 0xFFFFFFF0u, 0, 0));
      const uint32_t kbc = min(kb, kbt - 1);                               // (clamped fragments meet zeros)
      w[u] = *reinterpret_cast<const v4i*>(wf + static_cast<uint64_t>(kbc) * 1024);
    }
  };
  auto multiply_group = [&](uint32_t kb0, v4i (&a)[kGwUnroll], const v4i (&w)[kGwUnroll]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < kGwUnroll; u++) {
      rs = __builtin_amdgcn_sad_u8(a[u].x, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].y, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].z, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].w, 0u, rs);
      // a piece beyond K arrived as zeros and stays zero (a' == 0: the clamped weight fragment beside it is a real one)
      const int flip = static_cast<int>((kb0 + u) * 32 + khalf * 16 < p.k_total ? kFlip : 0u);
      a[u].x ^= flip;
      a[u].y ^= flip;
      a[u].z ^= flip;
      a[u].w ^= flip;
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[u], a[u], acc, 0, 0, 0);
    }
  };
  v4i a0[kGwUnroll], w0[kGwUnroll], a1[kGwUnroll], w1[kGwUnroll];
  load_group(0, a0, w0);
  for (uint32_t kb0 = 0; kb0 < kbt; kb0 += 2 * kGwUnroll) {
    load_group(kb0 + kGwUnroll, a1, w1);
    multiply_group(kb0, a0, w0);
    load_group(kb0 + 2 * kGwUnroll, a0, w0);
    multiply_group(kb0 + kGwUnroll, a1, w1);
  }
  rs += __shfl_xor(rs, 32);                                   // the other K half of the same row
  // (pieces beyond K read as zeros: the raw sum is over the k_total real bytes)
  const int32_t rowterm = p.row_coeff * static_cast<int32_t>(rs - 128u * p.k_total);
  uint8_t* out_row = p.output + static_cast<uint64_t>(rb * 32u + row_in_block) * p.output_stride;
  requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
    if constexpr (RES) {
      const uint8_t* res_row = p.residual + static_cast<uint64_t>(row_ok ? rb * 32u + row_in_block : 0u) * p.residual_stride;
      igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 0, true>(
          acc, bias4, with_rq_offset<decltype(shift0)::value>(rowterm), out_row, nb * 32, khalf, row_ok, p, res_row, &p.add);
      return;
    }
    igemm_store_tile<decltype(shift0)::value, decltype(full)::value>(
        acc, bias4, with_rq_offset<decltype(shift0)::value>(rowterm), out_row, nb * 32, khalf, row_ok, p);
  });
}

/*
 * The same with the reduction SPLIT over the waves of a workgroup (one workgroup = one 32x32 output block): with one
 * wave per block a K = 960 layer is a chain of 8 dependent (load 4 K blocks -> wait -> 4 MFMAs) rounds, ~1.2 us of
 * L2 latency each, and the whole launch lasts exactly as long as that chain (10.6 us for 7x7x960 -> 320 at batch
 * 128, which is 1 us of MFMA work). Here wave w takes K blocks w, w + 4, ... and issues up to eight blocks of loads
 * (16 x 16 bytes per lane) before the first multiply; the partial accumulators and row sums of waves 1-3 meet wave
 * 0's through LDS (12 KiB), and wave 0 requantizes and stores. int32 partial sums are exact, so the split is
 * bit-invisible.
 */
template <int kGwkDepth, bool RES>       // K blocks a wave has in flight: 4 or 8
__global__ __launch_bounds__(kThreads, 4)
void q8_pw_stream_gwk_kernel(const IgemmParams p)
{
  __shared__ __attribute__((aligned(16))) int32_t part[(kWaves - 1) * 17 * 64];   // [wave-1][register | row sum][lane]
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t row_in_block = lane & 31u;
  const uint32_t khalf = lane >> 5;
  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t unit = blockIdx.x;                         // channel block fastest: neighbours share the rows
  const uint32_t rb = unit / nblocks;
  const uint32_t nb = unit - rb * nblocks;

  uint32_t m = rb * 32u + row_in_block;
  const bool row_ok = m < p.rows;
  if (!row_ok) m = p.rows - 1;
  // (launcher: the input tensor is addressable with 32-bit offsets; pieces beyond K read as zeros, see the kernel above)
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>((p.rows - 1u) * p.input_stride + p.k_total), 0x00020000);
  const uint32_t row_off = m * p.input_stride;
  const int8_t* wf = p.packed_w + static_cast<uint64_t>(nb) * kblocks * 1024 + lane * 16;
  const uint32_t kbt = (p.k_total + 31u) / 32u;

  v16i acc;
#pragma unroll
  for (int r = 0; r < 16; r++) acc[r] = 0;
  uint32_t rs = 0;
  for (uint32_t kb0 = wave; kb0 < kbt; kb0 += kWaves * kGwkDepth) {
    v4i a[kGwkDepth], w[kGwkDepth];
#pragma unroll
    for (int u = 0; u < kGwkDepth; u++) {
      const uint32_t kb = kb0 + u * kWaves;
      const uint32_t koff = kb * 32 + khalf * 16;                          // k_total % 16 == 0: whole piece or none
      a[u] = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(in_rsrc, koff < p.k_total ? row_off + koff : 0xFFFFFFF0u, 0, 0));
      const uint32_t kbc = min(kb, kbt - 1);                               // (clamped fragments meet zeros)
      w[u] = *reinterpret_cast<const v4i*>(wf + static_cast<uint64_t>(kbc) * 1024);
    }
#pragma unroll
    for (int u = 0; u < kGwkDepth; u++) {
      rs = __builtin_amdgcn_sad_u8(a[u].x, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].y, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].z, 0u, rs);
      rs = __builtin_amdgcn_sad_u8(a[u].w, 0u, rs);
      // a piece beyond K arrived as zeros and stays zero (a' == 0: the clamped weight fragment beside it is a real one)
      const int flip = static_cast<int>((kb0 + u * kWaves) * 32 + khalf * 16 < p.k_total ? kFlip : 0u);
      a[u].x ^= flip;
      a[u].y ^= flip;
      a[u].z ^= flip;
      a[u].w ^= flip;
      acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[u], a[u], acc, 0, 0, 0);
    }
  }
  // raw sums of the wave's K blocks; sum(a') = sum(a) - 128 * k_total once per row, below
  int32_t rsc = static_cast<int32_t>(rs);
  if (wave != 0) {
    int32_t* mine = part + (wave - 1) * 17 * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; r++) mine[r * 64] = acc[r];
    mine[16 * 64] = rsc;
  }
  __syncthreads();
  if (wave != 0) return;
#pragma unroll
  for (int w = 0; w < kWaves - 1; w++) {
    const int32_t* other = part + w * 17 * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; r++) acc[r] += other[r * 64];
    rsc += other[16 * 64];
  }
  rsc += __shfl_xor(rsc, 32);                                 // the other K half of the same row
  rsc -= static_cast<int32_t>(128u * p.k_total);
  int4 bias4[4];
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    bias4[rg] = *reinterpret_cast<const int4*>(p.bias2 + nb * 32 + rg * 8 + khalf * 4);
  }
  uint8_t* out_row = p.output + static_cast<uint64_t>(rb * 32u + row_in_block) * p.output_stride;
  requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
    const int32_t rowterm = with_rq_offset<decltype(shift0)::value>(p.row_coeff * rsc);
    if constexpr (RES) {
      const uint8_t* res_row = p.residual + static_cast<uint64_t>(row_ok ? rb * 32u + row_in_block : 0u) * p.residual_stride;
      igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 0, true>(
          acc, bias4, rowterm, out_row, nb * 32, khalf, row_ok, p, res_row, &p.add);
      return;
    }
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

  requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
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
      const int32_t rowterm = with_rq_offset<decltype(shift0)::value>(p.row_coeff * static_cast<int32_t>(rs - raw_to_centred));

      const uint32_t m = unit * 32u + row_in_block;
      uint8_t* out_row = p.output + static_cast<uint64_t>(m) * p.output_stride;
      const bool row_ok = m < p.rows;
      for (uint32_t nb = 0; nb < nblocks; nb++) {
        v16i acc;
        // (read per block, beside the weight fragment: carried over from the previous trip it cost 16 v_mov_b64 per block)
        int4 bias4[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[nb * 8 + rg * 2 + khalf];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          acc[rg * 4 + 0] = bias4[rg].x; acc[rg * 4 + 1] = bias4[rg].y;
          acc[rg * 4 + 2] = bias4[rg].z; acc[rg * 4 + 3] = bias4[rg].w;
        }
        const uint8_t* wf = lds_w + nb * (KB * 1024);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a[kb], acc, 0, 0, 0);
        }
        igemm_store_tile<decltype(shift0)::value, decltype(full)::value, false, 2>(
            acc, bias4, rowterm, out_row, nb * 32, khalf, row_ok, p);
      }
    }
  });
}

/*
 * The same gather for 16-byte aligned output rows (n % 16 == 0: every real first layer), built like the staged
 * pointwise kernel: every load of the loop is issued unconditionally (buffer loads: 32-bit offsets, clamped unit
 * index), the unit's block is requantized into a per-wave LDS image, and the one vmcnt wait sits between the
 * multiply phase and the unit's stores. Per unit and lane: 8 table entries + 8 pixel dwords (two units / one unit
 * ahead), ~4 vector instructions per tap -- the first version spent ~22 per tap on 64-bit addresses, per-load
 * branches and the bytewise tail, and waited for its "prefetched" loads at once (vmcnt(0) behind them: the branch
 * around each load made the outstanding count path-dependent).
 * The last pixel of the tensor cannot be read as a dword: units that may touch the last image take a flavour of the
 * gather that clamps the offset to the last whole dword and shifts (wave-uniform choice, both flavours issue the
 * same loads).
 */
template <int SEQ, bool FULL>       // the requantization flavour, chosen on the host (as the staged pointwise kernel)
__global__ __launch_bounds__(kThreads, 5)
void q8_conv_stream_c3s_kernel(const IgemmParams p, const uint32_t log_cpr)
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
  const bool whole_dense = p.output_stride == p.n;
  const uint32_t pitch = whole_dense ? p.n : (16u << log_cpr);
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
  }
  const uint8_t* lds_w = lds + lane * 16;
  const uint32_t bias_bytes = (p.n_pad * 4u + 1023u) & ~1023u;
  const int4* lds_bias4 = reinterpret_cast<const int4*>(lds + nblocks * KB * 1024);
  uint8_t* stage = lds + nblocks * KB * 1024 + bias_bytes + wave * (32u * pitch);

  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t unit_stride = gridDim.x * kWaves;
  const uint32_t raw_to_centred = 128u * 32u * KB;
  const uint32_t in_bytes = static_cast<uint32_t>(p.input_end - p.input);          // (launcher: < 2^31)
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint8_t*>(p.input), 0, static_cast<int>(in_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t off_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<int32_t*>(p.offsets), 0, static_cast<int>(p.rows_per_image * p.ks * 4u + 16u), 0x00020000);   // (+16: convolution.c allocates the slack)
  // first output row (flattened) whose window may hold the tensor's last pixel: conservatively the last image
  const uint32_t tail_first = p.rows - p.rows_per_image;

  // what a slot that is not a pixel multiplies with: the padding pixel {izp, izp, izp, 0x80} for a tap of the
  // kernel, a' == 0 for a slot beyond it (K padding)
  uint32_t fill[KB * 4];
  bool is_tap[KB * 4];
#pragma unroll
  for (int i = 0; i < KB * 4; i++) {
    const uint32_t tap = (i >> 2) * 8 + khalf * 4 + (i & 3);       // kb*8 + khalf*4 + j
    is_tap[i] = tap < p.ks;
    fill[i] = is_tap[i] ? p.izp_fill : 0x80808080u;
  }

  struct Taps { int32_t off[KB * 4]; uint32_t img_off; };
  // table entries of the lane's 8 slots; slots beyond the kernel read the neighbouring entries (or 0 beyond the
  // table) and are replaced by `fill`
  auto load_offsets = [&](uint32_t unit, Taps& t) __attribute__((always_inline)) {
    uint32_t m = unit * 32u + row_in_block;
    if (m >= p.rows) m = p.rows - 1;                       // clamped rows are never stored
    const uint32_t img = m / p.rows_per_image;
    const uint32_t pix = m - img * p.rows_per_image;
    t.img_off = img * static_cast<uint32_t>(p.image_stride);
    const uint32_t voff = pix * p.ks * 4u + khalf * 16u;
    // one 16-byte load per K block (the lane's four consecutive entries): with a dword per load every instruction
    // walked the same ~18 cache lines again (36 bytes between the pixels of neighbouring lanes)
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      const v4i e = __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(off_rsrc, voff + kb * 32, 0, 0));
      t.off[kb * 4 + 0] = e.x; t.off[kb * 4 + 1] = e.y; t.off[kb * 4 + 2] = e.z; t.off[kb * 4 + 3] = e.w;
    }
  };
  // pixel dwords of the slots: offsets < 0 (padding, K padding) read some other in-range dword or 0; replaced later
  auto load_pixels = [&](auto tail_tag, const Taps& t, uint32_t (&x)[KB * 4]) __attribute__((always_inline)) {
    constexpr bool TAIL = decltype(tail_tag)::value;
#pragma unroll
    for (int i = 0; i < KB * 4; i++) {
      const uint32_t voff = t.img_off + static_cast<uint32_t>(t.off[i]);
      if constexpr (TAIL) {
        const uint32_t vc = min(voff, in_bytes - 4u);      // (negative offsets wrap to huge values: clamped too)
        const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, vc, 0, 0);
        x[i] = v >> (((voff - vc) & 3u) * 8u);
      } else {
        x[i] = __builtin_amdgcn_raw_buffer_load_b32(in_rsrc, voff, 0, 0);
      }
    }
  };
  // slots -> MFMA operand (raw bytes), row sum, re-centred at 128
  auto finish_rows = [&](const uint32_t (&x)[KB * 4], const Taps& t, v4i (&a)[KB]) __attribute__((always_inline)) -> uint32_t {
    uint32_t rs = 0;
#pragma unroll
    for (int kb = 0; kb < KB; kb++) {
      uint32_t v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int i = kb * 4 + j;
        const uint32_t pixel = (x[i] & 0x00FFFFFFu) | 0x80000000u;   // the slot's 4th byte is K padding: a' == 0
        v[j] = (is_tap[i] && t.off[i] >= 0) ? pixel : fill[i];
        rs = __builtin_amdgcn_sad_u8(v[j], 0u, rs);
        v[j] ^= kFlip;
      }
      a[kb] = v4i{static_cast<int>(v[0]), static_cast<int>(v[1]), static_cast<int>(v[2]), static_cast<int>(v[3])};
    }
    rs += __shfl_xor(rs, 32);
    return rs;
  };
  auto gather = [&](uint32_t unit, const Taps& t, uint32_t (&x)[KB * 4]) __attribute__((always_inline)) {
    if (unit * 32u + 32u > tail_first) {                   // wave-uniform
      load_pixels(std::true_type{}, t, x);
    } else {
      load_pixels(std::false_type{}, t, x);
    }
  };

  uint32_t unit = blockIdx.x * kWaves + wave;
  const uint32_t last = units - 1u;
  Taps t_cur, t_next;
  uint32_t x[KB * 4];
  v4i a[KB];
  load_offsets(min(unit, last), t_cur);
  load_offsets(min(unit + unit_stride, last), t_next);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // weights + bias are in LDS, the table entries landed
  gather(min(unit, last), t_cur, x);
  __syncthreads();
  if (unit >= units) return;

  {
    const std::integral_constant<int, SEQ> shift0{};
    const std::integral_constant<bool, FULL> full{};
    (void) shift0; (void) full;
    uint32_t rs = finish_rows(x, t_cur, a);
    for (;;) {
      const uint32_t next = unit + unit_stride;
      // pixels of the next unit (its table entries landed one iteration ago), table entries of the one after
      gather(min(next, last), t_next, x);
      Taps t_after;
      load_offsets(min(next + unit_stride, last), t_after);

      const int32_t rowterm = with_rq_offset<decltype(shift0)::value>(p.row_coeff * static_cast<int32_t>(rs - raw_to_centred));
      uint8_t* img = stage + row_in_block * pitch;
      for (uint32_t nb = 0; nb < nblocks; nb++) {
        v16i acc;
        // (read per block, beside the weight fragment: carried over from the previous trip it cost 16 v_mov_b64 per block)
        int4 bias4[4];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) bias4[rg] = lds_bias4[nb * 8 + rg * 2 + khalf];
#pragma unroll
        for (int rg = 0; rg < 4; rg++) {
          acc[rg * 4 + 0] = bias4[rg].x; acc[rg * 4 + 1] = bias4[rg].y;
          acc[rg * 4 + 2] = bias4[rg].z; acc[rg * 4 + 3] = bias4[rg].w;
        }
        const uint8_t* wf = lds_w + nb * (KB * 1024);
#pragma unroll
        for (int kb = 0; kb < KB; kb++) {
          const v4i w = *reinterpret_cast<const v4i*>(wf + kb * 1024);
          acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(w, a[kb], acc, 0, 0, 0);
        }
        igemm_stage_tile<decltype(shift0)::value, decltype(full)::value, false, 2>(
            acc, bias4, rowterm, img, nb * 32, khalf, p, nb * 32 + khalf * 16 < p.n);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // image complete before it is read back
      __builtin_amdgcn_sched_barrier(0);
      rs = finish_rows(x, t_next, a);                            // first use of the loads: the wait lands here
      t_next = t_after;
      __builtin_amdgcn_sched_barrier(0);
      stream_copy_out(stage, whole_dense, log_cpr, unit, 0u, p.n, p, lane);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // read back before the next unit overwrites it
      if (next >= units) break;
      unit = next;
    }
  }
}

template <int KB, int VEC, bool D2S = false, bool RES = false>
int launch_pw(const IgemmParams& p, uint32_t lds_bytes, hipStream_t stream)
{
  if constexpr (!RES && !D2S) {
    if (p.residual != nullptr) return launch_pw<KB, VEC, false, true>(p, lds_bytes, stream);
  }
  auto kernel = q8_pw_stream_mfma_kernel<KB, VEC, D2S, RES>;
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
  }
  // LDS bounds the residency: kMaxLds -> 2 per CU, half of that -> 4
  uint32_t per_cu = (KB <= 5) ? 4u : 2u;
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

template <int VEC>
int dispatch_kb(const IgemmParams& p, uint32_t kb, uint32_t lds_bytes, hipStream_t stream)
{
  switch (kb) {
    case 1: return launch_pw<1, VEC>(p, lds_bytes, stream);
    case 2: return launch_pw<2, VEC>(p, lds_bytes, stream);
    case 3: return launch_pw<3, VEC>(p, lds_bytes, stream);
    case 4: return launch_pw<4, VEC>(p, lds_bytes, stream);
    case 5: return launch_pw<5, VEC>(p, lds_bytes, stream);
    case 6: return launch_pw<6, VEC>(p, lds_bytes, stream);
    case 7: return launch_pw<7, VEC>(p, lds_bytes, stream);
    default: return launch_pw<8, VEC>(p, lds_bytes, stream);
  }
}

struct StagedPlan {
  uint32_t nbp;        // channel blocks per workgroup column
  uint32_t nsplit;     // workgroup columns
  uint32_t log_cpr;    // log2(16-byte pieces per image row) of the chunked image
  uint32_t lds_bytes;
  uint32_t per_cu;     // resident workgroups per CU the launch is sized for
  uint32_t dense8;     // 1: dense rows of n % 16 == 8 bytes, whole rows per unit (stream_copy_out's third mode)
};

uint32_t staged_lds_bytes(uint32_t kb, uint32_t nbp, uint32_t pitch)
{
  return nbp * kb * 1024u + ((nbp * 128u + 1023u) & ~1023u) + kWaves * 32u * pitch;
}

/* How the staged kernel covers an N: whole rows per unit when that fills the chip (and fits), else columns of 2 / 4 /
 * 8 channel blocks -- the widest that still gives every wave slot about two units. */
bool plan_staged(const IgemmParams& p, uint32_t kb, StagedPlan* plan, uint32_t vec = 16)
{
  plan->dense8 = 0;
  // dense rows of 8 (mod 16) bytes from a 16-byte aligned base: every 32-row block starts 16-byte aligned
  const bool dense8 = p.n % 16u == 8u && p.output_stride == p.n && reinterpret_cast<uintptr_t>(p.output) % 16u == 0 &&
      p.d2s_sh == 0 && p.rows >= 32u &&
      // (a residual rides in the 16-byte-load flavour only: launch_pw_staged_as)
      (p.residual == nullptr || (vec == 16 && p.residual_stride == p.n && reinterpret_cast<uintptr_t>(p.residual) % 16u == 0));
  if (dense8) {
    const uint32_t nb = p.n_pad / 32u;
    uint32_t lg = 1;
    while ((16u << lg) < nb * 32u) lg++;
    const uint32_t lds = staged_lds_bytes(kb, nb, 16u << lg);
    if (lds > kMaxLds) return false;
    plan->nbp = nb; plan->nsplit = 1; plan->log_cpr = lg; plan->lds_bytes = lds; plan->dense8 = 1;
    uint32_t per_cu = static_cast<uint32_t>(staged_waves(static_cast<int>(kb)));
    const uint32_t by_lds = (160u * 1024u) / lds;
    if (by_lds < per_cu) per_cu = by_lds > 0 ? by_lds : 1u;
    plan->per_cu = per_cu;
    return true;
  }
  if (p.store_mode != 2 || p.n % 16u != 0 || p.d2s_sh != 0) return false;
  const uint32_t nblocks = p.n_pad / 32u;
  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t base_per_cu = static_cast<uint32_t>(staged_waves(static_cast<int>(kb)));
  const uint32_t slots = p.cu_count * base_per_cu * kWaves;
  const bool dense = p.output_stride == p.n;
  uint32_t log_whole = 1;                                       // chunked image of a whole row: pitch = 2^k * 16 >= n
  while ((16u << log_whole) < nblocks * 32u) log_whole++;
  const uint32_t whole_pitch = dense ? p.n : (16u << log_whole);
  const uint32_t whole_lds = staged_lds_bytes(kb, nblocks, whole_pitch);
  const bool whole_fits = whole_lds <= kMaxLds;
  uint32_t nbp = 0;
  if (whole_fits && (units >= slots || nblocks < 4u)) {
    nbp = nblocks;
  } else {
    for (uint32_t cand = 8; cand >= 2; cand >>= 1) {
      if (cand >= nblocks || staged_lds_bytes(kb, cand, cand * 32u) > kMaxLds) continue;
      nbp = cand;
      if (static_cast<uint64_t>(units) * ((nblocks + cand - 1) / cand) >= 2ull * slots) break;
    }
    if (nbp == 0) {
      if (!whole_fits) return false;
      nbp = nblocks;
    }
  }
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_PW_NBP")) {            // measurement builds: channel blocks per workgroup column
    const uint32_t forced = static_cast<uint32_t>(atoi(env));
    if (forced >= 1 && forced <= nblocks && staged_lds_bytes(kb, forced, forced >= nblocks ? whole_pitch : forced * 32u) <= kMaxLds) {
      nbp = forced >= nblocks ? nblocks : forced;
    }
  }
#endif
  plan->nbp = nbp;
  plan->nsplit = (nblocks + nbp - 1) / nbp;
  if (plan->nsplit == 1) {
    plan->log_cpr = log_whole;
    plan->lds_bytes = whole_lds;
  } else {
    uint32_t lg = 1;
    while ((16u << lg) < nbp * 32u) lg++;
    plan->log_cpr = lg;
    plan->lds_bytes = staged_lds_bytes(kb, nbp, nbp * 32u);
  }
  uint32_t per_cu = base_per_cu;
  const uint32_t by_lds = (160u * 1024u) / plan->lds_bytes;
  if (by_lds < per_cu) per_cu = by_lds > 0 ? by_lds : 1u;
  plan->per_cu = per_cu;
  return true;
}

template <int KB, int VEC, int SEQ, bool FULL, bool RES = false>
int launch_pw_staged_as(const IgemmParams& p, const StagedPlan& plan, hipStream_t stream)
{
  if constexpr (!RES && VEC == 16) {               // (a residual implies 16-byte aligned rows: q8igemm.hip)
    if (p.residual != nullptr) return launch_pw_staged_as<KB, VEC, SEQ, FULL, true>(p, plan, stream);
  }
  auto kernel = q8_pw_stream_staged_kernel<KB, VEC, SEQ, FULL, RES>;
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
  }
  const uint32_t units = (p.rows + 31u) / 32u;
  uint32_t per_cu = plan.per_cu;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_PW_BLOCKS")) per_cu = static_cast<uint32_t>(atoi(env));
#endif
  // persistent workgroups: the resident set, spread over the columns
  uint32_t gx = (p.cu_count * per_cu + plan.nsplit - 1) / plan.nsplit;
  const uint32_t needed = (units + kWaves - 1) / kWaves;
  if (gx > needed) gx = needed;
  IgemmParams pa = p;
  pa.tiles_n_magic = static_cast<uint32_t>((1ull << 32) / p.n) + 1u;      // dense8: byte offset / n, exact below 2^32 / n
  const uint32_t log_arg = plan.log_cpr | (plan.dense8 != 0 ? 0x80000000u : 0u);
#ifdef QNNP_ENABLE_ABLATION
  pa.izp_fill = 0;
  if (const char* env = getenv("QNNP_PW_ABL")) pa.izp_fill = static_cast<uint32_t>(atoi(env));
#endif
  hipLaunchKernelGGL(kernel, dim3(gx, plan.nsplit), dim3(kThreads), plan.lds_bytes, stream, pa, plan.nbp, log_arg);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int KB, int VEC>
int launch_pw_staged(const IgemmParams& p, const StagedPlan& plan, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    rc = launch_pw_staged_as<KB, VEC, decltype(seq)::value, decltype(full)::value>(p, plan, stream);
  });
  return rc;
}

template <int VEC>
int dispatch_kb_staged(const IgemmParams& p, uint32_t kb, const StagedPlan& plan, hipStream_t stream)
{
  switch (kb) {
    case 1: return launch_pw_staged<1, VEC>(p, plan, stream);
    case 2: return launch_pw_staged<2, VEC>(p, plan, stream);
    case 3: return launch_pw_staged<3, VEC>(p, plan, stream);
    case 4: return launch_pw_staged<4, VEC>(p, plan, stream);
    case 5: return launch_pw_staged<5, VEC>(p, plan, stream);
    case 6: return launch_pw_staged<6, VEC>(p, plan, stream);
    case 7: return launch_pw_staged<7, VEC>(p, plan, stream);
    default: return launch_pw_staged<8, VEC>(p, plan, stream);
  }
}

/* plan for the long-K flavour: like plan_staged with the run-time K block count and its own LDS limit */
bool plan_longk(const IgemmParams& p, StagedPlan* plan, int* kbmax)
{
  if (p.store_mode != 2 || p.n % 16u != 0 || p.d2s_sh != 0 || p.k_total % 16u != 0) return false;
  const uint32_t kb = (p.k_total + 31u) / 32u;
  if (kb <= 8 || kb > 32) return false;
  *kbmax = kb <= 12 ? 12 : (kb <= 20 ? 20 : 32);
  const uint32_t nblocks = p.n_pad / 32u;
  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t base_per_cu = static_cast<uint32_t>(longk_waves(*kbmax));
  const uint32_t slots = p.cu_count * base_per_cu * kWaves;
  const bool dense = p.output_stride == p.n;
  uint32_t log_whole = 1;
  while ((16u << log_whole) < nblocks * 32u) log_whole++;
  const uint32_t whole_pitch = dense ? p.n : (16u << log_whole);
  const uint32_t whole_lds = staged_lds_bytes(kb, nblocks, whole_pitch);
  const bool whole_fits = whole_lds <= kMaxLdsLongK;
  uint32_t nbp = 0;
  if (whole_fits && (units >= slots || nblocks < 4u)) {
    nbp = nblocks;
  } else {
    // (columns of a single block measured worse: 7x7x960 -> 320 12.7 -> 14.3 us)
    for (uint32_t cand = 8; cand >= 2; cand >>= 1) {
      if (cand >= nblocks || staged_lds_bytes(kb, cand, cand * 32u) > kMaxLdsLongK) continue;
      nbp = cand;
      if (static_cast<uint64_t>(units) * ((nblocks + cand - 1) / cand) >= slots) break;
    }
    if (nbp == 0) {
      if (!whole_fits) return false;
      nbp = nblocks;
    }
  }
  plan->nbp = nbp;
  plan->nsplit = (nblocks + nbp - 1) / nbp;
  if (plan->nsplit == 1) {
    plan->log_cpr = log_whole;
    plan->lds_bytes = whole_lds;
  } else {
    uint32_t lg = 1;
    while ((16u << lg) < nbp * 32u) lg++;
    plan->log_cpr = lg;
    plan->lds_bytes = staged_lds_bytes(kb, nbp, nbp * 32u);
  }
  uint32_t per_cu = base_per_cu;
  const uint32_t by_lds = (160u * 1024u) / plan->lds_bytes;
  if (by_lds < per_cu) per_cu = by_lds > 0 ? by_lds : 1u;
  plan->per_cu = per_cu;
  return true;
}

template <int KBMAX, int SEQ, bool FULL, bool RES = false, bool PF = false>
int launch_longk_as(const IgemmParams& p, const StagedPlan& plan, hipStream_t stream)
{
  if constexpr (!RES) {
    if (p.residual != nullptr) return launch_longk_as<KBMAX, SEQ, FULL, true>(p, plan, stream);
  }
  if constexpr (!RES && !PF && KBMAX <= 20) {
    // more units than waves: the flavour that requests the next unit's rows before it multiplies the current one
    const uint32_t units_all = (p.rows + 31u) / 32u;
    uint32_t per_cu = plan.per_cu < 2u ? plan.per_cu : 2u;
    const uint32_t gx_pf = (p.cu_count * per_cu + plan.nsplit - 1) / plan.nsplit;
    if (static_cast<uint64_t>(gx_pf) * kWaves * 5u <= static_cast<uint64_t>(units_all) * 4u) {      // >= 1.25 units per wave
      return launch_longk_as<KBMAX, SEQ, FULL, false, true>(p, plan, stream);
    }
  }
  auto kernel = q8_pw_stream_longk_kernel<KBMAX, SEQ, FULL, RES, PF>;
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
    (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLdsLongK);
  }
  const uint32_t units = (p.rows + 31u) / 32u;
  const uint32_t per_cu_now = PF && plan.per_cu > 2u ? 2u : plan.per_cu;       // (PF: 2 waves per SIMD by its registers)
  uint32_t gx = (p.cu_count * per_cu_now + plan.nsplit - 1) / plan.nsplit;
  const uint32_t needed = (units + kWaves - 1) / kWaves;
  if (gx > needed) gx = needed;
  hipLaunchKernelGGL(kernel, dim3(gx, plan.nsplit), dim3(kThreads), plan.lds_bytes, stream, p, plan.nbp, plan.log_cpr);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int KBMAX>
int launch_longk(const IgemmParams& p, const StagedPlan& plan, hipStream_t stream)
{
  int rc = QNNP_HIP_EINVAL;
  requant_dispatch_lane(p.rq, p.lane, [&](auto seq, auto full) {
    rc = launch_longk_as<KBMAX, decltype(seq)::value, decltype(full)::value>(p, plan, stream);
  });
  return rc;
}

int dispatch_kb_d2s(const IgemmParams& p, uint32_t kb, uint32_t lds_bytes, hipStream_t stream)
{
  switch (kb) {
    case 1: return launch_pw<1, 16, true>(p, lds_bytes, stream);
    case 2: return launch_pw<2, 16, true>(p, lds_bytes, stream);
    case 3: return launch_pw<3, 16, true>(p, lds_bytes, stream);
    case 4: return launch_pw<4, 16, true>(p, lds_bytes, stream);
    case 5: return launch_pw<5, 16, true>(p, lds_bytes, stream);
    case 6: return launch_pw<6, 16, true>(p, lds_bytes, stream);
    case 7: return launch_pw<7, 16, true>(p, lds_bytes, stream);
    default: return launch_pw<8, 16, true>(p, lds_bytes, stream);
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
  if ((p.offsets != nullptr && p.offsets_dense == 0) || groups != 1 || (vec != 16 && vec != 8)) return false;
  if (p.fill_table == nullptr || p.rows == 0 || p.k_total == 0 || p.k_total > 256u) return false;
  if (p.k_total % vec != 0) return false;
  StagedPlan plan;
  if (p.offsets != nullptr) {       // strided 1x1 convolution: the staged flavour only (the one that reads the table)
    return p.d2s_sh == 0 && p.n != 32 && plan_staged(p, (p.k_total + 31u) / 32u, &plan, vec);
  }
  if (pw_lds_bytes(p) <= kMaxLds) return true;
  return plan_staged(p, (p.k_total + 31u) / 32u, &plan, vec);     // any N, in workgroup columns
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
  // 16-byte aligned output rows, tensors addressable with 32-bit offsets: the staged flavour
  const uint64_t in_bytes = static_cast<uint64_t>(p.input_end - p.input);
  const uint64_t table_bytes = static_cast<uint64_t>(p.rows_per_image) * p.ks * 4u;
  if (p.store_mode == 2 && p.n % 16u == 0 && in_bytes >= 4 && in_bytes < (UINT64_C(1) << 31) &&
      table_bytes < (UINT64_C(1) << 31) && p.rows >= p.rows_per_image) {
    uint32_t log_cpr = 1;
    while ((16u << log_cpr) < p.n_pad) log_cpr++;
    const uint32_t pitch = p.output_stride == p.n ? p.n : (16u << log_cpr);
    const uint32_t staged_bytes = lds_bytes + kWaves * 32u * pitch;
    if (staged_bytes <= kMaxLds) {
      const uint32_t units = (p.rows + 31u) / 32u;
      uint32_t per_cu = 5;                                   // what the kernel's registers allow
      const uint32_t by_lds = (160u * 1024u) / staged_bytes;
      if (by_lds < per_cu) per_cu = by_lds > 0 ? by_lds : 1u;
      uint32_t grid = p.cu_count * per_cu;
      const uint32_t needed = (units + kWaves - 1) / kWaves;
      if (grid > needed) grid = needed;
      *name = "q8_conv_stream_c3_mfma";
      requant_dispatch_ofs(p.rq, [&](auto seq, auto full) {
        auto kernel = q8_conv_stream_c3s_kernel<decltype(seq)::value, decltype(full)::value>;
        static qnnp::PerDeviceOnce attr_once_s;   // (one per instantiation of this lambda body)
        if (auto once_scope = attr_once_s.begin()) {
          (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
        }
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(kThreads), staged_bytes, stream, p, log_cpr);
      });
      return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
    }
  }
  static qnnp::PerDeviceOnce attr_once;   // function attributes are per device
  if (auto once_scope = attr_once.begin()) {
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

/* long-K staged flavour: pointwise / fully-connected form, one group, 16-byte aligned rows both sides, 256 < K <= 1024 */
bool pwstream_longk_supported(const IgemmParams& p, uint32_t groups, uint32_t vec)
{
  if (p.offsets != nullptr || groups != 1 || vec != 16) return false;
  if (p.fill_table == nullptr || p.rows == 0 || p.k_total <= 256u) return false;
  StagedPlan plan;
  int kbmax = 0;
  return plan_longk(p, &plan, &kbmax);
}

int pwstream_longk_launch(const IgemmParams& p, hipStream_t stream, const char** name)
{
  StagedPlan plan;
  int kbmax = 0;
  if (!plan_longk(p, &plan, &kbmax)) return QNNP_HIP_EINVAL;
  *name = "q8_pw_stream_longk_mfma";
  switch (kbmax) {
    case 12: return launch_longk<12>(p, plan, stream);
    case 20: return launch_longk<20>(p, plan, stream);
    default: return launch_longk<32>(p, plan, stream);
  }
}

/* global-weights flavour: pointwise / fully-connected form, one group, 16-byte aligned rows, any K and N */
bool pwstream_gw_supported(const IgemmParams& p, uint32_t groups, uint32_t vec)
{
  if (p.offsets != nullptr || groups != 1 || vec != 16) return false;
  if (p.fill_table == nullptr || p.rows == 0 || p.k_total == 0 || p.k_total % 16 != 0) return false;
  const uint64_t units = static_cast<uint64_t>((p.rows + 31u) / 32u) * (p.n_pad / 32u);
  const uint64_t in_bytes = static_cast<uint64_t>(p.rows - 1u) * p.input_stride + p.k_total;   // buffer addressing
  return units < (UINT64_C(1) << 31) && in_bytes < (UINT64_C(1) << 31);
}

int pwstream_gw_launch(const IgemmParams& p, hipStream_t stream, const char** name)
{
  const uint32_t units = ((p.rows + 31u) / 32u) * (p.n_pad / 32u);
  // long reductions: K split over the four waves of a workgroup (one workgroup per block)
  // (only when the blocks alone cannot fill the chip -- classifier heads: with ~2000 blocks the one-wave-per-block
  //  kernel was 7-25 % faster, same box: the split costs three more waves' loads of the row block and a barrier)
  bool split_k = p.k_total >= 256u && units <= 2u * p.cu_count;
#ifdef QNNP_ENABLE_ABLATION
  if (const char* env = getenv("QNNP_GW_SPLITK")) split_k = atoi(env) != 0;
#endif
  if (split_k) {
    *name = "q8_pw_stream_gwk_mfma";
    const uint32_t per_wave = ((p.k_total + 31u) / 32u + kWaves - 1) / kWaves;
    const bool res = p.residual != nullptr;
    if (per_wave <= 4) {
      if (res) hipLaunchKernelGGL((q8_pw_stream_gwk_kernel<4, true>), dim3(units), dim3(kThreads), 0, stream, p);
      else hipLaunchKernelGGL((q8_pw_stream_gwk_kernel<4, false>), dim3(units), dim3(kThreads), 0, stream, p);
    } else {
      if (res) hipLaunchKernelGGL((q8_pw_stream_gwk_kernel<8, true>), dim3(units), dim3(kThreads), 0, stream, p);
      else hipLaunchKernelGGL((q8_pw_stream_gwk_kernel<8, false>), dim3(units), dim3(kThreads), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
  }
  *name = "q8_pw_stream_gw_mfma";
  if (p.residual != nullptr) {
    hipLaunchKernelGGL(q8_pw_stream_gw_kernel<true>, dim3((units + kWaves - 1) / kWaves), dim3(kThreads), 0, stream, p);
  } else {
    hipLaunchKernelGGL(q8_pw_stream_gw_kernel<false>, dim3((units + kWaves - 1) / kWaves), dim3(kThreads), 0, stream, p);
  }
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

int pwstream_launch(const IgemmParams& p0, uint32_t vec, hipStream_t stream, const char** name)
{
  const IgemmParams& p = p0;
  const uint32_t kb = (p.k_total + 31u) / 32u;
  const uint32_t lds_bytes = pw_lds_bytes(p);
  if (p.d2s_sh != 0) {
    if (vec != 16) return QNNP_HIP_EINVAL;
    *name = "q8_pw_stream_d2s_mfma";
    return dispatch_kb_d2s(p, kb, lds_bytes, stream);
  }
  *name = "q8_pw_stream_mfma";
  // 16-byte aligned rows wider than one channel block: the staged flavour (whole-line stores, N in columns)
  StagedPlan plan;
  // (one 32-channel block per row: the first kernel keeps it -- 28x28x192 -> 32 measured 7.1 us there, 8.1 here)
  if (p.n != 32 && plan_staged(p, kb, &plan, vec)) {
    return vec == 16 ? dispatch_kb_staged<16>(p, kb, plan, stream) : dispatch_kb_staged<8>(p, kb, plan, stream);
  }
  if (p.offsets != nullptr) return QNNP_HIP_EINVAL;        // (pwstream_supported said so)
  if (lds_bytes > kMaxLds) return QNNP_HIP_EINVAL;
  return vec == 16 ? dispatch_kb<16>(p, kb, lds_bytes, stream) : dispatch_kb<8>(p, kb, lds_bytes, stream);
}

}  // namespace qnnp
