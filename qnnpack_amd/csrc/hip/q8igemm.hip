/*
 * q8igemm.hip -- uint8 GEMM and implicit-GEMM convolution on CDNA4 matrix cores.
 *
 * Replaces, as whole-operator launches, the reference's per-tile CPU microkernels
 *   q8gemm_ukernel_4x4c2__sse2  (src/q8gemm/4x4c2-sse2.c:14-318)
 *   q8conv_ukernel_4x4c2__sse2  (src/q8conv/4x4c2-sse2.c:14-273)
 * and their pthreadpool tilers compute_q8gemm / compute_q8conv
 * (src/operator-run.c:39-70, 183-217, 797-802, 837-842).
 *
 * Arithmetic (exact int32, bit-identical to the reference definition
 * test/gemm-microkernel-tester.h:213-226 + qnnp_q31_requantize):
 *   v_mfma_i32_32x32x32_i8 multiplies SIGNED int8, so both operands are
 *   re-centred at 128 (a' = a ^ 0x80, w' = w ^ 0x80 -- the latter at pack time)
 *   and the cross terms are restored from a per-row sum of a' and a folded
 *   per-column bias (pack.h). This is the reference's own "XZP" algebra
 *   (src/operator-run.c:727-743) re-centred for a signed matrix core.
 *
 * Kernel 1 (this file, "generic"): q8_igemm_mfma_kernel
 *   - one workgroup = 4 waves; tile BM x BN x 64, BM = 128, BN in {32, 64, 128}
 *   - activations: global -> VGPR -> (^0x80, row-sum) -> LDS, double buffered,
 *     XOR-swizzled 16-B chunks so the ds_read_b128 fragment reads are conflict free
 *   - weights: pre-packed MFMA fragments (pack.h), one coalesced 1-KiB global read
 *     per fragment per wave straight into VGPRs (weights are L2 resident; no LDS)
 *   - convolution gathers activation rows through the device-side int32 offset
 *     table (indirection.c); im2col is never materialised; padding taps read the
 *     input zero point
 *   - fused epilogue: + bias2 + row_coeff * rowsum -> Q31 requantize -> clamp ->
 *     4 channels packed per dword store
 *   - any M/N/K, any pixel strides, any alignment: the activation load width VEC
 *     (16/8/4/1 bytes) is picked per launch from the actual alignment
 *
 * MFMA operand roles: weights are the "A" operand (32 output channels across
 * lanes 0-31), activations the "B" operand (32 rows across lanes 0-31), so each
 * lane ends up with 4 CONSECUTIVE output channels of one row per accumulator
 * quad: C/D layout col = lane & 31 (row m), reg r -> n = (r & 3) + 8*(r >> 2) +
 * 4*(lane >> 5).
 */
#include <hip/hip_runtime.h>

#include <stdint.h>

#include "igemm_epilogue.hip.h"
#include "igemm_params.h"
#include "qnnp_hip.h"
#include "requant.hip.h"


namespace {

using qnnp::IgemmParams;

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int BK = 64;                       // bytes of K per main-loop step
constexpr uint32_t kPadK = 0x80808080u;      // raw bytes whose a' = a ^ 0x80 is zero
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;


template <int VEC>
__device__ __forceinline__ void load_vec(const uint8_t* p, uint32_t (&w)[4], int j);

template <>
__device__ __forceinline__ void load_vec<16>(const uint8_t* p, uint32_t (&w)[4], int) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
}
template <>
__device__ __forceinline__ void load_vec<8>(const uint8_t* p, uint32_t (&w)[4], int j) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  w[2 * j] = v.x; w[2 * j + 1] = v.y;
}
template <>
__device__ __forceinline__ void load_vec<4>(const uint8_t* p, uint32_t (&w)[4], int j) {
  w[j] = *reinterpret_cast<const uint32_t*>(p);
}
template <>
__device__ __forceinline__ void load_vec<1>(const uint8_t* p, uint32_t (&w)[4], int j) {
  const uint32_t b = *p;
  const int sh = (j & 3) * 8;
  w[j >> 2] = (w[j >> 2] & ~(0xFFu << sh)) | (b << sh);
}

template <int VEC>
__device__ __forceinline__ void fill_vec(uint32_t fill, uint32_t (&w)[4], int j) {
  if constexpr (VEC == 16) {
    w[0] = fill; w[1] = fill; w[2] = fill; w[3] = fill;
  } else if constexpr (VEC == 8) {
    w[2 * j] = fill; w[2 * j + 1] = fill;
  } else if constexpr (VEC == 4) {
    w[j] = fill;
  } else {
    const int sh = (j & 3) * 8;
    w[j >> 2] = (w[j >> 2] & ~(0xFFu << sh)) | ((fill & 0xFFu) << sh);
  }
}

/*
 * WM x WN waves, each computing TM x TN MFMA tiles of 32x32. PERSISTENT over row tiles: a workgroup
 * keeps one channel tile and walks row tiles m = cta_m, cta_m + ctas_m, ...; the activation loads of the
 * next row tile are issued before the epilogue of the current one, so the (latency-bound) global reads
 * overlap the (VALU-bound) requantization instead of alternating with it.
 */
template <int WM, int WN, int TM, int TN, int VEC, bool IS_CONV, bool PAD3 = false>
__global__ __launch_bounds__(WM * WN * 64, (TM * TN >= 4) ? 2 : 3)
void q8_igemm_mfma_kernel(const IgemmParams p_in)
{
  // phase table (strided deconvolutions): blockIdx.y picks one of several GEMM descriptions sharing this launch
  IgemmParams p = p_in;
  if (p_in.phases != nullptr) {
    const qnnp_hip_igemm_phase ph = p_in.phases[blockIdx.y / p_in.phase_groups];
    p.packed_w = ph.packed_w;
    p.bias2 = ph.bias2;
    p.offsets = ph.offsets;
    p.out_rows = ph.out_rows;
    p.rows = ph.rows;
    p.rows_per_image = ph.rows_per_image;
    p.ks = ph.ks;
    p.k_total = ph.k_total;
    p.k_pad = ph.k_pad;
  }
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int CH = (BM * (BK / 16)) / NT;  // 16-byte activation chunks staged per thread per step
  static_assert((BM * (BK / 16)) % NT == 0, "tile/thread mismatch");
  static_assert(CH >= 1, "tile too small");

  // one LDS object only: [2][BM][BK] activation ring | [BM][BN+16] staged output tile | [BM] int32 row sums
  constexpr int OUT_PITCH = BN + 16;
  constexpr int RING_BYTES = 2 * BM * BK;
  constexpr int OUT_BYTES = BM * OUT_PITCH;
  __shared__ __attribute__((aligned(16))) uint8_t lds[RING_BYTES + OUT_BYTES + BM * 4];
  uint8_t* lds_out = lds + RING_BYTES;
  int32_t* lds_rowsum = reinterpret_cast<int32_t*>(lds + RING_BYTES + OUT_BYTES);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63;
  const uint32_t wave = tid >> 6;
  const uint32_t wm = wave / WN;
  const uint32_t wn = wave % WN;
  const uint32_t g = p_in.phases != nullptr ? blockIdx.y % p_in.phase_groups : blockIdx.y;

  // XCD-aware bijective remap: consecutive logical ids (which share activation row tiles) land on the
  // same XCD / L2 (hardware: block b -> XCD b % 8).
  const uint32_t tiles_n = (p.n_pad + BN - 1) / BN;
  const uint32_t tiles_m = (p.rows + BM - 1) / BM;
  uint32_t logical;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const uint32_t n_tile = logical % tiles_n;
  const uint32_t cta_m = logical / tiles_n;
  const uint32_t ctas_m = gridDim.x / tiles_n;

  const uint32_t ksteps = p.k_pad / BK;
  const uint32_t kblocks = p.k_pad / 32;   // 32-deep fragment blocks per column block
  const uint32_t nblocks = p.n_pad / 32;

  // ---- per-thread activation staging assignment: chunk slots are fixed, rows change per row tile ----
  struct RowCtx {
    const uint8_t* base[CH];   // gemm: row base (+group); conv: image base (+group)
    const int32_t* offs[CH];   // conv: this pixel's row of the offset table
    bool valid[CH];
  };
  uint32_t a_lds[CH];          // swizzled byte offset inside one LDS buffer
  uint32_t a_kchunk[CH];       // 0..3: which 16-byte chunk of the K step
#pragma unroll
  for (int q = 0; q < CH; q++) {
    const uint32_t id = q * NT + tid;
    const uint32_t row = id >> 2;
    const uint32_t c = id & 3u;
    a_kchunk[q] = c;
    a_lds[q] = row * BK + ((c ^ ((row >> 2) & 3u)) << 4);
  }
  auto make_ctx = [&](uint32_t m_tile, RowCtx& ctx) {
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const uint32_t row = (q * NT + tid) >> 2;
      const uint32_t m = m_tile * BM + row;
      ctx.valid[q] = m_tile < tiles_m && m < p.rows;
      const uint32_t mm = ctx.valid[q] ? m : 0u;
      if constexpr (IS_CONV) {
        const uint32_t img = mm / p.rows_per_image;
        const uint32_t pix = mm - img * p.rows_per_image;
        ctx.base[q] = p.input + static_cast<uint64_t>(img) * p.image_stride + static_cast<uint64_t>(g) * p.kc;
        ctx.offs[q] = p.offsets + static_cast<uint64_t>(pix) * p.ks;
      } else {
        ctx.base[q] = p.input + static_cast<uint64_t>(mm) * p.input_stride + static_cast<uint64_t>(g) * p.kc;
        ctx.offs[q] = nullptr;
      }
    }
  };

  // global -> registers for K step `kstep` (raw uint8, already filled for padding)
  auto load_chunks = [&](const RowCtx& ctx, uint32_t kstep, uint32_t (&regs)[CH][4]) {
#pragma unroll
    for (int q = 0; q < CH; q++) {
      const uint32_t kk0 = kstep * BK + a_kchunk[q] * 16;
      regs[q][0] = kPadK; regs[q][1] = kPadK; regs[q][2] = kPadK; regs[q][3] = kPadK;
      uint32_t tap = 0, ch = kk0;
      if constexpr (IS_CONV) {
        tap = kk0 / p.kc;
        ch = kk0 - tap * p.kc;
      }
      if constexpr (VEC == 1 && !PAD3) {
        // Byte gathers (channel counts that are not a multiple of 4), BRANCH-FREE: every load is issued -- an address
        // that must not be read is replaced by the row's / image's base, the value by the zero point or the K padding
        // afterwards. As sixteen `if (inside) load` per chunk this path held an exec mask per byte: 195 spilled SGPRs +
        // 285-339 spilled VGPRs inside the K loop, and a wild address when unrelated code moved its allocation (round 3).
#pragma unroll
        for (int d = 0; d < 4; d++) {
          uint32_t v = 0;
#pragma unroll
          for (int jj = 0; jj < 4; jj++) {
            const uint32_t kk = kk0 + d * 4 + jj;
            const bool in_k = ctx.valid[q] && kk < p.k_total;
            bool real = in_k;
            uint32_t rel = kk;
            if constexpr (IS_CONV) {
              const uint32_t tap_c = tap < p.ks ? tap : p.ks - 1u;     // (beyond k_total: any entry of the pixel's row)
              const int32_t off = ctx.offs[q][tap_c];
              real = in_k && off >= 0;
              rel = static_cast<uint32_t>(off) + ch;
              ch += 1;
              if (ch >= p.kc) { ch = 0; tap += 1; }
            }
            uint32_t b = ctx.base[q][real ? rel : 0u];
            b = real ? b : (in_k ? (p.izp_fill & 0xFFu) : 0x80u);
            v |= b << (8 * jj);
          }
          regs[q][d] = v;
        }
        continue;
      }
#pragma unroll
      for (int j = 0; j < 16 / VEC; j++) {
        const uint32_t kk = kk0 + j * VEC;
        if (ctx.valid[q] && kk < p.k_total) {
          if constexpr (IS_CONV) {
            const int32_t off = ctx.offs[q][tap];
            if (off >= 0) {
              if constexpr (PAD3) {
                // 3-channel pixel: one unaligned dword (the 4th byte belongs to the next pixel and is
                // replaced by K padding); the very last pixel of the tensor is fetched bytewise
                const uint8_t* src = ctx.base[q] + off;
                uint32_t v;
                if (src + 4 <= p.input_end) {
                  v = *reinterpret_cast<const u32_unaligned*>(src);
                } else {
                  v = static_cast<uint32_t>(src[0]) | (static_cast<uint32_t>(src[1]) << 8) |
                      (static_cast<uint32_t>(src[2]) << 16);
                }
                regs[q][j] = (v & 0x00FFFFFFu) | 0x80000000u;
              } else {
                load_vec<VEC>(ctx.base[q] + off + ch, regs[q], j);
              }
            } else {
              fill_vec<VEC>(p.izp_fill, regs[q], j);   // padding tap: a == input zero point
            }
          } else {
            load_vec<VEC>(ctx.base[q] + kk, regs[q], j);
          }
        }
        if constexpr (IS_CONV) {
          ch += VEC;
          if (ch >= p.kc) { ch = 0; tap += 1; }   // VEC divides kc, so taps never straddle a vector
        }
      }
    }
  };

  int32_t rowsum_part[CH];
  // registers -> LDS: recentre at 128, accumulate the row sum of a'
  auto store_chunks = [&](uint32_t buf, const uint32_t (&regs)[CH][4]) {
#pragma unroll
    for (int q = 0; q < CH; q++) {
      v4i x;
      x.x = static_cast<int>(regs[q][0] ^ kPadK);
      x.y = static_cast<int>(regs[q][1] ^ kPadK);
      x.z = static_cast<int>(regs[q][2] ^ kPadK);
      x.w = static_cast<int>(regs[q][3] ^ kPadK);
      int32_t s = rowsum_part[q];
      s = __builtin_amdgcn_sdot4(x.x, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4(x.y, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4(x.z, 0x01010101, s, false);
      s = __builtin_amdgcn_sdot4(x.w, 0x01010101, s, false);
      rowsum_part[q] = s;
      *reinterpret_cast<v4i*>(lds + buf * (BM * BK) + a_lds[q]) = x;
    }
  };

  // weight fragments: lane l reads its 16 bytes of the (column block, K block) panel
  const uint32_t nb0 = n_tile * (BN / 32) + wn * TN;   // first 32-column block of this wave
  const int8_t* w_lane = p.packed_w + (static_cast<uint64_t>(g) * nblocks * kblocks) * 1024 + lane * 16;
  auto load_wfrags = [&](uint32_t kstep, v4i (&wf)[TN][2]) {
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      const uint32_t nb = nb0 + tn;
#pragma unroll
      for (int ksub = 0; ksub < 2; ksub++) {
        if (nb < nblocks) {
          const uint32_t kb = kstep * 2 + ksub;
          wf[tn][ksub] = *reinterpret_cast<const v4i*>(
              w_lane + (static_cast<uint64_t>(nb) * kblocks + kb) * 1024);
        } else {
          wf[tn][ksub] = v4i{0, 0, 0, 0};
        }
      }
    }
  };

  const uint32_t frag_row0 = wm * (TM * 32) + (lane & 31u);
  const uint32_t frag_khalf = lane >> 5;

  RowCtx ctx, ctx_next;
  uint32_t a_regs[CH][4];
  v4i w_cur[TN][2];
  v4i w_nxt[TN][2];

  make_ctx(cta_m, ctx);
  load_chunks(ctx, 0, a_regs);
  load_wfrags(0, w_cur);
  uint32_t step = 0;                           // global K-step counter: LDS ring slot = step & 1

  for (uint32_t m_tile = cta_m; m_tile < tiles_m; m_tile += ctas_m) {
    v16i acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; tm++)
#pragma unroll
      for (int tn = 0; tn < TN; tn++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0;
#pragma unroll
    for (int q = 0; q < CH; q++) rowsum_part[q] = 0;

    for (uint32_t kstep = 0; kstep < ksteps; kstep++, step++) {
      const uint32_t buf = step & 1u;
      store_chunks(buf, a_regs);
      __syncthreads();
      if (kstep + 1 < ksteps) {
        load_chunks(ctx, kstep + 1, a_regs);
        load_wfrags(kstep + 1, w_nxt);
      } else {
        // last K step of this row tile: fetch the first step of the NEXT row tile now, so its latency
        // hides under this tile's MFMAs and epilogue
        make_ctx(m_tile + ctas_m, ctx_next);
        load_chunks(ctx_next, 0, a_regs);
        if (ksteps > 1) load_wfrags(0, w_nxt);
      }
      const uint8_t* a_tile = lds + buf * (BM * BK);
#pragma unroll
      for (int ksub = 0; ksub < 2; ksub++) {
        v4i af[TM];
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
          const uint32_t row = frag_row0 + tm * 32;
          const uint32_t chunk = (ksub * 2 + frag_khalf) ^ ((row >> 2) & 3u);
          af[tm] = *reinterpret_cast<const v4i*>(a_tile + row * BK + (chunk << 4));
        }
#pragma unroll
        for (int tm = 0; tm < TM; tm++)
#pragma unroll
          for (int tn = 0; tn < TN; tn++)
            acc[tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w_cur[tn][ksub], af[tm], acc[tm][tn], 0, 0, 0);
      }
      if (ksteps > 1) {
#pragma unroll
        for (int tn = 0; tn < TN; tn++) {
          w_cur[tn][0] = w_nxt[tn][0];
          w_cur[tn][1] = w_nxt[tn][1];
        }
      }
      // no second barrier: the next step writes the other ring slot, whose last readers all passed
      // this step's barrier
    }
    ctx = ctx_next;

    // bias of this lane's 4-channel groups (issued before the barrier so the latency hides; L1/L2 resident)
    int4 bias4[TN][4];
#pragma unroll
    for (int tn = 0; tn < TN; tn++) {
      uint32_t nb = nb0 + tn;
      if (nb >= nblocks) nb = nblocks - 1;       // clamped blocks are never stored
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const uint32_t ncol = nb * 32 + rg * 8 + frag_khalf * 4;
        bias4[tn][rg] = *reinterpret_cast<const int4*>(p.bias2 + static_cast<uint64_t>(g) * p.n_pad + ncol);
      }
    }

    // ---- row sums: the 4 threads that staged one row are adjacent lanes ----
#pragma unroll
    for (int q = 0; q < CH; q++) {
      int32_t s = rowsum_part[q];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      if (a_kchunk[q] == 0) lds_rowsum[(q * NT + tid) >> 2] = s;
    }
    __syncthreads();

    // ---- fused epilogue (igemm_epilogue.hip.h); the requantization flavour is chosen once per tile ----
    if (p.store_mode == 2 && p.out_rows == nullptr) {
      // staged: requantized tile -> LDS (row-major image of the output) -> line-sized coalesced stores
      qnnp::requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
          const uint32_t row = frag_row0 + tm * 32;
          const int32_t rowterm = qnnp::with_rq_offset<decltype(shift0)::value>(p.row_coeff * lds_rowsum[row]);
#pragma unroll
          for (int tn = 0; tn < TN; tn++) {
            if (nb0 + tn >= nblocks) continue;       // wave-uniform
            qnnp::igemm_stage_tile<decltype(shift0)::value, decltype(full)::value>(
                acc[tm][tn], bias4[tn], rowterm, lds_out + row * OUT_PITCH, (wn * TN + tn) * 32, frag_khalf, p);
          }
        }
      });
      __syncthreads();
      const uint32_t m0 = m_tile * BM;
      const uint32_t n0 = n_tile * BN;
      const uint32_t rows_valid = min(static_cast<uint32_t>(BM), p.rows - m0);
      const uint32_t n_valid = min(static_cast<uint32_t>(BN), p.n - n0);
      qnnp::igemm_copy_out<NT>(
          lds_out, OUT_PITCH, rows_valid, n_valid,
          p.output + static_cast<uint64_t>(m0) * p.output_stride + static_cast<uint64_t>(g) * p.n + n0,
          p.output_stride, tid);
    } else {
      qnnp::requant_dispatch_ofs(p.rq, [&](auto shift0, auto full) {
#pragma unroll
        for (int tm = 0; tm < TM; tm++) {
          const uint32_t row = frag_row0 + tm * 32;
          const uint32_t m = m_tile * BM + row;
          const int32_t rowterm = qnnp::with_rq_offset<decltype(shift0)::value>(p.row_coeff * lds_rowsum[row]);
          uint64_t out_pixel = m;
          if (p.out_rows != nullptr && m < p.rows) {          // scattered rows (deconvolution phases)
            const uint32_t img_m = m / p.rows_per_image;
            out_pixel = static_cast<uint64_t>(img_m) * p.out_image_rows +
                        static_cast<uint32_t>(p.out_rows[m - img_m * p.rows_per_image]);
          }
          uint8_t* out_row = p.output + out_pixel * p.output_stride + static_cast<uint64_t>(g) * p.n;
#pragma unroll
          for (int tn = 0; tn < TN; tn++) {
            const uint32_t nb = nb0 + tn;
            if (nb >= nblocks) continue;       // wave-uniform
            qnnp::igemm_store_tile<decltype(shift0)::value, decltype(full)::value>(
                acc[tm][tn], bias4[tn], rowterm, out_row, nb * 32, frag_khalf, m < p.rows, p);
          }
        }
      });
    }
  }
}

template <int WM, int WN, int TM, int TN, int VEC, bool IS_CONV, bool PAD3 = false>
int launch_generic(const IgemmParams& p, uint32_t groups, hipStream_t stream)
{
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  const uint32_t tiles_m = (p.rows + BM - 1) / BM;
  const uint32_t tiles_n = (p.n_pad + BN - 1) / BN;
  // persistent over row tiles: exactly as many workgroups as are co-resident on the chip, each walking
  // tiles_m / ctas_m row tiles
  static int blocks_per_cu = 0;       // per instantiation; benign race (same value)
  if (blocks_per_cu == 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
            &nb, q8_igemm_mfma_kernel<WM, WN, TM, TN, VEC, IS_CONV, PAD3>, WM * WN * 64, 0) != hipSuccess || nb < 1) {
      (void) hipGetLastError();
      nb = 2;
    }
    blocks_per_cu = nb;
  }
  const uint32_t target = p.cu_count * static_cast<uint32_t>(blocks_per_cu);
  uint32_t ctas_m = target / (tiles_n * groups);
  if (ctas_m < 1) ctas_m = 1;
  if (ctas_m > tiles_m) ctas_m = tiles_m;
  const dim3 grid(ctas_m * tiles_n, groups, 1);     // (with a phase table `groups` is the number of phases)
  const dim3 block(WM * WN * 64, 1, 1);
  hipLaunchKernelGGL((q8_igemm_mfma_kernel<WM, WN, TM, TN, VEC, IS_CONV, PAD3>), grid, block, 0, stream, p);
  return hipGetLastError() == hipSuccess ? QNNP_HIP_OK : QNNP_HIP_ELAUNCH;
}

template <int VEC, bool IS_CONV, bool PAD3 = false>
int dispatch_tile(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name)
{
  if (p.n_pad <= 32) {
    *name = PAD3 ? "q8_igemm_mfma_128x32_c3" : "q8_igemm_mfma_128x32";
    return launch_generic<4, 1, 1, 1, VEC, IS_CONV, PAD3>(p, groups, stream);
  }
  if (p.n_pad <= 64) {
    *name = PAD3 ? "q8_igemm_mfma_128x64_c3" : "q8_igemm_mfma_128x64";
    return launch_generic<4, 1, 1, 2, VEC, IS_CONV, PAD3>(p, groups, stream);
  }
  *name = PAD3 ? "q8_igemm_mfma_128x128_c3" : "q8_igemm_mfma_128x128";
  return launch_generic<2, 2, 2, 2, VEC, IS_CONV, PAD3>(p, groups, stream);
}

template <bool IS_CONV>
int dispatch_vec(const IgemmParams& p, uint32_t groups, uint32_t vec, hipStream_t stream, const char** name)
{
  switch (vec) {
    case 16: return dispatch_tile<16, IS_CONV>(p, groups, stream, name);
    case 8: return dispatch_tile<8, IS_CONV>(p, groups, stream, name);
    case 4: return dispatch_tile<4, IS_CONV>(p, groups, stream, name);
    default: return dispatch_tile<1, IS_CONV>(p, groups, stream, name);
  }
}

}  // namespace

extern "C" int qnnp_hip_igemm_run(const struct qnnp_hip_igemm_args* a, const char** kernel_name)
{
  if (a == nullptr || a->rows == 0 || a->groups == 0 || a->groups > 65535u) return QNNP_HIP_EINVAL;
  if (a->n_pad % 32 != 0 || a->k_pad % BK != 0 || a->k_pad < a->k_total) return QNNP_HIP_EINVAL;

  IgemmParams p;
  p.input = a->input;
  p.output = a->output;
  p.packed_w = a->packed_w;
  p.bias2 = a->bias2;
  p.offsets = a->offsets;
  p.d2s_sh = a->d2s_stride_h;
  p.d2s_sw = a->d2s_stride_w;
  p.d2s_in_h = a->d2s_input_h;
  p.d2s_in_w = a->d2s_input_w;
  p.d2s_nbpp = 0;
  if (a->d2s_stride_h != 0) {
    const uint32_t phases = a->d2s_stride_h * a->d2s_stride_w;
    if (a->offsets != nullptr || a->groups != 1 || phases == 0 || a->n_pad % (32u * phases) != 0 ||
        a->d2s_input_h == 0 || a->d2s_input_w == 0) return QNNP_HIP_EINVAL;
    p.d2s_nbpp = a->n_pad / 32u / phases;
  }
  p.out_rows = a->out_rows;
  p.out_image_rows = a->out_image_rows;
  p.phases = a->phases;
  p.phase_groups = a->groups;
  if (a->phases != nullptr && (a->nphases == 0 || a->nphases * a->groups > 65535u || a->variant != 1 ||
                               a->offsets == nullptr)) return QNNP_HIP_EINVAL;
  if (a->out_rows != nullptr && a->phases == nullptr && (a->offsets == nullptr || a->variant != 1 || a->rows_per_image == 0)) return QNNP_HIP_EINVAL;
  p.rows = a->rows;
  p.rows_per_image = a->rows_per_image;
  p.image_stride = a->image_stride;
  p.n = a->n;
  p.n_pad = a->n_pad;
  p.kc = a->kc_slot;
  p.input_end = a->input + a->input_bytes;
  p.ks = a->ks;
  p.k_total = a->k_total;
  p.k_pad = a->k_pad;
  p.input_stride = a->input_stride;
  p.output_stride = a->output_stride;
  p.row_coeff = a->row_coeff;
  const bool pad3 = a->offsets != nullptr && a->kc == 3 && a->kc_slot == 4 && a->groups == 1;
  if (a->kc_slot != a->kc && !pad3) return QNNP_HIP_EINVAL;
  // padding taps read the input zero point; in 3-channel slot mode the slot's 4th byte is K padding (a' = 0)
  p.izp_fill = pad3 ? ((a->input_zero_point & 0xFFu) * 0x00010101u) | 0x80000000u
                    : (a->input_zero_point & 0xFFu) * 0x01010101u;
  p.rq = qnnp::make_requant_dev(a->rq);
  p.lane = qnnp::make_requant_lane(a->rq);
  p.bias2u = a->bias2_pair != 0 ? a->bias2 + static_cast<size_t>(a->groups) * a->n_pad : nullptr;
  if (p.bias2u == nullptr) p.lane.kind = 0;      // no table to start from: the offset forms
  p.stream_out = a->streaming_mode == 0 ? (qnnp_hip_streaming_stores() != 0 ? 1u : 0u) : (a->streaming_mode == 2 ? 1u : 0u);
  p.a_flip = 0;
  // a table row per output pixel with its one tap inside the image: a strided 1x1 convolution without padding taps
  p.offsets_dense = (a->offsets != nullptr && a->ks == 1 && a->kernel_height == 1 && a->kernel_width == 1 && a->groups == 1 &&
                     a->pad_top == 0 && a->pad_left == 0 && a->rows_per_image > 0 && a->output_height > 0 && a->output_width > 0 &&
                     static_cast<uint64_t>(a->output_height - 1) * a->stride_height < a->input_height &&
                     static_cast<uint64_t>(a->output_width - 1) * a->stride_width < a->input_width &&
                     a->rows_per_image == a->output_height * a->output_width) ? 1u : 0u;
  // (rows_per_image == 1 -- one output pixel per image, e.g. a 2x2 input at stride 2 -- has no 32-bit magic: 2^32 / 1 + 1
  //  wraps to 1 and hi32(m * 1) is 0 for every row. It takes the divide, which is m / 1.)
  p.rpi_magic = (p.offsets_dense != 0 && a->rows_per_image > 1 &&
                 static_cast<uint64_t>(a->rows) * a->rows_per_image < (UINT64_C(1) << 32))
                    ? static_cast<uint32_t>((UINT64_C(1) << 32) / a->rows_per_image) + 1u : 0u;
  p.fill_table = qnnp_hip_fill_table();
  p.trace = nullptr;
#ifdef QNNP_ENABLE_ABLATION
  p.trace = static_cast<unsigned long long*>(qnnp_hip_trace_buffer());
#endif
  {
    const int cus = qnnp_hip_compute_units();
    p.cu_count = cus > 0 ? static_cast<uint32_t>(cus) : 256u;
  }

  // widest activation vector the actual alignment allows (a vector never straddles a tap)
  const uintptr_t in_addr = reinterpret_cast<uintptr_t>(a->input);
  uint32_t vec = 1;
  const uint32_t candidates[3] = {16u, 8u, 4u};
  for (uint32_t v : candidates) {
    if (a->kc % v == 0 && a->input_stride % v == 0 && in_addr % v == 0) {
      vec = v;
      break;
    }
  }
  if (pad3) vec = 4;   // one (unaligned) dword per tap
  const uintptr_t out_addr = reinterpret_cast<uintptr_t>(a->output);
  p.store_mode = 0;
  if (a->n % 16 == 0 && a->output_stride % 16 == 0 && out_addr % 16 == 0) {
    p.store_mode = 2;
  } else if (a->n % 4 == 0 && a->output_stride % 4 == 0 && out_addr % 4 == 0) {
    p.store_mode = 1;
  }

  // Fused residual add: carried by the pointwise streaming kernels' epilogues (what a MobileNet-style project layer runs
  // on) when the residual rows are laid out like the output rows; everything else reports "not folded" and the caller
  // adds in place with the stand-alone kernel.
  p.residual = nullptr;
  p.residual_stride = 0;
  if (a->residual_folded != nullptr) *a->residual_folded = 0;
  if (a->residual != nullptr) {
    if (a->residual_add == nullptr || a->residual_folded == nullptr) return QNNP_HIP_EINVAL;
    const uintptr_t res_addr = reinterpret_cast<uintptr_t>(a->residual);
    const uint32_t align = p.store_mode == 2 ? 16u : 4u;
    if (p.store_mode != 0 && a->groups == 1 && a->d2s_stride_h == 0 && a->offsets == nullptr &&
        a->residual_stride == a->output_stride && res_addr % align == 0) {
      p.residual = a->residual;
      p.residual_stride = a->residual_stride;
      p.add = *a->residual_add;
    }
  }
  auto folded = [&]() { if (p.residual != nullptr) *a->residual_folded = 1; };

  hipStream_t stream = reinterpret_cast<hipStream_t>(qnnp_hip_get_stream());
  const char* name = nullptr;
  // Dense convolutions with power-of-two channel counts: LDS-tiled direct convolution (input read once).
  qnnp::ConvGeom geom;
  geom.H = a->input_height; geom.W = a->input_width; geom.OH = a->output_height; geom.OW = a->output_width;
  geom.KH = a->kernel_height; geom.KW = a->kernel_width; geom.sh = a->stride_height; geom.sw = a->stride_width;
  geom.dh = a->dilation_height; geom.dw = a->dilation_width; geom.pad_top = a->pad_top; geom.pad_left = a->pad_left;
  const bool lds_ok = a->offsets != nullptr && !pad3 && a->rows_per_image > 0 &&
      a->output_stride % 16 == 0 && qnnp::convlds_supported(p, geom, a->groups, vec);
  // 3x3 / stride 1 / dilation 1 windows with 32 or 64 channels in and out (BASELINE configs[2]): one wave per 8x8 block of
  // positions, patches streamed by LDS-DMA, no barriers (q8convwave.hip) -- 31 us against 37.7 us for the LDS-tiled
  // kernel on configs[2], same box. Other windows the wave kernel accepts run on it only when forced ("gemm_kernel" = 8).
  const bool wave_shape = a->offsets != nullptr && !pad3 && a->rows_per_image > 0 && p.store_mode == 2 &&
      qnnp::convwave_supported(p, geom, a->groups, vec, a->rows / a->rows_per_image);
  const bool wave_k33 = geom.KH == 3 && geom.KW == 3 && geom.sh == 1 && geom.sw == 1 && geom.dh == 1 && geom.dw == 1;
  // ("gemm_kernel" = 12: the same family with the round-2 register-path kernel instead of the weight-stationary one)
  const bool wave_forced = a->variant == 8 || a->variant == 12 || a->variant == 27;   // 27: the 32x32x32 weight-stationary kernel (A/B)
  const bool wave_ok = wave_shape && (wave_forced || (a->variant == 0 && wave_k33 && a->rows >= 16384u));
  if (wave_forced && !wave_ok) return QNNP_HIP_EINVAL;
  if (wave_ok) {
    // kernel zero points 127 / 128: the zero-point-centred image (convolution.c builds it for single-group convolutions
    // without K padding), taken by the weight-stationary kernel
    IgemmParams pc = p;
    const bool centred = a->centre_flip != 0 && a->packed_w_centred != nullptr && a->bias2_centred != nullptr &&
        a->bias2_pair != 0 && a->groups == 1;
    if (centred) {
      pc.packed_w = a->packed_w_centred;
      pc.bias2 = a->bias2_centred;
      pc.bias2u = a->bias2_centred + static_cast<size_t>(a->groups) * a->n_pad;
      pc.a_flip = (a->centre_flip & 0xFFu) * 0x01010101u;
      pc.row_coeff = 0;
    }
    const int rc_wave = qnnp::convwave_launch(p, geom, a->rows / a->rows_per_image, stream, &name, a->variant == 12 ? 1 : (a->variant == 27 ? 2 : 0),
                                              centred ? &pc : nullptr);
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_wave;
  }
  // (round 6) dense 3x3 / stride 1 with 16 / 32 / 48 / 64 input channels and a centred image (SqueezeNet's fire modules): the weight-stationary
  // kernel with the channel count as a template argument (q8convws16s.hip); "gemm_kernel" = 32 forces it, 1 / 3 / 22 keep what it replaces
  {
    IgemmParams ps = p;
    const bool centred = a->centre_flip != 0 && a->packed_w_centred != nullptr && a->bias2_centred != nullptr &&
        a->bias2_pair != 0 && a->groups == 1;
    if (centred) {
      ps.packed_w = a->packed_w_centred;
      ps.bias2 = a->bias2_centred;
      ps.bias2u = a->bias2_centred + static_cast<size_t>(a->groups) * a->n_pad;
      ps.a_flip = (a->centre_flip & 0xFFu) * 0x01010101u;
      ps.row_coeff = 0;
    }
    const bool s_ok = centred && a->offsets != nullptr && !pad3 && a->rows_per_image > 0 &&
        qnnp::convws16s_supported(ps, geom, a->groups, vec, a->rows / a->rows_per_image);
    if (a->variant == 32 && !s_ok) return QNNP_HIP_EINVAL;
    // (auto: everything but 64 -> 32 / 64, which the block above has taken; 64 -> 256 leaves the patch kernel: 27 x 27 24.7 -> 19.5 us,
    //  13 x 13 9.4 -> 9.0: profiles/r06/conv3x3_small_channels_r06y.txt)
    if (s_ok && (a->variant == 32 || (a->variant == 0 && a->rows >= 16384u))) {
      const int rc_s = qnnp::convws16s_launch(ps, geom, a->rows / a->rows_per_image, stream, &name);
      if (kernel_name != nullptr) *kernel_name = name;
      return rc_s;
    }
  }
  // Dense 3x3 with many channels (ResNet's 128 / 256 / 512-channel layers): patch in LDS, weights streamed (q8convpatch.hip);
  // "gemm_kernel" = 22 forces it, 1 / 2 / 3 keep the kernels it replaces.
  const bool patch_ok = a->offsets != nullptr && !pad3 && a->rows_per_image > 0 &&
      qnnp::convpatch_supported(p, geom, a->groups, vec, a->rows / a->rows_per_image);
  if (a->variant == 22 && !patch_ok) return QNNP_HIP_EINVAL;
  if (patch_ok && (a->variant == 22 || (a->variant == 0 && a->rows >= 4096u))) {
    const int rc_patch = qnnp::convpatch_launch(p, geom, a->rows / a->rows_per_image, stream, &name);
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_patch;
  }
  if (a->variant == 3 && !lds_ok) return QNNP_HIP_EINVAL;
  if (lds_ok && (a->variant == 3 || (a->variant == 0 && a->kernel_height * a->kernel_width > 1))) {
    const int rc_lds = qnnp::convlds_launch(p, geom, a->rows / a->rows_per_image, stream, &name);
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_lds;
  }
  // 3-channel images (first layers) with dense pixels: one 16-byte fetch per kernel row (q8convc3.hip);
  // "gemm_kernel" = 14 forces it, 7 keeps the tap-gather kernel below
  // (store_mode 1 with channels % 8 == 0: ShuffleNet's 3 -> 24 layer -- the kernel checks the 8-byte alignment it needs itself)
  const bool rows16_ok = pad3 && (p.store_mode == 2 || (p.store_mode == 1 && a->n % 8u == 0)) &&
      qnnp::conv_c3rows_supported(p, geom, a->groups, a->packed_w_rows16, a->kc);
  if ((a->variant == 14 || a->variant == 30) && !rows16_ok && !(pad3 && p.store_mode == 2 && qnnp::conv_c3rows32_supported(p, geom, a->groups, a->packed_w_rows16, a->kc))) return QNNP_HIP_EINVAL;
  if (rows16_ok && (a->variant == 14 || a->variant == 30 || (a->variant == 0 && a->rows >= 2048))) {
    qnnp::IgemmParams p16 = p;
    if (a->bias2_rows != nullptr) {          // the image is centred on kernel zero point 127: its own bias pair, no row term
      p16.bias2 = a->bias2_rows;
      p16.bias2u = a->bias2_rows + a->n_pad;
      p16.row_coeff = 0;
      p16.a_flip = 0x7F7F7F7Fu;
    }
    const int rc_r16 = qnnp::conv_c3rows_launch(p16, geom, a->packed_w_rows16, stream, &name, a->variant == 14 ? 1 : (a->variant == 30 ? 2 : 0));
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_r16;
  }
  // ... and its 32-byte-slot flavour for 5- and 7-row windows (ResNet's 7x7 entry layer); "gemm_kernel" = 14 forces it too
  const bool rows32_ok = pad3 && p.store_mode == 2 && qnnp::conv_c3rows32_supported(p, geom, a->groups, a->packed_w_rows16, a->kc);
  // (30 = its LDS-staged flavour or nothing; 14 = the register-path kernel; auto: the LDS-staged one where its plan takes the shape)
  if (a->variant == 30 && !rows32_ok) return QNNP_HIP_EINVAL;
  if (rows32_ok && (a->variant == 14 || a->variant == 30 || (a->variant == 0 && a->rows >= 2048))) {
    qnnp::IgemmParams p32 = p;
    if (a->bias2_rows != nullptr) {          // the image is centred on kernel zero point 127: its own bias pair, no row term
      p32.bias2 = a->bias2_rows;
      p32.bias2u = a->bias2_rows + a->n_pad;
      p32.row_coeff = 0;
      p32.a_flip = 0x7F7F7F7Fu;
    }
    const int rc_r32 = qnnp::conv_c3rows32_launch(p32, geom, a->packed_w_rows16, stream, &name, a->variant == 14 ? 1 : (a->variant == 30 ? 2 : 0));
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_r32;
  }
  // 3-channel images (first layers): barrier-free streaming kernel with an in-register tap gather.
  const bool c3_ok = pad3 && qnnp::convstream_c3_supported(p, a->groups);
  if (a->variant == 7 && !c3_ok) return QNNP_HIP_EINVAL;
  if (c3_ok && (a->variant == 7 || (a->variant == 0 && a->rows >= 2048))) {
    const int rc_c3 = qnnp::convstream_c3_launch(p, stream, &name);
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_c3;
  }
  // Mid-size GEMMs on the zero-point-centred image: 128-row tiles of four waves, two or three workgroups per CU (q8gemm128x.hip,
  // round 6); "gemm_kernel" = 24 forces it (25 / 26: with 64- / 128-channel tiles). Where auto takes it (measured at batch 128,
  // profiles/r06/mid_gemm_by_forced_kernel_r06d.txt) is decided below, next to the kernel it replaces.
  qnnp::IgemmParams pmid = p;
  bool mid_ok = !pad3 && a->centre_flip != 0 && a->packed_w_centred != nullptr && a->bias2_centred != nullptr && a->bias2_pair != 0 &&
      a->groups == 1 && p.d2s_sh == 0;
  if (mid_ok) {
    pmid.packed_w = a->packed_w_centred;
    pmid.bias2 = a->bias2_centred;
    pmid.bias2u = a->bias2_centred + static_cast<size_t>(a->groups) * a->n_pad;
    pmid.a_flip = (a->centre_flip & 0xFFu) * 0x01010101u;
    pmid.row_coeff = 0;
    mid_ok = qnnp::gemm128x_supported(pmid, vec);
  }
  const bool mid_forced = a->variant == 24 || a->variant == 25 || a->variant == 26;     // 25 / 26: 64- / 128-wide tiles (A/B)
  if (mid_forced && !mid_ok) return QNNP_HIP_EINVAL;
  auto launch_mid = [&]() {
    const int rc_mid = qnnp::gemm128x_launch(pmid, a->groups, stream, &name, a->variant == 25 ? 64u : (a->variant == 26 ? 128u : 0u));
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_mid;
  };
  if (mid_forced) return launch_mid();
  // (strided 1x1 convolutions over few rows -- ResNet-18's 28x28 128 -> 256 and 14x14 256 -> 512 shortcuts, 25 k / 6 k output pixels:
  //  7.9 -> 6.1 us and 8.2 -> 5.4 against the streaming kernel's table rows; with 100 k rows the streaming kernel keeps them)
  if (a->variant == 0 && mid_ok && a->offsets != nullptr && a->rows <= 32768u && a->k_total >= 128u && a->k_total <= 256u && a->rows >= 2048u) return launch_mid();
  // (round 6: pointwise layers whose channel count forces BYTE stores on the streaming kernel -- ShuffleNet v2's 24 -> 58 / 122 at 56 x 56:
  //  56.3 / 122.1 us -- run on the register-staged 128-row GEMM instead: 36.3 / 78.6 us, profiles/r06/ugemm_by_forced_kernel_r06p.txt)
  // (... and, with the transposed 16-byte stores of that kernel, the streaming kernel's DWORD-store shapes -- channel counts of 4 mod 8 --
  //  from 56 channels up: 56 x 56 24 -> 60 / 68 20.9 / 24.5 -> 17.1 / 22.4 us (profiles/r06/ugemm_transposed_stores_r06v.txt); 24 -> 36
  //  stays, 12.0 against 13.8, and so do multiples of 8 -- 24 -> 88 12.9 against 23.8, 88 -> 88 8.2 against 10.0: run r06w's lists)
  // (SqueezeNet's 13 x 13 512 -> 1000 with a centred image: the 128-row centred GEMM with dword-aligned stores, 25.7 against 27.8 us)
  if (a->variant == 0 && mid_ok && p.store_mode == 1 && a->n >= 256u && a->k_total >= 256u && a->rows >= 2048u && a->offsets == nullptr) return launch_mid();
  // (... and wide ones: a channel run of 360 -- ShuffleNet v1 g8's 12 -> 45 as a dense 96 -> 360 -- 31.3 us on the streaming kernel, 21.0 here;
  //  SqueezeNet's 13 x 13 512 -> 1000 35.1 us on the 256-wide GEMM's padding path, 25.5 here: run r06z's lists against r06w's)
  if (a->variant == 0 && !pad3 && (p.store_mode == 0 || (p.store_mode == 1 && ((a->n % 8u != 0u && a->n >= 56u) || a->n >= 256u))) && a->rows >= 2048u && p.d2s_sh == 0 &&
      a->residual == nullptr && qnnp::gemm128u_supported(p)) {
    const int rc_u = qnnp::gemm128u_launch(p, a->groups, stream, &name, 0u);
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_u;
  }
  // Short-K pointwise / fully-connected layers over many rows: barrier-free streaming kernel.
  const bool pw_ok = !pad3 && qnnp::pwstream_supported(p, a->groups, vec) && (p.d2s_sh == 0 || vec == 16);
  if ((a->variant == 5 || p.d2s_sh != 0) && !pw_ok) return QNNP_HIP_EINVAL;   /* depth-to-space exists in this kernel only */
  if (p.d2s_sh != 0 && a->variant != 5) return QNNP_HIP_EINVAL;
  // (round 6: few rows x many channels -- ShuffleNet's last 1x1, 7x7 192 -> 1024 at 6 k rows: the streaming kernel runs one chain per
  //  wave, 8.3 us; the 128-row GEMMs 6.0 (centred image) / 6.8 us (standard image): profiles/r06/ugemm_transposed_stores_r06v.txt)
  const bool few_rows_wide = a->variant == 0 && !pad3 && a->rows >= 2048u && a->rows <= 8192u && a->n >= 512u && a->k_total >= 192u &&
      p.d2s_sh == 0 && a->residual == nullptr && a->offsets == nullptr;
  if (few_rows_wide && pw_ok && mid_ok) return launch_mid();
  if (few_rows_wide && pw_ok && qnnp::gemm128u_supported(p)) {
    const int rc_u = qnnp::gemm128u_launch(p, a->groups, stream, &name, 0u);
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_u;
  }
  if (pw_ok && (a->variant == 5 || (a->variant == 0 && a->rows >= 2048))) {
    const int rc_pw = qnnp::pwstream_launch(p, vec, stream, &name);
    if (kernel_name != nullptr) *kernel_name = name;
    if (rc_pw == QNNP_HIP_OK) folded();
    return rc_pw;
  }
  // Long reductions over few rows with 16-byte aligned rows on both sides: weights of a channel column in LDS, every
  // K block of a unit's rows in flight at once ("gemm_kernel" = 9 forces it).
  const bool lk_ok = !pad3 && qnnp::pwstream_longk_supported(p, a->groups, vec);
  if (a->variant == 9 && !lk_ok) return QNNP_HIP_EINVAL;
  // (auto where it measured ahead, batch 128 MobileNetV2: the 14x14 project layers -- 784 row blocks, K = 384 / 576:
  //  8.6 against 10.1 us, 10.8 against 13.0 -- and 7x7x320 -> 1280 with its 40 channel blocks, 11.1 against 12.5 on
  //  the tiled kernel; with ~200 row blocks and few channel blocks the one-wave-per-block kernel below keeps the lead:
  //  7x7x960 -> 160 11.1 against 6.6 us)
  const uint32_t lk_units = (a->rows + 31u) / 32u;
  // (round 5: problems the 256-wide LDS-DMA GEMM takes -- N >= 256, K >= 512, rows >= 2048 -- go THERE, not to the long-K / one-wave
  //  kernels: ResNet-50's 14x14 1024 -> 256 25.4 -> 14.6 us, 7x7 512 -> 2048 28.8 -> 12.0, 7x7 2048 -> 512 41.4 -> 22.4, same box,
  //  profiles/r05/pointwise_rows_by_forced_kernel_r05var.txt)
  //  -- when the channels fill whole 256-wide tiles: MobileNetV2's 7x7 960 -> 320 (a quarter of its second tile is padding) stays
  //  on the one-wave kernel, 11.2 against 13.0 us)
  const bool big_first = a->variant == 0 && !pad3 && a->n >= 256 && a->n % 256u == 0 && a->k_total >= 512 && a->rows >= 2048 &&
      qnnp::gemm256_supported(p, vec);
  // (round 5: many rows with FEW channels -- ResNet-50's 28x28 512 -> 128: the weights of the whole row fit LDS and the flavour with
  //  the next unit's rows in flight streams them; 35 us on the generic tile kernel before)
  const bool lk_many_rows = a->rows > 65536u && a->n_pad <= 128u && a->k_total <= 640u;
  const bool lk_auto = !big_first && (a->rows <= 65536u || lk_many_rows) && (lk_units >= 512u || a->n_pad >= 512u);
  // (round 6: what the long-K kernel took automatically goes to the 128-row centred GEMM where that exists -- MobileNetV2's 14x14 project
  //  layers 7.2 -> 5.6, 8.1 -> 6.2, 9.7 -> 7.3 us, 7x7x320 -> 1280 12.4 -> 7.9, ResNet-50's 28x28 512 -> 128 32.6 -> 21.2)
  if (a->variant == 0 && lk_ok && lk_auto && mid_ok && a->rows >= 2048u) return launch_mid();
  // (... and without a centred image -- K = 464: ShuffleNet v2 x1.0's 7x7 464 -> 1024 -- to the register-staged one: 13.5 -> 10.7 us)
  if (few_rows_wide && lk_ok && lk_auto && qnnp::gemm128u_supported(p)) {
    const int rc_u = qnnp::gemm128u_launch(p, a->groups, stream, &name, 0u);
    if (kernel_name != nullptr) *kernel_name = name;
    return rc_u;
  }
  if (lk_ok && (a->variant == 9 || (a->variant == 0 && lk_auto))) {
    const int rc_lk = qnnp::pwstream_longk_launch(p, stream, &name);
    if (kernel_name != nullptr) *kernel_name = name;
    if (rc_lk == QNNP_HIP_OK) folded();
    return rc_lk;
  }
  // Small problems with a long reduction (late MobileNet layers, classifier heads): one wave per 32x32 block,
  // operands from L2 -- the tiled kernels would launch fewer workgroups than there are CUs.
  const bool gw_ok = !pad3 && qnnp::pwstream_gw_supported(p, a->groups, vec);
  if (a->variant == 6 && !gw_ok) return QNNP_HIP_EINVAL;
  // (selected when the 128-row x 128-channel tiling of the generic kernel would not even give ~1.5 workgroups
  //  per CU; MobileNetV2 layer 30 -- 490 tiles -- measured faster on the tiled kernel, layers 19-29 on this one)
  const uint64_t generic_tiles = static_cast<uint64_t>((a->rows + 127u) / 128u) * ((a->n_pad + 127u) / 128u);
  // (round 6: with >= 256 channels the 128-row centred GEMM is ahead -- 7x7x960 -> 320 10.5 -> 8.0 us; 160 channels stay here, 5.9 / 7.0
  //  against 6.0 / 7.8)
  if (a->variant == 0 && gw_ok && generic_tiles <= 400u && !big_first && mid_ok && a->n >= 256u && a->rows >= 2048u) return launch_mid();
  if (gw_ok && (a->variant == 6 || (a->variant == 0 && generic_tiles <= 400u && !big_first))) {
    const int rc_gw = qnnp::pwstream_gw_launch(p, stream, &name);
    if (kernel_name != nullptr) *kernel_name = name;
    if (rc_gw == QNNP_HIP_OK) folded();
    return rc_gw;
  }
  // Large MFMA-bound problems take the 256x256 LDS-DMA kernel; everything else the generic one.
  const bool big_ok = !pad3 && qnnp::gemm256_supported(p, vec);
  // (strided 1x1 convolutions -- a table row per output pixel, one tap -- from K = 256: ResNet-50's 56x56 stride-2 256 -> 512
  //  70.8 -> 51.3 us on the offset-table flavour of the 256-wide kernel)
  const bool strided_pw = a->offsets != nullptr && a->ks == 1 && a->n % 256u == 0 && a->k_total >= 256;
  const bool big_auto = a->n >= 256 && (a->k_total >= 512 || strided_pw) && a->rows >= 2048;
  const bool big_forced = a->variant == 2 || a->variant == 4 || a->variant == 10 || a->variant == 11 || a->variant == 15 || a->variant == 16;   // 10: 128 x 256 tiles, two workgroups per CU; 11: ping-pong schedule; 15: lean flavour
  if (big_forced && !big_ok) return QNNP_HIP_EINVAL;
  int rc;
  // Operators with a zero-point-centred weight image (kernel zero point 127 or 128, q8gemm256c.hip): no row term at all.
  // Round 6: auto takes the v_mfma_i32_16x16x64_i8 flavour (q8gemm256x.hip: 59.1 -> 52.6 us on 4096^3, same box, interleaved;
  // "gemm_kernel" 23 forces it); 20 keeps the 32x32x32 one (q8gemm256c.hip), 21 = that one's A/B structure (fragment reads in one burst).
  const bool c_forced = a->variant == 20 || a->variant == 21 || a->variant == 23;
  // (round 6: launches of at most ~100 tiles of 256 x 256 leave most CUs idle for the length of a long K loop -- ResNet-50's 7x7 2048 -> 512,
  //  50 tiles: 22.1 -> 15.0 us on 128-row tiles, 14x14 1024 -> 256, 98 tiles: 17.1 -> 16.0; from ~200 tiles on the wide kernel leads)
  if (a->variant == 0 && big_auto && mid_ok &&
      static_cast<uint64_t>((a->rows + 255u) / 256u) * ((a->n_pad + 255u) / 256u) <= 100u) return launch_mid();
  if (c_forced || (a->variant == 0 && big_auto && a->centre_flip != 0)) {
    qnnp::IgemmParams pc = p;
    const uint32_t opt = a->variant == 21 ? 2u : 0u;
    bool c_ok = a->centre_flip != 0 && a->packed_w_centred != nullptr && a->bias2_centred != nullptr && a->bias2_pair != 0;
    if (c_ok) {
      pc.packed_w = a->packed_w_centred;
      pc.bias2 = a->bias2_centred;
      pc.bias2u = a->bias2_centred + static_cast<size_t>(a->groups) * a->n_pad;
      pc.a_flip = (a->centre_flip & 0xFFu) * 0x01010101u;
      pc.row_coeff = 0;
      c_ok = big_ok && qnnp::gemm256c_supported(pc, vec);
    }
    if (c_ok) {
      rc = (a->variant == 23 || a->variant == 0) ? qnnp::gemm256x_launch(pc, a->groups, stream, &name)
                            : qnnp::gemm256c_launch(pc, a->groups, stream, &name, opt);
      if (kernel_name != nullptr) *kernel_name = name;
      return rc;
    }
    if (c_forced) return QNNP_HIP_EINVAL;
  }
  // Round 6: every OTHER kernel zero point on the 16x16x64 kernel as well -- the standard image and its row term (ROWSUM flavour of
  // q8gemm256x.hip; "gemm_kernel" 28 forces it, 15 keeps the lean 32x32x32 kernel it replaces).
  if (a->variant == 28 || (a->variant == 0 && big_auto && big_ok)) {
    qnnp::IgemmParams pr = p;
    pr.a_flip = 0x80808080u;
    const bool r_ok = big_ok && p.bias2u != nullptr && a->groups >= 1 && qnnp::gemm256c_supported(pr, vec);
    if (r_ok) {
      rc = qnnp::gemm256x_launch(pr, a->groups, stream, &name);
      if (kernel_name != nullptr) *kernel_name = name;
      return rc;
    }
    if (a->variant == 28) return QNNP_HIP_EINVAL;
  }
  // Round 6: what is left of the 1x1 / fully connected class -- grouped, odd channel counts, unaligned rows -- on the 128 x 128 kernel
  // that stages its activation tile through registers (q8gemm128u.hip; "gemm_kernel" 29 forces it, 1 keeps the generic tile kernel).
  {
    const bool u_ok = !pad3 && qnnp::gemm128u_supported(p);
    if (a->variant == 29 && !u_ok) return QNNP_HIP_EINVAL;
    if (u_ok && (a->variant == 29 || (a->variant == 0 && !(big_ok && big_auto)))) {
      rc = qnnp::gemm128u_launch(p, a->groups, stream, &name, 0u);
      if (kernel_name != nullptr) *kernel_name = name;
      return rc;
    }
  }
  if (big_ok && (big_forced || (a->variant == 0 && big_auto))) {
    rc = qnnp::gemm256_launch(p, a->groups, stream, &name, a->variant == 4 || a->variant == 16, a->variant == 10, a->variant == 11,
                               (a->variant == 15 || a->variant == 16) ? 2 : (a->variant == 0 ? 1 : 0));   // 16: the 4-wave flavour, lean;   // ("gemm_kernel" = 2 keeps the general flavour for A/B)
  } else {
    if (pad3) {
      rc = dispatch_tile<4, true, true>(p, a->groups, stream, &name);
    } else {
      rc = (a->offsets != nullptr) ? dispatch_vec<true>(p, a->phases != nullptr ? a->nphases * a->groups : a->groups, vec, stream, &name)
                                   : dispatch_vec<false>(p, a->groups, vec, stream, &name);
    }
  }
  if (kernel_name != nullptr) *kernel_name = name;
  return rc;
}
