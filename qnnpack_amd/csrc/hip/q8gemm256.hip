/*
 * q8gemm256.hip -- the large-problem uint8 GEMM / implicit-GEMM kernel:
 * 256x256 output tile per workgroup, operands staged by LDS-DMA, int8 MFMA.
 *
 * Same arithmetic and operand roles as q8igemm.hip (which see): it replaces the
 * same reference microkernels (src/q8gemm/4x4c2-sse2.c:14-318,
 * src/q8conv/4x4c2-sse2.c:14-273) for shapes big enough to be MFMA-bound, i.e.
 * BASELINE.json configs[1] (q8gemm M=N=K=4096).
 *
 * Structure (per workgroup, 8 waves as 2 (rows) x 4 (channels), 128 x 64 outputs per wave):
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
 *     taken with v_dot4 on those same registers, each of the 4 channel-waves doing
 *     one quarter of K, combined through LDS at the end.
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

#include "igemm_params.h"
#include "requant.cuh"

namespace qnnp {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int kBM = 256;
constexpr int kBN = 256;
constexpr int kBK = 128;                       // bytes of K per tile
constexpr int kATile = kBM * kBK;              // 32 KiB
constexpr int kWTile = kBN * kBK;              // 32 KiB
constexpr int kStage = kATile + kWTile;        // 64 KiB
constexpr int kThreads = 512;
constexpr uint32_t kFlip = 0x80808080u;

__device__ __forceinline__ void dma16(const uint8_t* src, uint8_t* lds_wave_base)
{
  // 16 bytes per lane, LDS destination = wave-uniform base + lane * 16
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*) src,
      (__attribute__((address_space(3))) void*) lds_wave_base, 16, 0, 0);
}

template <bool IS_CONV>
__global__ __launch_bounds__(kThreads, 2)
void q8_gemm_mfma_256x256_kernel(const IgemmParams p)
{
  // single LDS object: 2 stages of {A, W} tiles, then 4 x 256 partial row sums
  __shared__ __attribute__((aligned(16))) uint8_t lds[2 * kStage + 4 * kBM * 4];
  int32_t* lds_rowsum = reinterpret_cast<int32_t*>(lds + 2 * kStage);

  const uint32_t tid = threadIdx.x;
  const uint32_t lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t wm = wave >> 2;       // 0..1: 128-row half
  const uint32_t wn = wave & 3u;       // 0..3: 64-channel quarter
  const uint32_t g = blockIdx.y;

  const uint32_t tiles_n = (p.n_pad + kBN - 1) / kBN;
  uint32_t logical;
  {
    const uint32_t nwg = gridDim.x;
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t idx = blockIdx.x >> 3;
    const uint32_t q = nwg >> 3, r = nwg & 7u;
    logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const uint32_t n_tile = logical % tiles_n;
  const uint32_t m_tile = logical / tiles_n;

  const uint32_t nblocks = p.n_pad / 32;
  const uint32_t kblocks = p.k_pad / 32;
  const uint32_t ktiles = (p.k_pad + kBK - 1) / kBK;
  const uint8_t* pad_k = p.fill_table + 0x80 * 16;                       // a' == 0
  const uint8_t* pad_zp = p.fill_table + (p.izp_fill & 0xFFu) * 16;      // a == input zero point
  const uint8_t* pad_w = p.fill_table;                                   // w' == 0

  // ---- activation DMA assignment: 4 chunks per thread, LDS linear index L = i*512 + tid ----
  const uint8_t* a_row[4];      // gemm: row base (+ group); conv: image base (+ group)
  const int32_t* a_offs[4];     // conv: offset-table row of this pixel
  uint32_t a_chunk[4];          // logical 16-byte chunk (0..7) this lane fetches for its slot
  int32_t a_off_next[4];        // conv: prefetched table entry for the NEXT tile
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const uint32_t L = i * kThreads + tid;
    const uint32_t r = L >> 3;
    const uint32_t s = L & 7u;
    a_chunk[i] = s ^ ((r >> 1) & 7u);
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
    a_off_next[i] = 0;
  }

  // conv: table entry of (this lane's chunk, tile kt)
  auto conv_lookup = [&](uint32_t kt, int i) -> int32_t {
    const uint32_t kk = kt * kBK + a_chunk[i] * 16;
    if (kk >= p.k_total) return 0;
    return a_offs[i][kk / p.kc];
  };

  // ---- weight DMA assignment: 32 fragments of 1 KiB per tile, 4 per wave ----
  const uint32_t nb0 = n_tile * (kBN / 32);
  const uint8_t* w_group = reinterpret_cast<const uint8_t*>(p.packed_w) +
      static_cast<uint64_t>(g) * nblocks * kblocks * 1024 + lane * 16;

  auto stage = [&](uint32_t buf, uint32_t kt) {
    uint8_t* a_dst = lds + buf * kStage;
    uint8_t* w_dst = a_dst + kATile;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t kk = kt * kBK + a_chunk[i] * 16;
      const uint8_t* src;
      if constexpr (IS_CONV) {
        const uint32_t tap = kk / p.kc;
        const uint32_t ch = kk - tap * p.kc;
        const int32_t off = a_off_next[i];
        src = off >= 0 ? a_row[i] + off + ch : pad_zp;
      } else {
        src = a_row[i] + kk;
      }
      if (kk >= p.k_total) src = pad_k;
      dma16(src, a_dst + (i * kThreads + wave * 64) * 16);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t F = i * 8 + wave;             // fragment slot: (32-channel block, 32-deep K block)
      const uint32_t nb = nb0 + (F >> 2);
      const uint32_t kb = kt * 4 + (F & 3u);
      const uint8_t* src = (nb < nblocks && kb < kblocks)
          ? w_group + (static_cast<uint64_t>(nb) * kblocks + kb) * 1024
          : pad_w;
      dma16(src, w_dst + F * 1024);
    }
  };

  v16i acc[4][2];
#pragma unroll
  for (int tm = 0; tm < 4; tm++)
#pragma unroll
    for (int tn = 0; tn < 2; tn++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[tm][tn][r] = 0;
  int32_t rs[4] = {0, 0, 0, 0};

  const uint32_t frag_row0 = wm * 128 + (lane & 31u);
  const uint32_t frag_khalf = lane >> 5;

  if constexpr (IS_CONV) {
#pragma unroll
    for (int i = 0; i < 4; i++) a_off_next[i] = conv_lookup(0, i);
  }
  stage(0, 0);
  if constexpr (IS_CONV) {
    if (ktiles > 1) {
#pragma unroll
      for (int i = 0; i < 4; i++) a_off_next[i] = conv_lookup(1, i);
    }
  }

  for (uint32_t kt = 0; kt < ktiles; kt++) {
    const uint32_t buf = kt & 1u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile kt (and the prefetched table entries) landed
    __syncthreads();                                    // ... for every wave; stage buf^1 is free again
    if (kt + 1 < ktiles) {
      stage(buf ^ 1u, kt + 1);
      if constexpr (IS_CONV) {
        if (kt + 2 < ktiles) {
#pragma unroll
          for (int i = 0; i < 4; i++) a_off_next[i] = conv_lookup(kt + 2, i);
        }
      }
    }
    const uint8_t* a_lds = lds + buf * kStage;
    const uint8_t* w_lds = a_lds + kATile + lane * 16;
#pragma unroll
    for (int ksub = 0; ksub < 4; ksub++) {
      v4i wf[2];
#pragma unroll
      for (int tn = 0; tn < 2; tn++) {
        wf[tn] = *reinterpret_cast<const v4i*>(w_lds + ((wn * 2 + tn) * 4 + ksub) * 1024);
      }
      v4i af[4];
#pragma unroll
      for (int tm = 0; tm < 4; tm++) {
        const uint32_t row = frag_row0 + tm * 32;
        const uint32_t chunk = (ksub * 2 + frag_khalf) ^ ((row >> 1) & 7u);
        v4i x = *reinterpret_cast<const v4i*>(a_lds + row * kBK + (chunk << 4));
        x.x ^= static_cast<int>(kFlip);
        x.y ^= static_cast<int>(kFlip);
        x.z ^= static_cast<int>(kFlip);
        x.w ^= static_cast<int>(kFlip);
        af[tm] = x;
      }
      if (static_cast<uint32_t>(ksub) == wn) {       // this wave's quarter of the row sums
#pragma unroll
        for (int tm = 0; tm < 4; tm++) {
          int32_t s = rs[tm];
          s = __builtin_amdgcn_sdot4(af[tm].x, 0x01010101, s, false);
          s = __builtin_amdgcn_sdot4(af[tm].y, 0x01010101, s, false);
          s = __builtin_amdgcn_sdot4(af[tm].z, 0x01010101, s, false);
          s = __builtin_amdgcn_sdot4(af[tm].w, 0x01010101, s, false);
          rs[tm] = s;
        }
      }
#pragma unroll
      for (int tm = 0; tm < 4; tm++)
#pragma unroll
        for (int tn = 0; tn < 2; tn++)
          acc[tm][tn] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[tn], af[tm], acc[tm][tn], 0, 0, 0);
    }
  }

  // ---- combine the row sums: 2 K halves (lane, lane+32) then the 4 channel-waves ----
#pragma unroll
  for (int tm = 0; tm < 4; tm++) {
    int32_t s = rs[tm];
    s += __shfl_xor(s, 32);
    if (lane < 32) lds_rowsum[wn * kBM + wm * 128 + tm * 32 + lane] = s;
  }
  __syncthreads();

  // ---- fused epilogue ----
#pragma unroll
  for (int tm = 0; tm < 4; tm++) {
    const uint32_t row = frag_row0 + tm * 32;
    const uint32_t m = m_tile * kBM + row;
    const int32_t rowsum = lds_rowsum[row] + lds_rowsum[kBM + row] + lds_rowsum[2 * kBM + row] + lds_rowsum[3 * kBM + row];
    const int32_t rowterm = p.row_coeff * rowsum;
    uint8_t* out_row = p.output + static_cast<uint64_t>(m) * p.output_stride + static_cast<uint64_t>(g) * p.n;
#pragma unroll
    for (int tn = 0; tn < 2; tn++) {
      const uint32_t nb = nb0 + wn * 2 + tn;
      if (nb >= nblocks) continue;
#pragma unroll
      for (int rg = 0; rg < 4; rg++) {
        const uint32_t ncol = nb * 32 + rg * 8 + frag_khalf * 4;
        const int4 b = *reinterpret_cast<const int4*>(p.bias2 + static_cast<uint64_t>(g) * p.n_pad + ncol);
        const int32_t v0 = acc[tm][tn][rg * 4 + 0] + rowterm + b.x;
        const int32_t v1 = acc[tm][tn][rg * 4 + 1] + rowterm + b.y;
        const int32_t v2 = acc[tm][tn][rg * 4 + 2] + rowterm + b.z;
        const int32_t v3 = acc[tm][tn][rg * 4 + 3] + rowterm + b.w;
        const uint32_t packed = q31_requantize_pack4(v0, v1, v2, v3, p.rq);
        if (m < p.rows && ncol < p.n) {
          if (p.store_dword) {
            *reinterpret_cast<uint32_t*>(out_row + ncol) = packed;
          } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (ncol + j < p.n) out_row[ncol + j] = static_cast<uint8_t>(packed >> (8 * j));
            }
          }
        }
      }
    }
  }
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
