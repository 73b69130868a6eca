/*
 * igemm_epilogue.cuh -- the fused output stage shared by the MFMA GEMM kernels:
 * accumulator tile (+ folded bias + kernel-zero-point row term) -> Q31 requantize
 * (requant.cuh, bit-exact with reference src/qnnpack/requantization.h:464-480) ->
 * uint8, 4 channels per dword -> global stores.
 *
 * One call handles one 32x32 MFMA accumulator tile of one wave. C/D layout of
 * v_mfma_i32_32x32x32_i8 with weights as operand A and activations as operand B:
 * lane l holds row m = (l & 31) and, in register r, channel
 *     n = (r & 3) + 8*(r >> 2) + 4*(l >> 5),
 * i.e. four groups ("rg" = r >> 2) of 4 consecutive channels per lane.
 *
 * store_mode 2: the two half-waves exchange dwords with v_permlane32_swap so that
 *   lane l owns channels 0..15 and lane l+32 channels 16..31 of the row -> one
 *   16-byte store per lane per tile (needs n % 16 == 0, 16-byte aligned rows).
 * store_mode 1: one dword store per 4-channel group (n % 4 == 0, 4-byte aligned rows).
 * store_mode 0: byte stores, any n / stride / alignment.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

#include "igemm_params.h"
#include "requant.cuh"

namespace qnnp {

typedef int epi_v16i __attribute__((ext_vector_type(16)));

template <bool NO_REQUANT = false>
__device__ __forceinline__ void igemm_store_tile(
    const epi_v16i& acc, const int4 (&bias)[4], int32_t rowterm,
    uint8_t* out_row,        /* output + m*stride + g*n */
    uint32_t ncol0,          /* first channel of this 32-channel tile */
    uint32_t khalf,          /* lane >> 5 */
    bool row_ok,             /* m < rows */
    const IgemmParams& p)
{
  uint32_t pk[4];
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    const int32_t v0 = acc[rg * 4 + 0] + rowterm + bias[rg].x;
    const int32_t v1 = acc[rg * 4 + 1] + rowterm + bias[rg].y;
    const int32_t v2 = acc[rg * 4 + 2] + rowterm + bias[rg].z;
    const int32_t v3 = acc[rg * 4 + 3] + rowterm + bias[rg].w;
    if constexpr (NO_REQUANT) {
      pk[rg] = static_cast<uint32_t>(v0 ^ v1 ^ v2 ^ v3);   // measurement-only ablation
    } else {
      pk[rg] = q31_requantize_pack4(v0, v1, v2, v3, p.rq);
    }
  }
  if (p.store_mode == 2) {
    // before: lane l      : pk[rg] = channels 8rg + 0..3     lane l+32: channels 8rg + 4..7
    // after : lane l      : {pk0, pk2, pk1, pk3} = channels 0..15
    //         lane l + 32 : {pk0, pk2, pk1, pk3} = channels 16..31
    const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
    const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
    const uint32_t c = ncol0 + khalf * 16;
    if (row_ok && c < p.n) {
      *reinterpret_cast<uint4*>(out_row + c) = make_uint4(s02[0], s02[1], s13[0], s13[1]);
    }
  } else {
#pragma unroll
    for (int rg = 0; rg < 4; rg++) {
      const uint32_t c = ncol0 + rg * 8 + khalf * 4;
      if (row_ok && c < p.n) {
        if (p.store_mode == 1) {
          *reinterpret_cast<uint32_t*>(out_row + c) = pk[rg];
        } else {
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (c + j < p.n) out_row[c + j] = static_cast<uint8_t>(pk[rg] >> (8 * j));
          }
        }
      }
    }
  }
}

}  // namespace qnnp
