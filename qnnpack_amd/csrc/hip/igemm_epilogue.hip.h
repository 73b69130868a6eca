/*
 * igemm_epilogue.hip.h -- the fused output stage shared by the MFMA GEMM kernels:
 * accumulator tile (+ folded bias + kernel-zero-point row term) -> Q31 requantize
 * (requant.hip.h, bit-exact with reference src/qnnpack/requantization.h:464-480) ->
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

#include "add_math.hip.h"
#include "igemm_params.h"
#include "requant.hip.h"

namespace qnnp {

typedef int epi_v16i __attribute__((ext_vector_type(16)));

/* PRE_BIASED: what the accumulator was INITIALISED with (the MFMA adds into it), i.e. which adds the epilogue
 * skips: 0 = nothing (add bias and row term here), 1 = bias + row term (`bias` / `rowterm` ignored),
 * 2 = bias only (add the row term here; `bias` ignored). */
/* RESIDUAL: the requantized bytes are then summed (qnnp_add_quantize, operand b) with the bytes at `res_row`
 * (operand a; same channel offsets as the output row, 4-byte aligned) before they are stored. */
template <int SHIFT0, bool FULL_RANGE, bool NO_REQUANT = false, int PRE_BIASED = 0, bool RESIDUAL = false>
__device__ __forceinline__ void igemm_store_tile(
    const epi_v16i& acc, const int4 (&bias)[4], int32_t rowterm,
    uint8_t* out_row,        /* output + m*stride + g*n */
    uint32_t ncol0,          /* first channel of this 32-channel tile */
    uint32_t khalf,          /* lane >> 5 */
    bool row_ok,             /* m < rows */
    const IgemmParams& p,
    const uint8_t* res_row = nullptr, const qnnp_hip_add_params* add = nullptr)
{
  uint32_t pk[4];
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    int32_t v0 = acc[rg * 4 + 0], v1 = acc[rg * 4 + 1], v2 = acc[rg * 4 + 2], v3 = acc[rg * 4 + 3];
    if constexpr (PRE_BIASED == 0) {
      v0 = add_wrap(v0, rowterm, bias[rg].x);    // (wrapping: the row term may carry the 2^31 offset, requant.hip.h)
      v1 = add_wrap(v1, rowterm, bias[rg].y);
      v2 = add_wrap(v2, rowterm, bias[rg].z);
      v3 = add_wrap(v3, rowterm, bias[rg].w);
    } else if constexpr (PRE_BIASED == 2) {
      v0 = add_wrap(v0, rowterm); v1 = add_wrap(v1, rowterm); v2 = add_wrap(v2, rowterm); v3 = add_wrap(v3, rowterm);
    }
    if constexpr (NO_REQUANT) {
      pk[rg] = static_cast<uint32_t>(v0 ^ v1 ^ v2 ^ v3);   // measurement-only ablation
    } else {
      pk[rg] = q31_requantize_pack4<SHIFT0, FULL_RANGE>(v0, v1, v2, v3, p.rq);
    }
    if constexpr (RESIDUAL) {
      const uint32_t c = ncol0 + rg * 8 + khalf * 4;
      if (c < p.n) {                                         // (n % 4 == 0 for a residual block)
        pk[rg] = add_quantize4(*reinterpret_cast<const uint32_t*>(res_row + c), pk[rg], *add);
      }
    }
  }
  // (the store code is spelled out here and not shared with igemm_store_pk4 on purpose: the generic byte-gather kernels
  //  that end in this function run at the register limit, and their code must not move when the lane forms change)
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

/* the four packed dwords of a lane's 32 x 32 tile row (pk[rg] = channels ncol0 + 8 rg + 4 khalf .. + 3) -> global */
__device__ __forceinline__ void igemm_store_pk4(
    const uint32_t (&pk)[4], uint8_t* out_row, uint32_t ncol0, uint32_t khalf, bool row_ok, const IgemmParams& p)
{
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

/* The same for the lane forms of the requantization (requant.hip.h): the accumulators started from the bias table
 * that carries 2^31, `addend` = lane_addend(row term of this lane's row). */
template <int SEQ, bool FULL_RANGE, bool RESIDUAL = false>
__device__ __forceinline__ void igemm_store_tile_lane(
    const epi_v16i& acc, uint64_t addend, uint8_t* out_row, uint32_t ncol0, uint32_t khalf, bool row_ok,
    const IgemmParams& p, const uint8_t* res_row = nullptr, const qnnp_hip_add_params* add = nullptr)
{
  uint32_t pk[4];
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    pk[rg] = q31_requantize_pack4_lane<SEQ, FULL_RANGE>(
        static_cast<uint32_t>(acc[rg * 4 + 0]), static_cast<uint32_t>(acc[rg * 4 + 1]),
        static_cast<uint32_t>(acc[rg * 4 + 2]), static_cast<uint32_t>(acc[rg * 4 + 3]), addend, p.lane, p.rq);
    if constexpr (RESIDUAL) {                                  // (as igemm_store_tile)
      const uint32_t c = ncol0 + rg * 8 + khalf * 4;
      if (c < p.n) pk[rg] = add_quantize4(*reinterpret_cast<const uint32_t*>(res_row + c), pk[rg], *add);
    }
  }
  igemm_store_pk4(pk, out_row, ncol0, khalf, row_ok, p);
}

/*
 * Staged variant for store_mode 2: instead of storing its 16 bytes straight to global memory (one
 * instruction then touches 32 different rows, 32 bytes each -- the L2 sees a stream of 32-byte partial
 * line writes), the lane drops them into a row-major LDS image of the workgroup's output tile; after a
 * barrier igemm_copy_out() streams the image out with consecutive lanes on consecutive 16-byte chunks
 * of a row, so full 128-byte lines (whole rows, for dense NHWC outputs) leave in one request.
 * `pitch` = tile width in bytes + 16 keeps the 8-lane ds_write_b128 groups on distinct banks.
 */
template <int SHIFT0, bool FULL_RANGE, bool NO_REQUANT = false, int PRE_BIASED = 0>
__device__ __forceinline__ void igemm_stage_tile_rq(
    const epi_v16i& acc, const int4 (&bias)[4], int32_t rowterm,
    uint8_t* lds_row,        /* LDS image + tile_row * pitch */
    uint32_t col0,           /* first channel of this 32-channel tile inside the workgroup tile */
    uint32_t khalf, const RequantDev& rq,
    bool write_ok = true)    /* false: take part in the half-wave exchange (all 64 lanes must), write nothing */
{
  uint32_t pk[4];
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    int32_t v0 = acc[rg * 4 + 0], v1 = acc[rg * 4 + 1], v2 = acc[rg * 4 + 2], v3 = acc[rg * 4 + 3];
    if constexpr (PRE_BIASED == 0) {
      v0 = add_wrap(v0, rowterm, bias[rg].x);    // (wrapping: the row term may carry the 2^31 offset, requant.hip.h)
      v1 = add_wrap(v1, rowterm, bias[rg].y);
      v2 = add_wrap(v2, rowterm, bias[rg].z);
      v3 = add_wrap(v3, rowterm, bias[rg].w);
    } else if constexpr (PRE_BIASED == 2) {
      v0 = add_wrap(v0, rowterm); v1 = add_wrap(v1, rowterm); v2 = add_wrap(v2, rowterm); v3 = add_wrap(v3, rowterm);
    }
    if constexpr (NO_REQUANT) {
      pk[rg] = static_cast<uint32_t>(v0 ^ v1 ^ v2 ^ v3);
    } else {
      pk[rg] = q31_requantize_pack4<SHIFT0, FULL_RANGE>(v0, v1, v2, v3, rq);
    }
  }
  const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
  const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
  if (write_ok) {
    *reinterpret_cast<uint4*>(lds_row + col0 + khalf * 16) = make_uint4(s02[0], s02[1], s13[0], s13[1]);
  }
}

/* The same for the lane forms of the requantization (requant.hip.h): the accumulators started from the bias table
 * that carries 2^31, `addend` = lane_addend(row term of this lane's row). */
template <int SEQ, bool FULL_RANGE>
__device__ __forceinline__ void igemm_stage_tile_lane(
    const epi_v16i& acc, uint64_t addend, uint8_t* lds_row, uint32_t col0, uint32_t khalf, const IgemmParams& p,
    bool write_ok = true)
{
  uint32_t pk[4];
#pragma unroll
  for (int rg = 0; rg < 4; rg++) {
    pk[rg] = q31_requantize_pack4_lane<SEQ, FULL_RANGE>(
        static_cast<uint32_t>(acc[rg * 4 + 0]), static_cast<uint32_t>(acc[rg * 4 + 1]),
        static_cast<uint32_t>(acc[rg * 4 + 2]), static_cast<uint32_t>(acc[rg * 4 + 3]), addend, p.lane, p.rq);
  }
  const auto s02 = __builtin_amdgcn_permlane32_swap(pk[0], pk[2], false, false);
  const auto s13 = __builtin_amdgcn_permlane32_swap(pk[1], pk[3], false, false);
  if (write_ok) {
    *reinterpret_cast<uint4*>(lds_row + col0 + khalf * 16) = make_uint4(s02[0], s02[1], s13[0], s13[1]);
  }
}

template <int SHIFT0, bool FULL_RANGE, bool NO_REQUANT = false, int PRE_BIASED = 0>
__device__ __forceinline__ void igemm_stage_tile(
    const epi_v16i& acc, const int4 (&bias)[4], int32_t rowterm, uint8_t* lds_row, uint32_t col0,
    uint32_t khalf, const IgemmParams& p, bool write_ok = true)
{
  igemm_stage_tile_rq<SHIFT0, FULL_RANGE, NO_REQUANT, PRE_BIASED>(acc, bias, rowterm, lds_row, col0, khalf, p.rq, write_ok);
}

/* Stream a staged [rows_valid][n_valid] uint8 tile (LDS, row pitch `pitch`) to global memory. */
template <int NT>
__device__ __forceinline__ void igemm_copy_out(
    const uint8_t* lds_tile, uint32_t pitch, uint32_t rows_valid, uint32_t n_valid,
    uint8_t* out_tile,       /* output + m0*stride + g*n + n0 */
    uint32_t out_stride, uint32_t tid)
{
  const uint32_t cpr = n_valid >> 4;                 // 16-byte chunks per row
  const uint32_t total = rows_valid * cpr;
  for (uint32_t idx = tid; idx < total; idx += NT) {
    const uint32_t r = idx / cpr;
    const uint32_t cc = idx - r * cpr;
    const uint4 v = *reinterpret_cast<const uint4*>(lds_tile + r * pitch + cc * 16);
    *reinterpret_cast<uint4*>(out_tile + static_cast<uint64_t>(r) * out_stride + cc * 16) = v;
  }
}

}  // namespace qnnp
