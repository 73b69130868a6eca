/*
 * requant.cuh -- the fused Q31 fixed-point down-convert, in registers.
 *
 * Bit-exact restatement of qnnp_q31_requantize (reference
 * src/qnnpack/requantization.h:464-480; stand-alone spec
 * src/requantization/q31-scalar.c:17-138), which every reference microkernel
 * fuses as its epilogue (e.g. src/q8gemm/4x4c2-sse2.c:178-278):
 *
 *   p   = (int64) n * multiplier                    multiplier in [2^30, 2^31)
 *   q   = (int32) ((uint64) (p + 2^30) >> 31)       Q31 product, round half up
 *   rem = (q & remainder_mask) - (n < 0)
 *   y   = (q >>arith shift) + (rem > remainder_threshold)   round half away from zero
 *   y   = min(max(y, omin - ozp), omax - ozp) + ozp
 *
 * Two roundings on purpose -- this is the reference's own definition of the
 * result, not the mathematically nearest one (test/requantization.cc:280).
 */
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

#include "qnnp_hip.h"

namespace qnnp {

__device__ __forceinline__ int32_t q31_requantize(int32_t n, const qnnp_hip_requant& rq)
{
  const int64_t product = static_cast<int64_t>(n) * static_cast<int64_t>(rq.multiplier) + INT64_C(0x40000000);
  const int32_t q31 = static_cast<int32_t>(static_cast<uint32_t>(static_cast<uint64_t>(product) >> 31));
  const int32_t remainder = (q31 & rq.remainder_mask) - static_cast<int32_t>(n < 0);
  int32_t y = (q31 >> rq.shift) + static_cast<int32_t>(remainder > rq.remainder_threshold);
  y = max(y, rq.output_min_less_zero_point);
  y = min(y, rq.output_max_less_zero_point);
  return y + rq.output_zero_point;  // in [0, 255]
}

/* four results packed little-endian into one dword (channel c at byte c) */
__device__ __forceinline__ uint32_t q31_requantize_pack4(
    int32_t n0, int32_t n1, int32_t n2, int32_t n3, const qnnp_hip_requant& rq)
{
  const uint32_t b0 = static_cast<uint32_t>(q31_requantize(n0, rq));
  const uint32_t b1 = static_cast<uint32_t>(q31_requantize(n1, rq));
  const uint32_t b2 = static_cast<uint32_t>(q31_requantize(n2, rq));
  const uint32_t b3 = static_cast<uint32_t>(q31_requantize(n3, rq));
  return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

}  // namespace qnnp
