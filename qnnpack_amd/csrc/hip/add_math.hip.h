/*
 * add_math.hip.h -- the quantized element-wise add, in registers (reference src/qnnpack/requantization.h:500-522,
 * qnnp_add_quantize). Shared by the stand-alone add kernel (q8pointwise.hip) and by epilogues that fuse a residual
 * add (q8fused.hip).
 */
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

#include "qnnp_hip.h"

namespace qnnp {

__device__ __forceinline__ uint32_t add_quantize(uint32_t a, uint32_t b, const qnnp_hip_add_params& q)
{
  int32_t acc = static_cast<int32_t>(static_cast<uint32_t>(q.zero_point_product) + a * q.a_multiplier + b * q.b_multiplier);
  const int32_t rem = (acc & q.remainder_mask) - static_cast<int32_t>(acc < 0);
  acc = (acc >> q.shift) + static_cast<int32_t>(rem > q.remainder_threshold);
  int32_t y = acc + q.y_zero_point;
  y = y >= q.y_max ? q.y_max : y;
  y = y <= q.y_min ? q.y_min : y;
  return static_cast<uint32_t>(y);
}

/* four bytes of `a` and `b` at once (byte i of each dword) */
__device__ __forceinline__ uint32_t add_quantize4(uint32_t a4, uint32_t b4, const qnnp_hip_add_params& q)
{
  uint32_t r = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    r |= add_quantize((a4 >> (8 * i)) & 0xFFu, (b4 >> (8 * i)) & 0xFFu, q) << (8 * i);
  }
  return r;
}

}  // namespace qnnp
