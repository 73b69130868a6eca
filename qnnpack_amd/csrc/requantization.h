/*
 * requantization.h -- host-side builder for the fused Q31 down-convert
 * parameters. Restates the scalar branch of
 * qnnp_compute_conv_quantization_params (reference src/qnnpack/requantization.h:122-198,
 * scalar members :183-196): a float scale in [2^-32, 1) becomes a Q31 multiplier
 * in [2^30, 2^31) and a right shift in [0, 31].
 *
 * The arithmetic that consumes these parameters (the normative rounding of
 * qnnp_q31_requantize, requantization.h:464-480) lives in hip/requant.hip.h.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "hip/qnnp_hip.h"

static inline struct qnnp_hip_requant qnnp_compute_requant(
    float scale, uint8_t output_zero_point, uint8_t output_min, uint8_t output_max)
{
  uint32_t bits;
  memcpy(&bits, &scale, sizeof(bits));
  struct qnnp_hip_requant rq;
  /* mantissa with the implicit one restored, moved up to bit 30 */
  rq.multiplier = (int32_t) (((bits & UINT32_C(0x007FFFFF)) | UINT32_C(0x00800000)) << 7);
  /* 2^-shift * multiplier * 2^-31 == scale */
  const uint32_t shift = 127 + 31 - 32 - (bits >> 23);
  const uint32_t mask = (UINT32_C(1) << shift) - UINT32_C(1);
  rq.shift = shift;
  rq.remainder_mask = (int32_t) mask;
  rq.remainder_threshold = (int32_t) (mask >> 1);
  rq.output_min_less_zero_point = (int32_t) output_min - (int32_t) output_zero_point;
  rq.output_max_less_zero_point = (int32_t) output_max - (int32_t) output_zero_point;
  rq.output_zero_point = (int32_t) output_zero_point;
  rq.accumulator_bits = 0;      /* unknown until the operator says otherwise (qnnp_accumulator_bits) */
  return rq;
}

/*
 * Smallest b with |bias[i] + sum_k (a - izp)(w - kzp)| < 2^b for every uint8 a, w: the value the fused epilogue
 * requantizes is exactly this accumulator (the zero-point algebra of the kernels cancels), and each of the
 * `reduction_length` products is at most 255 * 255 in magnitude. 0 = more than 31 bits (bound unusable).
 * Lets the device use the cheaper bounded rounding sequence (hip/requant_math.h).
 */
static inline uint32_t qnnp_accumulator_bits(const int32_t* bias, size_t count, size_t reduction_length)
{
  uint64_t max_bias = 0;
  for (size_t i = 0; i < count; i++) {
    const int64_t b = bias[i];
    const uint64_t mag = (uint64_t) (b < 0 ? -b : b);
    if (mag > max_bias) max_bias = mag;
  }
  const uint64_t bound = max_bias + (uint64_t) reduction_length * UINT64_C(65025) + 1;
  uint32_t bits = 0;
  while (bits < 40 && (UINT64_C(1) << bits) < bound) bits++;
  return bits <= 31 ? bits : 0;
}
