/*
 * requant.cuh -- the fused Q31 fixed-point down-convert, in registers.
 *
 * Bit-exact with qnnp_q31_requantize (reference src/qnnpack/requantization.h:464-480;
 * stand-alone spec src/requantization/q31-scalar.c:17-138), which every reference
 * microkernel fuses as its epilogue (e.g. src/q8gemm/4x4c2-sse2.c:178-278):
 *
 *   p   = (int64) n * multiplier                    multiplier in [2^30, 2^31)
 *   q   = (int32) ((uint64) (p + 2^30) >> 31)       Q31 product, round half up
 *   rem = (q & remainder_mask) - (n < 0)
 *   y   = (q >>arith shift) + (rem > remainder_threshold)   round half away from zero
 *   y   = min(max(y, omin - ozp), omax - ozp) + ozp
 *
 * Two roundings on purpose -- this is the reference's own definition of the result,
 * not the mathematically nearest one (test/requantization.cc:280).
 *
 * The device code evaluates the equivalent single-shift form of requant_math.h (one
 * v_mad_i64_i32 instead of the multiply + remainder/threshold compare chain); the CPU test
 * tier checks that form against the oracle through qnnp_debug_requant_fast.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

#include "qnnp_hip.h"
#include "requant_math.h"

namespace qnnp {

/* kernel-argument form of struct qnnp_hip_requant */
struct RequantDev {
  qnnp_requant_fast f;
  int32_t min_less_zp;
  int32_t max_less_zp;
  int32_t zp;
  uint32_t full_range;   /* 1: clamp is exactly [0, 255] -> saturating packs */
};

inline RequantDev make_requant_dev(const qnnp_hip_requant& rq)
{
  RequantDev d;
  d.f = qnnp_requant_fast_init(rq.multiplier, rq.shift);
  d.min_less_zp = rq.output_min_less_zero_point;
  d.max_less_zp = rq.output_max_less_zero_point;
  d.zp = rq.output_zero_point;
  d.full_range = (rq.output_min_less_zero_point + rq.output_zero_point == 0 &&
                  rq.output_max_less_zero_point + rq.output_zero_point == 255) ? 1u : 0u;
  return d;
}

__device__ __forceinline__ int32_t q31_requantize(int32_t n, const RequantDev& rq)
{
  int32_t y = qnnp_requant_scale(n, rq.f);
  y = max(y, rq.min_less_zp);
  y = min(y, rq.max_less_zp);
  return y + rq.zp;  // in [0, 255]
}

/* four results packed little-endian into one dword (channel c at byte c) */
__device__ __forceinline__ uint32_t q31_requantize_pack4(
    int32_t n0, int32_t n1, int32_t n2, int32_t n3, const RequantDev& rq)
{
  if (rq.full_range) {
    // clamp to [0, 255] == saturation, two values per instruction:
    //   i32 -> i16 (signed saturation) -> + zero point (saturating packed add) -> u8 (unsigned saturation).
    // Saturating BEFORE the zero-point add keeps y + zp from wrapping for |y| near 2^31, and cannot
    // change the result: a saturated +-32767 still lands outside [0, 255] on the correct side.
    const auto p01 = __builtin_amdgcn_cvt_pk_i16(qnnp_requant_scale(n0, rq.f), qnnp_requant_scale(n1, rq.f));
    const auto p23 = __builtin_amdgcn_cvt_pk_i16(qnnp_requant_scale(n2, rq.f), qnnp_requant_scale(n3, rq.f));
    const uint32_t zp2 = static_cast<uint32_t>(rq.zp) * 0x00010001u;
    uint32_t lo, hi;
    asm("v_pk_add_i16 %0, %1, %2 clamp\n\tv_sat_pk_u8_i16 %0, %0" : "=&v"(lo) : "v"(p01), "s"(zp2));
    asm("v_pk_add_i16 %0, %1, %2 clamp\n\tv_sat_pk_u8_i16 %0, %0" : "=&v"(hi) : "v"(p23), "s"(zp2));
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);   // {lo.b0, lo.b1, hi.b0, hi.b1}
  }
  const uint32_t b0 = static_cast<uint32_t>(q31_requantize(n0, rq));
  const uint32_t b1 = static_cast<uint32_t>(q31_requantize(n1, rq));
  const uint32_t b2 = static_cast<uint32_t>(q31_requantize(n2, rq));
  const uint32_t b3 = static_cast<uint32_t>(q31_requantize(n3, rq));
  return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
}

}  // namespace qnnp
