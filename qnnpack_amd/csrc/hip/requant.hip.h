/*
 * requant.hip.h -- the fused Q31 fixed-point down-convert, in registers.
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

#include <type_traits>

#include "qnnp_hip.h"
#include "requant_math.h"

namespace qnnp {

/* kernel-argument form of struct qnnp_hip_requant. The output zero point is folded into the rounding addend of `f`
 * (requant_math.h), so the scale functions return y + zp directly; `zp_late` is the zero point still to be added
 * (0 when folded, i.e. always except for scales below 2^-23 and the single largest one) after the clamp, whose
 * bounds are in the domain of what the scale functions return. */
struct RequantDev {
  qnnp_requant_fast f;
  int32_t qmin;
  int32_t qmax;
  int32_t zp_late;
  uint32_t full_range;   /* 1: zero point folded and clamp exactly [0, 255] -> saturating packs, nothing else */
};

inline RequantDev make_requant_dev(const qnnp_hip_requant& rq)
{
  RequantDev d;
  d.f = qnnp_requant_fast_init(rq.multiplier, rq.shift);
  const int folded = qnnp_requant_fast_fold_zero_point(&d.f, static_cast<uint32_t>(rq.output_zero_point));
  (void) qnnp_requant_fast_enable_bounded(&d.f, static_cast<uint32_t>(rq.output_zero_point), folded, rq.accumulator_bits);
  (void) qnnp_requant_fast_enable_offset(&d.f);
  d.zp_late = folded ? 0 : rq.output_zero_point;
  // bounds in the domain of what the scale functions return: output domain when folded, output - zp otherwise
  // (the late addition comes AFTER the clamp: y + zp could wrap for |y| near 2^31)
  d.qmin = rq.output_min_less_zero_point + (folded ? rq.output_zero_point : 0);
  d.qmax = rq.output_max_less_zero_point + (folded ? rq.output_zero_point : 0);
  if (d.qmin > d.qmax) d.qmin = d.qmax;          // min(max(y, lo), hi) == hi then; keeps the median form equivalent
  d.full_range = (folded && d.qmin == 0 && d.qmax == 255) ? 1u : 0u;
  return d;
}

/* clamp(x, lo, hi) as ONE instruction (lo <= hi): the compiler cannot prove the order of two run-time bounds and
 * emits max + min otherwise. gfx9 VOP3 reads at most one SGPR: lo comes from a scalar, hi from a vector register. */
__device__ __forceinline__ int32_t clamp_med3(int32_t x, int32_t lo, int32_t hi)
{
  int32_t r;
  asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "s"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ int32_t q31_requantize(int32_t n, const RequantDev& rq)
{
  return clamp_med3(qnnp_requant_scale(n, rq.f), rq.qmin, rq.qmax) + rq.zp_late;  // in [0, 255]
}

/*
 * Four results packed little-endian into one dword (channel c at byte c).
 * The rounding sequence (kRq*) and FULL_RANGE are the per-operator facts that change the instruction sequence;
 * kernels branch on them ONCE (requant_dispatch) instead of once per element.
 */
/* rounding sequences (first template argument, chosen once per operator by requant_dispatch) */
constexpr int kRqShift0 = 1;     // shift == 0
constexpr int kRqGeneral = 0;    // shift >= 1, any accumulator
constexpr int kRqBounded = 2;    // shift >= 1, accumulators bounded at create time (requant_math.h)
/* offset forms of the two above (requant_math.h): the kernel hands over n + 2^31 -- it adds rq_offset<SEQ>() to the
 * bias / row term its accumulators start from -- and saves one (shift 0) or two (bounded) instructions per value.
 * Only kernels that call requant_dispatch_ofs see them. */
constexpr int kRqShift0Ofs = 3;
constexpr int kRqBoundedOfs = 4;
/* lane forms (requant_math.h, qnnp_requant_lane_*): the kernel hands over a + 2^31 where a = bias + dot product -- its
 * accumulators start from a bias table that carries the 2^31 -- and the row term rides in a per-lane 64-bit addend of
 * the multiply-add (lane_addend below, once per row block): no add per output value at all. Only kernels that call
 * requant_dispatch_lane see them. */
constexpr int kRqShift0Lane = 5;
constexpr int kRqBoundedLane = 6;
/* kRqBoundedLane with shift <= 7 (requant_math.h, qnnp_requant_lane_sn_pk): the sign of the rounding correction is the
 * multiply-add's carry out, and the shift works on int16 pairs -- 3.75 instead of 5.25 instructions per value after
 * the multiply-add. Full [0, 255] clamp only (the saturating packs are the clamp). */
constexpr int kRqBoundedLanePk = 7;
template <int SEQ> constexpr bool rq_is_lane() { return SEQ == kRqShift0Lane || SEQ == kRqBoundedLane || SEQ == kRqBoundedLanePk; }

inline qnnp_requant_lane make_requant_lane(const qnnp_hip_requant& rq)
{
  qnnp_requant_fast f = qnnp_requant_fast_init(rq.multiplier, rq.shift);
  const int folded = qnnp_requant_fast_fold_zero_point(&f, static_cast<uint32_t>(rq.output_zero_point));
  (void) qnnp_requant_fast_enable_bounded(&f, static_cast<uint32_t>(rq.output_zero_point), folded, rq.accumulator_bits);
  return qnnp_requant_lane_init(f, static_cast<uint32_t>(rq.output_zero_point), folded, rq.accumulator_bits);
}

/* Two's-complement add. Accumulators that carry the 2^31 offset wrap around BY DESIGN, and a plain signed + lets the
 * compiler assume they do not: it turned bias + INT32_MIN into bias | 0x80000000 (right only for bias >= 0). Every
 * add on the way from the offset to q31_requantize_pack4 goes through this. */
__device__ __forceinline__ int32_t add_wrap(int32_t a, int32_t b)
{
  return static_cast<int32_t>(static_cast<uint32_t>(a) + static_cast<uint32_t>(b));
}
__device__ __forceinline__ int32_t add_wrap(int32_t a, int32_t b, int32_t c)
{
  return static_cast<int32_t>(static_cast<uint32_t>(a) + static_cast<uint32_t>(b) + static_cast<uint32_t>(c));
}

/* what a kernel must have added to the accumulators it hands to q31_requantize_pack4<SEQ, ...> */
template <int SEQ>
__device__ __forceinline__ constexpr uint32_t rq_offset()
{
  return (SEQ == kRqShift0Ofs || SEQ == kRqBoundedOfs) ? QNNP_REQUANT_OFFSET : 0u;
}
/* x + rq_offset<SEQ>() (wrapping) */
template <int SEQ>
__device__ __forceinline__ int32_t with_rq_offset(int32_t x)
{
  return static_cast<int32_t>(static_cast<uint32_t>(x) + rq_offset<SEQ>());
}

/* hi32(np * multiplier + addend) of the offset forms (requant_math.h qnnp_requant_scale_s0_ofs). A VOP3 instruction
 * reads ONE scalar operand on gfx9, and with both constants in SGPRs hipcc keeps the multiplier there and re-copies
 * the 64-bit addend into a VGPR pair in front of every v_mad_u64_u32 (16 v_mov_b64 per 32x32 output tile and lane in
 * the streaming kernels). The callers therefore hand the multiplier over in a VGPR (`mult_v`: made by an opaque asm,
 * so it is neither rematerialised nor moved back), which leaves the addend as the scalar operand. */
__device__ __forceinline__ uint32_t requant_mad_hi(uint32_t np, uint32_t mult_v, uint64_t addend)
{
  return static_cast<uint32_t>((static_cast<uint64_t>(np) * mult_v + addend) >> 32);
}

/* four scaled values -> clamp -> four bytes of one dword (channel c at byte c) */
template <bool FULL_RANGE>
__device__ __forceinline__ uint32_t clamp_pack4(int32_t y0, int32_t y1, int32_t y2, int32_t y3, const RequantDev& rq)
{
  if constexpr (FULL_RANGE) {
    // the y's already carry the zero point; clamp to [0, 255] == saturation, two values per instruction:
    //   i32 -> i16 (signed saturation) -> u8 (unsigned saturation).
    // The first saturation cannot change the result: +-32767 still lands outside [0, 255] on the correct side.
    const auto p01 = __builtin_amdgcn_cvt_pk_i16(y0, y1);
    const auto p23 = __builtin_amdgcn_cvt_pk_i16(y2, y3);
    uint32_t lo, hi;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(lo) : "v"(p01));
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(hi) : "v"(p23));
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);   // {lo.b0, lo.b1, hi.b0, hi.b1}
  } else {
    y0 = clamp_med3(y0, rq.qmin, rq.qmax) + rq.zp_late;
    y1 = clamp_med3(y1, rq.qmin, rq.qmax) + rq.zp_late;
    y2 = clamp_med3(y2, rq.qmin, rq.qmax) + rq.zp_late;
    y3 = clamp_med3(y3, rq.qmin, rq.qmax) + rq.zp_late;
    return static_cast<uint32_t>(y0) | (static_cast<uint32_t>(y1) << 8) | (static_cast<uint32_t>(y2) << 16) |
           (static_cast<uint32_t>(y3) << 24);
  }
}

/* VMULT = false: leave the operand placement to the compiler (the wave-per-block 3x3 convolution measured 25.3 -> 26.6
 * us with the multiplier pinned to a VGPR, the streaming kernels gain: sweep +1 %, same box) */
template <int SHIFT0, bool FULL_RANGE, bool VMULT = true>
__device__ __forceinline__ uint32_t q31_requantize_pack4(
    int32_t n0, int32_t n1, int32_t n2, int32_t n3, const RequantDev& rq)
{
  int32_t y0, y1, y2, y3;
  if constexpr (SHIFT0 == kRqShift0Ofs || SHIFT0 == kRqBoundedOfs) {
    uint32_t mult_v = rq.f.ofs_multiplier;
    if constexpr (VMULT) asm("" : "+v"(mult_v));
    const uint64_t addend = rq.f.ofs_addend;
    const uint32_t r0 = requant_mad_hi(static_cast<uint32_t>(n0), mult_v, addend);
    const uint32_t r1 = requant_mad_hi(static_cast<uint32_t>(n1), mult_v, addend);
    const uint32_t r2 = requant_mad_hi(static_cast<uint32_t>(n2), mult_v, addend);
    const uint32_t r3 = requant_mad_hi(static_cast<uint32_t>(n3), mult_v, addend);
    if constexpr (SHIFT0 == kRqShift0Ofs) {
      y0 = static_cast<int32_t>(r0); y1 = static_cast<int32_t>(r1); y2 = static_cast<int32_t>(r2); y3 = static_cast<int32_t>(r3);
    } else {
      // (qnnp_requant_scale_sn_bounded_ofs: the sign of n rides in bit 31 of np)
      const uint32_t sh = rq.f.shift;
      y0 = qnnp_asr32(static_cast<int32_t>(r0 + (static_cast<uint32_t>(n0) >> 31)), sh);
      y1 = qnnp_asr32(static_cast<int32_t>(r1 + (static_cast<uint32_t>(n1) >> 31)), sh);
      y2 = qnnp_asr32(static_cast<int32_t>(r2 + (static_cast<uint32_t>(n2) >> 31)), sh);
      y3 = qnnp_asr32(static_cast<int32_t>(r3 + (static_cast<uint32_t>(n3) >> 31)), sh);
    }
  } else if constexpr (SHIFT0 == kRqShift0) {
    y0 = qnnp_requant_scale_s0(n0, rq.f); y1 = qnnp_requant_scale_s0(n1, rq.f);
    y2 = qnnp_requant_scale_s0(n2, rq.f); y3 = qnnp_requant_scale_s0(n3, rq.f);
  } else if constexpr (SHIFT0 == kRqBounded) {
    y0 = qnnp_requant_scale_sn_bounded(n0, rq.f); y1 = qnnp_requant_scale_sn_bounded(n1, rq.f);
    y2 = qnnp_requant_scale_sn_bounded(n2, rq.f); y3 = qnnp_requant_scale_sn_bounded(n3, rq.f);
  } else {
    y0 = qnnp_requant_scale_sn(n0, rq.f); y1 = qnnp_requant_scale_sn(n1, rq.f);
    y2 = qnnp_requant_scale_sn(n2, rq.f); y3 = qnnp_requant_scale_sn(n3, rq.f);
  }
  return clamp_pack4<FULL_RANGE>(y0, y1, y2, y3, rq);
}

/* The same with the clamp class as a template argument (q8gemm256c.hip picks it on the host):
 *   CLAMP 0: zero point folded, clamp exactly [0, 255] -- the saturating packs (FULL_RANGE above);
 *   CLAMP 1: nothing to add after the clamp (rq.zp_late == 0: zero point folded, or zero) -- one v_med3 per value and
 *            three v_perm per dword (the clamped values are bytes already);
 *   CLAMP 2: the general form (clamp, + zero point, pack). */
template <int SEQ, int CLAMP>
__device__ __forceinline__ uint32_t q31_requantize_pack4_clamp(int32_t n0, int32_t n1, int32_t n2, int32_t n3, const RequantDev& rq)
{
  if constexpr (CLAMP == 0) {
    return q31_requantize_pack4<SEQ, true>(n0, n1, n2, n3, rq);
  } else if constexpr (CLAMP == 2) {
    return q31_requantize_pack4<SEQ, false>(n0, n1, n2, n3, rq);
  } else {
    int32_t y0, y1, y2, y3;
    if constexpr (SEQ == kRqShift0Ofs) {
      uint32_t mult_v = rq.f.ofs_multiplier;
      asm("" : "+v"(mult_v));
      const uint64_t addend = rq.f.ofs_addend;
      y0 = static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n0), mult_v, addend));
      y1 = static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n1), mult_v, addend));
      y2 = static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n2), mult_v, addend));
      y3 = static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n3), mult_v, addend));
    } else if constexpr (SEQ == kRqBoundedOfs) {
      // (round 5: bounded operators with an explicit clamp -- the reference GEMM bench's [1, 254] at a scale below 0.5 -- keep
      //  the four-instruction bounded form instead of falling to the general one: 4096^3 70.6 -> 6x us)
      uint32_t mult_v = rq.f.ofs_multiplier;
      asm("" : "+v"(mult_v));
      const uint64_t addend = rq.f.ofs_addend;
      const uint32_t sh = rq.f.shift;
      y0 = qnnp_asr32(static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n0), mult_v, addend) + (static_cast<uint32_t>(n0) >> 31)), sh);
      y1 = qnnp_asr32(static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n1), mult_v, addend) + (static_cast<uint32_t>(n1) >> 31)), sh);
      y2 = qnnp_asr32(static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n2), mult_v, addend) + (static_cast<uint32_t>(n2) >> 31)), sh);
      y3 = qnnp_asr32(static_cast<int32_t>(requant_mad_hi(static_cast<uint32_t>(n3), mult_v, addend) + (static_cast<uint32_t>(n3) >> 31)), sh);
    } else {
      static_assert(SEQ == kRqGeneral, "offset forms or the general sequence");
      y0 = qnnp_requant_scale_sn(n0, rq.f); y1 = qnnp_requant_scale_sn(n1, rq.f);
      y2 = qnnp_requant_scale_sn(n2, rq.f); y3 = qnnp_requant_scale_sn(n3, rq.f);
    }
    const uint32_t b0 = static_cast<uint32_t>(clamp_med3(y0, rq.qmin, rq.qmax));     // in [0, 255]: 0 <= qmin <= qmax <= 255 here
    const uint32_t b1 = static_cast<uint32_t>(clamp_med3(y1, rq.qmin, rq.qmax));
    const uint32_t b2 = static_cast<uint32_t>(clamp_med3(y2, rq.qmin, rq.qmax));
    const uint32_t b3 = static_cast<uint32_t>(clamp_med3(y3, rq.qmin, rq.qmax));
    // three byte permutes (written as shifts and ors hipcc re-associates them into four instructions)
    const uint32_t p01 = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u);   // {b0.byte0, b1.byte0, 0, 0}
    const uint32_t p23 = __builtin_amdgcn_perm(b3, b2, 0x0c0c0400u);
    return __builtin_amdgcn_perm(p23, p01, 0x05040100u);               // {p01.b0, p01.b1, p23.b0, p23.b1}
  }
}

/* L of this lane's row (requant_math.h): once per row block */
__device__ __forceinline__ uint64_t lane_addend(int32_t rowterm, const qnnp_requant_lane& l)
{
  return static_cast<uint64_t>(static_cast<uint32_t>(rowterm) + 0x80000000u) * l.mult2 + l.konst;
}

/* v = q + (k1 - 1) + (q >= 0) of qnnp_requant_lane_sn_pk for four values: the multiply-adds with their carries out, then
 * the adds that consume them (the carry takes the instruction's one scalar read, so k1 - 1 comes in a VGPR). Two asm
 * blocks of four, not eight of one: the compiler neither interleaves inline asm nor knows its latencies, and it pads
 * every block with an s_nop. */
__device__ __forceinline__ void lane_mad_round4(
    uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3, uint32_t m2, uint64_t addend, uint32_t k1m1_v,
    int32_t& v0, int32_t& v1, int32_t& v2, int32_t& v3)
{
  uint64_t t0, t1, t2, t3, c0, c1, c2, c3;
  asm("v_mad_u64_u32 %0, %4, %8, %12, %13\n\t"
      "v_mad_u64_u32 %1, %5, %9, %12, %13\n\t"
      "v_mad_u64_u32 %2, %6, %10, %12, %13\n\t"
      "v_mad_u64_u32 %3, %7, %11, %12, %13"
      : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
      : "v"(u0), "v"(u1), "v"(u2), "v"(u3), "s"(m2), "v"(addend));
  uint32_t r0, r1, r2, r3;
  asm("v_addc_co_u32_e64 %0, %4, %8, %12, %4\n\t"
      "v_addc_co_u32_e64 %1, %5, %9, %12, %5\n\t"
      "v_addc_co_u32_e64 %2, %6, %10, %12, %6\n\t"
      "v_addc_co_u32_e64 %3, %7, %11, %12, %7"
      : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "+s"(c0), "+s"(c1), "+s"(c2), "+s"(c3)
      : "v"(static_cast<uint32_t>(t0 >> 32)), "v"(static_cast<uint32_t>(t1 >> 32)), "v"(static_cast<uint32_t>(t2 >> 32)),
        "v"(static_cast<uint32_t>(t3 >> 32)), "v"(k1m1_v));
  v0 = static_cast<int32_t>(r0); v1 = static_cast<int32_t>(r1); v2 = static_cast<int32_t>(r2); v3 = static_cast<int32_t>(r3);
}

/* u = a + 2^31 for four channels of one row; `addend` = lane_addend of that row. The multiplier stays the scalar
 * operand of the multiply-add (one per VOP3 instruction on gfx9), the addend is the lane's register pair. */
template <int SEQ, bool FULL_RANGE>
__device__ __forceinline__ uint32_t q31_requantize_pack4_lane(
    uint32_t u0, uint32_t u1, uint32_t u2, uint32_t u3, uint64_t addend, const qnnp_requant_lane& l, const RequantDev& rq)
{
  static_assert(rq_is_lane<SEQ>(), "lane forms only");
  const uint32_t m2 = l.mult2;
  if constexpr (SEQ == kRqBoundedLanePk) {
    static_assert(FULL_RANGE, "the packed tail is the [0, 255] clamp");
    uint32_t k1m1 = l.k1 - 1u;
    asm("" : "+v"(k1m1));
    const uint32_t sh2 = l.shift * 0x10001u;          // the shift for both halves of a pair
    // The matrix-core hazard: gfx950 has no interlock between an MFMA's write of its accumulators and a VALU read of them
    // (8 passes + 3 wait states here); hipcc pads in front of the first reader it KNOWS, and it does not look inside
    // inline asm -- the first version of this tail read accumulators three instructions after the MFMA and returned
    // garbage. So instructions the compiler does know read EVERY one of the four inputs first (u | 0, the zero opaque: four
    // plain VALU ops, which hipcc pads behind whichever MFMAs produced them), and the asm takes their results: the tail is
    // safe for inputs from different MFMA results too (round 6; round 5 guarded u3 only and relied on all four coming from one).
    uint32_t zero = 0;
    asm("" : "+s"(zero));
    const uint32_t u0f = u0 | zero, u1f = u1 | zero, u2f = u2 | zero, u3f = u3 | zero;
    int32_t v0, v1, v2, v3;
    lane_mad_round4(u0f, u1f, u2f, u3f, m2, addend, k1m1, v0, v1, v2, v3);
    const auto p01 = __builtin_amdgcn_cvt_pk_i16(v0, v1);   // saturating: qnnp_requant_lane_sn_pk says why that is exact
    const auto p23 = __builtin_amdgcn_cvt_pk_i16(v2, v3);
    uint32_t lo, hi;
    asm("v_pk_ashrrev_i16 %0, %2, %3\n\t"
        "v_pk_ashrrev_i16 %1, %2, %4\n\t"
        "v_sat_pk_u8_i16 %0, %0\n\t"
        "v_sat_pk_u8_i16 %1, %1"
        : "=&v"(lo), "=&v"(hi) : "s"(sh2), "v"(p01), "v"(p23));
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
  }
  int32_t y0 = static_cast<int32_t>(static_cast<uint32_t>((static_cast<uint64_t>(u0) * m2 + addend) >> 32));
  int32_t y1 = static_cast<int32_t>(static_cast<uint32_t>((static_cast<uint64_t>(u1) * m2 + addend) >> 32));
  int32_t y2 = static_cast<int32_t>(static_cast<uint32_t>((static_cast<uint64_t>(u2) * m2 + addend) >> 32));
  int32_t y3 = static_cast<int32_t>(static_cast<uint32_t>((static_cast<uint64_t>(u3) * m2 + addend) >> 32));
  if constexpr (SEQ == kRqBoundedLane) {
    const uint32_t k1 = l.k1, sh = l.shift;
    y0 = qnnp_asr32(static_cast<int32_t>(static_cast<uint32_t>(y0) + static_cast<uint32_t>(qnnp_asr32(y0, 31)) + k1), sh);
    y1 = qnnp_asr32(static_cast<int32_t>(static_cast<uint32_t>(y1) + static_cast<uint32_t>(qnnp_asr32(y1, 31)) + k1), sh);
    y2 = qnnp_asr32(static_cast<int32_t>(static_cast<uint32_t>(y2) + static_cast<uint32_t>(qnnp_asr32(y2, 31)) + k1), sh);
    y3 = qnnp_asr32(static_cast<int32_t>(static_cast<uint32_t>(y3) + static_cast<uint32_t>(qnnp_asr32(y3, 31)) + k1), sh);
  }
  return clamp_pack4<FULL_RANGE>(y0, y1, y2, y3, rq);
}

/* Calls f(shift0_tag, full_range_tag) with the compile-time tags matching `rq` (one uniform branch tree). */
template <typename F>
__device__ __forceinline__ void requant_dispatch(const RequantDev& rq, F&& f)
{
  using Shift0 = std::integral_constant<int, kRqShift0>;
  using General = std::integral_constant<int, kRqGeneral>;
  using Bounded = std::integral_constant<int, kRqBounded>;
  if (rq.f.shift == 0) {
    if (rq.full_range) f(Shift0{}, std::true_type{}); else f(Shift0{}, std::false_type{});
  } else if (rq.f.bounded && rq.full_range) {
    f(Bounded{}, std::true_type{});            // (the bounded form is instantiated for the common clamp only)
  } else {
    if (rq.full_range) f(General{}, std::true_type{}); else f(General{}, std::false_type{});
  }
}

/* The same for kernels that fold rq_offset<SEQ>() into their accumulators: offset forms where they exist. */
template <typename F>
__host__ __device__ __forceinline__ void requant_dispatch_ofs(const RequantDev& rq, F&& f)   // (host: to pick a kernel instantiation)
{
  using Shift0Ofs = std::integral_constant<int, kRqShift0Ofs>;
  using BoundedOfs = std::integral_constant<int, kRqBoundedOfs>;
  using General = std::integral_constant<int, kRqGeneral>;
  if (rq.f.shift == 0) {
    if (rq.full_range) f(Shift0Ofs{}, std::true_type{}); else f(Shift0Ofs{}, std::false_type{});
  } else if (rq.f.bounded && rq.full_range) {
    f(BoundedOfs{}, std::true_type{});
  } else {
    if (rq.full_range) f(General{}, std::true_type{}); else f(General{}, std::false_type{});
  }
}

/* The same for kernels that implement the lane forms (host side: picks the kernel instantiation). The shift-0 corner
 * whose zero point cannot be folded (the single largest multiplier) keeps the offset form. */
template <typename F>
__host__ __device__ __forceinline__ void requant_dispatch_lane(const RequantDev& rq, const qnnp_requant_lane& lane, F&& f)
{
  using Shift0Lane = std::integral_constant<int, kRqShift0Lane>;
  using BoundedLane = std::integral_constant<int, kRqBoundedLane>;
  using BoundedLanePk = std::integral_constant<int, kRqBoundedLanePk>;
  using Shift0Ofs = std::integral_constant<int, kRqShift0Ofs>;
  using General = std::integral_constant<int, kRqGeneral>;
  if (lane.kind == 1) {
    if (rq.full_range) f(Shift0Lane{}, std::true_type{}); else f(Shift0Lane{}, std::false_type{});
  } else if (lane.kind == 2 && rq.full_range) {
    if (lane.shift <= QNNP_REQUANT_LANE_PK_MAX_SHIFT) f(BoundedLanePk{}, std::true_type{}); else f(BoundedLane{}, std::true_type{});
  } else if (rq.f.shift == 0) {
    if (rq.full_range) f(Shift0Ofs{}, std::true_type{}); else f(Shift0Ofs{}, std::false_type{});
  } else {
    if (rq.full_range) f(General{}, std::true_type{}); else f(General{}, std::false_type{});
  }
}

}  // namespace qnnp
