/*
 * requant_math.h -- the Q31 fixed-point down-convert as plain integer arithmetic,
 * written once and compiled both for the device (requant.hip.h) and for the host
 * (debug-hooks.c, so the CPU test tier can check it against the oracle over
 * hundreds of millions of accumulators without a GPU).
 *
 * Normative definition (reference src/qnnpack/requantization.h:464-480):
 *     P   = (int64) n * M + 2^30                       M in [2^30, 2^31)
 *     q   = (int32) (P >> 31)
 *     rem = (q & (2^s - 1)) - (n < 0)
 *     y   = (q >> s) + (rem > ((2^s - 1) >> 1))
 *     out = min(max(y, qmin - zp), qmax - zp) + zp
 *
 * Equivalent single-shift form used here (derivation in DESIGN.md "Requantization"):
 *   s == 0 :  y = q                                    (mask 0, threshold 0: nothing to add)
 *   s >= 1 :  (rem > thr)  <=>  (q mod 2^s) >= 2^(s-1) + (n < 0), hence
 *             y = floor((q + 2^(s-1) - (n<0)) / 2^s)
 *               = floor((n*M + 2^30 + 2^(30+s) - (n<0)*2^31) / 2^(31+s))
 *             and with n = u - (n<0)*2^31, u = n & 0x7FFFFFFF:
 *               n*M - (n<0)*2^31 = n*(M+1) - u
 *             so y = high32( n*(M+1) + (2^30 + 2^(30+s) - u) ) >> (s-1)      [arithmetic]
 *   No intermediate overflows: |n*(M+1)| < 2^62, 2^(30+s) <= 2^61.
 * One 32x32+64 multiply-add, no remainder/threshold compare chain.
 */
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define QNNP_HD __host__ __device__ __forceinline__
#else
#define QNNP_HD static inline
#endif

/* per-operator constants derived from (multiplier, shift) */
struct qnnp_requant_fast {
  int32_t multiplier_plus_1;   /* M + 1 (used when shift >= 1) */
  int32_t multiplier;          /* M */
  uint32_t addend_lo;          /* low / high word of 2^30 + 2^(30+s)   (shift >= 1) */
  uint32_t addend_hi;
  uint32_t shift;              /* s */
  uint32_t bounded_lo;         /* low / high word of the bounded form's addend (qnnp_requant_fast_enable_bounded) */
  uint32_t bounded_hi;
  uint32_t bounded;            /* 1: |accumulator| is known to be small enough for qnnp_requant_scale_sn_bounded */
  /* offset forms (qnnp_requant_fast_enable_offset): the kernel hands over n + 2^31 as an UNSIGNED word */
  uint64_t ofs_addend;         /* 64-bit addend of the unsigned multiply-add */
  uint32_t ofs_multiplier;     /* 2M as an unsigned word */
  uint32_t ofs_kind;           /* 0: none, 1: shift-0 form, 2: bounded shift >= 1 form */
};

#define QNNP_REQUANT_OFFSET UINT32_C(0x80000000)   /* what the offset forms expect added to the accumulator */

QNNP_HD struct qnnp_requant_fast qnnp_requant_fast_init(int32_t multiplier, uint32_t shift)
{
  struct qnnp_requant_fast f;
  f.multiplier = multiplier;
  f.multiplier_plus_1 = multiplier + 1;          /* <= 0x7FFFFF81, no overflow */
  const uint64_t addend = (UINT64_C(1) << 30) + (shift >= 1 ? (UINT64_C(1) << (30 + shift)) : 0);
  f.addend_lo = (uint32_t) addend;
  f.addend_hi = (uint32_t) (addend >> 32);
  f.shift = shift;
  f.bounded_lo = 0;
  f.bounded_hi = 0;
  f.bounded = 0;
  f.ofs_addend = 0;
  f.ofs_multiplier = 0;
  f.ofs_kind = 0;
  return f;
}

/* arithmetic shift right of a signed value (gcc, clang and hipcc all implement >> on signed as arithmetic) */
QNNP_HD int32_t qnnp_asr32(int32_t x, uint32_t n) { return x >> n; }

/*
 * Fold the output zero point into the rounding addend, so that the scale functions below return y + zp and the
 * kernels spend no instruction on the addition:
 *   s == 0 :  (n*M + 2^30 + zp*2^31) >> 31                       == q + zp
 *   s >= 1 :  high32(n*(M+1) + A - u + zp*2^(31+s)) >> (s-1)      == y + zp   (zp*2^(31+s) is a multiple of 2^(32+s-1))
 * Needs zp*2^(31+s) < 2^61 beside |n*(M+1)| < 2^62 and A <= 2^61: s <= 22 (zp < 2^8). Returns 1 if folded; 0 leaves
 * `f` untouched (scales below 2^-23, or the one scale whose q + zp could wrap: the caller adds the zero point itself).
 */
QNNP_HD int qnnp_requant_fast_fold_zero_point(struct qnnp_requant_fast* f, uint32_t zero_point)
{
  if (f->shift > 22 || zero_point > 255) return 0;
  /* s == 0 returns the 32-bit q + zp: with the single largest multiplier (scale 1 - 2^-24) q reaches 2^31 - 129 and
   * the sum would wrap; one mantissa step lower q <= 2^31 - 257 and q + 255 still fits */
  if (f->shift == 0 && (uint32_t) f->multiplier > UINT32_C(0x7FFFFF00)) return 0;
  const uint64_t addend = (((uint64_t) f->addend_hi << 32) | f->addend_lo) + ((uint64_t) zero_point << (31 + f->shift));
  f->addend_lo = (uint32_t) addend;
  f->addend_hi = (uint32_t) (addend >> 32);
  return 1;
}

/* y = requantized value BEFORE clamping (plus the zero point if it was folded in); shift == 0 form.
 * q = (n*M + A) >> 31 with A = 2^30 (+ zp*2^31), low 32 bits. Doubling numerator and denominator puts q in the HIGH
 * word of a 64-bit product, which saves the funnel shift -- but 2M does not fit a signed 32-bit operand. Use
 * 2M - 2^32 instead (it does: [-2^31, -2]):  n*(2M - 2^32) + 2A  =  (n*2M + 2A) - n*2^32, and n*2^32 is a whole
 * number of high-word units, so  q = high32(n*(2M - 2^32) + 2A) + n  (mod 2^32), for every n. One v_mad_i64_i32 and
 * one add; |n*(2M - 2^32)| <= 2^62, 2A < 2^41. */
QNNP_HD int32_t qnnp_requant_scale_s0(int32_t n, const struct qnnp_requant_fast f)
{
  const int64_t addend2 = (int64_t) ((((uint64_t) f.addend_hi << 32) | f.addend_lo) << 1);
  const int32_t m2 = (int32_t) ((uint32_t) f.multiplier << 1);            /* 2M - 2^32 as a two's-complement int32 */
  const int64_t t = (int64_t) n * (int64_t) m2 + addend2;
  return (int32_t) ((uint32_t) ((uint64_t) t >> 32) + (uint32_t) n);
}

/* shift >= 1 form */
QNNP_HD int32_t qnnp_requant_scale_sn(int32_t n, const struct qnnp_requant_fast f)
{
  const uint32_t u = (uint32_t) n & UINT32_C(0x7FFFFFFF);
  const uint64_t addend = (((uint64_t) f.addend_hi << 32) | f.addend_lo) - (uint64_t) u;
  const int64_t t = (int64_t) n * (int64_t) f.multiplier_plus_1 + (int64_t) addend;
  const int32_t hi = (int32_t) (uint32_t) ((uint64_t) t >> 32);
  return qnnp_asr32(hi, f.shift - 1);
}

/*
 * Bounded form for shift >= 1 -- four instructions per value instead of six -- for operators whose accumulators are
 * known (at create time, from |bias| and the reduction length) to stay below 2^accumulator_bits:
 *   with T0 = n*M + 2^30 + 2^(30+s) (+ zp*2^(31+s)),  y = floor((T0 - (n<0)*2^31) / 2^(31+s))
 *                                                       = floor((floor(T0 / 2^31) - (n<0)) / 2^s)      (nested floors)
 *   R = floor(T0 / 2^31) = high32(2*T0) = high32(n*(2M - 2^32) + 2*addend) + n       (as in the shift-0 form)
 *   y = (R + (n >> 31)) >> s                                                          (both shifts arithmetic)
 * R must fit 32 bits: |R| <= |n| + 2^(s-1) + zp*2^s + 1, hence the bound; s <= 20 keeps 2*addend below 2^62.
 * Enabled only together with the folded zero point. Returns 1 if enabled.
 */
QNNP_HD int qnnp_requant_fast_enable_bounded(struct qnnp_requant_fast* f, uint32_t zero_point, int zero_point_folded,
                                             uint32_t accumulator_bits)
{
  if (f->shift < 1 || f->shift > 20 || !zero_point_folded || zero_point > 255) return 0;
  if (accumulator_bits == 0 || accumulator_bits > 30) return 0;
  /* 2^30 + 2^19 + 255*2^20 + 1 < 2^31 */
  const uint64_t addend2 = ((((uint64_t) f->addend_hi << 32) | f->addend_lo)) << 1;   /* folded addend, doubled */
  f->bounded_lo = (uint32_t) addend2;
  f->bounded_hi = (uint32_t) (addend2 >> 32);
  f->bounded = 1;
  return 1;
}

QNNP_HD int32_t qnnp_requant_scale_sn_bounded(int32_t n, const struct qnnp_requant_fast f)
{
  int64_t addend2 = (int64_t) (((uint64_t) f.bounded_hi << 32) | f.bounded_lo);
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(addend2));      /* keep it one 64-bit operand of the multiply-add */
#endif
  const int32_t m2 = (int32_t) ((uint32_t) f.multiplier << 1);            /* 2M - 2^32 */
  const int64_t t = (int64_t) n * (int64_t) m2 + addend2;
  const int32_t r = (int32_t) ((uint32_t) ((uint64_t) t >> 32) + (uint32_t) n);
  return qnnp_asr32(r + qnnp_asr32(n, 31), f.shift);
}

/*
 * Offset forms. Both forms above compute floor((n*2M + 2*addend) / 2^32) with a SIGNED 32x32+64 multiply-add, and 2M
 * does not fit a signed operand -- hence the detour over 2M - 2^32 and the "+ n" afterwards. With the accumulator
 * handed over as the unsigned word n' = n + 2^31 (a kernel folds the 2^31 into the bias / row term its accumulators
 * start from, which costs nothing per value) the product n' * 2M is a plain UNSIGNED 32x32 -> 64 one:
 *     n' * 2M + (2*addend - 2^31 * 2M)  ==  n*2M + 2*addend                 (mod 2^64; the true value fits 63 bits)
 * so   s == 0          :  y = high32(n'*2M + C0)                             one v_mad_u64_u32
 *      s >= 1, bounded :  y = (high32(n'*2M + C0 - 2^32) + (n' >> 31)) >> s  (n' >> 31 == (n >= 0) == 1 + (n >>arith 31))
 * i.e. four instructions where the signed bounded form needs six. Same preconditions as the forms they replace.
 * Call after the zero point was folded / the bounded form enabled. Returns the kind (0: not applicable).
 */
QNNP_HD uint32_t qnnp_requant_fast_enable_offset(struct qnnp_requant_fast* f)
{
  const uint64_t m2 = (uint64_t) (uint32_t) f->multiplier << 1;                  /* 2M in [2^31, 2^32) */
  if (f->shift == 0) {
    const uint64_t addend2 = ((((uint64_t) f->addend_hi << 32) | f->addend_lo)) << 1;
    f->ofs_addend = addend2 - (m2 << 31);
    f->ofs_kind = 1;
  } else if (f->bounded) {
    const uint64_t addend2 = ((uint64_t) f->bounded_hi << 32) | f->bounded_lo;
    f->ofs_addend = addend2 - (m2 << 31) - (UINT64_C(1) << 32);
    f->ofs_kind = 2;
  } else {
    f->ofs_kind = 0;
    return 0;
  }
  f->ofs_multiplier = (uint32_t) m2;
  return f->ofs_kind;
}

/* np = n + 2^31 (mod 2^32) */
QNNP_HD int32_t qnnp_requant_scale_s0_ofs(uint32_t np, const struct qnnp_requant_fast f)
{
  const uint64_t t = (uint64_t) np * (uint64_t) f.ofs_multiplier + f.ofs_addend;
  return (int32_t) (uint32_t) (t >> 32);
}

QNNP_HD int32_t qnnp_requant_scale_sn_bounded_ofs(uint32_t np, const struct qnnp_requant_fast f)
{
  const uint64_t t = (uint64_t) np * (uint64_t) f.ofs_multiplier + f.ofs_addend;
  const int32_t r1 = (int32_t) (uint32_t) (t >> 32);
  return qnnp_asr32(r1 + (int32_t) (np >> 31), f.shift);
}

/*
 * Lane forms (round 3). The accumulator of the MFMA kernels is n = a + rowterm, with a = bias + dot product (what the
 * matrix cores deliver when they start from the bias) and rowterm = the kernel-zero-point term of the output ROW -- one
 * value per lane in the MFMA layout. The forms above need n itself, i.e. one add per output value; these do not:
 *     u = a + 2^31 (mod 2^32)      -- the bias the accumulators start from carries the 2^31, nothing per value
 *     t = u * 2M + L               -- one unsigned 32x32+64 multiply-add per value
 *     L = (rowterm + 2^31) * 2M + K   per lane and row block (one multiply-add), K = 2^31 + Z * 2^32 - 2^33 * M
 * so that t == n * 2M + 2^31 + Z * 2^32 (mod 2^64): high32(t) = floor((n*M + 2^30) / 2^31) + Z = q + Z, the Q31
 * product (|q| <= |n| always fits).
 *     shift == 0 (kind 1, Z = zero point): y = high32(t), nothing else.
 *     shift >= 1 (kind 2, Z = 0):          y = (q + (q >>arith 31) + 2^(s-1) + zp * 2^s) >>arith s
 * The second line takes the sign of the rounding correction from q instead of n: the normative (n < 0) and (q < 0)
 * differ only for n < 0 with q == 0 (products in [-1/2, 0)), and at q == 0 the correction cannot change the result
 * ((2^(s-1) - 1) >> s == 2^(s-1) >> s == 0). Needs q + 2^(s-1) + zp * 2^s within 32 bits: the bounded form's
 * preconditions (|n| < 2^30, s <= 20). Three instructions after the multiply-add like the bounded offset form -- the
 * saving is the add that made n. Both kinds need a itself inside int32 (not only n): operators with accumulators
 * bounded at create time. Kind 0: not applicable (the caller keeps its other sequence).
 */
struct qnnp_requant_lane {
  uint64_t konst;      /* K */
  uint32_t mult2;      /* 2M */
  uint32_t k1;         /* 2^(s-1) + zp * 2^s   (kind 2) */
  uint32_t shift;
  uint32_t kind;
};

QNNP_HD struct qnnp_requant_lane qnnp_requant_lane_init(const struct qnnp_requant_fast f, uint32_t zero_point, int zero_point_folded,
                                                         uint32_t accumulator_bits)
{
  struct qnnp_requant_lane l;
  const uint64_t m = (uint64_t) (uint32_t) f.multiplier;
  l.mult2 = (uint32_t) (m << 1);
  l.shift = f.shift;
  l.k1 = 0;
  l.kind = 0;
  l.konst = 0;
  /* both kinds need a = n - rowterm inside int32 (u is a + 2^31 reduced mod 2^32): guaranteed when the operator's
   * accumulators are bounded at create time (|n| < 2^30 with the reduction length that bound implies, K < 2^14, so
   * |rowterm| <= 128 * 255 * K < 2^15 * 2^14 = 2^29, so |a| <= |n| + |rowterm| < 2^30 + 2^29 < 2^31); without a bound the caller keeps the forms that only need n itself */
  const int bounded_acc = accumulator_bits >= 1 && accumulator_bits <= 30;
  if (f.shift == 0 && zero_point_folded && bounded_acc) {
    l.kind = 1;
    l.konst = (UINT64_C(1) << 31) + ((uint64_t) zero_point << 32) - (m << 33);
  } else if (f.shift >= 1 && f.bounded) {
    l.kind = 2;
    l.konst = (UINT64_C(1) << 31) - (m << 33);
    l.k1 = (UINT32_C(1) << (f.shift - 1)) + (zero_point << f.shift);
  }
  return l;
}

/* L of a lane: rowterm is the row's kernel-zero-point term (any int32) */
QNNP_HD uint64_t qnnp_requant_lane_addend(int32_t rowterm, const struct qnnp_requant_lane l)
{
  return (uint64_t) ((uint32_t) rowterm + UINT32_C(0x80000000)) * (uint64_t) l.mult2 + l.konst;
}

/* u = a + 2^31 (mod 2^32); returns y (zero point included), before the clamp */
QNNP_HD int32_t qnnp_requant_lane_s0(uint32_t u, uint64_t addend, const struct qnnp_requant_lane l)
{
  return (int32_t) (uint32_t) (((uint64_t) u * (uint64_t) l.mult2 + addend) >> 32);
}

QNNP_HD int32_t qnnp_requant_lane_sn(uint32_t u, uint64_t addend, const struct qnnp_requant_lane l)
{
  const int32_t q = (int32_t) (uint32_t) (((uint64_t) u * (uint64_t) l.mult2 + addend) >> 32);
  const uint32_t v = (uint32_t) q + (uint32_t) qnnp_asr32(q, 31) + l.k1;
  return qnnp_asr32((int32_t) v, l.shift);
}

/*
 * Kind 2 with shift <= 7 under the full [0, 255] clamp: the packed tail (round 5). Returns the output BYTE.
 *   - the sign of q comes out of the multiply-add itself: L (as a 64-bit pattern) is 2^64 + 2M (rowterm - 2^31) + 2^31,
 *     inside (0, 2^64) for every |rowterm| <= 2^30, so u * 2M + L == 2^64 + (2M n + 2^31) exactly and the addition's
 *     carry out is (2M n + 2^31 >= 0) == (q >= 0): v = q + (k1 - 1) + carry is one add-with-carry instead of a shift and
 *     a three-operand add;
 *   - two v's are then saturated to int16 in one instruction, shifted as a pair, and saturated to bytes as a pair: an
 *     in-range result needs 0 <= v < 256 * 2^s <= 2^15, so the first saturation only ever touches values the clamp
 *     sends to 0 or 255 anyway (-32768 >> s < 0, 32767 >> s >= 255 for s <= 7).
 * 3.75 instructions per value after the multiply-add against 5.25.
 */
QNNP_HD uint8_t qnnp_requant_lane_sn_pk(uint32_t u, uint64_t addend, const struct qnnp_requant_lane l)
{
  const uint64_t prod = (uint64_t) u * (uint64_t) l.mult2;
  const uint64_t t = prod + addend;
  const uint32_t carry = t < prod ? 1u : 0u;
  const int32_t v = (int32_t) ((uint32_t) (t >> 32) + (l.k1 - 1u) + carry);
  int32_t h = v < -32768 ? -32768 : (v > 32767 ? 32767 : v);
  h = qnnp_asr32(h, l.shift);
  return (uint8_t) (h < 0 ? 0 : (h > 255 ? 255 : h));
}
#define QNNP_REQUANT_LANE_PK_MAX_SHIFT 7u

QNNP_HD int32_t qnnp_requant_scale(int32_t n, const struct qnnp_requant_fast f)
{
  if (f.shift == 0) return qnnp_requant_scale_s0(n, f);
  return f.bounded ? qnnp_requant_scale_sn_bounded(n, f) : qnnp_requant_scale_sn(n, f);
}

/* the same through the offset forms where they apply (what a kernel that opted into them evaluates) */
QNNP_HD int32_t qnnp_requant_scale_via_offset(int32_t n, const struct qnnp_requant_fast f)
{
  const uint32_t np = (uint32_t) n + QNNP_REQUANT_OFFSET;
  if (f.ofs_kind == 1) return qnnp_requant_scale_s0_ofs(np, f);
  if (f.ofs_kind == 2) return qnnp_requant_scale_sn_bounded_ofs(np, f);
  return qnnp_requant_scale(n, f);
}
