/*
 * debug-hooks.c -- exports the create/setup-time HOST logic (weight packing,
 * offset-table construction, requantization parameters) as plain C-ABI functions
 * so the CPU-only test tier can check it without a GPU (tests/test_host_logic.py
 * replays the device kernels' documented index arithmetic in numpy on these
 * images and compares with the oracle). No compute path uses these entry points.
 *
 * Reference counterparts of the logic under test: src/qnnpack/pack.h:12-91,
 * 135-167; src/indirection.c:18-79; src/qnnpack/requantization.h:122-198.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "indirection.h"
#include "operator.h"
#include "pack.h"
#include "requantization.h"
#include "hip/requant_math.h"

void qnnp_debug_pack_igemm_w(
    uint32_t groups, uint32_t n, uint32_t k_total, uint32_t n_pad, uint32_t k_pad,
    uint8_t izp, uint8_t kzp, const uint8_t* kernel, const int32_t* bias,
    int8_t* packed, int32_t* bias2)
{
  qnnp_pack_igemm_w(groups, n, k_total, n_pad, k_pad, izp, kzp, kernel, bias, packed, bias2);
}

void qnnp_debug_pack_dwconv_w(
    uint32_t channels, uint32_t c_pad, uint32_t kh, uint32_t kw,
    uint8_t izp, uint8_t kzp, const uint8_t* kernel, const int32_t* bias,
    int16_t* wadj, int32_t* bias1)
{
  qnnp_pack_dwconv_w(channels, c_pad, kh, kw, izp, kzp, kernel, bias, wadj, bias1);
}

uint32_t qnnp_debug_pack_dwconv_mfma(
    uint32_t channels, uint32_t c_pad32, uint32_t kh, uint32_t kw,
    uint8_t izp, uint8_t kzp, const uint8_t* kernel, const int32_t* bias,
    int8_t* xparts, int32_t* biasm)
{
  return qnnp_pack_dwconv_mfma(channels, c_pad32, kh, kw, izp, kzp, kernel, bias, xparts, biasm);
}

/* range class of the depthwise weights (0 / 1 / 2) and, for classes 1 and 2, the register image of the int8
 * dot-product walk of the 3x3 column kernel (pack.h); `image` holds 4 * c_pad words */
uint32_t qnnp_debug_pack_dwconv_dot4(
    uint32_t channels, uint32_t c_pad, uint8_t izp, uint8_t kzp, const uint8_t* kernel, const int32_t* bias,
    int16_t* wadj /* [9][c_pad] scratch */, int32_t* bias1 /* [c_pad] scratch */, uint32_t* image)
{
  qnnp_pack_dwconv_w(channels, c_pad, 3, 3, izp, kzp, kernel, bias, wadj, bias1);
  const uint32_t range = qnnp_dwconv_weight_range(wadj, (size_t) 9 * c_pad);
  if (range != 0) qnnp_pack_dwconv_dot4(c_pad, range, wadj, bias1, image);
  return range;
}

/* the same for 5x5 operators (kernel H): `image` holds 8 * c_pad words (pack.h qnnp_pack_dwconv_dot4_5x5) */
uint32_t qnnp_debug_pack_dwconv_dot4_5x5(
    uint32_t channels, uint32_t c_pad, uint8_t izp, uint8_t kzp, const uint8_t* kernel, const int32_t* bias,
    int16_t* wadj /* [25][c_pad] scratch */, int32_t* bias1 /* [c_pad] scratch */, uint32_t* image)
{
  qnnp_pack_dwconv_w(channels, c_pad, 5, 5, izp, kzp, kernel, bias, wadj, bias1);
  const uint32_t range = qnnp_dwconv_weight_range(wadj, (size_t) 25 * c_pad);
  if (range != 0) qnnp_pack_dwconv_dot4_5x5(c_pad, range, wadj, bias1, image);
  return range;
}

void qnnp_debug_conv2d_offsets(
    size_t input_height, size_t input_width, size_t input_pixel_stride,
    size_t output_height, size_t output_width,
    uint32_t kernel_height, uint32_t kernel_width,
    uint32_t stride_height, uint32_t stride_width,
    uint32_t dilation_height, uint32_t dilation_width,
    uint32_t pad_top, uint32_t pad_left,
    int32_t* table)
{
  struct qnnp_operator op;
  memset(&op, 0, sizeof(op));
  op.input_height = input_height;
  op.input_width = input_width;
  op.input_pixel_stride = input_pixel_stride;
  op.output_height = output_height;
  op.output_width = output_width;
  op.kernel_height = kernel_height;
  op.kernel_width = kernel_width;
  op.stride_height = stride_height;
  op.stride_width = stride_width;
  op.dilation_height = dilation_height;
  op.dilation_width = dilation_width;
  op.input_padding_top = pad_top;
  op.input_padding_left = pad_left;
  qnnp_indirection_init_conv2d_offsets(&op, table);
}

void qnnp_debug_compute_requant(
    float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax, struct qnnp_hip_requant* out)
{
  *out = qnnp_compute_requant(scale, zero_point, qmin, qmax);
}

/* the device requantization arithmetic (hip/requant_math.h) evaluated on the host: scale, clamp, add zero point */
void qnnp_debug_requant_fast_bits(
    size_t count, const int32_t* acc, float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax,
    uint32_t accumulator_bits, uint8_t* out, int* bounded_out);

void qnnp_debug_requant_fast(
    size_t count, const int32_t* acc, float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax, uint8_t* out)
{
  qnnp_debug_requant_fast_bits(count, acc, scale, zero_point, qmin, qmax, 0, out, NULL);
}

uint32_t qnnp_debug_accumulator_bits(const int32_t* bias, size_t count, size_t reduction_length)
{
  return qnnp_accumulator_bits(bias, count, reduction_length);
}

/* same with a caller-stated accumulator bound (0 = unknown): exercises the bounded rounding sequence on the host */
void qnnp_debug_requant_fast_bits(
    size_t count, const int32_t* acc, float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax,
    uint32_t accumulator_bits, uint8_t* out, int* bounded_out)
{
  const struct qnnp_hip_requant rq = qnnp_compute_requant(scale, zero_point, qmin, qmax);
  /* exactly what make_requant_dev (hip/requant.hip.h) hands the kernels: zero point folded into the addend when it
   * fits, clamp bounds in the output domain */
  struct qnnp_requant_fast f = qnnp_requant_fast_init(rq.multiplier, rq.shift);
  const int folded = qnnp_requant_fast_fold_zero_point(&f, (uint32_t) rq.output_zero_point);
  const int bounded = qnnp_requant_fast_enable_bounded(&f, (uint32_t) rq.output_zero_point, folded, accumulator_bits);
  if (bounded_out != NULL) *bounded_out = bounded;
  const int32_t zp_late = folded ? 0 : rq.output_zero_point;
  int32_t lo = rq.output_min_less_zero_point + (folded ? rq.output_zero_point : 0);
  const int32_t hi = rq.output_max_less_zero_point + (folded ? rq.output_zero_point : 0);
  if (lo > hi) lo = hi;
  for (size_t i = 0; i < count; i++) {
    int32_t y = qnnp_requant_scale(acc[i], f);
    if (y < lo) y = lo;
    if (y > hi) y = hi;
    out[i] = (uint8_t) (y + zp_late);
  }
}

/* the offset forms (hip/requant_math.h, qnnp_requant_fast_enable_offset) through the same path: what a kernel that
 * hands over n + 2^31 evaluates. `kind_out`: 0 none (general form used), 1 shift-0 form, 2 bounded form */
void qnnp_debug_requant_fast_offset(
    size_t count, const int32_t* acc, float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax,
    uint32_t accumulator_bits, uint8_t* out, int* kind_out)
{
  const struct qnnp_hip_requant rq = qnnp_compute_requant(scale, zero_point, qmin, qmax);
  struct qnnp_requant_fast f = qnnp_requant_fast_init(rq.multiplier, rq.shift);
  const int folded = qnnp_requant_fast_fold_zero_point(&f, (uint32_t) rq.output_zero_point);
  (void) qnnp_requant_fast_enable_bounded(&f, (uint32_t) rq.output_zero_point, folded, accumulator_bits);
  const uint32_t kind = qnnp_requant_fast_enable_offset(&f);
  if (kind_out != NULL) *kind_out = (int) kind;
  const int32_t zp_late = folded ? 0 : rq.output_zero_point;
  int32_t lo = rq.output_min_less_zero_point + (folded ? rq.output_zero_point : 0);
  const int32_t hi = rq.output_max_less_zero_point + (folded ? rq.output_zero_point : 0);
  if (lo > hi) lo = hi;
  for (size_t i = 0; i < count; i++) {
    int32_t y = qnnp_requant_scale_via_offset(acc[i], f);
    if (y < lo) y = lo;
    if (y > hi) y = hi;
    out[i] = (uint8_t) (y + zp_late);
  }
}

/* the lane forms (hip/requant_math.h, qnnp_requant_lane_*): the kernel hands over a + 2^31 with acc = a + rowterm; the
 * row term enters through the per-lane addend. `kind_out`: 0 none (the other sequences answer), 1 shift 0, 2 bounded */
void qnnp_debug_requant_lane(
    size_t count, const int32_t* acc, const int32_t* rowterm, float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax,
    uint32_t accumulator_bits, uint8_t* out, int* kind_out)
{
  const struct qnnp_hip_requant rq = qnnp_compute_requant(scale, zero_point, qmin, qmax);
  struct qnnp_requant_fast f = qnnp_requant_fast_init(rq.multiplier, rq.shift);
  const int folded = qnnp_requant_fast_fold_zero_point(&f, (uint32_t) rq.output_zero_point);
  (void) qnnp_requant_fast_enable_bounded(&f, (uint32_t) rq.output_zero_point, folded, accumulator_bits);
  (void) qnnp_requant_fast_enable_offset(&f);
  const struct qnnp_requant_lane l = qnnp_requant_lane_init(f, (uint32_t) rq.output_zero_point, folded, accumulator_bits);
  if (kind_out != NULL) *kind_out = (int) l.kind;
  const int32_t zp_late = folded ? 0 : rq.output_zero_point;
  int32_t lo = rq.output_min_less_zero_point + (folded ? rq.output_zero_point : 0);
  const int32_t hi = rq.output_max_less_zero_point + (folded ? rq.output_zero_point : 0);
  if (lo > hi) lo = hi;
  for (size_t i = 0; i < count; i++) {
    int32_t y;
    if (l.kind == 0) {
      y = qnnp_requant_scale_via_offset(acc[i], f);
    } else {
      const uint32_t u = (uint32_t) acc[i] - (uint32_t) rowterm[i] + UINT32_C(0x80000000);   /* a + 2^31 */
      const uint64_t addend = qnnp_requant_lane_addend(rowterm[i], l);
      if (l.kind == 2 && l.shift <= QNNP_REQUANT_LANE_PK_MAX_SHIFT && folded && lo == 0 && hi == 255) {
        out[i] = qnnp_requant_lane_sn_pk(u, addend, l);   /* the sequence requant_dispatch_lane picks for this operator */
        continue;
      }
      y = l.kind == 1 ? qnnp_requant_lane_s0(u, addend, l) : qnnp_requant_lane_sn(u, addend, l);
    }
    if (y < lo) y = lo;
    if (y > hi) y = hi;
    out[i] = (uint8_t) (y + zp_late);
  }
}

void qnnp_debug_pack_igemm_w_slots(
    uint32_t groups, uint32_t n, uint32_t ks, uint32_t kc, uint32_t kc_slot, uint32_t n_pad, uint32_t k_pad,
    uint8_t izp, uint8_t kzp, const uint8_t* kernel, const int32_t* bias,
    int8_t* packed, int32_t* bias2)
{
  qnnp_pack_igemm_w_slots(groups, n, ks, kc, kc_slot, n_pad, k_pad, izp, kzp, kernel, bias, packed, bias2);
}

void qnnp_debug_pack_igemm_w_centred127(
    uint32_t n, uint32_t k_total, uint32_t n_pad, uint8_t izp, const uint8_t* kernel, const int32_t* bias,
    int8_t* packed, int32_t* biasc)
{
  qnnp_pack_igemm_w_centred127(n, k_total, k_total, n_pad, izp, kernel, bias, packed, biasc);
}

void qnnp_debug_strip_pointwise_images(
    const int8_t* std_w, const int32_t* std_bias2, uint32_t k_pad_std, uint32_t n, uint32_t k, uint8_t izp, uint8_t kzp,
    int offset_form, int8_t* frags, int32_t* biasc)
{
  qnnp_strip_pointwise_images(std_w, std_bias2, k_pad_std, n, k, izp, kzp, offset_form, frags, biasc);
}

void qnnp_debug_strip_depthwise_images(
    const int16_t* wadj, const int32_t* bias1, uint32_t c_pad, uint32_t ch, uint32_t hidden_pad, uint8_t izp, uint8_t kzp,
    int offset_form, int8_t* w2, int32_t* biasc)
{
  qnnp_strip_depthwise_images(wadj, bias1, c_pad, ch, hidden_pad, izp, kzp, offset_form, w2, biasc);
}
