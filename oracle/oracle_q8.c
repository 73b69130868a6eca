/*
 * oracle_q8.c -- scalar CPU restatement of QNNPACK's q8 GEMM/conv hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle_q8.h). Not linked into the product.
 */
#include "oracle_q8.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 1;

void oracle_set_threads(int n) { g_threads = n < 1 ? 1 : n; }
int oracle_get_threads(void) { return g_threads; }

static inline uint32_t f32_bits(float f) {
  uint32_t u;
  memcpy(&u, &f, sizeof(u));
  return u;
}

/* src/qnnpack/scalar-utils.h:41-51: arithmetic shift right of a signed value */
static inline int32_t asr32(int32_t x, uint32_t n) {
  return x >= 0 ? (int32_t) ((uint32_t) x >> n) : (int32_t) ~(~(uint32_t) x >> n);
}

/* src/qnnpack/requantization.h:22-54 */
int oracle_q31_params_init(
    float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax,
    struct oracle_q31_params* p)
{
  if (!(scale < 1.0f) || !(scale >= 0x1.0p-32f)) {
    return -1;
  }
  const uint32_t bits = f32_bits(scale);
  /* :31  multiplier in [0x40000000, 0x7FFFFF80] */
  p->multiplier = (int32_t) (((bits & UINT32_C(0x007FFFFF)) | UINT32_C(0x00800000)) << 7);
  /* :36  shift in [0, 31] */
  const int32_t shift = 127 + 31 - 32 - (int32_t) (bits >> 23);
  const uint32_t mask = (UINT32_C(1) << shift) - UINT32_C(1);
  p->remainder_mask = (int32_t) mask;
  p->remainder_threshold = (int32_t) (mask >> 1);
  p->shift = (uint32_t) shift;
  p->min_less_zero_point = (int32_t) qmin - (int32_t) zero_point;
  p->max_less_zero_point = (int32_t) qmax - (int32_t) zero_point;
  p->zero_point = (int32_t) zero_point;
  return 0;
}

/* src/qnnpack/requantization.h:464-480 */
uint8_t oracle_q31_requantize(int32_t n, const struct oracle_q31_params* p)
{
  const int64_t product = (int64_t) n * (int64_t) p->multiplier;
  /* :469 round-half-up Q31 product, truncated to 32 bits */
  const int32_t q31 = (int32_t) (uint32_t) ((uint64_t) (product + INT64_C(0x40000000)) >> 31);
  /* :470 remainder, biased down by one for negative inputs */
  const int32_t remainder = (q31 & p->remainder_mask) - (int32_t) (n < 0);
  /* :471 shift with round-half-away-from-zero */
  int32_t y = asr32(q31, p->shift) + (int32_t) (remainder > p->remainder_threshold);
  if (y < p->min_less_zero_point) y = p->min_less_zero_point;
  if (y > p->max_less_zero_point) y = p->max_less_zero_point;
  return (uint8_t) (y + p->zero_point);
}

int oracle_q31_requantize_array(
    size_t n, const int32_t* input, float scale, uint8_t zero_point,
    uint8_t qmin, uint8_t qmax, uint8_t* output)
{
  struct oracle_q31_params p;
  if (oracle_q31_params_init(scale, zero_point, qmin, qmax, &p) != 0) return -1;
  for (size_t i = 0; i < n; i++) output[i] = oracle_q31_requantize(input[i], &p);
  return 0;
}

/* test/gemm-microkernel-tester.h:213-226. int32 arithmetic wraps mod 2^32, as
 * the reference's packed-bias + pmaddwd accumulation does. */
void oracle_gemm_acc(
    size_t M, size_t N, size_t K,
    const uint8_t* a, size_t a_stride,
    const uint8_t* w, const int32_t* bias,
    uint8_t izp, uint8_t kzp, int32_t* acc)
{
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (ptrdiff_t m = 0; m < (ptrdiff_t) M; m++) {
    const uint8_t* am = a + (size_t) m * a_stride;
    for (size_t n = 0; n < N; n++) {
      const uint8_t* wn = w + n * K;
      uint32_t s = (uint32_t) bias[n];
      for (size_t k = 0; k < K; k++) {
        s += (uint32_t) (((int32_t) am[k] - (int32_t) izp) * ((int32_t) wn[k] - (int32_t) kzp));
      }
      acc[(size_t) m * N + n] = (int32_t) s;
    }
  }
}

/* src/convolution.c:29-37 */
size_t oracle_conv_output_dim(size_t padded_input, size_t kernel, size_t dilation, size_t stride)
{
  const size_t effective = (kernel - 1) * dilation + 1;
  return (padded_input - effective) / stride + 1;
}

/* test/convolution-operator-tester.h:367-403 */
void oracle_conv2d_acc(
    const struct oracle_conv_shape* s,
    const uint8_t* input, const uint8_t* kernel, const int32_t* bias,
    uint8_t izp, uint8_t kzp, int32_t* acc)
{
  const size_t OH = oracle_conv_output_dim(
      s->pad_top + s->input_height + s->pad_bottom, s->kernel_height, s->dilation_height, s->stride_height);
  const size_t OW = oracle_conv_output_dim(
      s->pad_left + s->input_width + s->pad_right, s->kernel_width, s->dilation_width, s->stride_width);
  const size_t G = s->groups, GIC = s->group_input_channels, GOC = s->group_output_channels;
  const size_t KH = s->kernel_height, KW = s->kernel_width;
  const ptrdiff_t rows = (ptrdiff_t) (s->batch * OH);

#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (ptrdiff_t row = 0; row < rows; row++) {
    const size_t i = (size_t) row / OH;
    const size_t oy = (size_t) row % OH;
    for (size_t ox = 0; ox < OW; ox++) {
      int32_t* out = acc + (((i * OH + oy) * OW + ox) * G) * GOC;
      for (size_t g = 0; g < G; g++) {
        for (size_t oc = 0; oc < GOC; oc++) {
          uint32_t sum = (uint32_t) bias[g * GOC + oc];
          for (size_t ky = 0; ky < KH; ky++) {
            /* unsigned wrap-around compare, as :381-382 */
            const size_t iy = oy * s->stride_height + ky * s->dilation_height - s->pad_top;
            if (iy >= s->input_height) continue;
            for (size_t kx = 0; kx < KW; kx++) {
              const size_t ix = ox * s->stride_width + kx * s->dilation_width - s->pad_left;
              if (ix >= s->input_width) continue;
              const uint8_t* in = input +
                  ((i * s->input_height + iy) * s->input_width + ix) * s->input_pixel_stride + g * GIC;
              const uint8_t* kw = kernel + (((g * GOC + oc) * KH + ky) * KW + kx) * GIC;
              for (size_t ic = 0; ic < GIC; ic++) {
                sum += (uint32_t) (((int32_t) in[ic] - (int32_t) izp) * ((int32_t) kw[ic] - (int32_t) kzp));
              }
            }
          }
          out[g * GOC + oc] = (int32_t) sum;
        }
      }
    }
  }
}

/* src/deconvolution.c:25-37 */
size_t oracle_deconv_output_dim(
    size_t input, size_t padding, size_t adjustment, size_t kernel, size_t dilation, size_t stride)
{
  const size_t effective = (kernel - 1) * dilation + 1;
  return stride * (input - 1) + adjustment + effective - padding;
}

/* test/deconvolution-operator-tester.h:383-419 */
void oracle_deconv2d_acc(
    const struct oracle_conv_shape* s, uint32_t adjustment_height, uint32_t adjustment_width,
    const uint8_t* input, const uint8_t* kernel, const int32_t* bias,
    uint8_t izp, uint8_t kzp, int32_t* acc)
{
  const size_t OH = oracle_deconv_output_dim(s->input_height, (size_t) s->pad_top + s->pad_bottom, adjustment_height,
      s->kernel_height, s->dilation_height, s->stride_height);
  const size_t OW = oracle_deconv_output_dim(s->input_width, (size_t) s->pad_left + s->pad_right, adjustment_width,
      s->kernel_width, s->dilation_width, s->stride_width);
  const size_t G = s->groups, GIC = s->group_input_channels, GOC = s->group_output_channels;
  const size_t KH = s->kernel_height, KW = s->kernel_width;
  const ptrdiff_t rows = (ptrdiff_t) (s->batch * OH);

#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (ptrdiff_t row = 0; row < rows; row++) {
    const size_t i = (size_t) row / OH;
    const size_t oy = (size_t) row % OH;
    for (size_t ox = 0; ox < OW; ox++) {
      int32_t* out = acc + (((i * OH + oy) * OW + ox) * G) * GOC;
      for (size_t g = 0; g < G; g++) {
        for (size_t oc = 0; oc < GOC; oc++) {
          uint32_t sum = (uint32_t) bias[g * GOC + oc];        /* :388-389 */
          for (size_t ky = 0; ky < KH; ky++) {
            /* :399-401, size_t arithmetic: a negative y wraps, its quotient fails `< input_height` */
            const size_t y = oy + s->pad_top - ky * s->dilation_height;
            const size_t iy = y / s->stride_height;
            if (iy * s->stride_height != y || iy >= s->input_height) continue;
            for (size_t kx = 0; kx < KW; kx++) {
              const size_t x = ox + s->pad_left - kx * s->dilation_width;   /* :403-405 */
              const size_t ix = x / s->stride_width;
              if (ix * s->stride_width != x || ix >= s->input_width) continue;
              const uint8_t* in = input +
                  ((i * s->input_height + iy) * s->input_width + ix) * s->input_pixel_stride + g * GIC;
              for (size_t ic = 0; ic < GIC; ic++) {
                const uint8_t kv = kernel[(((g * GIC + ic) * KH + ky) * KW + kx) * GOC + oc];   /* :411 */
                sum += (uint32_t) (((int32_t) in[ic] - (int32_t) izp) * ((int32_t) kv - (int32_t) kzp));
              }
            }
          }
          out[g * GOC + oc] = (int32_t) sum;
        }
      }
    }
  }
}

int oracle_requantize_rows(
    size_t rows, size_t cols, const int32_t* acc,
    float scale, uint8_t ozp, uint8_t omin, uint8_t omax,
    uint8_t* out, size_t out_stride)
{
  struct oracle_q31_params p;
  if (oracle_q31_params_init(scale, ozp, omin, omax, &p) != 0) return -1;
#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (ptrdiff_t r = 0; r < (ptrdiff_t) rows; r++) {
    for (size_t c = 0; c < cols; c++) {
      out[(size_t) r * out_stride + c] = oracle_q31_requantize(acc[(size_t) r * cols + c], &p);
    }
  }
  return 0;
}

int oracle_fully_connected_q8(
    size_t batch, size_t input_channels, size_t output_channels,
    uint8_t izp, float input_scale, uint8_t kzp, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias,
    uint8_t ozp, float output_scale, uint8_t omin, uint8_t omax,
    const uint8_t* input, size_t input_stride,
    uint8_t* output, size_t output_stride)
{
  if (batch == 0) return 0;
  /* src/fully-connected.c:71 */
  const float scale = input_scale * kernel_scale / output_scale;
  int32_t* acc = (int32_t*) malloc(sizeof(int32_t) * batch * output_channels);
  if (acc == NULL) return -2;
  oracle_gemm_acc(batch, output_channels, input_channels, input, input_stride, kernel, bias, izp, kzp, acc);
  const int rc = oracle_requantize_rows(batch, output_channels, acc, scale, ozp, omin, omax, output, output_stride);
  free(acc);
  return rc;
}

int oracle_convolution2d_q8(
    const struct oracle_conv_shape* s,
    uint8_t izp, float input_scale, uint8_t kzp, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias,
    uint8_t ozp, float output_scale, uint8_t omin, uint8_t omax,
    const uint8_t* input, uint8_t* output, size_t output_pixel_stride)
{
  if (s->batch == 0) return 0;
  const size_t OH = oracle_conv_output_dim(
      s->pad_top + s->input_height + s->pad_bottom, s->kernel_height, s->dilation_height, s->stride_height);
  const size_t OW = oracle_conv_output_dim(
      s->pad_left + s->input_width + s->pad_right, s->kernel_width, s->dilation_width, s->stride_width);
  const size_t cols = (size_t) s->groups * s->group_output_channels;
  const size_t rows = s->batch * OH * OW;
  /* src/convolution.c:161 */
  const float scale = input_scale * kernel_scale / output_scale;
  int32_t* acc = (int32_t*) malloc(sizeof(int32_t) * rows * cols);
  if (acc == NULL) return -2;
  oracle_conv2d_acc(s, input, kernel, bias, izp, kzp, acc);
  const int rc = oracle_requantize_rows(rows, cols, acc, scale, ozp, omin, omax, output, output_pixel_stride);
  free(acc);
  return rc;
}

/* ---- element-wise add ------------------------------------------------------------------------------- */

static inline float f32_from_bits(uint32_t bits) {
  float f;
  memcpy(&f, &bits, sizeof(f));
  return f;
}

int oracle_add_q8(
    size_t batch, size_t channels,
    uint8_t a_zero_point, float a_scale, uint8_t b_zero_point, float b_scale,
    uint8_t y_zero_point, float y_scale, uint8_t y_min, uint8_t y_max,
    const uint8_t* a, size_t a_stride, const uint8_t* b, size_t b_stride, uint8_t* y, size_t y_stride)
{
  /* src/add.c:73-89 */
  const float a_output_scale = a_scale / y_scale;
  const float b_output_scale = b_scale / y_scale;
  if (a_output_scale < 0x1.0p-14f || a_output_scale >= 0x1.0p+8f) return -1;
  if (b_output_scale < 0x1.0p-14f || b_output_scale >= 0x1.0p+8f) return -1;
  /* src/qnnpack/requantization.h:341-359 */
  const float max_output_scale = a_output_scale > b_output_scale ? a_output_scale : b_output_scale;
  const int32_t max_scale_exponent = (int32_t) (f32_bits(max_output_scale) >> 23) - 127;
  const uint32_t shift = (uint32_t) (21 - max_scale_exponent);
  const float scale_multiplier = f32_from_bits((uint32_t) (21 - max_scale_exponent + 127) << 23);
  const uint32_t a_multiplier = (uint32_t) (int32_t) lrintf(a_output_scale * scale_multiplier);
  const uint32_t b_multiplier = (uint32_t) (int32_t) lrintf(b_output_scale * scale_multiplier);
  /* :400-413 (scalar member) */
  const uint32_t remainder_mask = (UINT32_C(1) << shift) - UINT32_C(1);
  const uint32_t remainder_threshold = remainder_mask >> 1;
  const int32_t zero_point_product =
      (int32_t) -(a_multiplier * (uint32_t) a_zero_point + b_multiplier * (uint32_t) b_zero_point);

#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (ptrdiff_t r = 0; r < (ptrdiff_t) batch; r++) {
    for (size_t c = 0; c < channels; c++) {
      /* :500-522 */
      const uint8_t av = a[(size_t) r * a_stride + c], bv = b[(size_t) r * b_stride + c];
      int32_t acc = (int32_t) ((uint32_t) zero_point_product + (uint32_t) av * a_multiplier + (uint32_t) bv * b_multiplier);
      const int32_t rem = (acc & (int32_t) remainder_mask) - (int32_t) (acc < 0);
      acc = asr32(acc, shift) + (int32_t) (rem > (int32_t) remainder_threshold);
      int32_t v = acc + (int32_t) y_zero_point;
      if (v >= (int32_t) y_max) v = (int32_t) y_max;
      if (v <= (int32_t) y_min) v = (int32_t) y_min;
      y[(size_t) r * y_stride + c] = (uint8_t) v;
    }
  }
  return 0;
}

/* ---- global average pooling ------------------------------------------------------------------------- */

int oracle_global_average_pooling_q8(
    size_t batch, size_t width, size_t channels,
    uint8_t input_zero_point, float input_scale, uint8_t output_zero_point, float output_scale,
    uint8_t output_min, uint8_t output_max,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride)
{
  /* src/global-average-pooling.c:138-145 */
  const int32_t bias = -(int32_t) width * (int32_t) (uint32_t) input_zero_point;
  const float scale = input_scale / (output_scale * (float) width);
  if (!(scale >= 0x1.0p-32f) || !(scale < 256.0f)) return -1;
  /* src/qnnpack/requantization.h:210-222, :252-265 */
  const uint32_t scale_bits = f32_bits(scale);
  const int32_t multiplier = ((int32_t) scale_bits & INT32_C(0x007FFFFF)) | INT32_C(0x00800000);
  const uint32_t right_shift = (uint32_t) (127 + 23 - (int32_t) (scale_bits >> 23));
  const int64_t rounding = INT64_C(1) << (right_shift - 1);
  const int32_t min_less_zp = (int32_t) (uint32_t) output_min - (int32_t) (uint32_t) output_zero_point;
  const int32_t max_less_zp = (int32_t) (uint32_t) output_max - (int32_t) (uint32_t) output_zero_point;

#pragma omp parallel for num_threads(g_threads) schedule(static)
  for (ptrdiff_t i = 0; i < (ptrdiff_t) batch; i++) {
    for (size_t c = 0; c < channels; c++) {
      int32_t n = bias;
      for (size_t w = 0; w < width; w++) {
        n += (int32_t) input[((size_t) i * width + w) * input_stride + c];
      }
      /* :482-498 */
      const int64_t product = (int64_t) n * (int64_t) multiplier;
      const int64_t adjusted = product - (int64_t) (n < 0);
      int64_t shifted = adjusted + rounding;
      shifted = shifted >= 0 ? shifted >> right_shift : ~(~shifted >> right_shift);   /* asr_s64 */
      n = (int32_t) shifted;
      if (n < min_less_zp) n = min_less_zp;
      if (n > max_less_zp) n = max_less_zp;
      output[(size_t) i * output_stride + c] = (uint8_t) (n + (int32_t) output_zero_point);
    }
  }
  return 0;
}
