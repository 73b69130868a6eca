/*
 * oracle_q8.h -- CPU restatement ("O1") of QNNPACK's uint8 GEMM / conv2d /
 * depthwise hot path, used ONLY as the checker for the gfx950 HIP path.
 *
 * TEST INFRASTRUCTURE. Nothing under qnnpack_amd/ (the product) may include,
 * link or call this. Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the pytorch/QNNPACK tree). The restatement is scalar on purpose: it follows
 * the arithmetic the reference's own test harnesses use as ground truth
 * (naive int32 accumulate, then qnnp_q31_requantize), not the SSE2/NEON
 * microkernels.
 *
 * Parity pinning: tests/test_oracle_requant.py re-hosts the reference's
 * deterministic known-answer tests (test/requantization-tester.h:84-246 as
 * driven for Q31 in test/requantization.cc:250-296); tests/golden/ holds
 * outputs of the compiled reference library (oracle/_ref, built from
 * /root/reference by oracle/Makefile) on seeded inputs, which this oracle
 * must reproduce byte-for-byte.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* src/qnnpack/params.h:94-104 (scalar member of qnnp_q31_requantization_params) */
struct oracle_q31_params {
  int32_t multiplier;
  int32_t remainder_mask;
  int32_t remainder_threshold;
  uint32_t shift;
  int32_t min_less_zero_point;
  int32_t max_less_zero_point;
  int32_t zero_point;
};

/* src/qnnpack/requantization.h:22-54. Returns 0, or -1 if scale is outside
 * [2^-32, 1) (the reference asserts). */
int oracle_q31_params_init(
    float scale, uint8_t zero_point, uint8_t qmin, uint8_t qmax,
    struct oracle_q31_params* params);

/* src/qnnpack/requantization.h:464-480 (normative rounding). */
uint8_t oracle_q31_requantize(int32_t n, const struct oracle_q31_params* params);

/* Same contract as qnnp_requantize_q31__scalar, src/requantization/q31-scalar.c:17-138. */
int oracle_q31_requantize_array(
    size_t n, const int32_t* input, float scale, uint8_t zero_point,
    uint8_t qmin, uint8_t qmax, uint8_t* output);

/* test/gemm-microkernel-tester.h:213-226 / test/fully-connected-operator-tester.h:133-146:
 * acc[m*N + n] = bias[n] + sum_k (a[m*a_stride + k] - izp) * (w[n*K + k] - kzp). */
void oracle_gemm_acc(
    size_t M, size_t N, size_t K,
    const uint8_t* a, size_t a_stride,
    const uint8_t* w, const int32_t* bias,
    uint8_t izp, uint8_t kzp, int32_t* acc);

struct oracle_conv_shape {
  size_t batch, input_height, input_width;
  uint32_t pad_top, pad_right, pad_bottom, pad_left;
  uint32_t kernel_height, kernel_width;
  uint32_t stride_height, stride_width;
  uint32_t dilation_height, dilation_width;
  uint32_t groups;
  size_t group_input_channels, group_output_channels;
  size_t input_pixel_stride;
};

/* src/convolution.c:29-37 */
size_t oracle_conv_output_dim(size_t padded_input, size_t kernel, size_t dilation, size_t stride);

/* test/convolution-operator-tester.h:367-403 (7-loop direct convolution).
 * kernel layout [g][oc][ky][kx][ic]; acc layout [n][oy][ox][g*GOC + oc]. */
void oracle_conv2d_acc(
    const struct oracle_conv_shape* s,
    const uint8_t* input, const uint8_t* kernel, const int32_t* bias,
    uint8_t izp, uint8_t kzp, int32_t* acc);

/* src/deconvolution.c:25-37: stride*(input-1) + adjustment + (kernel-1)*dilation + 1 - padding */
size_t oracle_deconv_output_dim(
    size_t input, size_t padding, size_t adjustment, size_t kernel, size_t dilation, size_t stride);

/* test/deconvolution-operator-tester.h:383-419 (transposed convolution, the tester's ground truth).
 * kernel layout [g][ic][ky][kx][oc] (:411); acc layout [n][oy][ox][g*GOC + oc]. `s` holds the INPUT
 * geometry; pads are the amounts removed from the full output; stride_* is the deconvolution stride. */
void oracle_deconv2d_acc(
    const struct oracle_conv_shape* s, uint32_t adjustment_height, uint32_t adjustment_width,
    const uint8_t* input, const uint8_t* kernel, const int32_t* bias,
    uint8_t izp, uint8_t kzp, int32_t* acc);

/* Requantize a [rows][cols] accumulator matrix into out rows of out_stride bytes. */
int oracle_requantize_rows(
    size_t rows, size_t cols, const int32_t* acc,
    float scale, uint8_t ozp, uint8_t omin, uint8_t omax,
    uint8_t* out, size_t out_stride);

/* Whole operators = the two steps above, with scale = in_scale*k_scale/out_scale
 * (src/convolution.c:161, src/fully-connected.c:71). Return 0 / -1 (bad scale). */
int oracle_fully_connected_q8(
    size_t batch, size_t input_channels, size_t output_channels,
    uint8_t izp, float input_scale, uint8_t kzp, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias,
    uint8_t ozp, float output_scale, uint8_t omin, uint8_t omax,
    const uint8_t* input, size_t input_stride,
    uint8_t* output, size_t output_stride);

int oracle_convolution2d_q8(
    const struct oracle_conv_shape* s,
    uint8_t izp, float input_scale, uint8_t kzp, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias,
    uint8_t ozp, float output_scale, uint8_t omin, uint8_t omax,
    const uint8_t* input, uint8_t* output, size_t output_pixel_stride);

/* Quantized element-wise add, the whole operator. Parameters: the scalar member of
 * qnnp_compute_add_quantization_params (src/qnnpack/requantization.h:327-360, :400-413); arithmetic:
 * qnnp_add_quantize (:500-522), which test/vadd-microkernel-tester.h:180,194 asserts every microkernel equals.
 * Validation as src/add.c:73-89. Returns 0, or -1 when a scale ratio is outside [2^-14, 2^8). */
int oracle_add_q8(
    size_t batch, size_t channels,
    uint8_t a_zero_point, float a_scale, uint8_t b_zero_point, float b_scale,
    uint8_t y_zero_point, float y_scale, uint8_t y_min, uint8_t y_max,
    const uint8_t* a, size_t a_stride, const uint8_t* b, size_t b_stride, uint8_t* y, size_t y_stride);

/* Global average pooling, the whole operator: n = -width*izp + sum over the image's `width` pixels
 * (src/global-average-pooling.c:138-145, src/operator-run.c:981-1016), then qnnp_avgpool_quantize
 * (src/qnnpack/requantization.h:482-498) with the scalar parameters of :200-222, :252-265 for
 * scale = input_scale / (output_scale * width); test/gavgpool-microkernel-tester.h:177,198 asserts every
 * microkernel equals it. Returns 0, or -1 when the scale is outside [2^-32, 256). */
int oracle_global_average_pooling_q8(
    size_t batch, size_t width, size_t channels,
    uint8_t input_zero_point, float input_scale, uint8_t output_zero_point, float output_scale,
    uint8_t output_min, uint8_t output_max,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride);

/* OpenMP thread count used by the loops above (cpu_baseline "cores"). */
void oracle_set_threads(int n);
int oracle_get_threads(void);

#ifdef __cplusplus
}
#endif
