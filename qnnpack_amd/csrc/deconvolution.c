/*
 * deconvolution.c -- qnnp_create_deconvolution2d_nhwc_q8 / qnnp_setup_deconvolution2d_nhwc_q8
 * for the gfx950 build ("next" row 3).
 *
 * Replaces reference src/deconvolution.c:39-211 (create) and :213-277 (setup). A transposed convolution is
 * an implicit GEMM like any other once the (output pixel, tap) -> input pixel map is a table: output pixel
 * (oy, ox) takes tap (ky, kx) from input pixel ((oy + pad_top - ky*dil) / stride, ...) when the division is
 * exact and the pixel exists, and the input zero point otherwise (reference src/indirection.c:171-182). So
 *   create: the kernel arrives as [g][ic][ky][kx][oc] (reference pack.h:93-133 reads
 *           k[((ic * ks + ki) * n + oc]); it is transposed on the host to the convolution order
 *           [g][oc][ky][kx][ic] and goes through the SAME MFMA-fragment packer and bias folding as a
 *           convolution (pack.h; the reference's pack_q8deconv_w folds the bias exactly like pack_q8conv_w:
 *           b + ks*kc*izp*kzp - izp * sum(k), pack.h:105,123);
 *   setup : output extent per reference deconvolution.c:25-37; the batch-invariant int32 offset table of
 *           indirection.c (deconvolution flavour);
 *   run   : the generic offset-table MFMA implicit-GEMM kernel (q8igemm.hip) -- the geometry-derived
 *           convolution kernels do not apply, the operator pins "gemm_kernel" = 1.
 * Status codes and their order follow the reference (deconvolution.c:69-129, :225-243).
 */
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <qnnpack.h>

#include "hip/qnnp_hip.h"
#include "indirection.h"
#include "log.h"
#include "operator.h"
#include "pack.h"
#include "requantization.h"
#include "state.h"

/* reference src/deconvolution.c:25-37 */
static inline size_t compute_output_dimension(
    size_t input, size_t padding, size_t adjustment, size_t kernel, size_t dilation, size_t stride)
{
  const size_t effective_kernel = (kernel - 1) * dilation + 1;
  return stride * (input - 1) + adjustment + effective_kernel - padding;
}

static inline bool scale_is_valid(float scale)
{
  return scale > 0.0f && isnormal(scale);
}

enum qnnp_status qnnp_create_deconvolution2d_nhwc_q8(
    uint32_t input_padding_top,
    uint32_t input_padding_right,
    uint32_t input_padding_bottom,
    uint32_t input_padding_left,
    uint32_t adjustment_height,
    uint32_t adjustment_width,
    uint32_t kernel_height,
    uint32_t kernel_width,
    uint32_t stride_height,
    uint32_t stride_width,
    uint32_t dilation_height,
    uint32_t dilation_width,
    uint32_t groups,
    size_t group_input_channels,
    size_t group_output_channels,
    uint8_t input_zero_point,
    float input_scale,
    uint8_t kernel_zero_point,
    float kernel_scale,
    const uint8_t* kernel,
    const int32_t* bias,
    uint8_t output_zero_point,
    float output_scale,
    uint8_t output_min,
    uint8_t output_max,
    uint32_t flags,
    qnnp_operator_t* deconvolution_out)
{
  (void) flags;
  qnnp_operator_t op = NULL;
  uint8_t* conv_order = NULL;
  void* host_weights = NULL;
  int32_t* host_bias = NULL;
  enum qnnp_status status = qnnp_status_uninitialized;

  /* reference deconvolution.c:69-72 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_create_deconvolution2d_nhwc_q8 failed because QNNPACK is not properly initialized");
    goto error;
  }

  /* reference deconvolution.c:74-116 */
  status = qnnp_status_invalid_parameter;
  if (kernel_width == 0 || kernel_height == 0) {
    qnnp_log_error("failed to create deconvolution with %" PRIu32 "x%" PRIu32 " kernel: kernel dimensions must be non-zero",
        kernel_width, kernel_height);
    goto error;
  }
  if (stride_width == 0 || stride_height == 0) {
    qnnp_log_error("failed to create deconvolution with %" PRIu32 "x%" PRIu32 " stride: stride dimensions must be non-zero",
        stride_width, stride_height);
    goto error;
  }
  if (dilation_width == 0 || dilation_height == 0) {
    qnnp_log_error("failed to create deconvolution with %" PRIu32 "x%" PRIu32 " dilation: dilation dimensions must be non-zero",
        dilation_width, dilation_height);
    goto error;
  }
  if (!scale_is_valid(input_scale)) {
    qnnp_log_error("failed to create deconvolution with %.7g input scale: scale must be finite and positive", input_scale);
    goto error;
  }
  if (!scale_is_valid(kernel_scale)) {
    qnnp_log_error("failed to create deconvolution with %.7g kernel scale: scale must be finite and positive", kernel_scale);
    goto error;
  }
  if (!scale_is_valid(output_scale)) {
    qnnp_log_error("failed to create deconvolution with %.7g output scale: scale must be finite and positive", output_scale);
    goto error;
  }
  if (groups == 0 || group_input_channels == 0 || group_output_channels == 0 || kernel == NULL || bias == NULL) {
    qnnp_log_error("failed to create deconvolution: groups, channel counts, kernel and bias must be non-zero");
    goto error;
  }

  /* reference deconvolution.c:118-128 */
  status = qnnp_status_unsupported_parameter;
  const float deconvolution_scale = input_scale * kernel_scale / output_scale;
  if (deconvolution_scale >= 1.0f) {
    qnnp_log_error(
        "failed to create deconvolution with %.7g input scale, %.7g kernel scale, and %.7g output scale: "
        "deconvolution scale %.7g is greater or equal to 1.0",
        input_scale, kernel_scale, output_scale, deconvolution_scale);
    goto error;
  }
  if (!(deconvolution_scale >= 0x1.0p-32f)) {
    qnnp_log_error("failed to create deconvolution: deconvolution scale %.7g is below 2**-32", deconvolution_scale);
    goto error;
  }
  const size_t kernel_size = (size_t) kernel_height * kernel_width;
  if (kernel_size * group_input_channels > (size_t) UINT32_MAX / 4 ||
      (size_t) groups * group_output_channels > (size_t) UINT32_MAX / 4) {
    qnnp_log_error("failed to create deconvolution: channel / kernel extents exceed the 32-bit index range of the device kernels");
    goto error;
  }

  status = qnnp_status_out_of_memory;
  op = calloc(1, sizeof(struct qnnp_operator));
  if (op == NULL) {
    qnnp_log_error("failed to allocate %zu bytes for qnnp_operator structure", sizeof(struct qnnp_operator));
    goto error;
  }

  /* [g][ic][tap][oc] -> [g][oc][tap][ic] */
  const size_t gic = group_input_channels, goc = group_output_channels;
  const size_t group_weights = goc * kernel_size * gic;
  conv_order = (uint8_t*) malloc(group_weights * groups);
  if (conv_order == NULL) {
    qnnp_log_error("failed to allocate %zu bytes for the transposed kernel", group_weights * groups);
    goto error;
  }
  for (size_t g = 0; g < groups; g++) {
    const uint8_t* src = kernel + g * group_weights;
    uint8_t* dst = conv_order + g * group_weights;
    for (size_t ic = 0; ic < gic; ic++) {
      for (size_t tap = 0; tap < kernel_size; tap++) {
        const uint8_t* row = src + (ic * kernel_size + tap) * goc;
        for (size_t oc = 0; oc < goc; oc++) {
          dst[(oc * kernel_size + tap) * gic + ic] = row[oc];
        }
      }
    }
  }

  const uint32_t kc_slot = (uint32_t) gic;
  const uint32_t k_total = (uint32_t) (kernel_size * kc_slot);
  const uint32_t n_pad = qnnp_round_up_u32((uint32_t) goc, 32);
  const uint32_t k_pad = qnnp_round_up_u32(k_total, 64);
  const size_t w_bytes = qnnp_igemm_packed_weights_size(groups, n_pad, k_pad);
  const size_t b_bytes = sizeof(int32_t) * (size_t) groups * n_pad;
  host_weights = malloc(w_bytes);
  host_bias = malloc(b_bytes);
  if (host_weights == NULL || host_bias == NULL) {
    qnnp_log_error("failed to allocate %zu bytes for packed weights", w_bytes + b_bytes);
    goto error;
  }
  qnnp_pack_igemm_w_slots(groups, (uint32_t) goc, (uint32_t) kernel_size, (uint32_t) gic, kc_slot, n_pad, k_pad,
      input_zero_point, kernel_zero_point, conv_order, bias, (int8_t*) host_weights, host_bias);
  op->n_pad = n_pad;
  op->k_pad = k_pad;
  op->kc_slot = kc_slot;
  op->d_weights = qnnp_hip_alloc(w_bytes);
  op->d_bias = (int32_t*) qnnp_hip_alloc(b_bytes);
  if (op->d_weights == NULL || op->d_bias == NULL ||
      qnnp_hip_h2d(op->d_weights, host_weights, w_bytes, 0) != QNNP_HIP_OK ||
      qnnp_hip_h2d(op->d_bias, host_bias, b_bytes, 0) != QNNP_HIP_OK) {
    qnnp_log_error("failed to place %zu bytes of packed weights on the device", w_bytes + b_bytes);
    goto error;
  }
  free(conv_order);
  free(host_weights);
  free(host_bias);
  conv_order = NULL;
  host_weights = NULL;
  host_bias = NULL;

  op->input_padding_top = input_padding_top;
  op->input_padding_right = input_padding_right;
  op->input_padding_bottom = input_padding_bottom;
  op->input_padding_left = input_padding_left;
  op->adjustment_height = adjustment_height;
  op->adjustment_width = adjustment_width;
  op->kernel_height = kernel_height;
  op->kernel_width = kernel_width;
  op->stride_height = stride_height;
  op->stride_width = stride_width;
  op->dilation_height = dilation_height;
  op->dilation_width = dilation_width;
  op->groups = groups;
  op->group_input_channels = group_input_channels;
  op->group_output_channels = group_output_channels;
  op->input_zero_point = input_zero_point;
  op->kernel_zero_point = kernel_zero_point;
  op->requant = qnnp_compute_requant(deconvolution_scale, output_zero_point, output_min, output_max);
  op->ukernel_type = qnnp_ukernel_type_conv;   /* reference deconvolution.c:203 */
  op->transposed = 1;

  *deconvolution_out = op;
  return qnnp_status_success;

error:
  free(conv_order);
  free(host_weights);
  free(host_bias);
  qnnp_delete_operator(op);
  return status;
}

enum qnnp_status qnnp_setup_deconvolution2d_nhwc_q8(
    qnnp_operator_t op,
    size_t batch_size,
    size_t input_height,
    size_t input_width,
    const uint8_t* input,
    size_t input_pixel_stride,
    uint8_t* output,
    size_t output_pixel_stride,
    pthreadpool_t threadpool)
{
  (void) threadpool;

  /* reference deconvolution.c:225-228 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_setup_deconvolution2d_nhwc_q8 failed because QNNPACK is not properly initialized");
    return qnnp_status_uninitialized;
  }
  if (op == NULL || !op->transposed) {
    return qnnp_status_invalid_parameter;
  }

  /* reference deconvolution.c:230-233 */
  if (batch_size == 0) {
    op->batch_size = 0;
    return qnnp_status_success;
  }

  /* reference deconvolution.c:235-241 */
  if (input_width == 0 || input_height == 0) {
    qnnp_log_error("failed to setup deconvolution with %zux%zu input: input dimensions must be non-zero",
        input_width, input_height);
    return qnnp_status_invalid_parameter;
  }
  const size_t in_channels = (size_t) op->groups * op->group_input_channels;
  const size_t out_channels = (size_t) op->groups * op->group_output_channels;
  if (input == NULL || output == NULL || input_pixel_stride < in_channels || output_pixel_stride < out_channels) {
    qnnp_log_error("failed to setup deconvolution: NULL tensor or pixel stride smaller than the channel count");
    return qnnp_status_invalid_parameter;
  }
  const size_t pad_h = (size_t) op->input_padding_top + op->input_padding_bottom;
  const size_t pad_w = (size_t) op->input_padding_left + op->input_padding_right;
  const size_t full_h = compute_output_dimension(input_height, 0, op->adjustment_height, op->kernel_height,
      op->dilation_height, op->stride_height);
  const size_t full_w = compute_output_dimension(input_width, 0, op->adjustment_width, op->kernel_width,
      op->dilation_width, op->stride_width);
  if (pad_h >= full_h || pad_w >= full_w) {
    qnnp_log_error("failed to setup deconvolution with %zux%zu input: the padding removes the whole output",
        input_width, input_height);
    return qnnp_status_invalid_parameter;
  }

  /* reference deconvolution.c:243-262 */
  op->batch_size = batch_size;
  op->input_height = input_height;
  op->input_width = input_width;
  op->input = input;
  op->input_pixel_stride = input_pixel_stride;
  op->output_height = full_h - pad_h;
  op->output_width = full_w - pad_w;
  op->output = output;
  op->output_pixel_stride = output_pixel_stride;

  const size_t output_size = op->output_height * op->output_width;
  const size_t input_size = input_height * input_width;
  if (batch_size * output_size > (size_t) UINT32_MAX / 2 || input_size * input_pixel_stride > (size_t) INT32_MAX) {
    qnnp_log_error("failed to setup deconvolution: %zu output pixels / %zu-byte images exceed the device kernels' index range",
        batch_size * output_size, input_size * input_pixel_stride);
    return qnnp_status_unsupported_parameter;
  }

  op->input_span = (batch_size * input_size - 1) * input_pixel_stride + in_channels;
  op->output_span = (batch_size * output_size - 1) * output_pixel_stride + out_channels;
  if (qnnp_bind_endpoint(input, op->input_span, &op->input_on_device, &op->d_stage_in, &op->stage_in_capacity) != 0 ||
      qnnp_bind_endpoint(output, op->output_span, &op->output_on_device, &op->d_stage_out, &op->stage_out_capacity) != 0) {
    qnnp_log_error("failed to allocate device staging for host tensors (%zu + %zu bytes)",
        op->input_span, op->output_span);
    return qnnp_status_out_of_memory;
  }

  op->variant = 1;   /* the offset-table kernel: the table, not the geometry, defines this operator */
  const size_t kernel_size = (size_t) op->kernel_height * op->kernel_width;
  const size_t entries = output_size * kernel_size;
  const bool same_geometry = op->d_offsets != NULL &&
      op->offsets_in_h == input_height && op->offsets_in_w == input_width &&
      op->offsets_in_stride == input_pixel_stride;
  if (same_geometry) {
    return qnnp_status_success;  /* the table is pointer- and batch-invariant */
  }
  int32_t* host_table = (int32_t*) malloc(sizeof(int32_t) * entries);
  if (host_table == NULL) {
    qnnp_log_error("failed to allocate %zu bytes for the offset table", sizeof(int32_t) * entries);
    return qnnp_status_out_of_memory;
  }
  if (op->offsets_capacity < entries) {
    qnnp_hip_free(op->d_offsets);
    op->offsets_capacity = 0;
    op->d_offsets = (int32_t*) qnnp_hip_alloc(sizeof(int32_t) * entries);
    if (op->d_offsets == NULL) {
      free(host_table);
      qnnp_log_error("failed to allocate %zu bytes for the device offset table", sizeof(int32_t) * entries);
      return qnnp_status_out_of_memory;
    }
    op->offsets_capacity = entries;
  }
  qnnp_indirection_init_deconv2d_offsets(op, host_table);
  const int rc = qnnp_hip_h2d(op->d_offsets, host_table, sizeof(int32_t) * entries, 0);
  free(host_table);
  if (rc != QNNP_HIP_OK) {
    op->offsets_in_h = 0;
    qnnp_log_error("failed to upload the offset table");
    return qnnp_status_out_of_memory;
  }
  op->offsets_in_h = input_height;
  op->offsets_in_w = input_width;
  op->offsets_in_stride = input_pixel_stride;
  return qnnp_status_success;
}
