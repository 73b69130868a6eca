/*
 * fully-connected.c -- qnnp_create_fully_connected_nc_q8 / qnnp_setup_fully_connected_nc_q8.
 *
 * Replaces reference src/fully-connected.c:25-129 (create) and :131-161 (setup).
 * As there, a fully connected layer is a single-group GEMM operator whose rows
 * are the batch (setup maps batch -> rows, :149-158). The weights go to the
 * device as MFMA operand fragments (pack.h) instead of the 4x4c2 CPU panels of
 * pack_q8gemm_w (src/qnnpack/pack.h:12-49).
 */
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <qnnpack.h>

#include "hip/qnnp_hip.h"
#include "log.h"
#include "operator.h"
#include "bias-pair.h"
#include "pack.h"
#include "requantization.h"
#include "state.h"

static inline bool scale_is_valid(float scale)
{
  return scale > 0.0f && isnormal(scale);
}

static enum qnnp_status qnnp_create_fully_connected_nc_q8_impl(
    size_t input_channels,
    size_t output_channels,
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
    qnnp_operator_t* fully_connected_out)
{
  (void) flags; /* ignored, reference fully-connected.c:38 */
  qnnp_operator_t op = NULL;
  int8_t* host_weights = NULL;
  int32_t* host_bias = NULL;
  enum qnnp_status status = qnnp_status_uninitialized;

  /* reference fully-connected.c:44-47 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_create_fully_connected_nc_q8 called before qnnp_initialize succeeded");
    goto error;
  }

  /* reference fully-connected.c:49-67 */
  status = qnnp_status_invalid_parameter;
  if (!scale_is_valid(input_scale)) {
    qnnp_log_error("cannot create fully connected operator with %.7g input scale: a scale has to be a finite number above zero", input_scale);
    goto error;
  }
  if (!scale_is_valid(kernel_scale)) {
    qnnp_log_error("cannot create fully connected operator with %.7g kernel scale: a scale has to be a finite number above zero", kernel_scale);
    goto error;
  }
  if (!scale_is_valid(output_scale)) {
    qnnp_log_error("cannot create fully connected operator with %.7g output scale: a scale has to be a finite number above zero", output_scale);
    goto error;
  }
  if (input_channels == 0 || output_channels == 0 || kernel == NULL || bias == NULL) {
    qnnp_log_error("cannot create fully connected operator: channel counts, kernel and bias may not be zero");
    goto error;
  }

  /* reference fully-connected.c:69-78 */
  status = qnnp_status_unsupported_parameter;
  const float requantization_scale = input_scale * kernel_scale / output_scale;
  if (requantization_scale >= 1.0f) {
    qnnp_log_error(
        "cannot create fully connected operator with %.7g input scale, %.7g kernel scale, and %.7g output scale: "
        "requantization scale %.7g is greater or equal to 1.0",
        input_scale, kernel_scale, output_scale, requantization_scale);
    goto error;
  }
  if (!(requantization_scale >= 0x1.0p-32f)) {
    qnnp_log_error("cannot create fully connected operator: requantization scale %.7g is below 2**-32", requantization_scale);
    goto error;
  }
  if (input_channels > (size_t) UINT32_MAX / 4 || output_channels > (size_t) UINT32_MAX / 4) {
    qnnp_log_error("cannot create fully connected operator: channel counts exceed the device kernels' 32-bit index range");
    goto error;
  }

  status = qnnp_status_out_of_memory;
  op = calloc(1, sizeof(struct qnnp_operator));
  if (op != NULL) op->device = qnnp_hip_device();   /* the context this create runs in (entry point below) */
  if (op == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for qnnp_operator structure", sizeof(struct qnnp_operator));
    goto error;
  }

  const uint32_t n_pad = qnnp_round_up_u32((uint32_t) output_channels, 32);
  const uint32_t k_pad = qnnp_round_up_u32((uint32_t) input_channels, 64);
  const size_t w_bytes = qnnp_igemm_packed_weights_size(1, n_pad, k_pad);
  const size_t b_bytes = sizeof(int32_t) * n_pad;
  host_weights = (int8_t*) malloc(w_bytes);
  host_bias = (int32_t*) malloc(b_bytes);
  if (host_weights == NULL || host_bias == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for packed weights", w_bytes + b_bytes);
    goto error;
  }
  qnnp_pack_igemm_w(1, (uint32_t) output_channels, (uint32_t) input_channels, n_pad, k_pad,
      input_zero_point, kernel_zero_point, kernel, bias, host_weights, host_bias);
  op->n_pad = n_pad;
  op->k_pad = k_pad;
  op->kc_slot = (uint32_t) input_channels;
  op->d_weights = qnnp_hip_alloc(w_bytes);
  op->d_bias = qnnp_upload_bias_pair(host_bias, n_pad);      /* bias-pair.h */
  if (op->d_weights == NULL || op->d_bias == NULL ||
      qnnp_hip_h2d(op->d_weights, host_weights, w_bytes, 0) != QNNP_HIP_OK) {
    qnnp_log_error("device allocation or upload failed: %zu bytes of packed weights on the device", w_bytes + 2 * b_bytes);
    goto error;
  }
  /* The big GEMM kernel's zero-point-centred image (pack.h, hip/q8gemm256c.hip): kernel zero point 128 IS the
   * standard image; 127 (the reference benchmarks' value, bench/q8gemm.cc:60-64) gets its own, for the shapes that
   * kernel takes (no K padding, channel blocks of 256, MFMA-bound sizes). */
  if (kernel_zero_point == 128) {
    op->centre_flip = 0x80;
  } else if (kernel_zero_point == 127 && k_pad == input_channels &&
             /* (the 256-wide tiling's shapes, or the 128-wide one's: hip/q8gemm128x.hip takes any K % 64 == 0) */
             ((n_pad % 256 == 0 && input_channels >= 512) || (input_channels % 64 == 0 && output_channels % 16 == 0))) {
    qnnp_pack_igemm_w_centred127((uint32_t) output_channels, (uint32_t) input_channels, (uint32_t) input_channels, n_pad,
        input_zero_point, kernel, bias, host_weights, host_bias);
    op->d_weights_centred = qnnp_hip_alloc(w_bytes);
    op->d_bias_centred = qnnp_upload_bias_pair(host_bias, n_pad);
    if (op->d_weights_centred == NULL || op->d_bias_centred == NULL ||
        qnnp_hip_h2d(op->d_weights_centred, host_weights, w_bytes, 0) != QNNP_HIP_OK) {
      /* the centred image is an optimisation, not a requirement: without it the operator runs on the standard image
       * (the lean kernel with its row term) -- drop what was placed and carry on */
      qnnp_log_warning("no room for %zu bytes of centred weights on the device: the operator keeps the standard image", w_bytes + 2 * b_bytes);
      qnnp_hip_free(op->d_weights_centred);
      qnnp_hip_free(op->d_bias_centred);
      op->d_weights_centred = NULL;
      op->d_bias_centred = NULL;
    } else {
      op->centre_flip = 0x7F;
    }
  }
  free(host_weights);
  free(host_bias);
  host_weights = NULL;
  host_bias = NULL;

  /* reference fully-connected.c:109-121 */
  op->kernel_height = 1;
  op->kernel_width = 1;
  op->stride_height = 1;
  op->stride_width = 1;
  op->dilation_height = 1;
  op->dilation_width = 1;
  op->groups = 1;
  op->group_input_channels = input_channels;
  op->group_output_channels = output_channels;
  op->input_zero_point = input_zero_point;
  op->kernel_zero_point = kernel_zero_point;
  op->requant = qnnp_compute_requant(requantization_scale, output_zero_point, output_min, output_max);
  op->requant.accumulator_bits = qnnp_accumulator_bits(bias, output_channels, input_channels);
  op->ukernel_type = qnnp_ukernel_type_gemm;

  *fully_connected_out = op;
  return qnnp_status_success;

error:
  free(host_weights);
  free(host_bias);
  qnnp_delete_operator(op);
  return status;
}

static enum qnnp_status qnnp_setup_fully_connected_nc_q8_impl(
    qnnp_operator_t op,
    size_t batch_size,
    const uint8_t* input,
    size_t input_stride,
    uint8_t* output,
    size_t output_stride)
{
  /* reference fully-connected.c:139-142 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_setup_fully_connected_nc_q8 called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  if (op == NULL || op->ukernel_type != qnnp_ukernel_type_gemm || op->transposed || op->group_input_channels == 0) {
    return qnnp_status_invalid_parameter;   /* not a handle from a fully connected / 1x1 create */
  }

  /* reference fully-connected.c:144-147 */
  if (batch_size == 0) {
    op->batch_size = 0;
    return qnnp_status_success;
  }
  if (input == NULL || output == NULL ||
      input_stride < op->group_input_channels || output_stride < op->group_output_channels) {
    qnnp_log_error("cannot set up fully connected operator: NULL tensor or stride smaller than the channel count");
    return qnnp_status_invalid_parameter;
  }
  if (batch_size > (size_t) UINT32_MAX / 2) {
    qnnp_log_error("cannot set up fully connected operator: batch %zu exceeds the device kernels' index range", batch_size);
    return qnnp_status_unsupported_parameter;
  }

  /* reference fully-connected.c:149-158: the batch becomes the row dimension */
  op->setup_valid = 0;   /* until every check, allocation and upload below has succeeded */
  op->residual = NULL;   /* an attached residual add belongs to the previous binding (residual.c), as in convolution.c */
  op->residual_folded = 0;
  op->batch_size = 1;
  op->input_height = batch_size;
  op->input_width = 1;
  op->input = input;
  op->input_pixel_stride = input_stride;
  op->output_height = batch_size;
  op->output_width = 1;
  op->output = output;
  op->output_pixel_stride = output_stride;

  op->input_span = (batch_size - 1) * input_stride + op->group_input_channels;
  op->output_span = (batch_size - 1) * output_stride + op->group_output_channels;
  {
    enum qnnp_status bound = qnnp_bind_endpoint(input, op->input_span, &op->input_on_device, &op->d_stage_in, &op->stage_in_capacity);
    if (bound == qnnp_status_success) bound = qnnp_bind_endpoint(output, op->output_span, &op->output_on_device, &op->d_stage_out, &op->stage_out_capacity);
    if (bound != qnnp_status_success) {
      qnnp_log_error("failed to bind the tensors: device staging for host memory could not be allocated, or a tensor "
          "lives on a different device than the operator");
      return bound;
    }
  }
  op->variant = qnnp_state.opt_gemm_kernel;
  return qnnp_status_success;
}

/* ---- public entry points: run the implementation inside the right device context ------------------
 * create: the calling thread's selected device (qnnp_gfx950_set_device, default = the primary one) becomes the
 * operator's device; setup: the operator's device. The previous HIP device of the thread is restored on return. */

enum qnnp_status qnnp_create_fully_connected_nc_q8(
    size_t input_channels,
    size_t output_channels,
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
    qnnp_operator_t* fully_connected_out)
{
  if (!qnnp_state.initialized) {
    return qnnp_create_fully_connected_nc_q8_impl(input_channels, output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias, output_zero_point, output_scale, output_min, output_max, flags, fully_connected_out);   /* logs and answers qnnp_status_uninitialized */
  }
  const int token = qnnp_hip_enter(qnnp_hip_device());
  if (token < 0) {
    return qnnp_status_unsupported_hardware;
  }
  if (qnnp_hip_graph_capturing()) {
    /* inside qnnp_gfx950_graph_begin ... graph_end on this device only operator launches are recordable: an upload
     * would become a graph node reading host memory that is freed right after this call */
    qnnp_hip_leave(token);
    return qnnp_status_invalid_parameter;
  }
  const enum qnnp_status status = qnnp_create_fully_connected_nc_q8_impl(input_channels, output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias, output_zero_point, output_scale, output_min, output_max, flags, fully_connected_out);
  qnnp_hip_leave(token);
  return status;
}

enum qnnp_status qnnp_setup_fully_connected_nc_q8(
    qnnp_operator_t op,
    size_t batch_size,
    const uint8_t* input,
    size_t input_stride,
    uint8_t* output,
    size_t output_stride)
{
  if (!qnnp_state.initialized || op == NULL) {
    return qnnp_setup_fully_connected_nc_q8_impl(op, batch_size, input, input_stride, output, output_stride);   /* answers qnnp_status_uninitialized / invalid_parameter */
  }
  const int token = qnnp_hip_enter(op->device);
  if (token < 0) {
    return qnnp_status_invalid_parameter;   /* not a live operator of this library instance */
  }
  if (qnnp_hip_graph_capturing()) {
    /* inside qnnp_gfx950_graph_begin ... graph_end on this device only operator launches are recordable: an upload
     * would become a graph node reading host memory that is freed right after this call */
    qnnp_hip_leave(token);
    return qnnp_status_invalid_parameter;
  }
  const enum qnnp_status status = qnnp_setup_fully_connected_nc_q8_impl(op, batch_size, input, input_stride, output, output_stride);
  /* the implementation cleared setup_valid where it began to change the operator: a failed setup leaves it
   * unrunnable instead of half updated (run answers invalid_parameter) */
  if (status == qnnp_status_success) {
    op->setup_valid = 1;
  }
  qnnp_hip_leave(token);
  return status;
}
