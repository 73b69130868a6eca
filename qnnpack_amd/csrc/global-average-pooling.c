/*
 * global-average-pooling.c -- qnnp_create_global_average_pooling_nwc_q8 /
 * qnnp_setup_global_average_pooling_nwc_q8 for the gfx950 build (the pooling in
 * front of MobileNetV2's classifier).
 *
 * Replaces reference src/global-average-pooling.c:20-107 (create) and :109-148 (setup): same validation
 * order and status codes. The quantization parameters are the scalar member of
 * qnnp_compute_avgpool_quantization_params (reference src/qnnpack/requantization.h:200-222, :252-265),
 * recomputed at every setup from the pooled width exactly as the reference does (:138-145). The
 * reference's zero buffer (:79-85) has no equivalent: the device kernel never reads beyond `width` pixels.
 */
#include <math.h>
#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <qnnpack.h>

#include "hip/qnnp_hip.h"
#include "log.h"
#include "operator.h"
#include "state.h"

static inline bool scale_is_valid(float scale)
{
  return scale > 0.0f && isnormal(scale);
}

/* reference requantization.h:200-222 + scalar members :252-265 */
static struct qnnp_hip_avgpool_params compute_avgpool_params(
    int32_t bias, float scale, uint8_t output_zero_point, uint8_t output_min, uint8_t output_max)
{
  uint32_t scale_bits;
  memcpy(&scale_bits, &scale, sizeof(scale_bits));
  struct qnnp_hip_avgpool_params p;
  p.bias = bias;
  p.multiplier = ((int32_t) scale_bits & INT32_C(0x007FFFFF)) | INT32_C(0x00800000);   /* [2^23, 2^24) */
  const int32_t shift = 127 + 23 - (int32_t) (scale_bits >> 23);                        /* [16, 55] */
  p.right_shift = (uint32_t) shift;
  p.rounding = INT64_C(1) << (p.right_shift - 1);
  p.output_min_less_zero_point = (int32_t) (uint32_t) output_min - (int32_t) (uint32_t) output_zero_point;
  p.output_max_less_zero_point = (int32_t) (uint32_t) output_max - (int32_t) (uint32_t) output_zero_point;
  p.output_zero_point = (int32_t) (uint32_t) output_zero_point;
  return p;
}

static enum qnnp_status qnnp_create_global_average_pooling_nwc_q8_impl(
    size_t channels,
    uint8_t input_zero_point,
    float input_scale,
    uint8_t output_zero_point,
    float output_scale,
    uint8_t output_min,
    uint8_t output_max,
    uint32_t flags,
    qnnp_operator_t* global_average_pooling_out)
{
  (void) flags;
  /* reference global-average-pooling.c:34-37 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_create_global_average_pooling_nwc_q8 called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  /* reference global-average-pooling.c:39-59 */
  if (channels == 0) {
    qnnp_log_error("cannot create global average pooling operator with %zu channels: number of channels may not be zero",
        channels);
    return qnnp_status_invalid_parameter;
  }
  if (!scale_is_valid(input_scale)) {
    qnnp_log_error("cannot create global average pooling operator with %.7g input scale: a scale has to be a finite number above zero",
        input_scale);
    return qnnp_status_invalid_parameter;
  }
  if (!scale_is_valid(output_scale)) {
    qnnp_log_error("cannot create global average pooling operator with %.7g output scale: a scale has to be a finite number above zero",
        output_scale);
    return qnnp_status_invalid_parameter;
  }
  /* reference global-average-pooling.c:61-70 */
  const float input_output_scale = input_scale / output_scale;
  if (input_output_scale < 0x1.0p-8f || input_output_scale >= 0x1.0p+8f) {
    qnnp_log_error("cannot create global average pooling operator with %.7g input-to-output scale ratio: "
        "scale ratio must be in [2**-8, 2**8) range", input_output_scale);
    return qnnp_status_unsupported_parameter;
  }
  if (channels > (size_t) UINT32_MAX / 4) {
    qnnp_log_error("cannot create global average pooling operator: %zu channels exceed the device kernels' index range",
        channels);
    return qnnp_status_unsupported_parameter;
  }

  qnnp_operator_t op = calloc(1, sizeof(struct qnnp_operator));
  if (op != NULL) op->device = qnnp_hip_device();   /* the context this create runs in (entry point below) */
  if (op == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for qnnp_operator structure", sizeof(struct qnnp_operator));
    return qnnp_status_out_of_memory;
  }
  op->channels = channels;
  op->input_zero_point = input_zero_point;
  op->output_zero_point = output_zero_point;
  op->input_scale = input_scale;
  op->output_scale = output_scale;
  op->output_min = output_min;
  op->output_max = output_max;
  op->ukernel_type = qnnp_ukernel_type_global_average_pooling;
  *global_average_pooling_out = op;
  return qnnp_status_success;
}

static enum qnnp_status qnnp_setup_global_average_pooling_nwc_q8_impl(
    qnnp_operator_t op,
    size_t batch_size,
    size_t width,
    const uint8_t* input,
    size_t input_stride,
    uint8_t* output,
    size_t output_stride)
{
  /* reference global-average-pooling.c:118-121 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_setup_global_average_pooling_nwc_q8 called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  if (op == NULL || op->ukernel_type != qnnp_ukernel_type_global_average_pooling) {
    return qnnp_status_invalid_parameter;
  }
  /* reference global-average-pooling.c:123-126 */
  if (batch_size == 0) {
    op->batch_size = 0;
    return qnnp_status_success;
  }
  /* reference global-average-pooling.c:128-131 */
  if (width == 0) {
    qnnp_log_error("cannot set up global average pooling operator with width %zu: width may not be zero", width);
    return qnnp_status_invalid_parameter;
  }
  const size_t channels = op->channels;
  if (input == NULL || output == NULL || input_stride < channels || output_stride < channels) {
    qnnp_log_error("cannot set up global average pooling operator: NULL tensor or stride smaller than the channel count");
    return qnnp_status_invalid_parameter;
  }
  /* the accumulator of one output is width * 255 at most; the scale must stay inside the parameter builder's
   * range [2^-32, 256) (asserted by the reference, requantization.h:208-209) */
  const float scale = op->input_scale / (op->output_scale * (float) width);
  if (width > (size_t) INT32_MAX / 255 || !(scale >= 0x1.0p-32f) || !(scale < 256.0f)) {
    qnnp_log_error("cannot set up global average pooling operator with width %zu: outside the supported range", width);
    return qnnp_status_unsupported_parameter;
  }

  /* reference global-average-pooling.c:133-145 */
  op->setup_valid = 0;   /* until every check, allocation and upload below has succeeded */
  op->batch_size = batch_size;
  op->input_width = width;
  op->input = input;
  op->input_pixel_stride = input_stride;
  op->output = output;
  op->output_pixel_stride = output_stride;
  op->avgpool_params = compute_avgpool_params(
      -(int32_t) width * (int32_t) (uint32_t) op->input_zero_point, scale,
      op->output_zero_point, op->output_min, op->output_max);

  op->input_span = (batch_size * width - 1) * input_stride + channels;
  op->output_span = (batch_size - 1) * output_stride + channels;
  {
    enum qnnp_status bound = qnnp_status_success;
    if (bound == qnnp_status_success) bound = qnnp_bind_endpoint(input, op->input_span, &op->input_on_device, &op->d_stage_in, &op->stage_in_capacity);
    if (bound == qnnp_status_success) bound = qnnp_bind_endpoint(output, op->output_span, &op->output_on_device, &op->d_stage_out, &op->stage_out_capacity);
    if (bound != qnnp_status_success) {
      qnnp_log_error("failed to bind the tensors: device staging for host memory could not be allocated, or a tensor "
          "lives on a different device than the operator");
      return bound;
    }
  }
  return qnnp_status_success;
}

/* ---- public entry points: run the implementation inside the right device context ------------------
 * create: the calling thread's selected device (qnnp_gfx950_set_device, default = the primary one) becomes the
 * operator's device; setup: the operator's device. The previous HIP device of the thread is restored on return. */

enum qnnp_status qnnp_create_global_average_pooling_nwc_q8(
    size_t channels,
    uint8_t input_zero_point,
    float input_scale,
    uint8_t output_zero_point,
    float output_scale,
    uint8_t output_min,
    uint8_t output_max,
    uint32_t flags,
    qnnp_operator_t* global_average_pooling_out)
{
  if (!qnnp_state.initialized) {
    return qnnp_create_global_average_pooling_nwc_q8_impl(channels, input_zero_point, input_scale, output_zero_point, output_scale, output_min, output_max, flags, global_average_pooling_out);   /* logs and answers qnnp_status_uninitialized */
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
  const enum qnnp_status status = qnnp_create_global_average_pooling_nwc_q8_impl(channels, input_zero_point, input_scale, output_zero_point, output_scale, output_min, output_max, flags, global_average_pooling_out);
  qnnp_hip_leave(token);
  return status;
}

enum qnnp_status qnnp_setup_global_average_pooling_nwc_q8(
    qnnp_operator_t op,
    size_t batch_size,
    size_t width,
    const uint8_t* input,
    size_t input_stride,
    uint8_t* output,
    size_t output_stride)
{
  if (!qnnp_state.initialized || op == NULL) {
    return qnnp_setup_global_average_pooling_nwc_q8_impl(op, batch_size, width, input, input_stride, output, output_stride);   /* answers qnnp_status_uninitialized / invalid_parameter */
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
  const enum qnnp_status status = qnnp_setup_global_average_pooling_nwc_q8_impl(op, batch_size, width, input, input_stride, output, output_stride);
  /* the implementation cleared setup_valid where it began to change the operator: a failed setup leaves it
   * unrunnable instead of half updated (run answers invalid_parameter) */
  if (status == qnnp_status_success) {
    op->setup_valid = 1;
  }
  qnnp_hip_leave(token);
  return status;
}
