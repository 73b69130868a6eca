/*
 * add.c -- qnnp_create_add_nc_q8 / qnnp_setup_add_nc_q8 for the gfx950 build (SURVEY.md section 8f, row 4:
 * the residual add between MobileNetV2's convolutions).
 *
 * Replaces reference src/add.c:20-116 (create) and :118-149 (setup): same validation order and status
 * codes; the quantization parameters are the scalar member of qnnp_compute_add_quantization_params
 * (reference src/qnnpack/requantization.h:327-360, :400-413), consumed on the device by q8pointwise.hip.
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

/* reference requantization.h:327-360 + scalar members :400-413 */
static struct qnnp_hip_add_params compute_add_params(
    uint8_t a_zero_point, uint8_t b_zero_point, uint8_t output_zero_point,
    float a_output_scale, float b_output_scale, uint8_t output_min, uint8_t output_max)
{
  const float max_output_scale = a_output_scale > b_output_scale ? a_output_scale : b_output_scale;
  uint32_t max_scale_bits;
  memcpy(&max_scale_bits, &max_output_scale, sizeof(max_scale_bits));
  const int32_t max_scale_exponent = (int32_t) (max_scale_bits >> 23) - 127;
  const uint32_t shift = (uint32_t) (21 - max_scale_exponent);          /* in [13, 31] */
  const uint32_t multiplier_bits = (uint32_t) (21 - max_scale_exponent + 127) << 23;
  float scale_multiplier;
  memcpy(&scale_multiplier, &multiplier_bits, sizeof(scale_multiplier));
  /* multipliers in [0, 2^22), the larger one in [2^21, 2^22); lrintf = round to nearest even */
  const uint32_t a_multiplier = (uint32_t) (int32_t) lrintf(a_output_scale * scale_multiplier);
  const uint32_t b_multiplier = (uint32_t) (int32_t) lrintf(b_output_scale * scale_multiplier);
  const uint32_t remainder_mask = (UINT32_C(1) << shift) - UINT32_C(1);

  struct qnnp_hip_add_params p;
  p.a_multiplier = a_multiplier;
  p.b_multiplier = b_multiplier;
  p.zero_point_product =
      (int32_t) -(a_multiplier * (uint32_t) a_zero_point + b_multiplier * (uint32_t) b_zero_point);
  p.shift = shift;
  p.remainder_mask = (int32_t) remainder_mask;
  p.remainder_threshold = (int32_t) (remainder_mask >> 1);
  p.y_zero_point = (int32_t) (uint32_t) output_zero_point;
  p.y_min = (int32_t) (uint32_t) output_min;
  p.y_max = (int32_t) (uint32_t) output_max;
  return p;
}

static enum qnnp_status qnnp_create_add_nc_q8_impl(
    size_t channels,
    uint8_t a_zero_point,
    float a_scale,
    uint8_t b_zero_point,
    float b_scale,
    uint8_t sum_zero_point,
    float sum_scale,
    uint8_t sum_min,
    uint8_t sum_max,
    uint32_t flags,
    qnnp_operator_t* add_out)
{
  (void) flags;
  /* reference add.c:36-39 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_create_add_nc_q8 called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  /* reference add.c:41-71 */
  if (channels == 0) {
    qnnp_log_error("cannot create add operator with %zu channels: number of channels may not be zero", channels);
    return qnnp_status_invalid_parameter;
  }
  if (!scale_is_valid(a_scale)) {
    qnnp_log_error("cannot create add operator with %.7g A scale: a scale has to be a finite number above zero", a_scale);
    return qnnp_status_invalid_parameter;
  }
  if (!scale_is_valid(b_scale)) {
    qnnp_log_error("cannot create add operator with %.7g B scale: a scale has to be a finite number above zero", b_scale);
    return qnnp_status_invalid_parameter;
  }
  if (!scale_is_valid(sum_scale)) {
    qnnp_log_error("cannot create add operator with %.7g output scale: a scale has to be a finite number above zero", sum_scale);
    return qnnp_status_invalid_parameter;
  }
  if (sum_min >= sum_max) {
    qnnp_log_error("cannot create add operator with [%" PRIu8 ", %" PRIu8 "] output range: range min must be below range max",
        sum_min, sum_max);
    return qnnp_status_invalid_parameter;
  }
  if (channels > (size_t) UINT32_MAX / 4) {
    qnnp_log_error("cannot create add operator: %zu channels exceed the device kernels' index range", channels);
    return qnnp_status_unsupported_parameter;
  }
  /* reference add.c:73-89 */
  const float a_output_scale = a_scale / sum_scale;
  if (a_output_scale < 0x1.0p-14f || a_output_scale >= 0x1.0p+8f) {
    qnnp_log_error("cannot create add operator with %.7g A-to-output scale ratio: scale ratio must be in [2**-14, 2**8) range",
        a_output_scale);
    return qnnp_status_unsupported_parameter;
  }
  const float b_output_scale = b_scale / sum_scale;
  if (b_output_scale < 0x1.0p-14f || b_output_scale >= 0x1.0p+8f) {
    qnnp_log_error("cannot create add operator with %.7g B-to-output scale ratio: scale ratio must be in [2**-14, 2**8) range",
        b_output_scale);
    return qnnp_status_unsupported_parameter;
  }

  qnnp_operator_t op = calloc(1, sizeof(struct qnnp_operator));
  if (op != NULL) op->device = qnnp_hip_device();   /* the context this create runs in (entry point below) */
  if (op == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for qnnp_operator structure", sizeof(struct qnnp_operator));
    return qnnp_status_out_of_memory;
  }
  op->channels = channels;
  op->add_params = compute_add_params(a_zero_point, b_zero_point, sum_zero_point,
      a_output_scale, b_output_scale, sum_min, sum_max);
  op->ukernel_type = qnnp_ukernel_type_add;
  *add_out = op;
  return qnnp_status_success;
}

static enum qnnp_status qnnp_setup_add_nc_q8_impl(
    qnnp_operator_t op,
    size_t batch_size,
    const uint8_t* a,
    size_t a_stride,
    const uint8_t* b,
    size_t b_stride,
    uint8_t* sum,
    size_t sum_stride)
{
  /* reference add.c:128-131 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_setup_add_nc_q8 called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  if (op == NULL || op->ukernel_type != qnnp_ukernel_type_add) {
    return qnnp_status_invalid_parameter;
  }
  /* reference add.c:133-136 */
  if (batch_size == 0) {
    op->batch_size = 0;
    return qnnp_status_success;
  }
  const size_t channels = op->channels;
  if (a == NULL || b == NULL || sum == NULL || a_stride < channels || b_stride < channels || sum_stride < channels) {
    qnnp_log_error("cannot set up add operator: NULL tensor or stride smaller than the channel count");
    return qnnp_status_invalid_parameter;
  }

  /* reference add.c:138-146 */
  op->setup_valid = 0;   /* until every check, allocation and upload below has succeeded */
  op->batch_size = batch_size;
  op->input = a;
  op->input_pixel_stride = a_stride;
  op->input2 = b;
  op->input2_pixel_stride = b_stride;
  op->output = sum;
  op->output_pixel_stride = sum_stride;

  op->input_span = (batch_size - 1) * a_stride + channels;
  op->input2_span = (batch_size - 1) * b_stride + channels;
  op->output_span = (batch_size - 1) * sum_stride + channels;
  {
    enum qnnp_status bound = qnnp_status_success;
    if (bound == qnnp_status_success) bound = qnnp_bind_endpoint(a, op->input_span, &op->input_on_device, &op->d_stage_in, &op->stage_in_capacity);
    if (bound == qnnp_status_success) bound = qnnp_bind_endpoint(b, op->input2_span, &op->input2_on_device, &op->d_stage_in2, &op->stage_in2_capacity);
    if (bound == qnnp_status_success) bound = qnnp_bind_endpoint(sum, op->output_span, &op->output_on_device, &op->d_stage_out, &op->stage_out_capacity);
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

enum qnnp_status qnnp_create_add_nc_q8(
    size_t channels,
    uint8_t a_zero_point,
    float a_scale,
    uint8_t b_zero_point,
    float b_scale,
    uint8_t sum_zero_point,
    float sum_scale,
    uint8_t sum_min,
    uint8_t sum_max,
    uint32_t flags,
    qnnp_operator_t* add_out)
{
  if (!qnnp_state.initialized) {
    return qnnp_create_add_nc_q8_impl(channels, a_zero_point, a_scale, b_zero_point, b_scale, sum_zero_point, sum_scale, sum_min, sum_max, flags, add_out);   /* logs and answers qnnp_status_uninitialized */
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
  const enum qnnp_status status = qnnp_create_add_nc_q8_impl(channels, a_zero_point, a_scale, b_zero_point, b_scale, sum_zero_point, sum_scale, sum_min, sum_max, flags, add_out);
  qnnp_hip_leave(token);
  return status;
}

enum qnnp_status qnnp_setup_add_nc_q8(
    qnnp_operator_t op,
    size_t batch_size,
    const uint8_t* a,
    size_t a_stride,
    const uint8_t* b,
    size_t b_stride,
    uint8_t* sum,
    size_t sum_stride)
{
  if (!qnnp_state.initialized || op == NULL) {
    return qnnp_setup_add_nc_q8_impl(op, batch_size, a, a_stride, b, b_stride, sum, sum_stride);   /* answers qnnp_status_uninitialized / invalid_parameter */
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
  const enum qnnp_status status = qnnp_setup_add_nc_q8_impl(op, batch_size, a, a_stride, b, b_stride, sum, sum_stride);
  /* the implementation cleared setup_valid where it began to change the operator: a failed setup leaves it
   * unrunnable instead of half updated (run answers invalid_parameter) */
  if (status == qnnp_status_success) {
    op->setup_valid = 1;
  }
  qnnp_hip_leave(token);
  return status;
}
