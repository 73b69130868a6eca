/*
 * residual.c -- qnnp_gfx950_attach_residual_add (qnnpack_gfx950.h): a convolution that also performs the quantized
 * add behind it. No reference counterpart: in the reference a residual connection is a convolution operator followed
 * by an add operator (src/add.c + q8vadd, the MobileNetV2 pattern of SURVEY.md section 8f row 4), i.e. one more pass
 * over the tensor -- HBM traffic that is the whole cost of the add on this chip. Here the add rides in the epilogue
 * of the convolution's kernel where that kernel carries it (the pointwise streaming kernels, q8pwconv.hip: every
 * MobileNetV2 project layer), and runs as an in-place launch of the add kernel behind the convolution otherwise;
 * either way the bytes are those of convolution -> qnnp_setup_add_nc_q8(a = residual, b = convolution output).
 *
 * The arithmetic is the add operator's own parameter block (add.c, reference requantization.h:327-360), copied at
 * attach time: the add operator may be deleted afterwards.
 */
#include <stddef.h>
#include <stdint.h>

#include <qnnpack.h>
#include <qnnpack_gfx950.h>

#include "hip/qnnp_hip.h"
#include "log.h"
#include "operator.h"
#include "state.h"

enum qnnp_status qnnp_gfx950_attach_residual_add(
    qnnp_operator_t convolution, qnnp_operator_t add, const uint8_t* residual, size_t residual_stride)
{
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_gfx950_attach_residual_add called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  if (convolution == NULL || add == NULL) {
    return qnnp_status_invalid_parameter;
  }
  const int is_conv = convolution->ukernel_type == qnnp_ukernel_type_conv ||
      convolution->ukernel_type == qnnp_ukernel_type_gemm || convolution->ukernel_type == qnnp_ukernel_type_dwconv;
  if (!is_conv || convolution->transposed || add->ukernel_type != qnnp_ukernel_type_add) {
    qnnp_log_error("failed to attach residual add: needs a convolution / fully connected operator and an add operator");
    return qnnp_status_invalid_parameter;
  }
  if (convolution->device != add->device) {
    qnnp_log_error("failed to attach residual add: the operators belong to different devices");
    return qnnp_status_invalid_parameter;
  }
  if (!convolution->setup_valid) {
    qnnp_log_error("failed to attach residual add: the convolution has no valid setup (attach after setup)");
    return qnnp_status_invalid_parameter;
  }
  const size_t channels = (size_t) convolution->groups * convolution->group_output_channels;
  if (add->channels != channels) {
    qnnp_log_error("failed to attach residual add: add operator of %zu channels on a convolution with %zu output channels",
        add->channels, channels);
    return qnnp_status_invalid_parameter;
  }
  if (convolution->batch_size == 0) {
    return qnnp_status_success;             /* nothing will run (reference operator-run.c:642-644) */
  }
  if (residual == NULL || residual_stride < channels) {
    qnnp_log_error("failed to attach residual add: residual stride %zu below %zu channels (or NULL residual)",
        residual_stride, channels);
    return qnnp_status_invalid_parameter;
  }
  if (residual_stride > UINT32_MAX) {
    return qnnp_status_unsupported_parameter;
  }
  const int token = qnnp_hip_enter(convolution->device);
  if (token < 0) {
    return qnnp_status_invalid_parameter;
  }
  enum qnnp_status status = qnnp_status_success;
  if (qnnp_hip_graph_capturing()) {
    status = qnnp_status_invalid_parameter;           /* only launches are recordable (convolution.c) */
  } else if (!convolution->input_on_device || !convolution->output_on_device ||
             qnnp_hip_is_device_pointer(residual) != 1) {
    /* the fused form exists for device-resident pipelines; host endpoints keep the two-operator form */
    qnnp_log_error("failed to attach residual add: input, output and residual must be memory of the operator's device");
    status = qnnp_status_unsupported_parameter;
  }
  if (status == qnnp_status_success) {
    /* The kernels read the residual through __restrict__ / streaming loads while they write the output: the two ranges
     * must not overlap -- not even exactly (the in-place form is the stand-alone add operator's business,
     * operator-run.c). Checked only here, where both are known to be memory of this device: `convolution->output` is
     * then the address the kernels write (a host-staged output was refused above with its own status). */
    const size_t pixels = convolution->batch_size * convolution->output_height * convolution->output_width;
    const uintptr_t r0 = (uintptr_t) residual;
    const uintptr_t r1 = r0 + (pixels - 1) * residual_stride + channels;
    const uintptr_t o0 = (uintptr_t) convolution->output;
    const uintptr_t o1 = o0 + (pixels - 1) * convolution->output_pixel_stride + channels;
    if (pixels != 0 && r0 < o1 && o0 < r1) {
      qnnp_log_error("failed to attach residual add: the residual tensor overlaps the convolution's output");
      status = qnnp_status_invalid_parameter;
    }
  }
  if (status == qnnp_status_success) {
    convolution->residual_params = add->add_params;
    convolution->residual_pixel_stride = residual_stride;
    convolution->residual = residual;
    convolution->residual_folded = 0;
  }
  qnnp_hip_leave(token);
  return status;
}

int qnnp_gfx950_operator_residual_folded(qnnp_operator_t op)
{
  if (op == NULL || op->residual == NULL) {
    return -1;
  }
  return (int) op->residual_folded;
}
