/*
 * operator-run.c -- qnnp_run_operator for the gfx950 build.
 *
 * Replaces the hot cases of reference src/operator-run.c:639-844. The reference
 * switches on op->ukernel_type, builds a stack context and fans MRxNR microkernel
 * calls out over a pthreadpool (:675 dwconv, :797 gemm, :837 conv). Here the same
 * switch builds a POD argument block and makes ONE call through the C-ABI HIP
 * shim per operator (hip/qnnp_hip.h): a whole-operator kernel launch on the
 * library stream. `threadpool` is ignored.
 *
 * Like the reference, nothing is allocated and nothing is logged on this path
 * (device pointers). Host pointers given to setup are staged through device
 * scratch sized at setup.
 *
 * Threading: as in the reference (run contexts are stack-local, src/operator-run.c:783-795), distinct operators
 * may be run from different threads at the same time. Nothing on this path is process-global mutable state: the
 * operator names its device, the launch stream / asynchrony flag belong to that device's context, and a hipGraph
 * capture belongs to the capturing thread (hip/runtime.hip).
 */
#include <stddef.h>
#include <stdint.h>

#include <qnnpack.h>
#include <qnnpack_gfx950.h>

#include "hip/qnnp_hip.h"
#include "operator.h"
#include "state.h"

#define QNNP_TIMING_SAMPLES 5

static enum qnnp_status status_from_hip(int rc)
{
  switch (rc) {
    case QNNP_HIP_OK: return qnnp_status_success;
    case QNNP_HIP_ENOMEM: return qnnp_status_out_of_memory;
    case QNNP_HIP_EINVAL: return qnnp_status_unsupported_parameter;
    default: return qnnp_status_unsupported_hardware;
  }
}

enum qnnp_status qnnp_bind_endpoint(const void* ptr, size_t span, int* on_device, void** stage, size_t* capacity)
{
  const int where = qnnp_hip_is_device_pointer(ptr);
  if (where < 0) {
    return qnnp_status_invalid_parameter;   /* memory of a GPU other than the operator's */
  }
  *on_device = where;
  if (where) return qnnp_status_success;
  if (*capacity < span) {
    qnnp_hip_free(*stage);
    *capacity = 0;
    *stage = qnnp_hip_alloc(span);
    if (*stage == NULL) return qnnp_status_out_of_memory;
    *capacity = span;
  }
  return qnnp_status_success;
}

/* Enqueue the operator's kernel on the library stream of the active context, reading `input` and
 * writing `output` (both device pointers). */

static int launch_kernel(struct qnnp_operator* op, const void* input, const void* input2, void* output)
{
  switch (op->ukernel_type) {
    case qnnp_ukernel_type_dwconv:
    {
      /* reference operator-run.c:647-710 (context) + :238-284 (trampolines) */
      const struct qnnp_hip_dwconv_args args = {
        .input = (const uint8_t*) input,
        .output = (uint8_t*) output,
        .streaming_mode = op->streaming_mode,
        .wadj = (const int16_t*) op->d_weights,
        .bias1 = op->d_bias,
        .dwm_x = (const int8_t*) op->d_dwm_x,
        .dwm_bias = op->d_dwm_bias,
        .dwm_parts = op->dwm_parts,
        .w_range = op->dw_wrange,
        .dot4 = (const uint32_t*) op->d_dw_dot4,
        .c_pad32 = op->c_pad32,
        .batch = (uint32_t) op->batch_size,
        .input_height = (uint32_t) op->input_height,
        .input_width = (uint32_t) op->input_width,
        .output_height = (uint32_t) op->output_height,
        .output_width = (uint32_t) op->output_width,
        .channels = op->groups,
        .c_pad = op->c_pad,
        .kernel_height = op->kernel_height,
        .kernel_width = op->kernel_width,
        .stride_height = op->stride_height,
        .stride_width = op->stride_width,
        .dilation_height = op->dilation_height,
        .dilation_width = op->dilation_width,
        .pad_top = op->input_padding_top,
        .pad_left = op->input_padding_left,
        .input_stride = (uint32_t) op->input_pixel_stride,
        .output_stride = (uint32_t) op->output_pixel_stride,
        .input_zero_point = op->input_zero_point,
        .rq = op->requant,
        .variant = op->variant,
        .plan = &op->dw_plan,
      };
      return qnnp_hip_dwconv_run(&args, &op->kernel_name);
    }
    case qnnp_ukernel_type_gemm:
    case qnnp_ukernel_type_conv:
    {
      /* reference operator-run.c:770-804 (gemm) and :805-844 (conv) */
      const int is_conv = op->ukernel_type == qnnp_ukernel_type_conv;
      const uint32_t output_size = (uint32_t) (op->output_height * op->output_width);
      const uint32_t taps = is_conv ? op->kernel_height * op->kernel_width : 1;
      if (op->transposed && op->deconv_d2s) {
        /* kernel == stride: one pointwise GEMM over the input pixels, stored depth-to-space (deconvolution.c) */
        const uint32_t phases = op->stride_height * op->stride_width;
        const struct qnnp_hip_igemm_args dargs = {
          .input = (const uint8_t*) input,
          .output = (uint8_t*) output,
          .packed_w = (const int8_t*) op->d_weights,
          .bias2 = op->d_bias,
          .bias2_pair = 1,
          .offsets = NULL,
          .rows = (uint32_t) (op->batch_size * op->input_height * op->input_width),
          .rows_per_image = (uint32_t) (op->input_height * op->input_width),
          .image_stride = 0,
          .groups = 1,
          .n = (uint32_t) op->group_output_channels,
          .n_pad = op->n_pad * phases,
          .kc = (uint32_t) op->group_input_channels,
          .kc_slot = op->kc_slot,
          .input_bytes = op->input_span,
          .ks = 1,
          .k_total = op->kc_slot,
          .k_pad = op->k_pad,
          .input_stride = (uint32_t) op->input_pixel_stride,
          .output_stride = (uint32_t) op->output_pixel_stride,
          .row_coeff = 128 - (int32_t) op->kernel_zero_point,
          .input_zero_point = op->input_zero_point,
          .rq = op->requant,
          .variant = 5,
          .d2s_stride_h = op->stride_height,
          .d2s_stride_w = op->stride_width,
          .d2s_input_h = (uint32_t) op->input_height,
          .d2s_input_w = (uint32_t) op->input_width,
          .streaming_mode = op->streaming_mode,
        };
        const int rc_d2s = qnnp_hip_igemm_run(&dargs, &op->kernel_name);
        if (rc_d2s != QNNP_HIP_EINVAL) {
          return rc_d2s;
        }
        /* the streaming kernel cannot take these tensors (alignment): the phase GEMMs below can */
      }
      if (op->transposed && op->deconv_phases == 4 && op->deconv_stream != 1 && op->groups == 1 &&
          op->stride_height == 2 && op->stride_width == 2 && op->dilation_height == 1 && op->dilation_width == 1 &&
          op->kernel_height == op->kernel_width && (op->kernel_height == 3 || op->kernel_height == 4)) {
        /* stride 2, 3x3 / 4x4: one streaming kernel over the input pixels, all four phases (q8deconv.hip) */
        struct qnnp_hip_deconv_s2_args sargs = {
          .input = (const uint8_t*) input,
          .output = (uint8_t*) output,
          .batch = (uint32_t) op->batch_size,
          .input_height = (uint32_t) op->input_height,
          .input_width = (uint32_t) op->input_width,
          .output_height = (uint32_t) op->output_height,
          .output_width = (uint32_t) op->output_width,
          .kernel_height = op->kernel_height,
          .kernel_width = op->kernel_width,
          .pad_top = op->input_padding_top,
          .pad_left = op->input_padding_left,
          .channels = (uint32_t) op->group_input_channels,
          .n = (uint32_t) op->group_output_channels,
          .n_pad = op->n_pad,
          .input_stride = (uint32_t) op->input_pixel_stride,
          .output_stride = (uint32_t) op->output_pixel_stride,
          .row_coeff = 128 - (int32_t) op->kernel_zero_point,
          .input_zero_point = op->input_zero_point,
          .rq = op->requant,
          .bias2_pair = 1,
        };
        for (int ph = 0; ph < 4; ph++) {
          sargs.packed_w[ph] = (const int8_t*) op->phase[ph].d_weights;
          sargs.bias2[ph] = op->phase[ph].d_bias;
          sargs.k_pad[ph] = op->phase[ph].k_pad;
        }
        const int rc_s2 = qnnp_hip_deconv_s2_run(&sargs, &op->kernel_name);
        if (rc_s2 != QNNP_HIP_EINVAL || op->deconv_stream == 2) {
          return rc_s2;
        }
        /* outside the streaming kernel's range (channel multiple, LDS, alignment): the phase GEMMs below */
      } else if (op->transposed && op->deconv_stream == 2) {
        return QNNP_HIP_EINVAL;
      }
      if (op->transposed && op->deconv_phases != 0) {
        /* strided deconvolution: one dense implicit GEMM per output phase, all in one launch (deconvolution.c) */
        if (op->phase_table_entries == 0) return QNNP_HIP_OK;
        const struct qnnp_deconv_phase* ph0 = &op->phase[0];
        const struct qnnp_hip_igemm_args pargs = {
          .input = (const uint8_t*) input,
          .output = (uint8_t*) output,
          .packed_w = (const int8_t*) ph0->d_weights,         /* per-phase fields come from the table */
          .bias2 = ph0->d_bias,
          .offsets = (const int32_t*) op->d_phase_table,        /* non-NULL marks the convolution form */
          .rows = (uint32_t) (op->batch_size * op->phase_max_rows),
          .rows_per_image = (uint32_t) op->phase_max_rows,
          .image_stride = (uint64_t) op->input_height * op->input_width * op->input_pixel_stride,
          .groups = op->groups,
          .n = (uint32_t) op->group_output_channels,
          .n_pad = op->n_pad,
          .kc = (uint32_t) op->group_input_channels,
          .kc_slot = op->kc_slot,
          .input_bytes = op->input_span,
          .ks = 1,
          .k_total = op->kc_slot,
          .k_pad = op->phase_max_k_pad,
          .input_stride = (uint32_t) op->input_pixel_stride,
          .output_stride = (uint32_t) op->output_pixel_stride,
          .row_coeff = 128 - (int32_t) op->kernel_zero_point,
          .input_zero_point = op->input_zero_point,
          .rq = op->requant,
          .variant = 1,
          .out_image_rows = output_size,
          .phases = (const struct qnnp_hip_igemm_phase*) op->d_phase_table,
          .nphases = op->phase_table_entries,
          .streaming_mode = op->streaming_mode,
        };
        return qnnp_hip_igemm_run(&pargs, &op->kernel_name);
      }
      const struct qnnp_hip_igemm_args args = {
        .input = (const uint8_t*) input,
        .output = (uint8_t*) output,
        .packed_w = (const int8_t*) op->d_weights,
        .packed_w_rows16 = is_conv ? (const int8_t*) op->d_weights_rows16 : NULL,
        .bias2_rows = is_conv ? op->d_bias_rows : NULL,
        .bias2 = op->d_bias,
        .bias2_pair = 1,
        .offsets = is_conv ? op->d_offsets : NULL,
        .rows = (uint32_t) op->batch_size * output_size,
        .rows_per_image = output_size,
        .image_stride = (uint64_t) op->input_height * op->input_width * op->input_pixel_stride,
        .groups = op->groups,
        .n = (uint32_t) op->group_output_channels,
        .n_pad = op->n_pad,
        .kc = (uint32_t) op->group_input_channels,
        .kc_slot = op->kc_slot,
        .input_bytes = op->input_span,
        .ks = taps,
        .k_total = taps * op->kc_slot,
        .k_pad = op->k_pad,
        .input_stride = (uint32_t) op->input_pixel_stride,
        .output_stride = (uint32_t) op->output_pixel_stride,
        .row_coeff = 128 - (int32_t) op->kernel_zero_point,
        .input_zero_point = op->input_zero_point,
        .rq = op->requant,
        .variant = op->variant,
        .input_height = (uint32_t) op->input_height,
        .input_width = (uint32_t) op->input_width,
        .output_height = (uint32_t) op->output_height,
        .output_width = (uint32_t) op->output_width,
        .kernel_height = op->kernel_height,
        .kernel_width = op->kernel_width,
        .stride_height = op->stride_height,
        .stride_width = op->stride_width,
        .dilation_height = op->dilation_height,
        .dilation_width = op->dilation_width,
        .pad_top = op->input_padding_top,
        .pad_left = op->input_padding_left,
        .residual = (const uint8_t*) op->residual,
        .residual_stride = (uint32_t) op->residual_pixel_stride,
        .residual_add = &op->residual_params,
        .residual_folded = &op->residual_folded,
        /* zero-point-centred image (fully-connected.c): its own, or the standard one when that is centred already */
        .packed_w_centred = (const int8_t*) (op->d_weights_centred != NULL ? op->d_weights_centred : op->d_weights),
        .bias2_centred = op->d_bias_centred != NULL ? op->d_bias_centred : op->d_bias,
        .centre_flip = op->centre_flip,
        .streaming_mode = op->streaming_mode,
      };
      /* grouped 1x1 with a dense image (convolution.c): many rows -> the dense GEMM; "gemm_kernel" 31 forces it, any other
       * forced kernel keeps the grouped image */
      op->ran_dense = 0;
      if (op->variant == 31 && op->d_weights_dense == NULL) return QNNP_HIP_EINVAL;
      if (op->d_weights_dense != NULL && op->residual == NULL &&
          (op->variant == 31 || (op->variant == 0 && (uint64_t) op->batch_size * output_size >= 65536u))) {
        struct qnnp_hip_igemm_args dargs = args;
        dargs.packed_w = (const int8_t*) op->d_weights_dense;
        dargs.bias2 = op->d_bias_dense;
        dargs.groups = 1;
        dargs.n = (uint32_t) (op->groups * op->group_output_channels);
        dargs.n_pad = op->dense_n_pad;
        dargs.kc = (uint32_t) (op->groups * op->group_input_channels);
        dargs.kc_slot = dargs.kc;
        dargs.k_total = dargs.kc;
        dargs.k_pad = op->dense_k_pad;
        dargs.variant = 0;
        dargs.packed_w_centred = dargs.packed_w;
        dargs.bias2_centred = dargs.bias2;
        dargs.centre_flip = 0;
        op->ran_dense = 1;
        return qnnp_hip_igemm_run(&dargs, &op->kernel_name);
      }
      if (op->variant == 31) return QNNP_HIP_EINVAL;
      return qnnp_hip_igemm_run(&args, &op->kernel_name);
    }
    case qnnp_ukernel_type_fused_block:
      return qnnp_fused_block_launch(op, input, output);
    case qnnp_ukernel_type_add:
    {
      /* reference operator-run.c, case qnnp_ukernel_type_add (q8vadd over rows of `channels` bytes) */
      const struct qnnp_hip_vadd_args args = {
        .streaming_mode = op->streaming_mode,
        .a = (const uint8_t*) input,
        .b = (const uint8_t*) input2,
        .sum = (uint8_t*) output,
        .rows = op->batch_size,
        .channels = (uint32_t) op->channels,
        .a_stride = op->input_pixel_stride,
        .b_stride = op->input2_pixel_stride,
        .sum_stride = op->output_pixel_stride,
        .params = op->add_params,
      };
      return qnnp_hip_vadd_run(&args, &op->kernel_name);
    }
    case qnnp_ukernel_type_global_average_pooling:
    {
      /* reference operator-run.c:981-1016 */
      const struct qnnp_hip_gavgpool_args args = {
        .input = (const uint8_t*) input,
        .output = (uint8_t*) output,
        .batch = op->batch_size,
        .width = op->input_width,
        .channels = (uint32_t) op->channels,
        .input_stride = op->input_pixel_stride,
        .output_stride = op->output_pixel_stride,
        .params = op->avgpool_params,
      };
      return qnnp_hip_gavgpool_run(&args, &op->kernel_name);
    }
    default:
      return QNNP_HIP_EINVAL;
  }
}

static int launch(struct qnnp_operator* op, const void* input, const void* input2, void* output)
{
  op->residual_folded = 0;
  const int rc = launch_kernel(op, input, input2, output);
  if (rc != QNNP_HIP_OK || op->residual == NULL || op->residual_folded) {
    return rc;
  }
  /* attached residual add (residual.c) the convolution kernel does not carry: the add kernel, in place on the output */
  const size_t out_channels = (size_t) op->groups * op->group_output_channels;
  const struct qnnp_hip_vadd_args args = {
    .streaming_mode = op->streaming_mode,
    .a = (const uint8_t*) op->residual,
    .b = (const uint8_t*) output,
    .sum = (uint8_t*) output,
    .rows = op->batch_size * op->output_height * op->output_width,
    .channels = (uint32_t) out_channels,
    .a_stride = op->residual_pixel_stride,
    .b_stride = op->output_pixel_stride,
    .sum_stride = op->output_pixel_stride,
    .params = op->residual_params,
  };
  const char* add_kernel = NULL;
  return qnnp_hip_vadd_run(&args, &add_kernel);
}

static enum qnnp_status run_operator(qnnp_operator_t op)
{
  if (!op->setup_valid) {
    return qnnp_status_invalid_parameter;  /* run before setup, or after a setup that failed */
  }
  /* reference operator-run.c:642-644: nothing to do for an empty batch */
  if (op->batch_size == 0) {
    return qnnp_status_success;
  }
  if (op->input == NULL || op->output == NULL) {
    return qnnp_status_invalid_parameter;
  }

  const void* input = op->input;
  const void* input2 = op->input2;
  void* output = op->output;
  const int staged = !op->input_on_device || !op->output_on_device || (op->input2 != NULL && !op->input2_on_device);
  const int capturing = qnnp_hip_graph_capturing();
  if (capturing && staged) {
    return qnnp_status_invalid_parameter;  /* a graph can only hold device-pointer launches */
  }

  if (!op->input_on_device) {
    if (qnnp_hip_h2d(op->d_stage_in, op->input, op->input_span, 1) != QNNP_HIP_OK) {
      return qnnp_status_invalid_parameter;
    }
    input = op->d_stage_in;
  }
  if (op->input2 != NULL && !op->input2_on_device) {
    if (qnnp_hip_h2d(op->d_stage_in2, op->input2, op->input2_span, 1) != QNNP_HIP_OK) {
      return qnnp_status_invalid_parameter;
    }
    input2 = op->d_stage_in2;
  }
  if (!op->output_on_device) {
    const size_t out_channels = op->channels != 0 ? op->channels : (size_t) op->groups * op->group_output_channels;
    if (op->output_pixel_stride != out_channels) {
      /* keep the caller's bytes between pixels intact across the round trip */
      if (qnnp_hip_h2d(op->d_stage_out, op->output, op->output_span, 1) != QNNP_HIP_OK) {
        return qnnp_status_invalid_parameter;
      }
    }
    output = op->d_stage_out;
  }

  const int rc = launch(op, input, input2, output);
  if (rc != QNNP_HIP_OK) {
    return status_from_hip(rc);
  }

  if (!op->output_on_device) {
    if (qnnp_hip_d2h(op->output, op->d_stage_out, op->output_span, 1) != QNNP_HIP_OK) {
      return qnnp_status_invalid_parameter;
    }
  }
  if (capturing) {
    return qnnp_status_success;            /* recorded into the graph; nothing has run yet */
  }
  if (staged || !qnnp_hip_get_async()) {
    /* reference semantics: outputs are complete when run returns */
    return status_from_hip(qnnp_hip_stream_sync());
  }
  return qnnp_status_success;
}

enum qnnp_status qnnp_run_operator(qnnp_operator_t op, pthreadpool_t threadpool)
{
  (void) threadpool;
  if (op == NULL) {
    return qnnp_status_invalid_parameter;
  }
  if (!qnnp_state.initialized) {
    return qnnp_status_uninitialized;
  }
  const int token = qnnp_hip_enter(op->device);
  if (token < 0) {
    return qnnp_status_invalid_parameter;
  }
  const enum qnnp_status status = run_operator(op);
  qnnp_hip_leave(token);
  return status;
}

/* ---- qnnpack_gfx950.h timing helpers ---------------------------------- */

static enum qnnp_status time_operator_rotating(
    qnnp_operator_t op, size_t nsets, const void* const* inputs, void* const* outputs,
    int warmup, int iters, float* avg_ms_out)
{
  if (!op->setup_valid) {
    return qnnp_status_invalid_parameter;
  }
  if (op->batch_size == 0) {
    *avg_ms_out = 0.0f;
    return qnnp_status_success;
  }
  for (size_t s = 0; s < nsets; s++) {
    if (qnnp_hip_is_device_pointer(inputs[s]) != 1 || qnnp_hip_is_device_pointer(outputs[s]) != 1) {
      return qnnp_status_invalid_parameter;
    }
  }
  if (op->input2 != NULL && !op->input2_on_device) {
    return qnnp_status_invalid_parameter;   /* add: the second operand is not rotated and must be device memory */
  }
  /* Preferred: record the `iters` launches into a hipGraph and time its replay -- one submission, so the
   * figure is kernel time, not the host's per-launch dispatch gap (5-8 us, as large as the small layers).
   * The replay is timed QNNP_TIMING_SAMPLES times, each sample its own event pair, and the MEDIAN sample is
   * reported (one sample is at the mercy of the clock state of the moment). Falls back to a plain launch loop
   * if the capture is refused. */
  if (qnnp_state.opt_timing_graph && !qnnp_hip_graph_capturing() && qnnp_hip_graph_begin() == QNNP_HIP_OK) {
    int rc = QNNP_HIP_OK;
    size_t gset = 0;
    for (int i = 0; i < iters && rc == QNNP_HIP_OK; i++) {
      rc = launch(op, inputs[gset], op->input2, outputs[gset]);
      gset = (gset + 1) % nsets;
    }
    void* graph = NULL;
    const int rc_end = qnnp_hip_graph_end(&graph);
    if (rc == QNNP_HIP_OK && rc_end == QNNP_HIP_OK) {
      float ms = 0.0f;
      const int reps = warmup > 0 ? 1 : 0;
      rc = qnnp_hip_graph_time_median(graph, reps, 1, QNNP_TIMING_SAMPLES, &ms);
      qnnp_hip_graph_destroy(graph);
      if (rc == QNNP_HIP_OK) {
        *avg_ms_out = ms / (float) iters;
        return qnnp_status_success;
      }
    } else if (rc_end == QNNP_HIP_OK) {
      qnnp_hip_graph_destroy(graph);
    }
  }
  void* timer = NULL;
  if (qnnp_hip_timer_create(&timer) != QNNP_HIP_OK) {
    return qnnp_status_out_of_memory;
  }
  enum qnnp_status status = qnnp_status_success;
  size_t set = 0;
  for (int i = 0; i < warmup && status == qnnp_status_success; i++) {
    status = status_from_hip(launch(op, inputs[set], op->input2, outputs[set]));
    set = (set + 1) % nsets;
  }
  if (status == qnnp_status_success) {
    qnnp_hip_timer_start(timer);
    for (int i = 0; i < iters && status == qnnp_status_success; i++) {
      status = status_from_hip(launch(op, inputs[set], op->input2, outputs[set]));
      set = (set + 1) % nsets;
    }
    float ms = 0.0f;
    const int rc = qnnp_hip_timer_stop_ms(timer, &ms);
    if (status == qnnp_status_success) status = status_from_hip(rc);
    *avg_ms_out = ms / (float) iters;
  }
  qnnp_hip_timer_destroy(timer);
  return status;
}

enum qnnp_status qnnp_gfx950_time_operator_rotating(
    qnnp_operator_t op, size_t nsets, const void* const* inputs, void* const* outputs,
    int warmup, int iters, float* avg_ms_out)
{
  if (op == NULL || avg_ms_out == NULL || iters <= 0 || nsets == 0 || inputs == NULL || outputs == NULL) {
    return qnnp_status_invalid_parameter;
  }
  if (!qnnp_state.initialized) {
    return qnnp_status_uninitialized;
  }
  const int token = qnnp_hip_enter(op->device);
  if (token < 0) {
    return qnnp_status_invalid_parameter;
  }
  const enum qnnp_status status = time_operator_rotating(op, nsets, inputs, outputs, warmup, iters, avg_ms_out);
  qnnp_hip_leave(token);
  return status;
}

enum qnnp_status qnnp_gfx950_time_operator(
    qnnp_operator_t op, int warmup, int iters, float* avg_ms_out)
{
  if (op == NULL) {
    return qnnp_status_invalid_parameter;
  }
  const void* in = op->input;
  void* out = op->output;
  return qnnp_gfx950_time_operator_rotating(op, 1, &in, &out, warmup, iters, avg_ms_out);
}

/* ---- qnnpack_gfx950.h graph capture ------------------------------------ */

enum qnnp_status qnnp_gfx950_graph_begin(void)
{
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  /* the calling thread records launches on its selected device until graph_end */
  const int token = qnnp_hip_enter(qnnp_hip_device());
  if (token < 0) return qnnp_status_unsupported_hardware;
  const enum qnnp_status status = status_from_hip(qnnp_hip_graph_begin());
  qnnp_hip_leave(token);
  return status;
}

enum qnnp_status qnnp_gfx950_graph_end(void** graph_out)
{
  if (graph_out == NULL) return qnnp_status_invalid_parameter;
  if (!qnnp_state.initialized) return qnnp_status_uninitialized;
  return status_from_hip(qnnp_hip_graph_end(graph_out));
}

enum qnnp_status qnnp_gfx950_graph_launch(void* graph)
{
  if (graph == NULL) return qnnp_status_invalid_parameter;
  const int rc = qnnp_hip_graph_launch(graph);
  if (rc != QNNP_HIP_OK) return status_from_hip(rc);
  const int token = qnnp_hip_enter(qnnp_hip_graph_device(graph));
  const int async = token >= 0 ? qnnp_hip_get_async() : 0;
  qnnp_hip_leave(token);
  return async ? qnnp_status_success : status_from_hip(qnnp_hip_graph_sync(graph));
}

enum qnnp_status qnnp_gfx950_graph_time(void* graph, int warmup, int iters, float* avg_ms_out)
{
  if (graph == NULL || avg_ms_out == NULL || iters <= 0) return qnnp_status_invalid_parameter;
  /* QNNP_TIMING_SAMPLES event-bracketed batches of `iters` replays each; the median batch / iters */
  return status_from_hip(qnnp_hip_graph_time_median(graph, warmup, iters, QNNP_TIMING_SAMPLES, avg_ms_out));
}

enum qnnp_status qnnp_gfx950_graph_synchronize(void* graph)
{
  if (graph == NULL) return qnnp_status_invalid_parameter;
  return status_from_hip(qnnp_hip_graph_sync(graph));
}

void qnnp_gfx950_graph_destroy(void* graph)
{
  qnnp_hip_graph_destroy(graph);
}
