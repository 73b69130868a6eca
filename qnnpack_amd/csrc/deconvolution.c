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
 * STRIDED deconvolutions are split by output phase: all output pixels with the same
 * ((oy + pad_top) % stride_h, (ox + pad_left) % stride_w) see the same taps -- those with
 * ky*dil_h = phase_y (mod stride_h), likewise in x -- so each of the stride_h*stride_w phases is a dense
 * implicit GEMM over its own sub-kernel, offset table and output-pixel list (the kernel scatters GEMM rows
 * through that list). One table over all taps would spend (stride_h*stride_w - 1)/(stride_h*stride_w) of the
 * MFMA work and of the operand gathers on taps that are padding by construction (3/4 at stride 2).
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
#include "bias-pair.h"
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

static enum qnnp_status qnnp_create_deconvolution2d_nhwc_q8_impl(
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
    qnnp_log_error("qnnp_create_deconvolution2d_nhwc_q8 called before qnnp_initialize succeeded");
    goto error;
  }

  /* reference deconvolution.c:74-116 */
  status = qnnp_status_invalid_parameter;
  if (kernel_width == 0 || kernel_height == 0) {
    qnnp_log_error("cannot create deconvolution with %" PRIu32 "x%" PRIu32 " kernel: kernel dimensions may not be zero",
        kernel_width, kernel_height);
    goto error;
  }
  if (stride_width == 0 || stride_height == 0) {
    qnnp_log_error("cannot create deconvolution with %" PRIu32 "x%" PRIu32 " stride: stride dimensions may not be zero",
        stride_width, stride_height);
    goto error;
  }
  if (dilation_width == 0 || dilation_height == 0) {
    qnnp_log_error("cannot create deconvolution with %" PRIu32 "x%" PRIu32 " dilation: dilation dimensions may not be zero",
        dilation_width, dilation_height);
    goto error;
  }
  if (!scale_is_valid(input_scale)) {
    qnnp_log_error("cannot create deconvolution with %.7g input scale: a scale has to be a finite number above zero", input_scale);
    goto error;
  }
  if (!scale_is_valid(kernel_scale)) {
    qnnp_log_error("cannot create deconvolution with %.7g kernel scale: a scale has to be a finite number above zero", kernel_scale);
    goto error;
  }
  if (!scale_is_valid(output_scale)) {
    qnnp_log_error("cannot create deconvolution with %.7g output scale: a scale has to be a finite number above zero", output_scale);
    goto error;
  }
  if (groups == 0 || group_input_channels == 0 || group_output_channels == 0 || kernel == NULL || bias == NULL) {
    qnnp_log_error("cannot create deconvolution: groups, channel counts, kernel and bias may not be zero");
    goto error;
  }

  /* reference deconvolution.c:118-128 */
  status = qnnp_status_unsupported_parameter;
  const float deconvolution_scale = input_scale * kernel_scale / output_scale;
  if (deconvolution_scale >= 1.0f) {
    qnnp_log_error(
        "cannot create deconvolution with %.7g input scale, %.7g kernel scale, and %.7g output scale: "
        "deconvolution scale %.7g is greater or equal to 1.0",
        input_scale, kernel_scale, output_scale, deconvolution_scale);
    goto error;
  }
  if (!(deconvolution_scale >= 0x1.0p-32f)) {
    qnnp_log_error("cannot create deconvolution: deconvolution scale %.7g is below 2**-32", deconvolution_scale);
    goto error;
  }
  const size_t kernel_size = (size_t) kernel_height * kernel_width;
  if (kernel_size * group_input_channels > (size_t) UINT32_MAX / 4 ||
      (size_t) groups * group_output_channels > (size_t) UINT32_MAX / 4) {
    qnnp_log_error("cannot create deconvolution: channel / kernel extents exceed the 32-bit index range of the device kernels");
    goto error;
  }

  status = qnnp_status_out_of_memory;
  op = calloc(1, sizeof(struct qnnp_operator));
  if (op != NULL) op->device = qnnp_hip_device();   /* the context this create runs in (entry point below) */
  if (op == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for qnnp_operator structure", sizeof(struct qnnp_operator));
    goto error;
  }

  /* [g][ic][tap][oc] -> [g][oc][tap][ic] */
  const size_t gic = group_input_channels, goc = group_output_channels;
  const size_t group_weights = goc * kernel_size * gic;
  conv_order = (uint8_t*) malloc(group_weights * groups);
  if (conv_order == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for the transposed kernel", group_weights * groups);
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
  const uint32_t n_pad = qnnp_round_up_u32((uint32_t) goc, 32);
  const uint32_t phases = stride_height * stride_width;
  if (phases > 1 && phases <= QNNP_MAX_DECONV_PHASES && kernel_size <= 64) {
    /* one packed sub-kernel per output phase */
    const size_t b_bytes = sizeof(int32_t) * (size_t) groups * n_pad;
    uint8_t* sub = (uint8_t*) malloc(group_weights * groups + gic * goc * groups);
    host_bias = malloc(b_bytes);
    if (sub == NULL || host_bias == NULL) {
      free(sub);
      qnnp_log_error("out of host memory: %zu bytes for the phase kernels", group_weights * groups);
      goto error;
    }
    for (uint32_t py = 0; py < stride_height; py++) {
      for (uint32_t px = 0; px < stride_width; px++) {
        struct qnnp_deconv_phase* ph = &op->phase[py * stride_width + px];
        uint32_t taps = 0;
        for (uint32_t ky = 0; ky < kernel_height; ky++) {
          if ((ky * dilation_height) % stride_height != py) continue;
          for (uint32_t kx = 0; kx < kernel_width; kx++) {
            if ((kx * dilation_width) % stride_width != px) continue;
            ph->tap_ky[taps] = (uint8_t) ky;
            ph->tap_kx[taps] = (uint8_t) kx;
            taps++;
          }
        }
        const int empty = taps == 0;
        if (empty) {
          /* no tap ever reaches this phase: one tap of weight == kernel zero point, always padding (255 = no tap) */
          ph->tap_ky[0] = ph->tap_kx[0] = 255;
          taps = 1;
        }
        ph->taps = taps;
        for (size_t g = 0; g < groups; g++) {
          for (size_t oc = 0; oc < goc; oc++) {
            for (uint32_t t = 0; t < taps; t++) {
              uint8_t* dst = sub + ((g * goc + oc) * taps + t) * gic;
              if (empty) {
                memset(dst, kernel_zero_point, gic);
              } else {
                const size_t tap = (size_t) ph->tap_ky[t] * kernel_width + ph->tap_kx[t];
                memcpy(dst, conv_order + g * group_weights + (oc * kernel_size + tap) * gic, gic);
              }
            }
          }
        }
        ph->k_pad = qnnp_round_up_u32(taps * kc_slot, 64);
        const size_t w_bytes = qnnp_igemm_packed_weights_size(groups, n_pad, ph->k_pad);
        void* packed = malloc(w_bytes);
        int ok = packed != NULL;
        if (ok) {
          qnnp_pack_igemm_w_slots(groups, (uint32_t) goc, taps, (uint32_t) gic, kc_slot, n_pad, ph->k_pad,
              input_zero_point, kernel_zero_point, sub, bias, (int8_t*) packed, host_bias);
          ph->d_weights = qnnp_hip_alloc(w_bytes);
          ph->d_bias = qnnp_upload_bias_pair((const int32_t*) host_bias, (size_t) groups * n_pad);   /* bias-pair.h */
          ok = ph->d_weights != NULL && ph->d_bias != NULL &&
              qnnp_hip_h2d(ph->d_weights, packed, w_bytes, 0) == QNNP_HIP_OK;
        }
        free(packed);
        op->deconv_phases = py * stride_width + px + 1;   /* so that delete frees what exists so far */
        if (!ok) {
          free(sub);
          qnnp_log_error("device allocation or upload failed: %zu bytes of packed phase weights on the device", w_bytes + b_bytes);
          goto error;
        }
      }
    }
    free(sub);
    op->n_pad = n_pad;
    op->kc_slot = kc_slot;

    /* Kernel == stride (the usual 2x upsampling): every output pixel has exactly one tap and every input pixel
     * feeds stride_h*stride_w output pixels, so the whole operator is ONE pointwise GEMM over the input pixels with
     * phases * n_pad columns (phase-major) whose 32-channel blocks are stored depth-to-space (q8pwconv.hip).
     * Packed beside the phase kernels; the run falls back to those if the streaming kernel cannot take the tensors. */
    const bool any_padding =
        (input_padding_top | input_padding_right | input_padding_bottom | input_padding_left) != 0;
    const uint32_t d2s_cols = phases * n_pad;
    const uint32_t d2s_k_pad = qnnp_round_up_u32((uint32_t) gic, 64);
    if (groups == 1 && kernel_height == stride_height && kernel_width == stride_width &&
        dilation_height == 1 && dilation_width == 1 && !any_padding &&
        adjustment_height == 0 && adjustment_width == 0 && gic <= 256 && gic % 16 == 0 &&
        (size_t) d2s_cols * ((gic + 31) / 32 * 32) + (size_t) d2s_cols * 4 + 1024 <= 64 * 1024) {
      uint8_t* mat = (uint8_t*) malloc((size_t) d2s_cols * gic);
      int32_t* cols_bias = (int32_t*) calloc(d2s_cols, sizeof(int32_t));
      const size_t dw_bytes = qnnp_igemm_packed_weights_size(1, d2s_cols, d2s_k_pad);
      const size_t db_bytes = sizeof(int32_t) * d2s_cols;
      void* packed = malloc(dw_bytes);
      int32_t* packed_bias = (int32_t*) malloc(db_bytes);
      int ok = mat != NULL && cols_bias != NULL && packed != NULL && packed_bias != NULL;
      if (ok) {
        memset(mat, kernel_zero_point, (size_t) d2s_cols * gic);     /* padding columns: w - kzp == 0 */
        for (uint32_t ph = 0; ph < phases; ph++) {
          const size_t tap = (size_t) (ph / stride_width) * kernel_width + ph % stride_width;   /* (ky, kx) = (py, px) */
          for (size_t oc = 0; oc < goc; oc++) {
            memcpy(mat + ((size_t) ph * n_pad + oc) * gic, conv_order + (oc * kernel_size + tap) * gic, gic);
            cols_bias[(size_t) ph * n_pad + oc] = bias[oc];
          }
        }
        qnnp_pack_igemm_w_slots(1, d2s_cols, 1, (uint32_t) gic, kc_slot, d2s_cols, d2s_k_pad,
            input_zero_point, kernel_zero_point, mat, cols_bias, (int8_t*) packed, packed_bias);
        op->d_weights = qnnp_hip_alloc(dw_bytes);
        op->d_bias = qnnp_upload_bias_pair(packed_bias, db_bytes / sizeof(int32_t));   /* bias-pair.h */
        ok = op->d_weights != NULL && op->d_bias != NULL &&
            qnnp_hip_h2d(op->d_weights, packed, dw_bytes, 0) == QNNP_HIP_OK;
      }
      free(mat);
      free(cols_bias);
      free(packed);
      free(packed_bias);
      if (!ok) {
        qnnp_log_error("device allocation or upload failed: %zu bytes of packed depth-to-space weights on the device", dw_bytes + db_bytes);
        goto error;
      }
      op->k_pad = d2s_k_pad;
      op->deconv_d2s = 1;
    }
    goto packed_done;
  }
  const uint32_t k_total = (uint32_t) (kernel_size * kc_slot);
  const uint32_t k_pad = qnnp_round_up_u32(k_total, 64);
  const size_t w_bytes = qnnp_igemm_packed_weights_size(groups, n_pad, k_pad);
  const size_t b_bytes = sizeof(int32_t) * (size_t) groups * n_pad;
  host_weights = malloc(w_bytes);
  host_bias = malloc(b_bytes);
  if (host_weights == NULL || host_bias == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for packed weights", w_bytes + b_bytes);
    goto error;
  }
  qnnp_pack_igemm_w_slots(groups, (uint32_t) goc, (uint32_t) kernel_size, (uint32_t) gic, kc_slot, n_pad, k_pad,
      input_zero_point, kernel_zero_point, conv_order, bias, (int8_t*) host_weights, host_bias);
  op->n_pad = n_pad;
  op->k_pad = k_pad;
  op->kc_slot = kc_slot;
  op->d_weights = qnnp_hip_alloc(w_bytes);
  op->d_bias = qnnp_upload_bias_pair((const int32_t*) host_bias, (size_t) groups * n_pad);   /* bias-pair.h */
  if (op->d_weights == NULL || op->d_bias == NULL ||
      qnnp_hip_h2d(op->d_weights, host_weights, w_bytes, 0) != QNNP_HIP_OK) {
    qnnp_log_error("device allocation or upload failed: %zu bytes of packed weights on the device", w_bytes + b_bytes);
    goto error;
  }
packed_done:
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
  op->requant.accumulator_bits = qnnp_accumulator_bits(bias, (size_t) groups * group_output_channels,
      kernel_size * group_input_channels);
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

/* The device-side description of the phase GEMMs of the geometry and batch bound to `op`: all non-empty phases run
 * as ONE launch of the offset-table kernel (blockIdx.y = phase), so that the small per-phase problems fill the chip
 * together and a workgroup's next tile overlaps its current epilogue. */
static enum qnnp_status upload_phase_table(struct qnnp_operator* op)
{
  struct qnnp_hip_igemm_phase table[QNNP_MAX_DECONV_PHASES];
  uint32_t n = 0;
  op->phase_max_rows = 0;
  op->phase_max_k_pad = 0;
  for (uint32_t i = 0; i < op->deconv_phases; i++) {
    const struct qnnp_deconv_phase* ph = &op->phase[i];
    if (ph->rows == 0) continue;
    table[n].packed_w = (const int8_t*) ph->d_weights;
    table[n].bias2 = ph->d_bias;
    table[n].offsets = ph->d_offsets;
    table[n].out_rows = ph->d_out_rows;
    table[n].rows = (uint32_t) (op->batch_size * ph->rows);
    table[n].rows_per_image = (uint32_t) ph->rows;
    table[n].ks = ph->taps;
    table[n].k_total = ph->taps * op->kc_slot;
    table[n].k_pad = ph->k_pad;
    table[n].reserved = 0;
    if (ph->rows > op->phase_max_rows) op->phase_max_rows = ph->rows;
    if (ph->k_pad > op->phase_max_k_pad) op->phase_max_k_pad = ph->k_pad;
    n++;
  }
  op->phase_table_entries = n;
  if (n == 0) return qnnp_status_success;
  if (op->d_phase_table == NULL) {
    op->d_phase_table = qnnp_hip_alloc(sizeof(table));
    if (op->d_phase_table == NULL) {
      qnnp_log_error("out of host memory: %zu bytes for the phase table", sizeof(table));
      return qnnp_status_out_of_memory;
    }
  }
  if (qnnp_hip_h2d(op->d_phase_table, table, sizeof(struct qnnp_hip_igemm_phase) * n, 0) != QNNP_HIP_OK) {
    qnnp_log_error("failed to upload the phase table");
    return qnnp_status_out_of_memory;
  }
  return qnnp_status_success;
}

static enum qnnp_status qnnp_setup_deconvolution2d_nhwc_q8_impl(
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
    qnnp_log_error("qnnp_setup_deconvolution2d_nhwc_q8 called before qnnp_initialize succeeded");
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
    qnnp_log_error("cannot set up deconvolution with %zux%zu input: input dimensions may not be zero",
        input_width, input_height);
    return qnnp_status_invalid_parameter;
  }
  const size_t in_channels = (size_t) op->groups * op->group_input_channels;
  const size_t out_channels = (size_t) op->groups * op->group_output_channels;
  if (input == NULL || output == NULL || input_pixel_stride < in_channels || output_pixel_stride < out_channels) {
    qnnp_log_error("cannot set up deconvolution: NULL tensor or pixel stride smaller than the channel count");
    return qnnp_status_invalid_parameter;
  }
  const size_t pad_h = (size_t) op->input_padding_top + op->input_padding_bottom;
  const size_t pad_w = (size_t) op->input_padding_left + op->input_padding_right;
  const size_t full_h = compute_output_dimension(input_height, 0, op->adjustment_height, op->kernel_height,
      op->dilation_height, op->stride_height);
  const size_t full_w = compute_output_dimension(input_width, 0, op->adjustment_width, op->kernel_width,
      op->dilation_width, op->stride_width);
  if (pad_h >= full_h || pad_w >= full_w) {
    qnnp_log_error("cannot set up deconvolution with %zux%zu input: the padding removes the whole output",
        input_width, input_height);
    return qnnp_status_invalid_parameter;
  }

  /* reference deconvolution.c:243-262 */
  op->setup_valid = 0;   /* until every check, allocation and upload below has succeeded */
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
    qnnp_log_error("cannot set up deconvolution: %zu output pixels / %zu-byte images exceed the device kernels' index range",
        batch_size * output_size, input_size * input_pixel_stride);
    return qnnp_status_unsupported_parameter;
  }

  op->input_span = (batch_size * input_size - 1) * input_pixel_stride + in_channels;
  op->output_span = (batch_size * output_size - 1) * output_pixel_stride + out_channels;
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

  op->variant = 1;   /* the offset-table kernel: the table, not the geometry, defines this operator */
  op->deconv_stream = qnnp_state.opt_gemm_kernel == 13 ? 2 : (qnnp_state.opt_gemm_kernel == 1 ? 1 : 0);
  if (op->deconv_phases != 0) {
    if (op->offsets_in_h == input_height && op->offsets_in_w == input_width &&
        op->offsets_in_stride == input_pixel_stride) {
      return upload_phase_table(op);  /* offset tables are pointer- and batch-invariant; the row counts are not */
    }
    op->offsets_in_h = 0;
    const size_t sh = op->stride_height, sw = op->stride_width;
    for (uint32_t i = 0; i < op->deconv_phases; i++) {
      struct qnnp_deconv_phase* ph = &op->phase[i];
      const size_t py = i / sw, px = i % sw;
      /* first output row / column of the phase: (o + pad) % stride == phase */
      const size_t oy0 = (py + sh - op->input_padding_top % sh) % sh;
      const size_t ox0 = (px + sw - op->input_padding_left % sw) % sw;
      const size_t rows_y = oy0 < op->output_height ? (op->output_height - oy0 + sh - 1) / sh : 0;
      const size_t rows_x = ox0 < op->output_width ? (op->output_width - ox0 + sw - 1) / sw : 0;
      ph->rows = rows_y * rows_x;
      if (ph->rows == 0) continue;
      int32_t* host_offsets = (int32_t*) malloc(sizeof(int32_t) * ph->rows * ph->taps);
      int32_t* host_rows = (int32_t*) malloc(sizeof(int32_t) * ph->rows);
      if (host_offsets == NULL || host_rows == NULL) {
        free(host_offsets);
        free(host_rows);
        qnnp_log_error("out of host memory: %zu bytes for a phase table", sizeof(int32_t) * ph->rows * (ph->taps + 1));
        return qnnp_status_out_of_memory;
      }
      size_t r = 0;
      for (size_t oy = oy0; oy < op->output_height; oy += sh) {
        for (size_t ox = ox0; ox < op->output_width; ox += sw, r++) {
          host_rows[r] = (int32_t) (oy * op->output_width + ox);
          for (uint32_t t = 0; t < ph->taps; t++) {
            int32_t entry = QNNP_OFFSET_PADDING;
            if (ph->tap_ky[t] != 255) {
              /* reference src/indirection.c:171-177; the divisions are exact by construction of the phase */
              const size_t y = oy + op->input_padding_top - (size_t) ph->tap_ky[t] * op->dilation_height;
              const size_t x = ox + op->input_padding_left - (size_t) ph->tap_kx[t] * op->dilation_width;
              const size_t iy = y / sh, ix = x / sw;
              if (iy * sh == y && iy < input_height && ix * sw == x && ix < input_width) {
                entry = (int32_t) ((iy * input_width + ix) * input_pixel_stride);
              }
            }
            host_offsets[r * ph->taps + t] = entry;
          }
        }
      }
      if (ph->rows_capacity < ph->rows) {
        qnnp_hip_free(ph->d_offsets);
        qnnp_hip_free(ph->d_out_rows);
        ph->rows_capacity = 0;
        ph->d_offsets = (int32_t*) qnnp_hip_alloc(sizeof(int32_t) * ph->rows * ph->taps);
        ph->d_out_rows = (int32_t*) qnnp_hip_alloc(sizeof(int32_t) * ph->rows);
        if (ph->d_offsets != NULL && ph->d_out_rows != NULL) ph->rows_capacity = ph->rows;
      }
      const int ok = ph->rows_capacity >= ph->rows &&
          qnnp_hip_h2d(ph->d_offsets, host_offsets, sizeof(int32_t) * ph->rows * ph->taps, 0) == QNNP_HIP_OK &&
          qnnp_hip_h2d(ph->d_out_rows, host_rows, sizeof(int32_t) * ph->rows, 0) == QNNP_HIP_OK;
      free(host_offsets);
      free(host_rows);
      if (!ok) {
        qnnp_log_error("failed to place a phase table on the device");
        return qnnp_status_out_of_memory;
      }
    }
    op->offsets_in_h = input_height;
    op->offsets_in_w = input_width;
    op->offsets_in_stride = input_pixel_stride;
    return upload_phase_table(op);
  }
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
    qnnp_log_error("out of host memory: %zu bytes for the offset table", sizeof(int32_t) * entries);
    return qnnp_status_out_of_memory;
  }
  if (op->offsets_capacity < entries) {
    qnnp_hip_free(op->d_offsets);
    op->offsets_capacity = 0;
    op->d_offsets = (int32_t*) qnnp_hip_alloc(sizeof(int32_t) * entries);
    if (op->d_offsets == NULL) {
      free(host_table);
      qnnp_log_error("out of host memory: %zu bytes for the device offset table", sizeof(int32_t) * entries);
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

/* ---- public entry points: run the implementation inside the right device context ------------------
 * create: the calling thread's selected device (qnnp_gfx950_set_device, default = the primary one) becomes the
 * operator's device; setup: the operator's device. The previous HIP device of the thread is restored on return. */

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
  if (!qnnp_state.initialized) {
    return qnnp_create_deconvolution2d_nhwc_q8_impl(input_padding_top, input_padding_right, input_padding_bottom, input_padding_left, adjustment_height, adjustment_width, kernel_height, kernel_width, stride_height, stride_width, dilation_height, dilation_width, groups, group_input_channels, group_output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias, output_zero_point, output_scale, output_min, output_max, flags, deconvolution_out);   /* logs and answers qnnp_status_uninitialized */
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
  const enum qnnp_status status = qnnp_create_deconvolution2d_nhwc_q8_impl(input_padding_top, input_padding_right, input_padding_bottom, input_padding_left, adjustment_height, adjustment_width, kernel_height, kernel_width, stride_height, stride_width, dilation_height, dilation_width, groups, group_input_channels, group_output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias, output_zero_point, output_scale, output_min, output_max, flags, deconvolution_out);
  qnnp_hip_leave(token);
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
  if (!qnnp_state.initialized || op == NULL) {
    return qnnp_setup_deconvolution2d_nhwc_q8_impl(op, batch_size, input_height, input_width, input, input_pixel_stride, output, output_pixel_stride, threadpool);   /* answers qnnp_status_uninitialized / invalid_parameter */
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
  const enum qnnp_status status = qnnp_setup_deconvolution2d_nhwc_q8_impl(op, batch_size, input_height, input_width, input, input_pixel_stride, output, output_pixel_stride, threadpool);
  /* the implementation cleared setup_valid where it began to change the operator: a failed setup leaves it
   * unrunnable instead of half updated (run answers invalid_parameter) */
  if (status == qnnp_status_success) {
    op->setup_valid = 1;
  }
  qnnp_hip_leave(token);
  return status;
}
