/*
 * fused-block.c -- qnnp_gfx950_create_fused_block / qnnp_gfx950_setup_fused_block (include/qnnpack_gfx950.h):
 * an inverted-residual block [1x1 expand ->] 3x3 depthwise -> 1x1 project [-> + input] as ONE operator whose
 * intermediates never leave the chip (hip/q8fused.hip). SURVEY.md section 8f, row 2; no reference counterpart --
 * the reference runs the three (four) operators one after another (bench/convolution.cc:464-471 lists the shape
 * triples).
 *
 * The fused operator is BUILT FROM the stand-alone operators, created through the regular API: it borrows their
 * packed device weights, folded biases and requantization parameters, so its result is bit-identical to running them
 * in sequence by construction of the arithmetic, and a caller can fall back to exactly those operators when create or
 * setup reports unsupported_parameter. The source operators must outlive the fused one.
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <qnnpack.h>
#include <qnnpack_gfx950.h>

#include "hip/qnnp_hip.h"
#include "log.h"
#include "operator.h"
#include "pack.h"
#include "state.h"

static int is_pointwise(const struct qnnp_operator* op)
{
  return op != NULL && op->ukernel_type == qnnp_ukernel_type_gemm && !op->transposed && op->groups == 1 &&
      op->kernel_height == 1 && op->kernel_width == 1 && op->kc_slot == op->group_input_channels;
}

static struct qnnp_hip_fused_args fused_args(const struct qnnp_operator* op, const void* input, void* output)
{
  const struct qnnp_operator* ex = op->fused_expand;
  const struct qnnp_operator* dw = op->fused_depthwise;
  const struct qnnp_operator* pr = op->fused_project;
  struct qnnp_hip_fused_args a;
  a.input = (const uint8_t*) input;
  a.output = (uint8_t*) output;
  a.batch = (uint32_t) op->batch_size;
  a.input_height = (uint32_t) op->input_height;
  a.input_width = (uint32_t) op->input_width;
  a.output_height = (uint32_t) op->output_height;
  a.output_width = (uint32_t) op->output_width;
  a.input_channels = (uint32_t) (ex != NULL ? ex->group_input_channels : dw->groups);
  a.hidden_channels = dw->groups;
  a.output_channels = (uint32_t) pr->group_output_channels;
  a.input_stride = (uint32_t) op->input_pixel_stride;
  a.output_stride = (uint32_t) op->output_pixel_stride;
  a.stride = dw->stride_height;
  a.has_expand = ex != NULL;
  a.expand_w = ex != NULL ? (const int8_t*) ex->d_weights : NULL;
  a.expand_bias2 = ex != NULL ? ex->d_bias : NULL;
  a.expand_k_pad = ex != NULL ? ex->k_pad : 0;
  a.expand_n_pad = ex != NULL ? ex->n_pad : 0;
  a.expand_row_coeff = ex != NULL ? 128 - (int32_t) ex->kernel_zero_point : 0;
  a.expand_rq = ex != NULL ? ex->requant : dw->requant;
  a.dw_wadj = (const int16_t*) dw->d_weights;
  a.dw_bias1 = dw->d_bias;
  a.dw_c_pad = dw->c_pad;
  a.dw_input_zero_point = dw->input_zero_point;
  a.dw_rq = dw->requant;
  a.project_w = (const int8_t*) pr->d_weights;
  a.project_bias2 = pr->d_bias;
  a.project_k_pad = pr->k_pad;
  a.project_n_pad = pr->n_pad;
  a.project_row_coeff = 128 - (int32_t) pr->kernel_zero_point;
  a.project_rq = pr->requant;
  a.has_residual = op->fused_add != NULL;
  if (op->fused_add != NULL) {
    a.add = op->fused_add->add_params;
  } else {
    const struct qnnp_hip_add_params none = {0};
    a.add = none;
  }
  return a;
}

/* ---- the strip kernel's images (hip/q8fusedstrip.hip) -------------------------------------------------------------
 * Everything is derived from what the stand-alone operators already hold on the device (the caller's kernel / bias
 * arrays are gone by now): the standard fragment image w' = w - 128 and its folded bias2 (pack.h qnnp_pack_igemm_w), the
 * depthwise int16 image w - kzp and its bias1 (qnnp_pack_dwconv_w). Centred element: w ^ flip with flip = 0x80 for
 * kernel zero point 128 (= w') and 0x7F for 127 (= ~w' = 127 - w); padding positions stay zero weights.
 *   bias       = bias2 - (128 - izp) * sum w' - K (128 - izp)(128 - kzp)                  (undoing pack.h's folding)
 *   biasc[n]   = bias + (kzp - izp) * sum_k (w - kzp)   [+ 2^31 where the stage's rounding sequence is an offset form]
 * so that  bias + sum (a - izp)(w - kzp) = biasc + sum (a ^ flip)(w ^ flip)  with both factors valid int8. */
static uint32_t round_up32(uint32_t x) { return (x + 31u) / 32u * 32u; }

/* one pointwise stage (pack.h qnnp_strip_pointwise_images does the arithmetic on host copies of the device images) */
static int strip_pointwise(const struct qnnp_operator* op, uint32_t n, uint32_t k, int8_t* frags, int32_t* biasc)
{
  const size_t w_bytes = (size_t) op->n_pad * op->k_pad;
  int8_t* std_w = (int8_t*) malloc(w_bytes);
  int32_t* std_b = (int32_t*) malloc(sizeof(int32_t) * op->n_pad);
  const int ok = std_w != NULL && std_b != NULL &&
      qnnp_hip_d2h(std_w, op->d_weights, w_bytes, 0) == QNNP_HIP_OK &&
      qnnp_hip_d2h(std_b, op->d_bias, sizeof(int32_t) * op->n_pad, 0) == QNNP_HIP_OK;
  if (ok) {
    qnnp_strip_pointwise_images(std_w, std_b, op->k_pad, n, k, op->input_zero_point, op->kernel_zero_point,
        qnnp_hip_fused_strip_bias_offset(&op->requant), frags, biasc);
  }
  free(std_w);
  free(std_b);
  return ok;
}

static int strip_depthwise(const struct qnnp_operator* dw, uint32_t ch, uint32_t hidden_pad, int8_t* w2, int32_t* biasc)
{
  const size_t w_elems = (size_t) 9 * dw->c_pad;
  int16_t* wadj = (int16_t*) malloc(sizeof(int16_t) * w_elems);
  int32_t* bias1 = (int32_t*) malloc(sizeof(int32_t) * dw->c_pad);
  const int ok = wadj != NULL && bias1 != NULL &&
      qnnp_hip_d2h(wadj, dw->d_weights, sizeof(int16_t) * w_elems, 0) == QNNP_HIP_OK &&
      qnnp_hip_d2h(bias1, dw->d_bias, sizeof(int32_t) * dw->c_pad, 0) == QNNP_HIP_OK;
  if (ok) {
    qnnp_strip_depthwise_images(wadj, bias1, dw->c_pad, ch, hidden_pad, dw->input_zero_point, dw->kernel_zero_point,
        qnnp_hip_fused_strip_bias_offset(&dw->requant), w2, biasc);
  }
  free(wadj);
  free(bias1);
  return ok;
}

static int centred_zero_point(const struct qnnp_operator* op)
{
  return op == NULL || op->kernel_zero_point == 127 || op->kernel_zero_point == 128;
}

/* builds op->d_strip; leaves it NULL (no error) when the block is outside the strip kernel's arithmetic */
static void build_strip_images(struct qnnp_operator* op)
{
  const struct qnnp_operator* ex = op->fused_expand;
  const struct qnnp_operator* dw = op->fused_depthwise;
  const struct qnnp_operator* pr = op->fused_project;
  if (!centred_zero_point(ex) || !centred_zero_point(dw) || !centred_zero_point(pr)) return;
  const uint32_t ch = dw->groups;
  const uint32_t cin = ex != NULL ? (uint32_t) ex->group_input_channels : ch;
  const uint32_t cout = (uint32_t) pr->group_output_channels;
  if (cin > 160 || ch % 4 != 0 || cin % 4 != 0 || cout % 4 != 0) return;
  const uint32_t hidden_pad = round_up32(ch), output_pad = round_up32(cout);
  const uint32_t kb1 = (cin + 31u) / 32u, nb1 = hidden_pad / 32u, nb3 = output_pad / 32u;
  const size_t w1_bytes = ex != NULL ? (size_t) nb1 * kb1 * 1024u : 0, w3_bytes = (size_t) nb3 * nb1 * 1024u;
  const size_t w2_bytes = ((size_t) 9 * hidden_pad + 255u) & ~(size_t) 255u;
  op->strip_w1 = 0;
  op->strip_w3 = w1_bytes;
  op->strip_w2 = op->strip_w3 + w3_bytes;
  op->strip_b1 = op->strip_w2 + w2_bytes;
  op->strip_b2 = op->strip_b1 + sizeof(int32_t) * hidden_pad;
  op->strip_b3 = op->strip_b2 + sizeof(int32_t) * hidden_pad;
  const size_t total = op->strip_b3 + sizeof(int32_t) * output_pad;
  uint8_t* host = (uint8_t*) calloc(1, total);
  if (host == NULL) return;
  int ok = 1;
  if (ex != NULL) ok = strip_pointwise(ex, ch, cin, (int8_t*) (host + op->strip_w1), (int32_t*) (host + op->strip_b1));
  ok = ok && strip_depthwise(dw, ch, hidden_pad, (int8_t*) (host + op->strip_w2), (int32_t*) (host + op->strip_b2));
  ok = ok && strip_pointwise(pr, cout, ch, (int8_t*) (host + op->strip_w3), (int32_t*) (host + op->strip_b3));
  if (ok) {
    op->d_strip = qnnp_hip_alloc(total);
    if (op->d_strip != NULL && qnnp_hip_h2d(op->d_strip, host, total, 0) != QNNP_HIP_OK) {
      qnnp_hip_free(op->d_strip);
      op->d_strip = NULL;
    }
  }
  free(host);
  if (op->d_strip != NULL) {
    op->strip_hidden_pad = hidden_pad;
    op->strip_output_pad = output_pad;
    op->strip_flip1 = ex != NULL && ex->kernel_zero_point == 127 ? 0x7F : 0x80;
    op->strip_flip2 = dw->kernel_zero_point == 127 ? 0x7F : 0x80;
    op->strip_flip3 = pr->kernel_zero_point == 127 ? 0x7F : 0x80;
    op->strip_dw_pad = (uint8_t) (dw->input_zero_point ^ op->strip_flip2);
  }
}

static struct qnnp_hip_fused_strip_args strip_args(const struct qnnp_operator* op, const void* input, void* output)
{
  const struct qnnp_operator* ex = op->fused_expand;
  const struct qnnp_operator* dw = op->fused_depthwise;
  const struct qnnp_operator* pr = op->fused_project;
  const uint8_t* blob = (const uint8_t*) op->d_strip;
  struct qnnp_hip_fused_strip_args a;
  memset(&a, 0, sizeof(a));
  a.input = (const uint8_t*) input;
  a.output = (uint8_t*) output;
  a.batch = (uint32_t) op->batch_size;
  a.input_height = (uint32_t) op->input_height;
  a.input_width = (uint32_t) op->input_width;
  a.output_height = (uint32_t) op->output_height;
  a.output_width = (uint32_t) op->output_width;
  a.input_channels = (uint32_t) (ex != NULL ? ex->group_input_channels : dw->groups);
  a.hidden_channels = dw->groups;
  a.output_channels = (uint32_t) pr->group_output_channels;
  a.input_stride = (uint32_t) op->input_pixel_stride;
  a.output_stride = (uint32_t) op->output_pixel_stride;
  a.stride = dw->stride_height;
  a.has_expand = ex != NULL;
  a.has_residual = op->fused_add != NULL;
  a.expand_w = (const int8_t*) (blob + op->strip_w1);
  a.expand_bias = (const int32_t*) (blob + op->strip_b1);
  a.expand_flip = op->strip_flip1;
  a.expand_rq = ex != NULL ? ex->requant : dw->requant;
  a.dw_w = (const int8_t*) (blob + op->strip_w2);
  a.dw_bias = (const int32_t*) (blob + op->strip_b2);
  a.dw_flip = op->strip_flip2;
  a.dw_pad = op->strip_dw_pad;
  a.dw_rq = dw->requant;
  a.project_w = (const int8_t*) (blob + op->strip_w3);
  a.project_bias = (const int32_t*) (blob + op->strip_b3);
  a.project_flip = op->strip_flip3;
  a.project_rq = pr->requant;
  a.hidden_pad = op->strip_hidden_pad;
  a.output_pad = op->strip_output_pad;
  if (op->fused_add != NULL) a.add = op->fused_add->add_params;
  a.rows_per_strip = op->fused_rows_per_strip;
  a.weights_in_lds = op->fused_weights;
  return a;
}

static enum qnnp_status qnnp_gfx950_create_fused_block_impl(
    qnnp_operator_t expand, qnnp_operator_t depthwise, qnnp_operator_t project, qnnp_operator_t residual_add,
    qnnp_operator_t* fused_out)
{
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_gfx950_create_fused_block called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  if (depthwise == NULL || project == NULL || fused_out == NULL) {
    return qnnp_status_invalid_parameter;
  }
  /* the shapes the fused kernel is written for; anything else stays with the stand-alone operators */
  if (depthwise->ukernel_type != qnnp_ukernel_type_dwconv ||
      depthwise->kernel_height != 3 || depthwise->kernel_width != 3 ||
      depthwise->dilation_height != 1 || depthwise->dilation_width != 1 ||
      depthwise->stride_height != depthwise->stride_width || depthwise->stride_height > 2 ||
      depthwise->input_padding_top != 1 || depthwise->input_padding_left != 1 ||
      depthwise->input_padding_bottom != 1 || depthwise->input_padding_right != 1 ||
      !is_pointwise(project) || project->group_input_channels != depthwise->groups ||
      (expand != NULL && (!is_pointwise(expand) || expand->group_output_channels != depthwise->groups))) {
    qnnp_log_error("cannot create fused block: operators are not a [1x1 ->] 3x3 depthwise (pad 1) -> 1x1 chain");
    return qnnp_status_unsupported_parameter;
  }
  if (residual_add != NULL) {
    const size_t cin = expand != NULL ? expand->group_input_channels : depthwise->groups;
    if (residual_add->ukernel_type != qnnp_ukernel_type_add || residual_add->channels != project->group_output_channels ||
        cin != project->group_output_channels || depthwise->stride_height != 1) {
      qnnp_log_error("cannot create fused block: the residual add does not match the block's input / output");
      return qnnp_status_unsupported_parameter;
    }
  }
  if (project->device != depthwise->device || (expand != NULL && expand->device != depthwise->device)) {
    qnnp_log_error("cannot create fused block: the operators live on different devices");
    return qnnp_status_invalid_parameter;
  }
  qnnp_operator_t op = calloc(1, sizeof(struct qnnp_operator));
  if (op != NULL) op->device = depthwise->device;   /* it borrows their device images */
  if (op == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for qnnp_operator structure", sizeof(struct qnnp_operator));
    return qnnp_status_out_of_memory;
  }
  op->fused_expand = expand;
  op->fused_depthwise = depthwise;
  op->fused_project = project;
  op->fused_add = residual_add;
  op->channels = project->group_output_channels;
  op->ukernel_type = qnnp_ukernel_type_fused_block;
  build_strip_images(op);                       /* (optional: without them the tile kernel is all the block has) */
  *fused_out = op;
  return qnnp_status_success;
}

static enum qnnp_status qnnp_gfx950_setup_fused_block_impl(
    qnnp_operator_t op, size_t batch_size, size_t input_height, size_t input_width,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride)
{
  if (!qnnp_state.initialized) {
    return qnnp_status_uninitialized;
  }
  if (op == NULL || op->ukernel_type != qnnp_ukernel_type_fused_block) {
    return qnnp_status_invalid_parameter;
  }
  if (batch_size == 0) {
    op->batch_size = 0;
    return qnnp_status_success;
  }
  const size_t cin = op->fused_expand != NULL ? op->fused_expand->group_input_channels : op->fused_depthwise->groups;
  const size_t cout = op->fused_project->group_output_channels;
  if (input_height == 0 || input_width == 0 || input == NULL || output == NULL ||
      input_stride < cin || output_stride < cout) {
    qnnp_log_error("cannot set up fused block: zero extent, NULL tensor or stride smaller than the channel count");
    return qnnp_status_invalid_parameter;
  }
  const size_t s = op->fused_depthwise->stride_height;
  op->setup_valid = 0;   /* until every check, allocation and upload below has succeeded */
  op->batch_size = batch_size;
  op->input_height = input_height;
  op->input_width = input_width;
  op->input = input;
  op->input_pixel_stride = input_stride;
  op->output_height = (input_height + 2 - 3) / s + 1;
  op->output_width = (input_width + 2 - 3) / s + 1;
  op->output = output;
  op->output_pixel_stride = output_stride;
  const size_t in_pixels = batch_size * input_height * input_width;
  const size_t out_pixels = batch_size * op->output_height * op->output_width;
  if (in_pixels > (size_t) UINT32_MAX / 2 || in_pixels * input_stride > (size_t) UINT32_MAX) {
    qnnp_log_error("cannot set up fused block: %zu pixels exceed the device kernel's index range", in_pixels);
    return qnnp_status_unsupported_parameter;
  }
  op->input_span = (in_pixels - 1) * input_stride + cin;
  op->output_span = (out_pixels - 1) * output_stride + cout;
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
  /* does a kernel take this block (LDS plan, channel multiples, alignment)? The strip kernel first. */
  const void* in_dev = op->input_on_device ? input : op->d_stage_in;
  void* out_dev = op->output_on_device ? (void*) output : op->d_stage_out;
  op->fused_use_strip = 0;
  op->fused_rows_per_strip = (uint32_t) qnnp_state.opt_fused_rows;
  op->fused_weights = (uint32_t) qnnp_state.opt_fused_weights;
  if (op->d_strip != NULL && qnnp_state.opt_fused_kernel != 1) {
    const struct qnnp_hip_fused_strip_args probe = strip_args(op, in_dev, out_dev);
    if (qnnp_hip_fused_strip_supported(&probe)) op->fused_use_strip = 1;
  }
  if (!op->fused_use_strip) {
    const struct qnnp_hip_fused_args probe = fused_args(op, in_dev, out_dev);
    if (qnnp_state.opt_fused_kernel == 2 || !qnnp_hip_fused_block_supported(&probe)) {
      op->batch_size = 0;
      op->input = NULL;
      qnnp_log_info("fused block: shape outside the fused kernels' range; use the stand-alone operators");
      return qnnp_status_unsupported_parameter;
    }
  }
  return qnnp_status_success;
}

int qnnp_fused_block_launch(struct qnnp_operator* op, const void* input, void* output)
{
  if (op->fused_use_strip) {
    const struct qnnp_hip_fused_strip_args sargs = strip_args(op, input, output);
    return qnnp_hip_fused_strip_run(&sargs, &op->kernel_name);
  }
  const struct qnnp_hip_fused_args args = fused_args(op, input, output);
  return qnnp_hip_fused_block_run(&args, &op->kernel_name);
}

/* ---- public entry points: run the implementation inside the right device context ------------------
 * create: the device of the member operators (the block's own device images live beside theirs, whatever device the
 * calling thread has selected); setup: the operator's device. The previous HIP device of the thread is restored on return. */

enum qnnp_status qnnp_gfx950_create_fused_block(
    qnnp_operator_t expand, qnnp_operator_t depthwise, qnnp_operator_t project, qnnp_operator_t residual_add,
    qnnp_operator_t* fused_out)
{
  if (!qnnp_state.initialized) {
    return qnnp_gfx950_create_fused_block_impl(expand, depthwise, project, residual_add, fused_out);   /* logs and answers qnnp_status_uninitialized */
  }
  /* The block borrows its members' device images and adds its own (build_strip_images: alloc + upload), so everything
   * device-side must happen on THEIR device -- not on whichever one the calling thread has selected. Members on
   * different devices are refused by the implementation; a NULL depthwise as well. */
  const int token = qnnp_hip_enter(depthwise != NULL ? depthwise->device : qnnp_hip_device());
  if (token < 0) {
    return depthwise != NULL ? qnnp_status_invalid_parameter : qnnp_status_unsupported_hardware;
  }
  if (qnnp_hip_graph_capturing()) {
    /* inside qnnp_gfx950_graph_begin ... graph_end on this device only operator launches are recordable: an upload
     * would become a graph node reading host memory that is freed right after this call */
    qnnp_hip_leave(token);
    return qnnp_status_invalid_parameter;
  }
  const enum qnnp_status status = qnnp_gfx950_create_fused_block_impl(expand, depthwise, project, residual_add, fused_out);
  qnnp_hip_leave(token);
  return status;
}

enum qnnp_status qnnp_gfx950_setup_fused_block(
    qnnp_operator_t op, size_t batch_size, size_t input_height, size_t input_width,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride)
{
  if (!qnnp_state.initialized || op == NULL) {
    return qnnp_gfx950_setup_fused_block_impl(op, batch_size, input_height, input_width, input, input_stride, output, output_stride);   /* answers qnnp_status_uninitialized / invalid_parameter */
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
  const enum qnnp_status status = qnnp_gfx950_setup_fused_block_impl(op, batch_size, input_height, input_width, input, input_stride, output, output_stride);
  /* the implementation cleared setup_valid where it began to change the operator: a failed setup leaves it
   * unrunnable instead of half updated (run answers invalid_parameter) */
  if (status == qnnp_status_success) {
    op->setup_valid = 1;
  }
  qnnp_hip_leave(token);
  return status;
}
