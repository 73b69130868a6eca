/*
 * convolution.c -- qnnp_create_convolution2d_nhwc_q8 / qnnp_setup_convolution2d_nhwc_q8
 * for the gfx950 build.
 *
 * Replaces reference src/convolution.c:39-378 (create) and :380-492 (setup):
 * same argument validation and status codes, same operator-type selection idea,
 * but the products of create/setup are device-resident:
 *   create: weights re-laid-out as MFMA operand fragments (or a tap-major int16
 *           image for depthwise) with both zero points folded into an int32 bias,
 *           uploaded once; Q31 requantization parameters.
 *   setup : output geometry; for general convolutions a batch-invariant int32
 *           offset table (pixel x tap -> byte offset inside one image, -1 = padding)
 *           instead of the reference's per-image table of absolute host pointers
 *           (src/indirection.c:18-79) -- im2col is never materialised and the
 *           table does not depend on the input pointer or the batch size.
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

/* reference src/convolution.c:29-37 */
static inline size_t compute_output_dimension(
    size_t padded_input, size_t kernel, size_t dilation, size_t subsampling)
{
  const size_t effective_kernel = (kernel - 1) * dilation + 1;
  return (padded_input - effective_kernel) / subsampling + 1;
}

static inline bool scale_is_valid(float scale)
{
  return scale > 0.0f && isnormal(scale);
}

static enum qnnp_status qnnp_create_convolution2d_nhwc_q8_impl(
    uint32_t input_padding_top,
    uint32_t input_padding_right,
    uint32_t input_padding_bottom,
    uint32_t input_padding_left,
    uint32_t kernel_height,
    uint32_t kernel_width,
    uint32_t subsampling_height,
    uint32_t subsampling_width,
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
    qnnp_operator_t* convolution_out)
{
  (void) flags; /* accepted and ignored, as in the reference (convolution.c:63) */
  qnnp_operator_t op = NULL;
  void* host_weights = NULL;
  int32_t* host_bias = NULL;
  enum qnnp_status status = qnnp_status_uninitialized;

  /* reference convolution.c:69-72 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_create_convolution2d_nhwc_q8 called before qnnp_initialize succeeded");
    goto error;
  }

  /* reference convolution.c:74-115 */
  status = qnnp_status_invalid_parameter;
  if (kernel_width == 0 || kernel_height == 0) {
    qnnp_log_error("cannot create convolution with %" PRIu32 "x%" PRIu32 " kernel: kernel dimensions may not be zero",
        kernel_width, kernel_height);
    goto error;
  }
  if (subsampling_width == 0 || subsampling_height == 0) {
    qnnp_log_error("cannot create convolution with %" PRIu32 "x%" PRIu32 " subsampling: subsampling dimensions may not be zero",
        subsampling_width, subsampling_height);
    goto error;
  }
  if (dilation_width == 0 || dilation_height == 0) {
    qnnp_log_error("cannot create convolution with %" PRIu32 "x%" PRIu32 " dilation: dilation dimensions may not be zero",
        dilation_width, dilation_height);
    goto error;
  }
  if (!scale_is_valid(input_scale)) {
    qnnp_log_error("cannot create convolution with %.7g input scale: a scale has to be a finite number above zero", input_scale);
    goto error;
  }
  if (!scale_is_valid(kernel_scale)) {
    qnnp_log_error("cannot create convolution with %.7g kernel scale: a scale has to be a finite number above zero", kernel_scale);
    goto error;
  }
  if (!scale_is_valid(output_scale)) {
    qnnp_log_error("cannot create convolution with %.7g output scale: a scale has to be a finite number above zero", output_scale);
    goto error;
  }
  if (groups == 0 || group_input_channels == 0 || group_output_channels == 0 || kernel == NULL || bias == NULL) {
    qnnp_log_error("cannot create convolution: groups, channel counts, kernel and bias may not be zero");
    goto error;
  }

  /* reference convolution.c:117-168 (the "inefficiency" notes are informational there) */
  status = qnnp_status_unsupported_parameter;
  const float convolution_scale = input_scale * kernel_scale / output_scale;
  if (convolution_scale >= 1.0f) {
    qnnp_log_error(
        "cannot create convolution with %.7g input scale, %.7g kernel scale, and %.7g output scale: "
        "convolution scale %.7g is greater or equal to 1.0",
        input_scale, kernel_scale, output_scale, convolution_scale);
    goto error;
  }
  if (!(convolution_scale >= 0x1.0p-32f)) {
    /* the reference's parameter builder asserts this range (requantization.h:28, :139) */
    qnnp_log_error("cannot create convolution: convolution scale %.7g is below 2**-32", convolution_scale);
    goto error;
  }
  const size_t kernel_size = (size_t) kernel_height * kernel_width;
  if (kernel_size * group_input_channels > (size_t) UINT32_MAX / 4 ||
      (size_t) groups * group_output_channels > (size_t) UINT32_MAX / 4) {
    qnnp_log_error("cannot create convolution: channel / kernel extents exceed the 32-bit index range of the device kernels");
    goto error;
  }

  status = qnnp_status_out_of_memory;
  op = calloc(1, sizeof(struct qnnp_operator));
  if (op != NULL) op->device = qnnp_hip_device();   /* the context this create runs in (entry point below) */
  if (op == NULL) {
    qnnp_log_error("out of host memory: %zu bytes for qnnp_operator structure", sizeof(struct qnnp_operator));
    goto error;
  }

  /*
   * Operator-type selection. Reference convolution.c:180-189 sends 3x3 / 5x5
   * depthwise to dwconv, unpadded stride-1 1x1 to gemm and the rest to conv.
   * The device depthwise kernel is not limited to 9 or 25 taps, so every
   * depthwise convolution (one input and one output channel per group) takes
   * it; results are identical, only the kernel differs.
   */
  const bool any_padding =
      (input_padding_left | input_padding_top | input_padding_right | input_padding_bottom) != 0;
  enum qnnp_ukernel_type ukernel_type;
  if (group_input_channels == 1 && group_output_channels == 1 && groups > 1) {
    ukernel_type = qnnp_ukernel_type_dwconv;
  } else if (kernel_size == 1 && subsampling_height == 1 && subsampling_width == 1 && !any_padding) {
    ukernel_type = qnnp_ukernel_type_gemm;
  } else {
    ukernel_type = qnnp_ukernel_type_conv;
  }

  if (ukernel_type == qnnp_ukernel_type_dwconv) {
    const uint32_t c_pad = qnnp_round_up_u32(groups, 16);
    const size_t w_bytes = sizeof(int16_t) * kernel_size * c_pad;
    const size_t b_bytes = sizeof(int32_t) * c_pad;
    host_weights = malloc(w_bytes);
    host_bias = malloc(b_bytes);
    if (host_weights == NULL || host_bias == NULL) {
      qnnp_log_error("out of host memory: %zu bytes for packed weights", w_bytes + b_bytes);
      goto error;
    }
    qnnp_pack_dwconv_w(groups, c_pad, kernel_height, kernel_width,
        input_zero_point, kernel_zero_point, kernel, bias, (int16_t*) host_weights, host_bias);
    op->c_pad = c_pad;
    op->dw_wrange = qnnp_dwconv_weight_range((const int16_t*) host_weights, kernel_size * c_pad);
    op->d_weights = qnnp_hip_alloc(w_bytes);
    op->d_bias = (int32_t*) qnnp_hip_alloc(b_bytes);
    if (op->d_weights == NULL || op->d_bias == NULL ||
        qnnp_hip_h2d(op->d_weights, host_weights, w_bytes, 0) != QNNP_HIP_OK ||
        qnnp_hip_h2d(op->d_bias, host_bias, b_bytes, 0) != QNNP_HIP_OK) {
      qnnp_log_error("device allocation or upload failed: %zu bytes of packed depthwise weights on the device", w_bytes + b_bytes);
      goto error;
    }
    /* 3x3 with weights in int8 range: the register image of the int8 dot-product walk (pack.h) */
    if (kernel_height == 3 && kernel_width == 3 && op->dw_wrange != 0) {
      const size_t q_bytes = sizeof(uint32_t) * 4 * c_pad;
      uint32_t* host_q = (uint32_t*) malloc(q_bytes);
      int ok = host_q != NULL;
      if (ok) {
        qnnp_pack_dwconv_dot4(c_pad, op->dw_wrange, (const int16_t*) host_weights, host_bias, host_q);
        op->d_dw_dot4 = qnnp_hip_alloc(q_bytes);
        ok = op->d_dw_dot4 != NULL && qnnp_hip_h2d(op->d_dw_dot4, host_q, q_bytes, 0) == QNNP_HIP_OK;
      }
      free(host_q);
      if (!ok) {
        qnnp_log_error("device allocation or upload failed: %zu bytes of depthwise dot-product weights on the device", q_bytes);
        goto error;
      }
    }
    /* 5x5 likewise: eight rows (pack.h qnnp_pack_dwconv_dot4_5x5) */
    if (kernel_height == 5 && kernel_width == 5 && op->dw_wrange != 0) {
      const size_t q_bytes = sizeof(uint32_t) * 8 * c_pad;
      uint32_t* host_q = (uint32_t*) malloc(q_bytes);
      int ok = host_q != NULL;
      if (ok) {
        qnnp_pack_dwconv_dot4_5x5(c_pad, op->dw_wrange, (const int16_t*) host_weights, host_bias, host_q);
        op->d_dw_dot4 = qnnp_hip_alloc(q_bytes);
        ok = op->d_dw_dot4 != NULL && qnnp_hip_h2d(op->d_dw_dot4, host_q, q_bytes, 0) == QNNP_HIP_OK;
      }
      free(host_q);
      if (!ok) {
        qnnp_log_error("device allocation or upload failed: %zu bytes of depthwise dot-product weights on the device", q_bytes);
        goto error;
      }
    }
    /* second image: int8 weight parts + folded bias for the matrix-core depthwise kernel */
    {
      const uint32_t c_pad32 = qnnp_round_up_u32(groups, 32);
      const size_t x_bytes = (size_t) 3 * kernel_size * c_pad32;
      const size_t bm_bytes = sizeof(int32_t) * c_pad32;
      int8_t* host_x = (int8_t*) malloc(x_bytes);
      int32_t* host_bm = (int32_t*) malloc(bm_bytes);
      int ok = host_x != NULL && host_bm != NULL;
      if (ok) {
        op->dwm_parts = qnnp_pack_dwconv_mfma(groups, c_pad32, kernel_height, kernel_width,
            input_zero_point, kernel_zero_point, kernel, bias, host_x, host_bm);
        op->c_pad32 = c_pad32;
        op->d_dwm_x = qnnp_hip_alloc(x_bytes);
        op->d_dwm_bias = (int32_t*) qnnp_hip_alloc(bm_bytes);
        ok = op->d_dwm_x != NULL && op->d_dwm_bias != NULL &&
            qnnp_hip_h2d(op->d_dwm_x, host_x, x_bytes, 0) == QNNP_HIP_OK &&
            qnnp_hip_h2d(op->d_dwm_bias, host_bm, bm_bytes, 0) == QNNP_HIP_OK;
      }
      free(host_x);
      free(host_bm);
      if (!ok) {
        qnnp_log_error("device allocation or upload failed: %zu bytes of depthwise weight parts on the device", x_bytes + bm_bytes);
        goto error;
      }
    }
  } else {
    /* 3-channel inputs (first layers): one unaligned 4-byte fetch per tap, see pack.h "channel slots" */
    const uint32_t kc_slot = (ukernel_type == qnnp_ukernel_type_conv && groups == 1 && group_input_channels == 3) ?
        4u : (uint32_t) group_input_channels;
    const uint32_t k_total = (uint32_t) (kernel_size * kc_slot);
    const uint32_t n_pad = qnnp_round_up_u32((uint32_t) group_output_channels, 32);
    const uint32_t k_pad = qnnp_round_up_u32(k_total, 64);
    const size_t w_bytes = qnnp_igemm_packed_weights_size(groups, n_pad, k_pad);
    const size_t b_bytes = sizeof(int32_t) * (size_t) groups * n_pad;
    host_weights = malloc(w_bytes);
    host_bias = malloc(b_bytes);
    if (host_weights == NULL || host_bias == NULL) {
      qnnp_log_error("out of host memory: %zu bytes for packed weights", w_bytes + b_bytes);
      goto error;
    }
    qnnp_pack_igemm_w_slots(groups, (uint32_t) group_output_channels, (uint32_t) kernel_size,
        (uint32_t) group_input_channels, kc_slot, n_pad, k_pad,
        input_zero_point, kernel_zero_point, kernel, bias, (int8_t*) host_weights, host_bias);
    op->n_pad = n_pad;
    op->k_pad = k_pad;
    op->kc_slot = kc_slot;
    op->d_weights = qnnp_hip_alloc(w_bytes);
    op->d_bias = qnnp_upload_bias_pair((const int32_t*) host_bias, (size_t) groups * n_pad);   /* bias-pair.h */
    if (op->d_weights == NULL || op->d_bias == NULL ||
        qnnp_hip_h2d(op->d_weights, host_weights, w_bytes, 0) != QNNP_HIP_OK) {
      qnnp_log_error("device allocation or upload failed: %zu bytes of packed weights on the device", w_bytes + 2 * b_bytes);
      goto error;
    }
    /* Grouped 1x1 as one dense GEMM (round 6). A group of 12 ... 68 channels fills a fifth ... a half of a 64-byte K step and a
     * 128-channel tile, every group's tile stages its own copy of the rows, and its output pieces are 12 ... 68 bytes; the dense
     * [groups * GOC][groups * GIC] matrix with the groups' blocks on its diagonal and the KERNEL ZERO POINT everywhere else
     * ((w - kzp) = 0: the grouped result bit for bit) reads every input byte once and writes whole pixels, at `groups` times the
     * multiplies -- which these layers do not notice: ShuffleNet v1's 28x28 layers at batch 128 run 15 ... 59 us grouped and 12 ... 27
     * dense (profiles/r06/grouped_1x1_dense_equivalents_r06v.txt); from 14x14 down the two are level or the grouped form leads, so
     * operator-run.c takes this image from 65536 rows up. Built for up to 1024 channels on either side (<= 1 MiB of weights). */
    if (ukernel_type == qnnp_ukernel_type_gemm && groups > 1 && (size_t) groups * group_input_channels <= 1024 &&
        (size_t) groups * group_output_channels <= 1024) {
      const uint32_t cin = (uint32_t) (groups * group_input_channels), cout = (uint32_t) (groups * group_output_channels);
      const uint32_t n_pad_d = qnnp_round_up_u32(cout, 32), k_pad_d = qnnp_round_up_u32(cin, 64);
      const size_t wd_bytes = qnnp_igemm_packed_weights_size(1, n_pad_d, k_pad_d);
      uint8_t* dense = (uint8_t*) malloc((size_t) cout * cin);
      int8_t* host_wd = (int8_t*) malloc(wd_bytes);
      int32_t* host_bd = (int32_t*) malloc(sizeof(int32_t) * n_pad_d);
      int placed = dense != NULL && host_wd != NULL && host_bd != NULL;
      if (placed) {
        memset(dense, kernel_zero_point, (size_t) cout * cin);
        for (uint32_t g = 0; g < groups; g++) {
          for (uint32_t oc = 0; oc < group_output_channels; oc++) {
            memcpy(dense + ((size_t) g * group_output_channels + oc) * cin + (size_t) g * group_input_channels,
                   kernel + ((size_t) g * group_output_channels + oc) * group_input_channels, group_input_channels);
          }
        }
        qnnp_pack_igemm_w_slots(1, cout, 1, cin, cin, n_pad_d, k_pad_d, input_zero_point, kernel_zero_point, dense, bias,
            host_wd, host_bd);
        op->d_weights_dense = qnnp_hip_alloc(wd_bytes);
        op->d_bias_dense = qnnp_upload_bias_pair(host_bd, n_pad_d);
        placed = op->d_weights_dense != NULL && op->d_bias_dense != NULL &&
            qnnp_hip_h2d(op->d_weights_dense, host_wd, wd_bytes, 0) == QNNP_HIP_OK;
      }
      free(dense);
      free(host_wd);
      free(host_bd);
      if (placed) {
        op->dense_n_pad = n_pad_d;
        op->dense_k_pad = k_pad_d;
      } else {
        /* an optimisation, not a requirement: the operator keeps its grouped image */
        qnnp_log_warning("no room for %zu bytes of dense weights on the device: the operator keeps the grouped image", wd_bytes);
        qnnp_hip_free(op->d_weights_dense);
        qnnp_hip_free(op->d_bias_dense);
        op->d_weights_dense = NULL;
        op->d_bias_dense = NULL;
      }
    }
    /* Zero-point-centred image (pack.h qnnp_pack_igemm_w_centred127; hip/q8gemm256c.hip, the weight-stationary 3x3
     * kernel of hip/q8convwave.hip): single group, whole 32-deep K blocks, kernel zero point 128 (the standard image IS the centred
     * one) or 127 (a second image + bias pair). The [output channel][kh][kw][input channel] kernel tensor is the GEMM
     * layout the packer takes. */
    /* (round 6: hip/q8convws16s.hip -- 3x3 windows over 16 / 32 / 48 / 64 input channels, SqueezeNet's fire modules -- takes the image
     * WITH K padding: 9 x 16 and 9 x 48 bytes are not whole 32-byte blocks; the padding meets zero weights in either image) */
    const int small_3x3 = kernel_height == 3 && kernel_width == 3 && group_input_channels % 16 == 0 && group_input_channels <= 64 &&
        group_output_channels % 16 == 0 && group_output_channels <= 256;
    if (groups == 1 && kc_slot == (uint32_t) group_input_channels && (k_total % 32 == 0 || small_3x3)) {
      if (kernel_zero_point == 128) {
        op->centre_flip = 0x80;
      } else if (kernel_zero_point == 127 &&
                 /* (a second image only where a kernel takes it: the 3x3 weight-stationary kernel, the 256x256 GEMM) */
                 (small_3x3 ||
                  (kernel_size == 1 && k_total == k_pad && k_total >= 512 && n_pad % 256 == 0) ||
                  /* ... and the 128-wide tiling of hip/q8gemm128x.hip: any K % 64 == 0 */
                  (kernel_size == 1 && k_total == k_pad && k_total % 64 == 0 && group_output_channels % 4 == 0))) {
        int8_t* host_wc = (int8_t*) malloc(w_bytes);
        int32_t* host_bc = (int32_t*) malloc(b_bytes);
        int placed = host_wc != NULL && host_bc != NULL;
        if (placed) {
          qnnp_pack_igemm_w_centred127((uint32_t) group_output_channels, k_total, k_pad, n_pad, input_zero_point, kernel, bias,
              host_wc, host_bc);
          op->d_weights_centred = qnnp_hip_alloc(w_bytes);
          op->d_bias_centred = qnnp_upload_bias_pair(host_bc, n_pad);
          placed = op->d_weights_centred != NULL && op->d_bias_centred != NULL &&
              qnnp_hip_h2d(op->d_weights_centred, host_wc, w_bytes, 0) == QNNP_HIP_OK;
        }
        free(host_wc);
        free(host_bc);
        if (placed) {
          op->centre_flip = 0x7F;
        } else {
          /* the centred image is an optimisation, not a requirement: without it the operator runs on the standard
           * image (row term in the kernel) -- drop what was placed and carry on */
          qnnp_log_warning("no room for %zu bytes of centred weights on the device: the operator keeps the standard image", w_bytes + 2 * b_bytes);
          qnnp_hip_free(op->d_weights_centred);
          qnnp_hip_free(op->d_bias_centred);
          op->d_weights_centred = NULL;
          op->d_bias_centred = NULL;
        }
      }
    }
    if (kc_slot == 4 && kernel_height <= 4 && kernel_width * 3 <= 16 && dilation_height == 1 && dilation_width == 1 &&
        n_pad <= 64) {
      /* first layers: the row-slot image beside the tap-slot one (hip/q8convc3.hip takes it when pixels are dense) */
      const size_t r_bytes = (size_t) n_pad * 64;
      int8_t* host_rows = (int8_t*) malloc(r_bytes);
      if (host_rows == NULL) {
        qnnp_log_error("out of host memory: %zu bytes for packed weights", r_bytes);
        goto error;
      }
      int placed = 1;
      if (kernel_zero_point == 127) {
        /* centred on the kernel zero point: no row term in the kernel (pack.h) */
        int32_t* host_bc = (int32_t*) malloc((size_t) n_pad * sizeof(int32_t));
        placed = host_bc != NULL;
        if (placed) {
          qnnp_pack_conv_rows16_centred127((uint32_t) group_output_channels, kernel_height, kernel_width, 3, n_pad, input_zero_point,
              kernel, bias, host_rows, host_bc);
          op->d_bias_rows = qnnp_upload_bias_pair(host_bc, n_pad);
          placed = op->d_bias_rows != NULL;
        }
        free(host_bc);
      } else {
        qnnp_pack_conv_rows16((uint32_t) group_output_channels, kernel_height, kernel_width, 3, n_pad, kernel, host_rows);
      }
      op->d_weights_rows16 = qnnp_hip_alloc(r_bytes);
      placed = placed && op->d_weights_rows16 != NULL && qnnp_hip_h2d(op->d_weights_rows16, host_rows, r_bytes, 0) == QNNP_HIP_OK;
      free(host_rows);
      if (!placed) {
        qnnp_log_error("device allocation or upload failed: %zu bytes of packed weights on the device", r_bytes);
        goto error;
      }
    } else if (kc_slot == 4 && (kernel_height == 5 || kernel_height == 7) && kernel_width * 3 <= 32 &&
               dilation_height == 1 && dilation_width == 1 && n_pad <= (kernel_height == 7 ? 96u : 64u)) {
      /* ... and with 32-byte row slots for the larger windows (5x5, ResNet's 7x7 entry layer): same pointer, the window
       * decides which image it is (hip/q8convc3.hip) */
      const size_t r_bytes = qnnp_conv_rows32_size(n_pad, kernel_height);
      int8_t* host_rows = (int8_t*) malloc(r_bytes);
      if (host_rows == NULL) {
        qnnp_log_error("out of host memory: %zu bytes for packed weights", r_bytes);
        goto error;
      }
      int placed = 1;
      if (kernel_zero_point == 127) {
        /* centred on the kernel zero point: no row term in the kernel (pack.h) */
        int32_t* host_bc = (int32_t*) malloc((size_t) n_pad * sizeof(int32_t));
        placed = host_bc != NULL;
        if (placed) {
          qnnp_pack_conv_rows32_centred127((uint32_t) group_output_channels, kernel_height, kernel_width, 3, n_pad, input_zero_point,
              kernel, bias, host_rows, host_bc);
          op->d_bias_rows = qnnp_upload_bias_pair(host_bc, n_pad);
          placed = op->d_bias_rows != NULL;
        }
        free(host_bc);
      } else {
        qnnp_pack_conv_rows32((uint32_t) group_output_channels, kernel_height, kernel_width, 3, n_pad, kernel, host_rows);
      }
      op->d_weights_rows16 = qnnp_hip_alloc(r_bytes);
      placed = placed && op->d_weights_rows16 != NULL && qnnp_hip_h2d(op->d_weights_rows16, host_rows, r_bytes, 0) == QNNP_HIP_OK;
      free(host_rows);
      if (!placed) {
        qnnp_log_error("device allocation or upload failed: %zu bytes of packed weights on the device", r_bytes);
        goto error;
      }
    }
  }
  free(host_weights);
  free(host_bias);
  host_weights = NULL;
  host_bias = NULL;

  op->input_padding_top = input_padding_top;
  op->input_padding_right = input_padding_right;
  op->input_padding_bottom = input_padding_bottom;
  op->input_padding_left = input_padding_left;
  op->kernel_height = kernel_height;
  op->kernel_width = kernel_width;
  op->stride_height = subsampling_height;
  op->stride_width = subsampling_width;
  op->dilation_height = dilation_height;
  op->dilation_width = dilation_width;
  op->groups = groups;
  op->group_input_channels = group_input_channels;
  op->group_output_channels = group_output_channels;
  op->input_zero_point = input_zero_point;
  op->kernel_zero_point = kernel_zero_point;
  op->requant = qnnp_compute_requant(convolution_scale, output_zero_point, output_min, output_max);
  op->requant.accumulator_bits = qnnp_accumulator_bits(bias, (size_t) groups * group_output_channels,
      kernel_size * group_input_channels);
  op->ukernel_type = ukernel_type;

  /* reference convolution.c:372: the handle is written only on success */
  *convolution_out = op;
  return qnnp_status_success;

error:
  free(host_weights);
  free(host_bias);
  qnnp_delete_operator(op);
  return status;
}

static enum qnnp_status qnnp_setup_convolution2d_nhwc_q8_impl(
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
  (void) threadpool; /* unused by the reference's setup too (convolution.c:389) */

  /* reference convolution.c:391-394 */
  if (!qnnp_state.initialized) {
    qnnp_log_error("qnnp_setup_convolution2d_nhwc_q8 called before qnnp_initialize succeeded");
    return qnnp_status_uninitialized;
  }
  if (op == NULL || op->transposed || op->kernel_height == 0 ||
      (op->ukernel_type != qnnp_ukernel_type_conv && op->ukernel_type != qnnp_ukernel_type_dwconv &&
       op->ukernel_type != qnnp_ukernel_type_gemm)) {
    return qnnp_status_invalid_parameter;   /* not a handle from qnnp_create_convolution2d_nhwc_q8 */
  }

  /* reference convolution.c:396-399 */
  if (batch_size == 0) {
    op->batch_size = 0;
    return qnnp_status_success;
  }

  /* reference convolution.c:401-407 */
  if (input_width == 0 || input_height == 0) {
    qnnp_log_error("cannot set up convolution with %zux%zu input: input dimensions may not be zero",
        input_width, input_height);
    return qnnp_status_invalid_parameter;
  }
  const size_t in_channels = (size_t) op->groups * op->group_input_channels;
  const size_t out_channels = (size_t) op->groups * op->group_output_channels;
  if (input == NULL || output == NULL || input_pixel_stride < in_channels || output_pixel_stride < out_channels) {
    qnnp_log_error("cannot set up convolution: NULL tensor or pixel stride smaller than the channel count");
    return qnnp_status_invalid_parameter;
  }
  const size_t eff_kh = (size_t) (op->kernel_height - 1) * op->dilation_height + 1;
  const size_t eff_kw = (size_t) (op->kernel_width - 1) * op->dilation_width + 1;
  if (op->input_padding_top + input_height + op->input_padding_bottom < eff_kh ||
      op->input_padding_left + input_width + op->input_padding_right < eff_kw) {
    qnnp_log_error("cannot set up convolution with %zux%zu input: padded input is smaller than the dilated kernel",
        input_width, input_height);
    return qnnp_status_invalid_parameter;
  }

  /* reference convolution.c:409-426 */
  op->setup_valid = 0;   /* until every check, allocation and upload below has succeeded */
  op->residual = NULL;   /* an attached residual add belongs to the previous binding (residual.c) */
  op->dw_plan.key = 0;   /* depthwise launch plan: recomputed at the next run */
  op->batch_size = batch_size;
  op->input_height = input_height;
  op->input_width = input_width;
  op->input = input;
  op->input_pixel_stride = input_pixel_stride;
  op->output_height = compute_output_dimension(
      op->input_padding_top + input_height + op->input_padding_bottom,
      op->kernel_height, op->dilation_height, op->stride_height);
  op->output_width = compute_output_dimension(
      op->input_padding_left + input_width + op->input_padding_right,
      op->kernel_width, op->dilation_width, op->stride_width);
  op->output = output;
  op->output_pixel_stride = output_pixel_stride;

  const size_t output_size = op->output_height * op->output_width;
  const size_t input_size = input_height * input_width;
  if (batch_size * output_size > (size_t) UINT32_MAX / 2 || input_size * input_pixel_stride > (size_t) INT32_MAX) {
    qnnp_log_error("cannot set up convolution: %zu output pixels / %zu-byte images exceed the device kernels' index range",
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

  switch (op->ukernel_type) {
    case qnnp_ukernel_type_gemm:
      /* maps directly to GEMM, no table (reference convolution.c:429-431) */
      op->variant = qnnp_state.opt_gemm_kernel;
      return qnnp_status_success;
    case qnnp_ukernel_type_dwconv:
      /* the depthwise kernels derive tap coordinates arithmetically */
      op->variant = qnnp_state.opt_dwconv_kernel;
      return qnnp_status_success;
    case qnnp_ukernel_type_conv:
    {
      op->variant = qnnp_state.opt_gemm_kernel;
      const size_t kernel_size = (size_t) op->kernel_height * op->kernel_width;
      const size_t entries = output_size * kernel_size;
      const bool same_geometry = op->d_offsets != NULL &&
          op->offsets_in_h == input_height && op->offsets_in_w == input_width &&
          op->offsets_in_stride == input_pixel_stride;
      if (same_geometry) {
        return qnnp_status_success;  /* table is pointer- and batch-invariant */
      }
      int32_t* host_table = (int32_t*) malloc(sizeof(int32_t) * entries);
      if (host_table == NULL) {
        qnnp_log_error("out of host memory: %zu bytes for the offset table", sizeof(int32_t) * entries);
        return qnnp_status_out_of_memory;
      }
      if (op->offsets_capacity < entries) {
        qnnp_hip_free(op->d_offsets);
        op->offsets_capacity = 0;
        /* (+ 16 bytes: the 3-channel streaming kernel reads a lane's four consecutive entries with one 16-byte
         *  load, which for the last pixel's second K block starts on its 9th entry and runs past the table) */
        op->d_offsets = (int32_t*) qnnp_hip_alloc(sizeof(int32_t) * entries + 16);
        if (op->d_offsets == NULL) {
          free(host_table);
          qnnp_log_error("out of host memory: %zu bytes for the device offset table", sizeof(int32_t) * entries);
          return qnnp_status_out_of_memory;
        }
        op->offsets_capacity = entries;
      }
      qnnp_indirection_init_conv2d_offsets(op, host_table);
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
    default:
      return qnnp_status_invalid_parameter;
  }
}

/* ---- public entry points: run the implementation inside the right device context ------------------
 * create: the calling thread's selected device (qnnp_gfx950_set_device, default = the primary one) becomes the
 * operator's device; setup: the operator's device. The previous HIP device of the thread is restored on return. */

enum qnnp_status qnnp_create_convolution2d_nhwc_q8(
    uint32_t input_padding_top,
    uint32_t input_padding_right,
    uint32_t input_padding_bottom,
    uint32_t input_padding_left,
    uint32_t kernel_height,
    uint32_t kernel_width,
    uint32_t subsampling_height,
    uint32_t subsampling_width,
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
    qnnp_operator_t* convolution_out)
{
  if (!qnnp_state.initialized) {
    return qnnp_create_convolution2d_nhwc_q8_impl(input_padding_top, input_padding_right, input_padding_bottom, input_padding_left, kernel_height, kernel_width, subsampling_height, subsampling_width, dilation_height, dilation_width, groups, group_input_channels, group_output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias, output_zero_point, output_scale, output_min, output_max, flags, convolution_out);   /* logs and answers qnnp_status_uninitialized */
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
  const enum qnnp_status status = qnnp_create_convolution2d_nhwc_q8_impl(input_padding_top, input_padding_right, input_padding_bottom, input_padding_left, kernel_height, kernel_width, subsampling_height, subsampling_width, dilation_height, dilation_width, groups, group_input_channels, group_output_channels, input_zero_point, input_scale, kernel_zero_point, kernel_scale, kernel, bias, output_zero_point, output_scale, output_min, output_max, flags, convolution_out);
  qnnp_hip_leave(token);
  return status;
}

enum qnnp_status qnnp_setup_convolution2d_nhwc_q8(
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
    return qnnp_setup_convolution2d_nhwc_q8_impl(op, batch_size, input_height, input_width, input, input_pixel_stride, output, output_pixel_stride, threadpool);   /* answers qnnp_status_uninitialized / invalid_parameter */
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
  const enum qnnp_status status = qnnp_setup_convolution2d_nhwc_q8_impl(op, batch_size, input_height, input_width, input, input_pixel_stride, output, output_pixel_stride, threadpool);
  /* the implementation cleared setup_valid where it began to change the operator: a failed setup leaves it
   * unrunnable instead of half updated (run answers invalid_parameter) */
  if (status == qnnp_status_success) {
    op->setup_valid = 1;
  }
  qnnp_hip_leave(token);
  return status;
}
