/*
 * indirection.c -- builds the offset table that replaces the reference's
 * indirection buffer for qnnp_ukernel_type_conv.
 *
 * Reference: qnnp_indirection_init_conv2d (src/indirection.c:18-79) stores, for
 * every (group, image, output pixel, tap), an absolute host POINTER to the input
 * pixel, or a pointer to a zero-point-filled buffer for padding (:61-65); the
 * table is mr-tiled and sized 8 * batch * groups * round_up(OH*OW, mr) * taps
 * bytes (src/convolution.c:455), and must be rebuilt whenever the input pointer
 * changes.
 *
 * Here one table serves every image and group and survives pointer changes:
 *   table[pixel * taps + (ky * KW + kx)] = ((iy * W + ix) * input_pixel_stride)   in-bounds
 *                                        = QNNP_OFFSET_PADDING (-1)               padding
 * with iy = oy*stride_h + ky*dilation_h - pad_top (same index arithmetic and
 * unsigned wrap-around bounds test as src/indirection.c:56-60). The kernel adds
 * image * image_stride + group * group_input_channels + channel, and substitutes
 * the input zero point for padding entries. 4 * OH*OW * taps bytes in total.
 */
#include <stddef.h>
#include <stdint.h>

#include "indirection.h"

void qnnp_indirection_init_conv2d_offsets(const struct qnnp_operator* op, int32_t* table)
{
  const size_t input_height = op->input_height;
  const size_t input_width = op->input_width;
  const size_t output_height = op->output_height;
  const size_t output_width = op->output_width;
  const size_t kernel_height = op->kernel_height;
  const size_t kernel_width = op->kernel_width;
  const size_t taps = kernel_height * kernel_width;

  for (size_t oy = 0; oy < output_height; oy++) {
    for (size_t ox = 0; ox < output_width; ox++) {
      int32_t* entry = table + (oy * output_width + ox) * taps;
      for (size_t ky = 0; ky < kernel_height; ky++) {
        /* size_t arithmetic: a negative coordinate wraps to a huge value and fails `< extent` */
        const size_t iy = oy * op->stride_height + ky * op->dilation_height - op->input_padding_top;
        for (size_t kx = 0; kx < kernel_width; kx++) {
          const size_t ix = ox * op->stride_width + kx * op->dilation_width - op->input_padding_left;
          if (iy < input_height && ix < input_width) {
            entry[ky * kernel_width + kx] = (int32_t) ((iy * input_width + ix) * op->input_pixel_stride);
          } else {
            entry[ky * kernel_width + kx] = QNNP_OFFSET_PADDING;
          }
        }
      }
    }
  }
}

/*
 * Deconvolution: replaces qnnp_indirection_init_deconv2d (reference src/indirection.c:134-190). Output pixel
 * (oy, ox) receives tap (ky, kx) from input pixel (y / stride_h, x / stride_w) with
 * y = oy + pad_top - ky*dilation_h, x likewise, when both divisions are exact and the pixel exists
 * (:171-177, same size_t arithmetic: a negative y wraps to a huge value whose quotient fails `< extent`);
 * every other tap reads the zero point (:181) = a padding entry here.
 */
void qnnp_indirection_init_deconv2d_offsets(const struct qnnp_operator* op, int32_t* table)
{
  const size_t input_height = op->input_height;
  const size_t input_width = op->input_width;
  const size_t output_height = op->output_height;
  const size_t output_width = op->output_width;
  const size_t kernel_height = op->kernel_height;
  const size_t kernel_width = op->kernel_width;
  const size_t taps = kernel_height * kernel_width;

  for (size_t oy = 0; oy < output_height; oy++) {
    for (size_t ox = 0; ox < output_width; ox++) {
      int32_t* entry = table + (oy * output_width + ox) * taps;
      for (size_t ky = 0; ky < kernel_height; ky++) {
        const size_t y = oy + op->input_padding_top - ky * op->dilation_height;
        const size_t iy = y / op->stride_height;
        const int row_ok = iy * op->stride_height == y && iy < input_height;
        for (size_t kx = 0; kx < kernel_width; kx++) {
          const size_t x = ox + op->input_padding_left - kx * op->dilation_width;
          const size_t ix = x / op->stride_width;
          if (row_ok && ix * op->stride_width == x && ix < input_width) {
            entry[ky * kernel_width + kx] = (int32_t) ((iy * input_width + ix) * op->input_pixel_stride);
          } else {
            entry[ky * kernel_width + kx] = QNNP_OFFSET_PADDING;
          }
        }
      }
    }
  }
}
