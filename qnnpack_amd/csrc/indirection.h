/*
 * indirection.h -- device-side indirection for general convolutions.
 * See indirection.c.
 */
#pragma once

#include <stdint.h>

#include "operator.h"

#define QNNP_OFFSET_PADDING (-1)

/* Fill table[pixel * taps + tap] for the geometry currently bound to `op`
 * (output_height/width, input_height/width, input_pixel_stride must be set). */
void qnnp_indirection_init_conv2d_offsets(const struct qnnp_operator* op, int32_t* table);

/* Same table for a deconvolution (transposed convolution): entry (output pixel, tap) is the input pixel
 * ((oy + pad_top - ky*dilation) / stride, ...) when that division is exact and in range, else padding. */
void qnnp_indirection_init_deconv2d_offsets(const struct qnnp_operator* op, int32_t* table);
