/*
 * qnnpack.h -- public C API of the gfx950 (MI355X) build of QNNPACK's uint8
 * convolution / GEMM hot path.
 *
 * This header is the DROP-IN BOUNDARY: every declaration below has the same
 * name, argument order, argument types and status-code values as the
 * reference's include/qnnpack.h, so a caller compiled against the reference
 * header links against libqnnpack_gfx950.so unchanged. Each prototype cites
 * the reference declaration it replaces (pytorch/QNNPACK tree).
 *
 * Scope (SURVEY.md sections 8b, 8f): the operators on the q8 conv/GEMM hot
 * path -- convolution2d_nhwc_q8 (which covers 1x1 "gemm", general "conv" and
 * depthwise "dwconv") and fully_connected_nc_q8 -- plus the rows ranked next:
 * deconvolution2d_nhwc_q8, add_nc_q8 and global_average_pooling_nwc_q8. The
 * other reference operators (windowed pooling, clamp, LUT ops, ...) are not
 * part of this library.
 *
 * Pointer contract specific to this build: `input` / `output` passed to
 * qnnp_setup_* may be either
 *   - device pointers (hipMalloc'ed on the library's device): used in place,
 *     zero copy -- the production path; or
 *   - ordinary host pointers: staged to/from device scratch inside
 *     qnnp_run_operator, so unmodified reference-style callers still work.
 * `kernel` / `bias` passed to qnnp_create_* are host pointers, copied at create.
 */
#pragma once

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

/* The reference header includes <pthreadpool.h> for this one type
 * (include/qnnpack.h:15). The gfx950 build ignores the thread pool, so the
 * dependency is reduced to the (identical) opaque typedef. */
#if defined(__has_include)
#  if __has_include(<pthreadpool.h>)
#    include <pthreadpool.h>
#    define QNNP_HAVE_PTHREADPOOL_H 1
#  endif
#endif
#ifndef QNNP_HAVE_PTHREADPOOL_H
typedef struct pthreadpool* pthreadpool_t;
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* reference include/qnnpack.h:24-32 -- identical enumerators and values */
enum qnnp_status {
  qnnp_status_success = 0,
  qnnp_status_uninitialized = 1,
  qnnp_status_invalid_parameter = 2,
  qnnp_status_unsupported_parameter = 3,
  qnnp_status_unsupported_hardware = 4,
  qnnp_status_out_of_memory = 5,
};

/* reference include/qnnpack.h:34. Binds the library to one gfx950 device.
 * Returns unsupported_hardware when no gfx950 GPU is usable -- there is no CPU
 * fallback. */
enum qnnp_status qnnp_initialize(void);

/* reference include/qnnpack.h:36 */
enum qnnp_status qnnp_deinitialize(void);

/* reference include/qnnpack.h:38 */
typedef struct qnnp_operator* qnnp_operator_t;

/*
 * Argument groups shared by the create functions below (same order and types as the reference):
 *   padding          top, right, bottom, left (uint32 each)
 *   window           kernel h, w; subsampling / stride h, w; dilation h, w (uint32 each)
 *   channels         groups (uint32), input channels per group, output channels per group (size_t)
 *   quantization     input zero point (uint8) + scale (float), kernel zero point + scale,
 *                    [kernel, bias: host pointers, copied], output zero point + scale, output min, max (uint8)
 *   flags            accepted and ignored, as in the reference
 *   last argument    receives the handle, written only on success
 */

/* reference include/qnnpack.h:40-65. kernel: [groups][group_output_channels][kh][kw][group_input_channels] uint8;
 * bias: [groups * group_output_channels] int32 */
enum qnnp_status qnnp_create_convolution2d_nhwc_q8(
    uint32_t input_padding_top, uint32_t input_padding_right, uint32_t input_padding_bottom, uint32_t input_padding_left,
    uint32_t kernel_height, uint32_t kernel_width, uint32_t subsampling_height, uint32_t subsampling_width,
    uint32_t dilation_height, uint32_t dilation_width,
    uint32_t groups, size_t group_input_channels, size_t group_output_channels,
    uint8_t input_zero_point, float input_scale, uint8_t kernel_zero_point, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias,
    uint8_t output_zero_point, float output_scale, uint8_t output_min, uint8_t output_max,
    uint32_t flags, qnnp_operator_t* convolution);

/* reference include/qnnpack.h:67-76. Strides are in bytes between pixels; `threadpool` is ignored. */
enum qnnp_status qnnp_setup_convolution2d_nhwc_q8(
    qnnp_operator_t convolution, size_t batch_size, size_t input_height, size_t input_width,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride, pthreadpool_t threadpool);

/* reference include/qnnpack.h:78-105. Transposed convolution:
 *   output extent = stride * (input - 1) + adjustment + (kernel - 1) * dilation + 1 - (padding before + after)
 * kernel: [groups][group_input_channels][kh][kw][group_output_channels] uint8
 *         (test/deconvolution-operator-tester.h:411 -- input channel OUTERMOST, unlike convolution)
 * bias:   [groups * group_output_channels] int32 */
enum qnnp_status qnnp_create_deconvolution2d_nhwc_q8(
    uint32_t input_padding_top, uint32_t input_padding_right, uint32_t input_padding_bottom, uint32_t input_padding_left,
    uint32_t adjustment_height, uint32_t adjustment_width,
    uint32_t kernel_height, uint32_t kernel_width, uint32_t stride_height, uint32_t stride_width,
    uint32_t dilation_height, uint32_t dilation_width,
    uint32_t groups, size_t group_input_channels, size_t group_output_channels,
    uint8_t input_zero_point, float input_scale, uint8_t kernel_zero_point, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias,
    uint8_t output_zero_point, float output_scale, uint8_t output_min, uint8_t output_max,
    uint32_t flags, qnnp_operator_t* deconvolution);

/* reference include/qnnpack.h:107-116 */
enum qnnp_status qnnp_setup_deconvolution2d_nhwc_q8(
    qnnp_operator_t deconvolution, size_t batch_size, size_t input_height, size_t input_width,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride, pthreadpool_t threadpool);

/* reference include/qnnpack.h:118-132. kernel: [output_channels][input_channels] uint8; bias: [output_channels] */
enum qnnp_status qnnp_create_fully_connected_nc_q8(
    size_t input_channels, size_t output_channels,
    uint8_t input_zero_point, float input_scale, uint8_t kernel_zero_point, float kernel_scale,
    const uint8_t* kernel, const int32_t* bias,
    uint8_t output_zero_point, float output_scale, uint8_t output_min, uint8_t output_max,
    uint32_t flags, qnnp_operator_t* fully_connected);

/* reference include/qnnpack.h:134-140. Strides are in bytes between rows. */
enum qnnp_status qnnp_setup_fully_connected_nc_q8(
    qnnp_operator_t fully_connected, size_t batch_size,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride);

/* reference include/qnnpack.h:142-151. Averages `width` pixels of `channels` bytes per image (NWC). */
enum qnnp_status qnnp_create_global_average_pooling_nwc_q8(
    size_t channels, uint8_t input_zero_point, float input_scale, uint8_t output_zero_point, float output_scale,
    uint8_t output_min, uint8_t output_max, uint32_t flags, qnnp_operator_t* global_average_pooling);

/* reference include/qnnpack.h:153-160. input_stride: bytes between pixels; output_stride: between images. */
enum qnnp_status qnnp_setup_global_average_pooling_nwc_q8(
    qnnp_operator_t global_average_pooling, size_t batch_size, size_t width,
    const uint8_t* input, size_t input_stride, uint8_t* output, size_t output_stride);

/* reference include/qnnpack.h:234-245. Quantized element-wise sum of two [batch][channels] tensors. */
enum qnnp_status qnnp_create_add_nc_q8(
    size_t channels, uint8_t a_zero_point, float a_scale, uint8_t b_zero_point, float b_scale,
    uint8_t sum_zero_point, float sum_scale, uint8_t sum_min, uint8_t sum_max,
    uint32_t flags, qnnp_operator_t* add);

/* reference include/qnnpack.h:247-255 */
enum qnnp_status qnnp_setup_add_nc_q8(
    qnnp_operator_t add, size_t batch_size,
    const uint8_t* a, size_t a_stride, const uint8_t* b, size_t b_stride, uint8_t* sum, size_t sum_stride);

/* reference include/qnnpack.h:327-329. `threadpool` is accepted and ignored: the launch covers the whole operator.
 * Synchronous by default (outputs complete on return); see qnnpack_gfx950.h for the asynchronous mode. */
enum qnnp_status qnnp_run_operator(qnnp_operator_t op, pthreadpool_t threadpool);

/* reference include/qnnpack.h:331-332. NULL -> invalid_parameter (src/operator-delete.c:17-19). */
enum qnnp_status qnnp_delete_operator(qnnp_operator_t op);

#ifdef __cplusplus
} /* extern "C" */
#endif
