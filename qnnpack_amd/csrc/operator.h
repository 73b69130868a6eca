/*
 * operator.h -- the opaque `struct qnnp_operator` behind qnnp_operator_t.
 *
 * Role-equivalent to the reference's src/qnnpack/operator.h:39-102, but laid out
 * for the gfx950 build: the host-side packed weights / pointer indirection
 * buffer / zero buffer of the reference become device allocations (an MFMA
 * fragment-panel weight image, an int32 offset table, folded int32 bias).
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <qnnpack.h>

#include "hip/qnnp_hip.h"

/* subset of reference enum qnnp_ukernel_type (src/qnnpack/operator.h:23-37) */
enum qnnp_ukernel_type {
  qnnp_ukernel_type_none = 0,
  qnnp_ukernel_type_conv,
  qnnp_ukernel_type_dwconv,
  qnnp_ukernel_type_gemm,
  qnnp_ukernel_type_add,
  qnnp_ukernel_type_global_average_pooling,
  qnnp_ukernel_type_fused_block,
};

/* One output phase of a strided deconvolution (deconvolution.c): the output pixels whose (oy + pad_top) % stride_h
 * and (ox + pad_left) % stride_w are (py, px) see the same sub-kernel, so each phase is a dense implicit GEMM. */
#define QNNP_MAX_DECONV_PHASES 16
struct qnnp_deconv_phase {
  void* d_weights;          /* packed sub-kernel (MFMA fragment panels) */
  int32_t* d_bias;          /* bias folded for the sub-kernel's taps */
  uint32_t k_pad;
  uint32_t taps;            /* sub-kernel taps (>= 1; a phase without taps gets one all-padding tap) */
  uint8_t tap_ky[64], tap_kx[64];
  int32_t* d_offsets;       /* [rows][taps] */
  int32_t* d_out_rows;      /* [rows] output pixel inside the image */
  size_t rows;              /* output pixels of this phase per image (0: phase absent for this geometry) */
  size_t rows_capacity;
};

struct qnnp_operator {
  /* geometry fixed at create (reference operator.h:40-57) */
  size_t batch_size;
  uint32_t input_padding_top;
  uint32_t input_padding_right;
  uint32_t input_padding_bottom;
  uint32_t input_padding_left;
  uint32_t kernel_height;
  uint32_t kernel_width;
  uint32_t stride_height;
  uint32_t stride_width;
  uint32_t dilation_height;
  uint32_t dilation_width;
  uint32_t groups;
  size_t group_input_channels;
  size_t group_output_channels;
  uint32_t adjustment_height;  /* deconvolution only (reference operator.h:45-46) */
  uint32_t adjustment_width;
  int transposed;              /* 1: deconvolution -- the offset table is the transposed-convolution one */
  uint32_t deconv_phases;      /* 0: one table over all taps; else stride_h * stride_w phase GEMMs */
  void* d_phase_table;         /* device array of struct qnnp_hip_igemm_phase, one per non-empty phase (setup) */
  uint32_t phase_table_entries;
  size_t phase_max_rows;       /* largest phase, rows per image */
  uint32_t phase_max_k_pad;
  int deconv_d2s;              /* 1: kernel == stride, no padding: also packed as ONE pointwise GEMM with
                                * depth-to-space stores (d_weights / d_bias / n_pad / k_pad), tried first at run */
  int deconv_stream;           /* stride-2 streaming kernel (q8deconv.hip): 0 auto, 1 never, 2 forced ("gemm_kernel" 1 / 13 at setup) */
  struct qnnp_deconv_phase phase[QNNP_MAX_DECONV_PHASES];

  /* bound at setup (reference operator.h:59-73): caller-owned, not copied */
  size_t input_height;
  size_t input_width;
  size_t input_pixel_stride;
  const void* input;
  size_t output_height;
  size_t output_width;
  size_t output_pixel_stride;
  void* output;

  /* fused inverted-residual block (fused-block.c): BORROWED handles of the stand-alone operators it was built from */
  const struct qnnp_operator* fused_expand;     /* may be NULL */
  const struct qnnp_operator* fused_depthwise;
  const struct qnnp_operator* fused_project;
  const struct qnnp_operator* fused_add;        /* may be NULL */
  /* the strip kernel's own images (hip/q8fusedstrip.hip), derived at create from the members' device images when every
   * kernel zero point is 127 or 128: ONE device allocation, owned */
  void* d_strip;                                /* NULL: the block only has the tile kernel (hip/q8fused.hip) */
  size_t strip_w1, strip_b1, strip_w2, strip_b2, strip_w3, strip_b3;   /* byte offsets into d_strip */
  uint32_t strip_hidden_pad, strip_output_pad;
  uint8_t strip_flip1, strip_flip2, strip_flip3, strip_dw_pad;
  int fused_use_strip;                          /* decided at setup */
  uint32_t fused_rows_per_strip;                /* "fused_rows" option at setup: 0 = the kernel's choice */
  uint32_t fused_weights;                       /* "fused_weights" option at setup (hip/qnnp_hip.h weights_in_lds) */

  /* residual add attached to a convolution (residual.c, qnnp_gfx950_attach_residual_add): the operator then writes
   * add(a = residual pixel, b = convolution output) -- in the convolution kernel's epilogue where that kernel carries
   * it, else by an in-place launch of the add kernel behind it. Caller-owned device memory; cleared by the next
   * convolution setup. */
  const void* residual;
  size_t residual_pixel_stride;
  struct qnnp_hip_add_params residual_params;
  uint32_t residual_folded;   /* last launch: 1 = in the epilogue, 0 = separate add launch */

  /* add / global average pooling (reference operator.h:58, 66-67, 75-100) */
  size_t channels;
  const void* input2;
  size_t input2_pixel_stride;
  uint8_t output_zero_point;
  uint8_t output_min;
  uint8_t output_max;
  float input_scale;
  float output_scale;
  struct qnnp_hip_add_params add_params;
  struct qnnp_hip_avgpool_params avgpool_params;

  uint8_t input_zero_point;
  uint8_t kernel_zero_point;
  struct qnnp_hip_requant requant;
  enum qnnp_ukernel_type ukernel_type;

  /* ---- device-side state owned by the operator ---- */
  void* d_weights;        /* igemm: int8 fragment panels; dwconv: int16 [taps][c_pad] */
  void* d_weights_rows16; /* igemm, 3-channel first layers: the [ky][16-byte row slot] fragment image (pack.h), or NULL */
  /* grouped 1x1 convolutions (ukernel gemm, groups > 1): the same operator as ONE dense GEMM -- the groups' weight blocks on the diagonal of
   * a [groups * output channels][groups * input channels] matrix whose other elements are the kernel zero point (w - kzp = 0: bit for
   * bit the grouped result) -- packed like any single-group image; operator-run.c takes it for many rows (convolution.c) */
  void* d_weights_dense;
  int32_t* d_bias_dense;  /* bias pair table [2][dense_n_pad] */
  uint32_t dense_n_pad, dense_k_pad;
  uint32_t ran_dense;     /* 1: the last run took the dense image (include/qnnpack_gfx950_test.h) */
  int32_t* d_bias_rows;   /* its bias pair table when that image is the zero-point-centred one (32-byte slots, kernel zero point 127:
                           * pack.h qnnp_pack_conv_rows32_centred127), else NULL: the image goes with d_bias and the row term */
  int32_t* d_bias;        /* igemm: bias2 [groups][n_pad]; dwconv: bias1 [c_pad] */
  uint32_t streaming_mode;  /* streaming-store hint of this operator's launches: 0 = the process default, 1 = off, 2 = on
                             * (qnnp_gfx950_operator_set_streaming_stores) */
  /* zero-point-centred image for the big GEMM kernel (pack.h qnnp_pack_igemm_w_centred, hip/q8gemm256c.hip):
   * centre_flip 0 = none; 0x80 = kernel zero point 128, the standard image IS centred (the two pointers stay NULL and
   * d_weights / d_bias serve); 0x7F = kernel zero point 127, own image + own bias pair table */
  void* d_weights_centred;
  int32_t* d_bias_centred;
  uint32_t centre_flip;
  uint32_t n_pad;         /* igemm */
  uint32_t k_pad;         /* igemm */
  uint32_t kc_slot;       /* igemm: K positions per tap (group_input_channels, or 4 for 3-channel inputs) */
  uint32_t c_pad;         /* dwconv */
  void* d_dwm_x;          /* dwconv, MFMA kernel: int8 [3][taps][c_pad32] weight parts (pack.h) */
  int32_t* d_dwm_bias;    /* dwconv, MFMA kernel: int32 [c_pad32] */
  uint32_t dwm_parts;     /* 1..3 */
  void* d_dw_dot4;        /* dwconv 3x3 / 5x5, dw_wrange != 0: uint32 [4 | 8][c_pad] register image of the int8 dot-product walk (pack.h) */
  uint32_t dw_wrange;     /* qnnp_dwconv_weight_range (pack.h): 0, 1 (w - kzp fits int8), 2 (kzp - w does) */
  uint32_t c_pad32;

  int32_t* d_offsets;     /* conv: [output pixels][taps] int32, -1 = padding */
  size_t offsets_capacity;     /* in entries */
  size_t offsets_in_h, offsets_in_w, offsets_in_stride;  /* geometry the table was built for */

  /* host-pointer staging (only used when setup() received host memory) */
  int input_on_device;
  int output_on_device;
  void* d_stage_in;
  size_t stage_in_capacity;
  void* d_stage_out;
  size_t stage_out_capacity;
  int input2_on_device;   /* add: the second operand */
  void* d_stage_in2;
  size_t stage_in2_capacity;
  size_t input2_span;
  size_t input_span;      /* bytes of caller input touched by the operator */
  size_t output_span;     /* bytes of caller output the operator may write */

  int variant;            /* kernel-variant option captured at setup */
  const char* kernel_name;

  int device;             /* HIP ordinal of the GPU that owns every d_* allocation above: create runs on the calling
                           * thread's selected device, setup / run / delete enter this one (runtime.hip) */
  int setup_valid;        /* 1 after a successful setup; cleared where a setup starts to change the operator, so a
                           * failed setup cannot be run against half-updated geometry / tables */
  struct qnnp_hip_dwconv_plan dw_plan;   /* depthwise launch plan, computed at the first run after a setup */
};

/* Decide where a caller pointer lives and (re)size the device staging buffer a host pointer needs:
 * success; out_of_memory (staging); invalid_parameter (device memory of a GPU other than the operator's).
 * (operator-run.c) */
enum qnnp_status qnnp_bind_endpoint(const void* ptr, size_t span, int* on_device, void** stage, size_t* capacity);

/* fused-block.c */
int qnnp_fused_block_launch(struct qnnp_operator* op, const void* input, void* output);
