/*
 * qnnp_hip.h -- the C-ABI seam between the plain-C host code (init.c,
 * convolution.c, fully-connected.c, operator-run.c, operator-delete.c) and the
 * gfx950 HIP translation units. Host C never includes a HIP header: everything
 * that crosses this seam is POD (sizes, strides, raw pointers, fixed-width ints).
 *
 * It replaces the reference's microkernel function-pointer table
 * (src/qnnpack/params.h:267-378, 520-538) and the pthreadpool fan-out in
 * src/operator-run.c:675, 797, 837: one call here = one whole-operator launch.
 *
 * Return convention: 0 on success, a negative QNNP_HIP_E* code otherwise.
 */
#pragma once

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QNNP_HIP_OK 0
#define QNNP_HIP_ENODEV (-1)   /* no usable gfx950 device */
#define QNNP_HIP_ENOMEM (-2)   /* hipMalloc failed */
#define QNNP_HIP_ELAUNCH (-3)  /* kernel launch / runtime error */
#define QNNP_HIP_EINVAL (-4)   /* argument the kernels cannot serve */

/* Fused fixed-point down-convert parameters: the scalar member of
 * union qnnp_conv_quantization_params (src/qnnpack/params.h:128-138) minus the
 * two zero points, which this build folds into the packed bias / row term. */
struct qnnp_hip_requant {
  int32_t multiplier;           /* [0x40000000, 0x7FFFFF80] */
  int32_t remainder_mask;       /* (1 << shift) - 1 */
  int32_t remainder_threshold;  /* remainder_mask >> 1 */
  uint32_t shift;               /* [0, 31] */
  int32_t output_min_less_zero_point;
  int32_t output_max_less_zero_point;
  int32_t output_zero_point;
  uint32_t accumulator_bits;    /* |accumulator| < 2^accumulator_bits for every input (host-side bound from |bias| and the
                                 * reduction length), or 0 = unknown: lets the device pick a cheaper rounding sequence */
};

/* ---- runtime (runtime.hip) ----------------------------------------------
 * One context per gfx950 device (stream, asynchrony flag, fill table, properties). qnnp_hip_init binds the
 * PRIMARY device; qnnp_hip_bind adds others. Every call below acts on the calling thread's ACTIVE context:
 * the one entered with qnnp_hip_enter (operators enter their own device), else the thread's selected device
 * (qnnp_hip_select), else the primary. Thread-local: selection, active context, hipGraph capture. */
int qnnp_hip_init(int device /* <0: the calling thread's current HIP device */);
int qnnp_hip_bind(int device);           /* bind one more device after init (idempotent) */
int qnnp_hip_shutdown(void);
int qnnp_hip_device_count(void);
int qnnp_hip_select(int device);         /* the calling thread's library device from now on (must be bound) */
int qnnp_hip_device(void);               /* active context's device ordinal, -1 if none */
/* enter: make `device`'s context active for this thread and its HIP device current; returns a token (< 0: the
 * device is not bound). leave(token) restores the previous context and the previous current HIP device. */
int qnnp_hip_enter(int device);
void qnnp_hip_leave(int token);
int qnnp_hip_device_info(char* arch, size_t arch_len, int* cus, int* clock_khz, size_t* mem_bytes);
int qnnp_hip_compute_units(void);
void qnnp_hip_set_streaming_stores(int on);   /* see qnnp_gfx950_set_option("streaming_stores") */
int qnnp_hip_streaming_stores(void);
void qnnp_hip_set_stream(void* stream);
void* qnnp_hip_get_stream(void);         /* the stream a launch of this thread goes to (its capture stream while recording) */
void qnnp_hip_set_async(int async);
int qnnp_hip_get_async(void);
int qnnp_hip_stream_sync(void);
/* device table [256][16]: entry v = sixteen bytes of value v (constant LDS-DMA sources) */
const uint8_t* qnnp_hip_fill_table(void);
#ifdef QNNP_ENABLE_ABLATION
/* measurement builds: device buffer for in-kernel cycle stamps (NULL unless env QNNP_GFX950_TRACE is set) and its dump */
void* qnnp_hip_trace_buffer(void);
int qnnp_hip_trace_dump(unsigned long long* host, size_t count);
#endif

void* qnnp_hip_alloc(size_t bytes);
void qnnp_hip_free(void* p);
/* async = 0: complete on return and ordered behind the work already enqueued on the library stream */
int qnnp_hip_h2d(void* dst, const void* src, size_t bytes, int async);
int qnnp_hip_d2h(void* dst, const void* src, size_t bytes, int async);
int qnnp_hip_memset(void* dst, int value, size_t bytes);
/* 1 = memory of the active device (or managed), 0 = host memory, -1 = memory of another device */
int qnnp_hip_is_device_pointer(const void* p);

/* hipEvent-based timing on the library stream (for qnnp_gfx950_time_operator) */
int qnnp_hip_timer_create(void** timer);
int qnnp_hip_timer_start(void* timer);
int qnnp_hip_timer_stop_ms(void* timer, float* ms);
void qnnp_hip_timer_destroy(void* timer);

/* hipGraph capture of the calling thread's operator launches (a private stream stands in for the default
 * stream, which cannot be captured); replay = one submission, no per-launch gaps */
int qnnp_hip_graph_capturing(void);
int qnnp_hip_graph_begin(void);
int qnnp_hip_graph_end(void** graph);
int qnnp_hip_graph_device(void* graph);
int qnnp_hip_graph_launch(void* graph);
int qnnp_hip_graph_time(void* graph, int warmup, int iters, float* avg_ms);
/* `samples` event-bracketed batches of `iters` replays; the median batch / iters */
int qnnp_hip_graph_time_median(void* graph, int warmup, int iters, int samples, float* avg_ms);
int qnnp_hip_graph_sync(void* graph);
void qnnp_hip_graph_destroy(void* graph);

/* ---- q8 GEMM / implicit-GEMM convolution (MFMA) ------------------------
 * Replaces q8gemm_ukernel_4x4c2__sse2 (src/q8gemm/4x4c2-sse2.c:14-318) and
 * q8conv_ukernel_4x4c2__sse2 (src/q8conv/4x4c2-sse2.c:14-273) together with
 * their tilers compute_q8gemm / compute_q8conv (src/operator-run.c:39-70, 183-217).
 *
 * For each group g, row m in [0, rows) and column n in [0, n):
 *   out[m*output_stride + g*n_ + n] = requant( bias2[g*n_pad + n]
 *        + row_coeff * sum_k a'(m,k) + sum_k a'(m,k) * w'(g,n,k) )
 * with a' = a - 128, w' = w - 128 (both valid int8), k flattened as
 * tap * group_input_channels + channel, and a(m,k) read from
 *   gemm : input[m*input_stride + g*kc + k]
 *   conv : input[img*image_stride + offsets[pix*ks + tap] + g*kc + ch], or the
 *          input zero point where offsets[...] < 0 (padding)   (img = m / rows_per_image).
 */
/* One entry of an optional device-resident phase table (deconvolution.c): the launch covers `nphases` independent
 * implicit GEMMs that share input, output, channel counts and requantization but have their own packed weights,
 * folded bias, offset table, output-pixel list and row count (blockIdx.y = phase * groups + group). */
struct qnnp_hip_igemm_phase {
  const int8_t* packed_w;
  const int32_t* bias2;
  const int32_t* offsets;
  const int32_t* out_rows;
  uint32_t rows;              /* batch * rows_per_image */
  uint32_t rows_per_image;
  uint32_t ks;
  uint32_t k_total;
  uint32_t k_pad;
  uint32_t reserved;
};

struct qnnp_hip_add_params;
struct qnnp_hip_igemm_args {
  const uint8_t* input;
  uint8_t* output;
  const int8_t* packed_w;     /* MFMA-fragment panels, see pack.h */
  const int8_t* packed_w_rows16; /* 3-channel first layers: the row-slot image (pack.h qnnp_pack_conv_rows16), or NULL */
  const int32_t* bias2_rows;  /* NULL, or: the row-slot image is centred on kernel zero point 127 (element 127 - w; the kernel re-centres the
                               * activations with ^ 0x7F and needs no row term) and this is its bias pair table [2][n_pad] */
  const int32_t* bias2;       /* [groups][n_pad] */
  const int32_t* offsets;     /* conv: [rows_per_image][ks]; NULL for gemm */
  uint32_t rows;              /* batch * output pixels */
  uint32_t rows_per_image;    /* output pixels per image (gemm: rows) */
  uint64_t image_stride;      /* bytes between consecutive input images (conv) */
  uint32_t groups;
  uint32_t n;                 /* group output channels */
  uint32_t n_pad;             /* round_up(n, 32) */
  uint32_t kc;                /* group input channels */
  uint32_t kc_slot;           /* K positions per tap: kc, or 4 when kc == 3 (one unaligned dword fetch per tap) */
  uint64_t input_bytes;       /* extent of the input tensor from `input` (bounds the 4-byte fetches of kc == 3) */
  uint32_t ks;                /* taps (1 for gemm) */
  uint32_t k_total;           /* ks * kc_slot */
  uint32_t k_pad;             /* round_up(k_total, 64) */
  uint32_t input_stride;      /* bytes between pixels */
  uint32_t output_stride;
  int32_t row_coeff;          /* 128 - kernel_zero_point */
  uint32_t input_zero_point;
  struct qnnp_hip_requant rq;
  int variant;                /* 0 auto, 1 generic, 2 big-tile LDS-DMA, 3 LDS-tiled direct convolution */
  /* optional scatter of the GEMM rows (conv + variant 1 only; deconvolution phases): row m of image img is written to
   * output pixel img * out_image_rows + out_rows[m % rows_per_image] instead of pixel m. NULL = rows are pixels. */
  const int32_t* out_rows;
  uint32_t out_image_rows;
  /* optional phase table (device memory, conv + variant 1 only): see struct qnnp_hip_igemm_phase. With it, `rows`
   * is the LARGEST phase's row count (grid sizing), `k_pad` the largest k_pad, and the per-phase fields above are
   * ignored. */
  const struct qnnp_hip_igemm_phase* phases;
  uint32_t nphases;
  /* optional depth-to-space stores (gemm form + the streaming kernel only; deconvolution with kernel == stride):
   * n_pad = stride_h*stride_w*round_up(n, 32) packed columns, phase-major; row m = input pixel (img, iy, ix) writes
   * its phase (py, px) block to output pixel (img, iy*stride_h + py, ix*stride_w + px). d2s_stride_h == 0: off. */
  uint32_t d2s_stride_h, d2s_stride_w, d2s_input_h, d2s_input_w;
  /* convolution geometry (conv only; lets the LDS-tiled kernel address the input directly) */
  uint32_t input_height, input_width, output_height, output_width;
  uint32_t kernel_height, kernel_width, stride_height, stride_width, dilation_height, dilation_width;
  uint32_t pad_top, pad_left;
  /* optional fused residual add (qnnpack_gfx950.h qnnp_gfx950_attach_residual_add):
   * output = add(a = residual pixel, b = the requantized convolution output) with the add operator's parameters.
   * Kernels that carry the add in their epilogue set *residual_folded = 1; for the others the caller launches the
   * stand-alone add kernel in place on `output` (operator-run.c). residual == NULL: off. */
  const uint8_t* residual;
  uint32_t residual_stride;   /* bytes between residual pixels */
  const struct qnnp_hip_add_params* residual_add;
  uint32_t* residual_folded;
  /* 1: `bias2` is a pair table (bias-pair.h) -- groups * n_pad values followed by the same values + 2^31, which the
   * kernels that use the lane forms of the requantization start their accumulators from. 0: those kernels keep the
   * offset forms. */
  uint32_t bias2_pair;
  /* optional zero-point-centred weight image (pack.h qnnp_pack_igemm_w_centred; q8gemm256c.hip): same fragment layout
   * as packed_w with w'' = (w ^ centre_flip) as int8, and its folded bias as a pair table (bias-pair.h). centre_flip =
   * 0x80 (kernel zero point 128: the two pointers may simply repeat packed_w / bias2) or 0x7F (127); 0 = none. */
  const int8_t* packed_w_centred;
  const int32_t* bias2_centred;
  uint32_t centre_flip;
  /* the streaming-store hint of THIS launch (qnnp_gfx950_operator_set_streaming_stores): 0 = the process default
   * ("streaming_stores" option), 1 = off, 2 = on */
  uint32_t streaming_mode;
};
int qnnp_hip_igemm_run(const struct qnnp_hip_igemm_args* args, const char** kernel_name);

/* ---- q8 deconvolution, stride 2, 3x3 / 4x4 kernels (q8deconv.hip) --------------------------------------
 * One streaming kernel over the input pixels in place of the phase-table GEMMs above (same packed per-phase
 * sub-kernels and folded biases, deconvolution.c; phase = py * 2 + px with py = (oy + pad_top) % 2).
 * QNNP_HIP_EINVAL = outside the kernel's range (channels % 32, channels > 128, weights beyond LDS, alignment):
 * the caller keeps the phase-table path. */
struct qnnp_hip_deconv_s2_args {
  const uint8_t* input;
  uint8_t* output;
  const int8_t* packed_w[4];
  const int32_t* bias2[4];
  uint32_t k_pad[4];
  uint32_t batch, input_height, input_width, output_height, output_width;
  uint32_t kernel_height, kernel_width, pad_top, pad_left;
  uint32_t channels;          /* input channels (one group) */
  uint32_t n, n_pad;          /* output channels, round_up(n, 32) */
  uint32_t input_stride, output_stride;
  int32_t row_coeff;          /* 128 - kernel_zero_point */
  uint32_t input_zero_point;
  struct qnnp_hip_requant rq;
  uint32_t bias2_pair;        /* 1: every bias2[ph] is a pair table (bias-pair.h): n_pad values, then the same + 2^31 */
};
int qnnp_hip_deconv_s2_run(const struct qnnp_hip_deconv_s2_args* args, const char** kernel_name);

/* ---- q8 depthwise convolution ------------------------------------------
 * Replaces q8dwconv_ukernel_up8x9__sse2 (src/q8dwconv/up8x9-sse2.c:14-372) and
 * q8dwconv_ukernel_mp8x25__sse2 (src/q8dwconv/mp8x25-sse2.c:14-742) plus
 * compute_dwconv_unipass/multiipass (src/operator-run.c:238-284).
 *   out[c] = requant( bias1[c] + sum_taps a(tap,c) * wadj[tap][c] )
 * wadj = w - kernel_zero_point (int16), bias1 = bias + taps*izp*kzp - izp*sum w
 * (the reference's own folding, src/qnnpack/pack.h:146-159); padding taps read
 * a = input_zero_point.
 */
/* Launch plan of a depthwise operator (kernel choice + band / slab geometry). It depends on what setup fixes, on the
 * variant and on the tensors' alignment only, so the operator keeps it: computed at the first run after a setup
 * (setup zeroes `key`), reused by every later run (the reference's run path plans nothing either, src/operator-run.c:647-710). */
struct qnnp_hip_dwconv_plan {
  uint32_t key;               /* 0 = not computed; else the (alignment, variant, batch) signature it is valid for */
  uint32_t kernel;
  uint32_t CS, TOH, IR, IC, PP, bands, slabs, store_mode, vec16;
};

struct qnnp_hip_dwconv_args {
  const uint8_t* input;
  uint8_t* output;
  const int16_t* wadj;        /* [taps][c_pad], tap = ky*kw + kx */
  const int32_t* bias1;       /* [c_pad] */
  const int8_t* dwm_x;        /* [3][taps][c_pad32] int8 weight parts for the MFMA kernel (pack.h) */
  const int32_t* dwm_bias;    /* [c_pad32] */
  uint32_t dwm_parts;         /* 1..3 parts in use */
  uint32_t w_range;           /* qnnp_dwconv_weight_range of wadj (pack.h): 0 unknown / neither, 1, 2 */
  const uint32_t* dot4;       /* w_range 1 / 2: [4][c_pad] qnnp_pack_dwconv_dot4 image (3x3) or [8][c_pad] qnnp_pack_dwconv_dot4_5x5 (5x5), else NULL */
  uint32_t c_pad32;
  uint32_t batch;
  uint32_t input_height, input_width;
  uint32_t output_height, output_width;
  uint32_t channels, c_pad;   /* c_pad = round_up(channels, 16) */
  uint32_t kernel_height, kernel_width;
  uint32_t stride_height, stride_width;
  uint32_t dilation_height, dilation_width;
  uint32_t pad_top, pad_left;
  uint32_t input_stride, output_stride;
  uint32_t input_zero_point;
  struct qnnp_hip_requant rq;
  int variant;                /* 0 auto, 1 generic direct, 2 LDS-tiled, 3 register sliding window (3x3), 4 matrix-core (gather), 5 matrix-core (LDS band), 6 column-sliding window (3x3) */
  struct qnnp_hip_dwconv_plan* plan;   /* optional plan cache owned by the caller (NULL: plan on every call) */
  uint32_t streaming_mode;    /* as qnnp_hip_igemm_args: 0 = process default, 1 = off, 2 = on */
};
int qnnp_hip_dwconv_run(const struct qnnp_hip_dwconv_args* args, const char** kernel_name);

/* ---- q8 element-wise add and global average pooling ("next" row 4) --------------
 * HBM-bound byte kernels (q8pointwise.hip).
 *
 * add: replaces q8vadd_ukernel__sse2 (src/q8vadd/sse2.c) + the qnnp_ukernel_type_add case of
 * qnnp_run_operator (src/operator-run.c). Normative arithmetic = qnnp_add_quantize
 * (src/qnnpack/requantization.h:500-522; the microkernel tester asserts equality with it,
 * test/vadd-microkernel-tester.h:180,194):
 *   acc = zero_point_product + a*a_multiplier + b*b_multiplier            (32-bit wrap-around)
 *   acc = asr(acc, shift) + (((acc & mask) - (acc < 0)) > (mask >> 1))
 *   sum = min(max(acc + y_zero_point, y_min), y_max)
 */
struct qnnp_hip_add_params {
  uint32_t a_multiplier, b_multiplier;
  int32_t zero_point_product;
  uint32_t shift;
  int32_t remainder_mask, remainder_threshold;
  int32_t y_zero_point, y_min, y_max;
};
struct qnnp_hip_vadd_args {
  const uint8_t* a;
  const uint8_t* b;
  uint8_t* sum;
  uint64_t rows;
  uint32_t channels;
  uint64_t a_stride, b_stride, sum_stride;
  struct qnnp_hip_add_params params;
  uint32_t streaming_mode;    /* as qnnp_hip_igemm_args: 0 = process default, 1 = off, 2 = on */
};
int qnnp_hip_vadd_run(const struct qnnp_hip_vadd_args* args, const char** kernel_name);

/* global average pooling: replaces q8gavgpool_ukernel_{up8x7,mp8x7p7q,up8xm}__sse2 (src/q8gavgpool/) + the
 * qnnp_ukernel_type_global_average_pooling case (src/operator-run.c:981-1016). For image i and channel c:
 *   n   = bias + sum_w input[(i*width + w)*input_stride + c]              bias = -width * input_zero_point
 *   out = qnnp_avgpool_quantize(n)  (src/qnnpack/requantization.h:482-498; asserted equal by
 *         test/gavgpool-microkernel-tester.h:177,198):
 *         n = asr64(n*multiplier - (n < 0) + rounding, right_shift); clamp to [min, max] - zp; + zp
 */
struct qnnp_hip_avgpool_params {
  int32_t bias;
  int32_t multiplier;
  int64_t rounding;
  uint32_t right_shift;
  int32_t output_min_less_zero_point, output_max_less_zero_point, output_zero_point;
};
struct qnnp_hip_gavgpool_args {
  const uint8_t* input;
  uint8_t* output;
  uint64_t batch;
  uint64_t width;             /* pixels averaged per image */
  uint32_t channels;
  uint64_t input_stride;      /* bytes between pixels */
  uint64_t output_stride;     /* bytes between images */
  struct qnnp_hip_avgpool_params params;
};
int qnnp_hip_gavgpool_run(const struct qnnp_hip_gavgpool_args* args, const char** kernel_name);

/* ---- fused inverted-residual block ---------------------------------------
 * [pointwise expand ->] depthwise 3x3 (pad 1, stride 1 | 2) -> pointwise project [-> + block input], one launch,
 * the expanded tensors live only in LDS (q8fused.hip). Arithmetic per stage is that of the stand-alone operators
 * (each intermediate is requantized to uint8 exactly as if it had been stored), so the result is bit-identical to
 * running them one after another. Weights / biases are the device images the stand-alone operators packed.
 */
struct qnnp_hip_fused_args {
  const uint8_t* input;
  uint8_t* output;
  uint32_t batch, input_height, input_width, output_height, output_width;
  uint32_t input_channels, hidden_channels, output_channels;
  uint32_t input_stride, output_stride;       /* bytes between pixels */
  uint32_t stride;                            /* of the depthwise stage */
  /* expand (has_expand == 0: the depthwise stage reads the input, hidden_channels == input_channels) */
  uint32_t has_expand;
  const int8_t* expand_w; const int32_t* expand_bias2; uint32_t expand_k_pad; uint32_t expand_n_pad;
  int32_t expand_row_coeff; struct qnnp_hip_requant expand_rq;
  /* depthwise */
  const int16_t* dw_wadj; const int32_t* dw_bias1; uint32_t dw_c_pad; uint32_t dw_input_zero_point;
  struct qnnp_hip_requant dw_rq;
  /* project */
  const int8_t* project_w; const int32_t* project_bias2; uint32_t project_k_pad; uint32_t project_n_pad;
  int32_t project_row_coeff; struct qnnp_hip_requant project_rq;
  /* residual: output = add(a = block input, b = project output) */
  uint32_t has_residual;
  struct qnnp_hip_add_params add;
};
/* QNNP_HIP_EINVAL when the block does not fit the kernel (LDS, channel multiples) -- callers keep the unfused form */
int qnnp_hip_fused_block_run(const struct qnnp_hip_fused_args* args, const char** kernel_name);
int qnnp_hip_fused_block_supported(const struct qnnp_hip_fused_args* args);

/* ---- the same block, strip kernel (q8fusedstrip.hip, round 4) ------------------------------------------------------
 * One workgroup of 16 waves owns a STRIP of output rows of one image (the whole image where it is small) and walks the
 * hidden channels in chunks: expand (MFMA) -> requantize -> LDS, depthwise as MFMAs against diagonal weight fragments
 * built in registers -> requantize -> LDS, project accumulated over the chunks (MFMA) -> requantize [-> + input] ->
 * global. All three stages use zero-point-CENTRED int8 images (as q8gemm256c.hip: element w ^ flip with flip = 0x7F for
 * kernel zero point 127, 0x80 for 128; activations recentred with the same mask), so no stage has a row term; the
 * folded biases carry both zero points and, where the stage's rounding sequence is an offset form, the 2^31.
 * Built by fused-block.c at create time from the stand-alone operators' device images. */
struct qnnp_hip_fused_strip_args {
  const uint8_t* input;
  uint8_t* output;
  uint32_t batch, input_height, input_width, output_height, output_width;
  uint32_t input_channels, hidden_channels, output_channels;
  uint32_t input_stride, output_stride;       /* bytes between pixels */
  uint32_t stride;                            /* of the depthwise stage */
  uint32_t has_expand, has_residual;
  /* expand: fragments [hidden blocks][input blocks] of 1 KiB (lane l: n = l & 31, k = (l >> 5) * 16 + j), bias [hidden_pad] */
  const int8_t* expand_w; const int32_t* expand_bias; uint32_t expand_flip; struct qnnp_hip_requant expand_rq;
  /* depthwise: int8 [9][hidden_pad], bias [hidden_pad]; dw_pad = (input zero point ^ flip) byte of the padding taps */
  const int8_t* dw_w; const int32_t* dw_bias; uint32_t dw_flip; uint32_t dw_pad; struct qnnp_hip_requant dw_rq;
  /* project: fragments [output blocks][hidden blocks], bias [output_pad] */
  const int8_t* project_w; const int32_t* project_bias; uint32_t project_flip; struct qnnp_hip_requant project_rq;
  uint32_t hidden_pad, output_pad;            /* channel counts rounded up to 32 */
  struct qnnp_hip_add_params add;             /* residual: output = add(a = block input, b = project output) */
  uint32_t rows_per_strip;                    /* 0 = the kernel's own choice; else forced (A/B, tests) */
  uint32_t weights_in_lds;                    /* 0 = the kernel's own choice (where they fit); 1 = only plans that stage the chunk's
                                               * expand / project fragments in LDS; 2 = fragments from L2 (round-4 first form; A/B, tests) */
};
/* 1: the stage's folded bias must carry + 2^31 (its rounding sequence is an offset form); host-side twin of the kernel's choice */
int qnnp_hip_fused_strip_bias_offset(const struct qnnp_hip_requant* rq);
int qnnp_hip_fused_strip_supported(const struct qnnp_hip_fused_strip_args* args);
int qnnp_hip_fused_strip_run(const struct qnnp_hip_fused_strip_args* args, const char** kernel_name);

#ifdef __cplusplus
}
#endif
