/*
 * igemm_params.h -- kernel-argument block shared by the MFMA GEMM / implicit-GEMM kernels
 * (q8igemm.hip: generic kernel + dispatch; q8gemm256.hip: 256x256 LDS-DMA kernel).
 */
#pragma once

#include <hip/hip_runtime.h>

#include <stdint.h>

#include "qnnp_hip.h"
#include "requant.hip.h"

namespace qnnp {

// cycle stamp of lane 0 / wave 0 into trace[(block * items + item) * 8 + slot]; compiled out of the product
#ifdef QNNP_ENABLE_ABLATION
#define QNNP_TRACE(p, block, item, slot)                                                        \
  do {                                                                                           \
    if ((p).trace != nullptr && threadIdx.x == 0 && (item) < 4)                                  \
      (p).trace[((block) * 4 + (item)) * 8 + (slot)] = __builtin_readcyclecounter();            \
  } while (0)
/* same, but the constant 100 MHz counter (s_memrealtime): calibrates the shader clock of a traced run */
#define QNNP_TRACE_WALL(p, block, item, slot)                                                   \
  do {                                                                                           \
    if ((p).trace != nullptr && threadIdx.x == 0 && (item) < 4)                                  \
      (p).trace[((block) * 4 + (item)) * 8 + (slot)] = wall_clock64();                           \
  } while (0)
/* per-wave stamp: lane 0 of the wave whose index is `item` */
#define QNNP_TRACE_WAVE(p, block, item, slot)                                                   \
  do {                                                                                           \
    if ((p).trace != nullptr && (threadIdx.x & 63u) == 0 && (item) < 4)                          \
      (p).trace[((block) * 4 + (item)) * 8 + (slot)] = __builtin_readcyclecounter();            \
  } while (0)
#else
#define QNNP_TRACE_WAVE(p, block, item, slot) do { } while (0)
#define QNNP_TRACE(p, block, item, slot) do { } while (0)
#define QNNP_TRACE_WALL(p, block, item, slot) do { } while (0)
#endif

struct IgemmParams {
  const uint8_t* input;
  uint8_t* output;
  const int8_t* packed_w;
  const int32_t* bias2;
  const int32_t* offsets;
  uint32_t rows;
  uint32_t rows_per_image;
  uint64_t image_stride;
  uint32_t n;
  uint32_t n_pad;
  uint32_t kc;             // K positions per tap as the kernel sees them (= kc_slot)
  const uint8_t* input_end; // one past the last readable input byte (pad3 mode bound)
  uint32_t ks;
  uint32_t k_total;
  uint32_t k_pad;
  uint32_t input_stride;
  uint32_t output_stride;
  int32_t row_coeff;
  uint32_t izp_fill;       // input zero point replicated into 4 bytes
  uint32_t store_mode;     // 2: 16-byte stores, 1: dword stores, 0: byte stores (igemm_epilogue.hip.h)
  uint32_t cu_count;       // compute units of the bound device (persistent-grid sizing)
  unsigned long long* trace;  // measurement builds only (QNNP_ENABLE_ABLATION + env QNNP_GFX950_TRACE): cycle stamps
  const uint8_t* fill_table; // [256][16]: entry v = 16 bytes of value v (LDS-DMA padding sources)
  const int32_t* out_rows;   // optional (generic kernel): output pixel of GEMM row `pix` inside its image, else NULL
  uint32_t out_image_rows;   // with out_rows: output pixels per image
  const qnnp_hip_igemm_phase* phases;  // optional (generic kernel): blockIdx.y = phase * phase_groups + group
  uint32_t phase_groups;
  // depth-to-space stores of the pointwise streaming kernel (deconvolution with kernel == stride): GEMM row m is
  // input pixel (img, iy, ix); its channel block nb belongs to phase nb / d2s_nbpp = (py, px) and is stored at output
  // pixel (img, iy*d2s_sh + py, ix*d2s_sw + px), channels (nb % d2s_nbpp)*32.. of p.n. d2s_sh == 0: off.
  uint32_t d2s_sh, d2s_sw, d2s_in_h, d2s_in_w, d2s_nbpp;
  RequantDev rq;
  // fused residual add (pointwise streaming kernels only, their RES flavours): the requantized bytes of pixel m are
  // summed, as operand b of qnnp_add_quantize, with the bytes at residual + m*residual_stride (operand a) before they
  // are stored. NULL: off. (Behind everything else: the kernels that never look at it keep their argument offsets.)
  const uint8_t* residual;
  uint32_t residual_stride;
  qnnp_hip_add_params add;
  // lane forms of the requantization (requant.hip.h; the kernels instantiated for kRqShift0Lane / kRqBoundedLane only)
  qnnp_requant_lane lane;
  const int32_t* bias2u;     // bias2 + 2^31, laid out like bias2 (bias-pair.h), or NULL (then lane.kind == 0)
  uint32_t stream_out;       // 1: whole-line stores that write a line exactly once carry the streaming hint ("streaming_stores")
  uint32_t a_flip;           // q8gemm256c.hip only: the activation recentring mask, 0x80808080 (image centred on 128) or
                             // 0x7F7F7F7F (on 127); 0 = the operator has no centred image
  uint32_t tiles_n_magic;    // q8gemm256c.hip only: floor(2^32 / tiles_n) + 1, so that x / tiles_n == hi32(x * magic) for the
                             // tile ids of a launch (x * tiles_n < 2^32); set by gemm256c_launch
  // strided 1x1 convolutions on the staged pointwise streaming kernel (round 5): `offsets` holds one VALID entry per
  // output pixel of an image (ks == 1, no tap on padding); row m reads input + (m / rows_per_image) * image_stride +
  // offsets[m % rows_per_image]. rpi_magic = floor(2^32 / rows_per_image) + 1 when rows_per_image > 1 and rows * rows_per_image < 2^32 (then
  // m / rows_per_image == hi32(m * magic)), else 0 = divide. offsets_dense == 0: the kernel never looks at `offsets`.
  uint32_t offsets_dense;
  uint32_t rpi_magic;
};

/* convolution geometry for the LDS-tiled direct-convolution kernel */
struct ConvGeom {
  uint32_t H, W, OH, OW;
  uint32_t KH, KW, sh, sw, dh, dw;
  uint32_t pad_top, pad_left;
};

/* q8convlds.hip */
bool convlds_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec);
int convlds_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name);

/* q8convwave.hip */
bool convwave_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch);
int convwave_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name, int flavour,
                    const IgemmParams* centred = nullptr);

/* q8convpatch.hip: dense 3x3 with 64 .. 512 input channels and output channels in multiples of 128 */
bool convpatch_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch);
int convpatch_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name);

/* q8pwconv.hip */
bool pwstream_supported(const IgemmParams& p, uint32_t groups, uint32_t vec);
int pwstream_launch(const IgemmParams& p, uint32_t vec, hipStream_t stream, const char** name);
bool convstream_c3_supported(const IgemmParams& p, uint32_t groups);
int convstream_c3_launch(const IgemmParams& p, hipStream_t stream, const char** name);
bool pwstream_longk_supported(const IgemmParams& p, uint32_t groups, uint32_t vec);
int pwstream_longk_launch(const IgemmParams& p, hipStream_t stream, const char** name);
bool pwstream_gw_supported(const IgemmParams& p, uint32_t groups, uint32_t vec);
int pwstream_gw_launch(const IgemmParams& p, hipStream_t stream, const char** name);

/* q8convc3.hip */
bool conv_c3rows_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, const int8_t* w_rows16, uint32_t real_kc);
int conv_c3rows_launch(const IgemmParams& p, const ConvGeom& g, const int8_t* w_rows16, hipStream_t stream, const char** name, int flavour);
/* q8convws16s.hip: dense 3x3 / stride 1 with 16 / 32 / 48 / 64 input channels, weights in registers (p: the centred parameters) */
bool convws16s_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, uint32_t vec, uint32_t batch);
int convws16s_launch(const IgemmParams& p, const ConvGeom& g, uint32_t batch, hipStream_t stream, const char** name);
bool conv_c3rows32_supported(const IgemmParams& p, const ConvGeom& g, uint32_t groups, const int8_t* w_rows32, uint32_t real_kc);
int conv_c3rows32_launch(const IgemmParams& p, const ConvGeom& g, const int8_t* w_rows32, hipStream_t stream, const char** name, int flavour);

/* q8gemm256c.hip: the zero-point-centred flavour (p carries the centred image, its bias pair table and a_flip) */
bool gemm256c_supported(const IgemmParams& p, uint32_t vec);
int gemm256c_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, uint32_t opt);
/* q8gemm256x.hip: the same GEMM on v_mfma_i32_16x16x64_i8 (round 6); takes what gemm256c_supported accepts */
int gemm256x_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name);
/* q8gemm128u.hip: the same matrix side for ANY channel counts / groups / alignment (activations staged by the threads, standard image) */
bool gemm128u_supported(const IgemmParams& p);
int gemm128u_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, uint32_t tile_n);
/* q8gemm128x.hip: 128 x 128 tiles of four waves on the same operand path -- mid-size problems, any K tile count */
bool gemm128x_supported(const IgemmParams& p, uint32_t vec);
int gemm128x_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, uint32_t tile_n);

/* q8gemm256.hip */
bool gemm256_supported(const IgemmParams& p, uint32_t vec);
int gemm256_launch(const IgemmParams& p, uint32_t groups, hipStream_t stream, const char** name, bool waves4, bool rows128, bool pingpong, int lean);   /* lean: 0 never, 1 where supported, 2 forced (EINVAL otherwise) */

}  // namespace qnnp
