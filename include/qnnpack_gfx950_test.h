/*
 * qnnpack_gfx950_test.h -- test and measurement hooks of libqnnpack_gfx950.so. NOT part of the product interface: nothing a
 * caller of include/qnnpack.h / include/qnnpack_gfx950.h needs, no stability promise for the codes. The GPU test tier and the
 * A/B tools force every kernel through the PRODUCT library with these (so that the library the driver records as loaded is the
 * one under test); operators set up afterwards keep the forced kernel, and a forced kernel refuses what it cannot take
 * (unsupported_parameter at run) instead of rerouting.
 */
#pragma once

#include <qnnpack.h>

#ifdef __cplusplus
extern "C" {
#endif

/* family / code:
 *   "gemm_kernel":   0 = auto, 1 = generic MFMA implicit-GEMM kernel, 2 = 256x256 LDS-DMA MFMA kernel,
 *                    3 = LDS-tiled direct-convolution MFMA kernel (convolutions only),
 *                    4 = the 256x256 kernel in its 4-wave flavour (one wave per SIMD, 128x128 per wave) [measurement
 *                        builds only since round 4, as 10, 11 and 16: structures that lost their A/B],
 *                    5 = barrier-free streaming kernel for short-K pointwise / fully-connected layers,
 *                    6 = its global-operand flavour (one wave per 32x32 block; small problems with long K),
 *                    7 = its 3-channel-image convolution flavour (first layers; in-register tap gather),
 *                    8 = wave-per-8x8-block direct-convolution MFMA kernel (small windows, <= 64 channels, dense output),
 *                    9 = long-K flavour of the streaming kernel (256 < K <= 1024, 16-byte aligned rows both sides:
 *                        a channel column's weights in LDS, every K block of a unit's rows in flight at once),
 *                    10 = 128x256 tiles of the LDS-DMA kernel, two workgroups per CU; 11 = its ping-pong schedule;
 *                    12 = the round-2 register-path flavour of kernel 8; 13 = stride-2 deconvolution streaming kernel
 *                    forced (1 keeps deconvolutions on the phase-table GEMMs); 14 = first-layer row-slot kernel;
 *                    15 = the lean flavour of kernel 2 (what auto picks when K % 64 == 0 and N % 256 == 0; 2 keeps the
 *                    general flavour), 16 = the lean flavour of kernel 4,
 *                    20 = the zero-point-centred 256x256 kernel (hip/q8gemm256c.hip: what auto picks for operators with
 *                    kernel zero point 127 or 128, K % 64 == 0, K >= 512, N % 256 == 0 -- until round 6), 21 = its A/B structure in
 *                    MEASUREMENT BUILDS ONLY (fragment reads in one burst),
 *                    23 = the centred kernel on v_mfma_i32_16x16x64_i8 (hip/q8gemm256x.hip, round 6: what auto picks now),
 *                    24 = its 128-row sibling for mid-size problems (hip/q8gemm128x.hip: any K % 64 == 0, two or three
 *                    workgroups per CU; tile width by the channel count), 25 / 26 = the same with 64- / 128-channel tiles,
 *                    27 = the 32x32x32 weight-stationary 3x3 convolution kernel for 64 input channels with a centred image (auto
 *                    takes its 16x16x64 flavour since round 6), 28 = the 16x16x64 GEMM with kernel-zero-point row sums (any kernel
 *                    zero point; what auto picks where 15 ran before),
 *                    29 = the 128-row GEMM for 1x1 / fully-connected shapes with NO alignment (hip/q8gemm128u.hip: grouped, odd channel
 *                    counts, unaligned pixels; what auto picks where 1 ran for them),
 *                    30 = the LDS-staged flavour of the 7x7 / 5x5 first-layer kernel (hip/q8convc3.hip; 14 keeps the register-path one),
 *                    31 = grouped 1x1 convolutions as ONE dense GEMM (block-diagonal weights, the kernel zero point off the diagonal:
 *                    convolution.c; auto takes it from 65536 rows up; the dense problem's kernel is chosen automatically),
 *                    32 = the weight-stationary 3x3 kernel for 16 / 32 / 48 / 64 input channels (hip/q8convws16s.hip; what auto picks for
 *                    SqueezeNet's fire modules).
 *   "fused_kernel":  fused inverted-residual blocks: 0 = auto (the strip kernel, hip/q8fusedstrip.hip, where it takes the
 *                    block -- kernel zero points 127 / 128 in all three members -- else the tile kernel of rounds 1-3),
 *                    1 = the tile kernel only, 2 = the strip kernel only (unsupported_parameter at setup otherwise)
 *   "fused_rows":    output rows per strip of the strip kernel, 0 = its planner's choice (tests, A/B)
 *   "fused_weights": 0 / 2 = a chunk's expand / project fragments fetched from L2 by the stage that multiplies them,
 *                    1 = staged in LDS one stage ahead by LDS-DMA where they fit (measured level: measurement builds only,
 *                    the product library ignores it)
 *   "dwconv_kernel": 0 = auto, 1 = generic direct kernel, 2 = LDS-tiled kernel, 3 = register sliding-window kernel (3x3),
 *                    4 = matrix-core kernel (diagonal MFMA operands; 3x3, channels % 16 == 0), tap operands gathered
 *                        from global memory, 5 = the same with the input band staged in LDS first,
 *                    6 = column-sliding register window (3x3, stride 1 | 2): tap pairs shared between output rows,
 *                    7 = matrix-core kernel on v_mfma_i32_16x16x64_i8 without data transposition (3x3, stride 1, channels % 16 == 0;
 *                        a negative result kept for its A/B: slower than 6 on every layer)
 *                    8 = the register sliding-window kernel (3) on UNALIGNED dwords: 3x3 windows, any channel count >= 4, any pixel
 *                        stride / base address (what auto picks where nothing aligned takes the shape), 9 = the four-channel generic
 *                        kernel kept for such shapes (A/B)
 *                    (1 keeps the byte-per-thread direct kernel; auto takes its four-channel flavour for C >= 4 and windows other than 3x3)
 * Unknown family or code -> invalid_parameter. 0 = the automatic choice, always. */
enum qnnp_status qnnp_gfx950_test_force_kernel(const char* family, int code);

/* 1: the operator's last run took the dense image of a grouped 1x1 convolution ("gemm_kernel" 31 / its automatic rule) */
int qnnp_gfx950_test_operator_ran_dense(qnnp_operator_t op);

#ifdef __cplusplus
}
#endif
