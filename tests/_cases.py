"""Shared test-case matrices and deterministic data generators.

The matrices re-host the reference's own operator test lists:
  test/convolution.cc (49 cases; the 7 xzp_* cases are ARM-only -- kthreshold is
  SIZE_MAX off-ARM, src/init.c:194-196 -- and are represented by their plain 1x1 twins),
  test/fully-connected.cc (11 cases).
Data follows the reference testers (uniform uint8, bias in [-10000, 10000],
zero points 127/127, output scale/zero point derived from the accumulator range:
test/convolution-operator-tester.h:345-413) but with FIXED seeds
(the reference uses std::random_device).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field, replace
from typing import Tuple

import numpy as np


@dataclass(frozen=True)
class ConvCase:
    name: str
    input_size: Tuple[int, int]                 # (height, width)
    kernel_size: Tuple[int, int] = (1, 1)       # (height, width)
    padding: Tuple[int, int, int, int] = (0, 0, 0, 0)   # top, right, bottom, left
    subsampling: Tuple[int, int] = (1, 1)       # (height, width)
    dilation: Tuple[int, int] = (1, 1)
    groups: int = 1
    gic: int = 1
    goc: int = 1
    batch: int = 1
    input_pixel_stride: int = 0                 # 0 -> groups * gic
    output_pixel_stride: int = 0                # 0 -> groups * goc
    qmin: int = 0
    qmax: int = 255
    izp: int = 127
    kzp: int = 127

    @property
    def in_stride(self):
        return self.input_pixel_stride or self.groups * self.gic

    @property
    def out_stride(self):
        return self.output_pixel_stride or self.groups * self.goc


def _pad(h=0, w=0, top=None, right=None, bottom=None, left=None):
    t, r, b, l = h, w, h, w
    if top is not None: t = top
    if right is not None: r = right
    if bottom is not None: b = bottom
    if left is not None: l = left
    return (t, r, b, l)


# test/convolution.cc, in file order
CONV_CASES = [
    ConvCase("zero_batch", (5, 5), (1, 1), gic=2, goc=2, batch=0),
    ConvCase("1x1", (27, 29), gic=23, goc=19),
    ConvCase("1x1_with_qmin", (27, 29), gic=23, goc=19, qmin=128),
    ConvCase("1x1_with_qmax", (27, 29), gic=23, goc=19, qmax=128),
    ConvCase("1x1_with_input_stride", (27, 29), gic=23, goc=19, input_pixel_stride=28),
    ConvCase("1x1_with_output_stride", (27, 29), gic=23, goc=19, output_pixel_stride=29),
    ConvCase("1x1_with_batch", (13, 14), gic=23, goc=19, batch=3),
    ConvCase("grouped_1x1", (24, 25), groups=2, gic=17, goc=19),
] + [
    # test/convolution.cc:103-198, the seven xzp_* cases: group input channels = q8conv_xzp.kthreshold + 1, where the
    # threshold is what src/init.c:66-80 sets per ARM core (64, 256, 32, 16; SIZE_MAX -- "never" -- elsewhere). The XZP
    # kernels are ARM-only; the shapes are plain API cases and run here for every threshold.
    case for kt in (16, 32, 64, 256) for case in (
        ConvCase(f"xzp_1x1_kt{kt}", (27, 29), gic=kt + 1, goc=19),
        ConvCase(f"xzp_1x1_with_qmin_kt{kt}", (27, 29), gic=kt + 1, goc=19, qmin=128),
        ConvCase(f"xzp_1x1_with_qmax_kt{kt}", (27, 29), gic=kt + 1, goc=19, qmax=128),
        ConvCase(f"xzp_1x1_with_input_stride_kt{kt}", (27, 29), gic=kt + 1, goc=19, input_pixel_stride=kt + 5),
        ConvCase(f"xzp_1x1_with_output_stride_kt{kt}", (27, 29), gic=kt + 1, goc=19, output_pixel_stride=29),
        ConvCase(f"xzp_1x1_with_batch_kt{kt}", (13, 14), gic=kt + 1, goc=19, batch=3),
        ConvCase(f"grouped_xzp_1x1_kt{kt}", (24, 25), groups=2, gic=kt + 1, goc=19),
    )
] + [
    ConvCase("1x3", (20, 19), (1, 3), _pad(w=1), gic=17, goc=15),
    ConvCase("grouped_1x3", (20, 19), (1, 3), _pad(w=1), groups=2, gic=17, goc=15),
    ConvCase("3x1", (19, 20), (3, 1), _pad(h=1), gic=17, goc=15),
    ConvCase("grouped_3x1", (19, 20), (3, 1), _pad(h=1), groups=2, gic=17, goc=15),
    ConvCase("3x3", (13, 12), (3, 3), _pad(1, 1), gic=15, goc=17),
    ConvCase("3x3_without_padding", (13, 12), (3, 3), gic=15, goc=17),
    ConvCase("3x3_with_left_padding", (13, 12), (3, 3), _pad(left=1), gic=15, goc=17),
    ConvCase("3x3_with_right_padding", (13, 12), (3, 3), _pad(right=1), gic=15, goc=17),
    ConvCase("3x3_with_top_padding", (13, 12), (3, 3), _pad(top=1), gic=15, goc=17),
    ConvCase("3x3_with_bottom_padding", (13, 12), (3, 3), _pad(bottom=1), gic=15, goc=17),
    ConvCase("3x3_with_input_stride", (13, 12), (3, 3), _pad(1, 1), gic=15, goc=17, input_pixel_stride=22),
    ConvCase("3x3_with_output_stride", (13, 12), (3, 3), _pad(1, 1), gic=15, goc=17, output_pixel_stride=23),
    ConvCase("3x3_with_batch", (10, 9), (3, 3), _pad(1, 1), gic=15, goc=17, batch=3),
    ConvCase("grouped_3x3", (10, 11), (3, 3), _pad(1, 1), groups=2, gic=14, goc=13),
    ConvCase("3x3s2", (19, 21), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=27, goc=19),
    ConvCase("3x3s1x2", (13, 13), (3, 3), _pad(1, 1), subsampling=(1, 2), gic=27, goc=19),
    ConvCase("3x3s2x1", (13, 13), (3, 3), _pad(1, 1), subsampling=(2, 1), gic=27, goc=19),
    ConvCase("3x3d2", (13, 14), (3, 3), _pad(2, 2), dilation=(2, 2), gic=27, goc=19),
    ConvCase("3x3d1x2", (14, 15), (3, 3), _pad(1, 2), dilation=(1, 2), gic=27, goc=19),
    ConvCase("3x3d2x1", (15, 14), (3, 3), _pad(2, 1), dilation=(2, 1), gic=27, goc=19),
    ConvCase("depthwise_3x3", (15, 14), (3, 3), _pad(1, 1), groups=27),
    ConvCase("depthwise_3x3s2", (15, 14), (3, 3), _pad(1, 1), subsampling=(2, 2), groups=27),
    ConvCase("depthwise_3x3s1x2", (15, 14), (3, 3), _pad(1, 1), subsampling=(1, 2), groups=27),
    ConvCase("depthwise_3x3s2x1", (15, 14), (3, 3), _pad(1, 1), subsampling=(2, 1), groups=27),
    ConvCase("depthwise_3x3d2", (15, 14), (3, 3), _pad(1, 1), dilation=(2, 2), groups=27),
    ConvCase("depthwise_3x3d1x2", (15, 14), (3, 3), _pad(1, 1), dilation=(1, 2), groups=27),
    ConvCase("depthwise_3x3d2x1", (15, 14), (3, 3), _pad(1, 1), dilation=(2, 1), groups=27),
    ConvCase("depthwise_5x5", (15, 14), (5, 5), _pad(2, 2), groups=27),
    ConvCase("depthwise_5x5s2", (15, 14), (5, 5), _pad(2, 2), subsampling=(2, 2), groups=27),
    ConvCase("depthwise_5x5s1x2", (15, 14), (5, 5), _pad(2, 2), subsampling=(1, 2), groups=27),
    ConvCase("depthwise_5x5s2x1", (15, 14), (5, 5), _pad(2, 2), subsampling=(2, 1), groups=27),
    ConvCase("depthwise_5x5d2", (15, 14), (5, 5), _pad(2, 2), dilation=(2, 2), groups=27),
    ConvCase("depthwise_5x5d1x2", (15, 14), (5, 5), _pad(2, 2), dilation=(1, 2), groups=27),
    ConvCase("depthwise_5x5d2x1", (15, 14), (5, 5), _pad(2, 2), dilation=(2, 1), groups=27),
]

# Shapes beyond the reference list that exercise the device kernels' fast paths
# (16/8/4-byte activation vectors, dword stores, every tile shape, LDS depthwise).
EXTRA_CONV_CASES = [
    ConvCase("x_1x1_k64_n64_vec16", (9, 11), gic=64, goc=64, batch=2),
    ConvCase("x_1x1_k24_n144_vec8", (8, 7), gic=24, goc=144, batch=2),
    ConvCase("x_1x1_k20_n36_vec4", (8, 7), gic=20, goc=36),
    ConvCase("x_1x1_k144_n24", (7, 9), gic=144, goc=24, batch=3),
    ConvCase("x_1x1_k320_n200", (5, 5), gic=320, goc=200, batch=2),
    ConvCase("x_1x1_zp_0_255", (6, 7), gic=32, goc=32, izp=0, kzp=255),
    ConvCase("x_1x1_zp_255_0", (6, 7), gic=32, goc=32, izp=255, kzp=0),
    ConvCase("x_1x1_zp_128_128", (6, 7), gic=48, goc=40, izp=128, kzp=128),
    ConvCase("x_grouped_1x1_k16", (6, 7), groups=3, gic=16, goc=8),
    ConvCase("x_3x3_c64_vec16", (12, 10), (3, 3), _pad(1, 1), gic=64, goc=64, batch=2),
    ConvCase("x_3x3_c16_s2", (15, 17), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=16, goc=48),
    ConvCase("x_3x3_c3_first_layer", (32, 32), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=2),
    ConvCase("x_3x3_c3_s1_pad", (13, 11), (3, 3), _pad(1, 1), gic=3, goc=24, batch=2),
    ConvCase("x_3x3_c3_pixel_stride5", (9, 10), (3, 3), _pad(1, 1), gic=3, goc=40, input_pixel_stride=5),
    ConvCase("x_5x5_c3_s2_zp", (17, 15), (5, 5), _pad(2, 2), subsampling=(2, 2), gic=3, goc=16, izp=9, kzp=200),
    ConvCase("x_1x1_c3_s2", (9, 9), (1, 1), subsampling=(2, 2), gic=3, goc=33),
    ConvCase("x_3x3_c8_vec8", (9, 9), (3, 3), _pad(1, 1), gic=8, goc=20),
    ConvCase("x_5x5_c4", (11, 12), (5, 5), _pad(2, 2), gic=4, goc=12),
    ConvCase("x_7x7_dw_c12", (13, 13), (7, 7), _pad(3, 3), groups=12),
    ConvCase("x_1x1_dw_c20", (6, 5), (1, 1), groups=20),
    ConvCase("x_dw3x3_c32", (16, 16), (3, 3), _pad(1, 1), groups=32, batch=2),
    ConvCase("x_dw3x3_c96_s2", (17, 19), (3, 3), _pad(1, 1), subsampling=(2, 2), groups=96, batch=2),
    ConvCase("x_dw3x3_c144", (14, 14), (3, 3), _pad(1, 1), groups=144),
    ConvCase("x_dw3x3_c20_vec4", (9, 10), (3, 3), _pad(1, 1), groups=20, batch=2),
    ConvCase("x_dw3x3_c960_7x7", (7, 7), (3, 3), _pad(1, 1), groups=960, batch=2),
    ConvCase("x_dw3x3_c32_strided", (10, 9), (3, 3), _pad(1, 1), groups=32, input_pixel_stride=48, output_pixel_stride=36),
    ConvCase("x_dw5x5_c64", (12, 13), (5, 5), _pad(2, 2), groups=64),
    ConvCase("x_dw3x3_c64_d2", (13, 12), (3, 3), _pad(2, 2), dilation=(2, 2), groups=64),
    ConvCase("x_dw3x3_c32_qmin_qmax", (9, 9), (3, 3), _pad(1, 1), groups=32, qmin=64, qmax=192),
    ConvCase("x_dw3x3_c64_zp", (9, 9), (3, 3), _pad(1, 1), groups=64, izp=3, kzp=250),
]


@dataclass(frozen=True)
class FcCase:
    name: str
    batch: int
    input_channels: int
    output_channels: int
    input_stride: int = 0
    output_stride: int = 0
    qmin: int = 0
    qmax: int = 255
    izp: int = 127
    kzp: int = 127

    @property
    def in_stride(self):
        return self.input_stride or self.input_channels

    @property
    def out_stride(self):
        return self.output_stride or self.output_channels


# test/fully-connected.cc, in file order
FC_CASES = [
    FcCase("zero_batch", 0, 2, 2),
    FcCase("unit_batch", 1, 23, 19),
    FcCase("unit_batch_with_qmin", 1, 23, 19, qmin=128),
    FcCase("unit_batch_with_qmax", 1, 23, 19, qmax=128),
    FcCase("unit_batch_with_input_stride", 1, 23, 19, input_stride=28),
    FcCase("unit_batch_with_output_stride", 1, 23, 19, output_stride=29),
    FcCase("small_batch", 12, 23, 19),
    FcCase("small_batch_with_qmin", 12, 23, 19, qmin=128),
    FcCase("small_batch_with_qmax", 12, 23, 19, qmax=128),
    FcCase("small_batch_with_input_stride", 12, 23, 19, input_stride=28),
    FcCase("small_batch_with_output_stride", 12, 23, 19, output_stride=29),
]

EXTRA_FC_CASES = [
    FcCase("x_c1_plumbing_1x1024x1000", 1, 1024, 1000),      # BASELINE.json configs[0]
    FcCase("x_m300_k256_n256", 300, 256, 256),
    FcCase("x_m129_k72_n33", 129, 72, 33),
    FcCase("x_m64_k1024_n1000", 64, 1024, 1000),
    FcCase("x_m257_k100_n260_strided", 257, 100, 260, input_stride=112, output_stride=264),
]


def seed_for(name: str) -> int:
    return 0x51A0 ^ (zlib.crc32(name.encode()) & 0x7FFFFFFF)


def conv_tensors(case: ConvCase):
    """Seeded input / kernel / bias in the layouts of the C API.

    input: flat uint8 buffer of the strided NHWC tensor (as the tester allocates it,
    test/convolution-operator-tester.h:350); kernel [g][oc][kh][kw][ic]; bias [g*oc].
    """
    rng = np.random.default_rng(seed_for(case.name))
    H, W = case.input_size
    pixels = case.batch * H * W
    in_len = max(pixels - 1, 0) * case.in_stride + case.groups * case.gic if pixels else 0
    inp = rng.integers(0, 256, size=in_len, dtype=np.uint8)
    kernel = rng.integers(0, 256, size=(case.groups, case.goc, case.kernel_size[0], case.kernel_size[1], case.gic),
                          dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=case.groups * case.goc, dtype=np.int32)
    return inp, kernel, bias


def fc_tensors(case: FcCase):
    rng = np.random.default_rng(seed_for(case.name))
    in_len = max(case.batch - 1, 0) * case.in_stride + case.input_channels if case.batch else 0
    inp = rng.integers(0, 256, size=in_len, dtype=np.uint8)
    kernel = rng.integers(0, 256, size=(case.output_channels, case.input_channels), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=case.output_channels, dtype=np.int32)
    return inp, kernel, bias


def output_quantization(acc: np.ndarray):
    """Output scale and zero point from the accumulator range, as the reference testers
    derive them (test/convolution-operator-tester.h:407-413; fully-connected tester :148-154)."""
    if acc.size == 0:
        amin, amax = 0, 900
    else:
        amin, amax = int(acc.min()), int(acc.max())
    scale = float(np.uint32(amax - amin)) / 255.0
    if scale < 1.0001:          # keep requantization scale = 1/scale below 1.0 (cf. gemm tester :236)
        scale = 1.00001
    zp = int(round(127.5 - 0.5 * float(amin + amax) / scale))
    zp = max(0, min(255, zp))
    return np.float32(scale), zp


def strided_view(flat: np.ndarray, rows: int, channels: int, stride: int) -> np.ndarray:
    """[rows, channels] view of a flat strided buffer."""
    if rows == 0:
        return np.zeros((0, channels), dtype=flat.dtype)
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, channels), strides=(stride, 1), writeable=False)


# ---------------------------------------------------------------------------------------------------------
# Deconvolution (transposed convolution): test/deconvolution.cc, in file order. `ConvCase.subsampling` is the
# deconvolution stride, `padding` the amounts removed from the full output.
@dataclass(frozen=True)
class DeconvCase(ConvCase):
    adjustment: Tuple[int, int] = (0, 0)


DECONV_CASES = [
    DeconvCase("d_zero_batch", (5, 5), (1, 1), gic=2, goc=2, batch=0),
    DeconvCase("d_1x1", (27, 29), (1, 1), gic=23, goc=19),
    DeconvCase("d_1x1_with_qmin", (27, 29), (1, 1), gic=23, goc=19, qmin=128),
    DeconvCase("d_1x1_with_qmax", (27, 29), (1, 1), gic=23, goc=19, qmax=128),
    DeconvCase("d_1x1_with_input_stride", (27, 29), (1, 1), gic=23, goc=19, input_pixel_stride=28),
    DeconvCase("d_1x1_with_output_stride", (27, 29), (1, 1), gic=23, goc=19, output_pixel_stride=29),
    DeconvCase("d_1x1_with_batch", (13, 14), (1, 1), gic=23, goc=19, batch=3),
    DeconvCase("d_grouped_1x1", (24, 25), (1, 1), groups=2, gic=17, goc=19),
    DeconvCase("d_1x3", (20, 19), (1, 3), _pad(w=1), gic=17, goc=15),
    DeconvCase("d_grouped_1x3", (20, 19), (1, 3), _pad(w=1), groups=2, gic=17, goc=15),
    DeconvCase("d_3x1", (19, 20), (3, 1), _pad(h=1), gic=17, goc=15),
    DeconvCase("d_grouped_3x1", (19, 20), (3, 1), _pad(h=1), groups=2, gic=17, goc=15),
    DeconvCase("d_3x3", (13, 12), (3, 3), _pad(1, 1), gic=15, goc=17),
    DeconvCase("d_3x3_with_input_stride", (13, 12), (3, 3), _pad(1, 1), gic=15, goc=17, input_pixel_stride=22),
    DeconvCase("d_3x3_with_output_stride", (13, 12), (3, 3), _pad(1, 1), gic=15, goc=17, output_pixel_stride=23),
    DeconvCase("d_3x3_with_batch", (10, 9), (3, 3), _pad(1, 1), gic=15, goc=17, batch=3),
    DeconvCase("d_grouped_3x3", (10, 11), (3, 3), _pad(1, 1), groups=2, gic=14, goc=13),
    DeconvCase("d_3x3s2", (19, 21), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=27, goc=19),
    DeconvCase("d_3x3s1x2", (13, 13), (3, 3), _pad(1, 1), subsampling=(1, 2), gic=27, goc=19),
    DeconvCase("d_3x3s2x1", (13, 13), (3, 3), _pad(1, 1), subsampling=(2, 1), gic=27, goc=19),
    DeconvCase("d_3x3d2", (13, 14), (3, 3), _pad(2, 2), dilation=(2, 2), gic=27, goc=19),
    DeconvCase("d_3x3d1x2", (14, 15), (3, 3), _pad(1, 2), dilation=(1, 2), gic=27, goc=19),
    DeconvCase("d_3x3d2x1", (15, 14), (3, 3), _pad(2, 1), dilation=(2, 1), gic=27, goc=19),
]

# Beyond the reference's list: output adjustment, the usual 2x upsampling layers (2x2 s2, 4x4 s2 p1), aligned
# channel counts (the 16-byte activation-vector path), zero-point / clamp corners, asymmetric padding.
EXTRA_DECONV_CASES = [
    DeconvCase("dx_3x3s2_adjust", (9, 8), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=8, goc=8, adjustment=(1, 1)),
    DeconvCase("dx_3x3s3_adjust_2x1", (6, 7), (3, 3), subsampling=(3, 3), gic=5, goc=7, adjustment=(2, 1)),
    DeconvCase("dx_2x2s2_c64_n32", (14, 14), (2, 2), subsampling=(2, 2), gic=64, goc=32, batch=2),
    DeconvCase("dx_4x4s2p1_c32_n16", (11, 13), (4, 4), _pad(1, 1), subsampling=(2, 2), gic=32, goc=16, batch=2),
    DeconvCase("dx_3x3_c128_n64", (8, 9), (3, 3), _pad(1, 1), gic=128, goc=64),
    DeconvCase("dx_3x3s2_zp_0_255", (7, 7), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=16, goc=16, izp=0, kzp=255),
    DeconvCase("dx_3x3s2_zp_255_0", (7, 7), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=16, goc=16, izp=255, kzp=0),
    DeconvCase("dx_3x3_asym_pad", (9, 10), (3, 3), (2, 0, 0, 1), gic=12, goc=20),
    DeconvCase("dx_5x5s2d2", (6, 6), (5, 5), _pad(2, 2), subsampling=(2, 2), dilation=(2, 2), gic=9, goc=11),
    DeconvCase("dx_grouped_2x2s2", (10, 10), (2, 2), subsampling=(2, 2), groups=4, gic=8, goc=6, batch=2),
    DeconvCase("dx_3x3s2_qmin_qmax", (8, 8), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=16, goc=24, qmin=40, qmax=200),
    DeconvCase("dx_1x1s2", (6, 5), (1, 1), subsampling=(2, 2), gic=10, goc=12),
    # kernel == stride, no padding: the single-GEMM depth-to-space path (q8_pw_stream_d2s_mfma)
    DeconvCase("dx_d2s_2x2_c16_n19", (9, 7), (2, 2), subsampling=(2, 2), gic=16, goc=19, batch=3),
    DeconvCase("dx_d2s_2x2_c32_n64_rows", (33, 35), (2, 2), subsampling=(2, 2), gic=32, goc=64, batch=2),
    DeconvCase("dx_d2s_3x3s3_c48_n8", (5, 6), (3, 3), subsampling=(3, 3), gic=48, goc=8),
    DeconvCase("dx_d2s_2x1_c64_n40", (8, 9), (2, 1), subsampling=(2, 1), gic=64, goc=40, batch=2),
    DeconvCase("dx_d2s_1x2_c16_n16", (4, 11), (1, 2), subsampling=(1, 2), gic=16, goc=16),
    DeconvCase("dx_d2s_strided_pixels", (7, 6), (2, 2), subsampling=(2, 2), gic=32, goc=24, batch=2,
               input_pixel_stride=48, output_pixel_stride=40),
    DeconvCase("dx_d2s_zp_qrange", (6, 6), (2, 2), subsampling=(2, 2), gic=16, goc=32, izp=3, kzp=250, qmin=30, qmax=220),
    DeconvCase("dx_d2s_c256_n32", (5, 5), (2, 2), subsampling=(2, 2), gic=256, goc=32),
    # same shape family but NOT eligible (unaligned channels / rows): falls back to the phase GEMMs
    DeconvCase("dx_phase_2x2_c10_n12", (6, 5), (2, 2), subsampling=(2, 2), gic=10, goc=12, batch=2),
    DeconvCase("dx_phase_2x2_unaligned_rows", (6, 5), (2, 2), subsampling=(2, 2), gic=16, goc=16, input_pixel_stride=20),
]


# Stride 2 with 3x3 / 4x4 kernels and channels % 32 == 0: the streaming kernel over the input pixels (q8deconv.hip)
STREAM_DECONV_CASES = [
    DeconvCase("ds_3x3s2_c64_n32_adjust", (28, 28), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=64, goc=32, batch=2,
               adjustment=(1, 1)),
    DeconvCase("ds_3x3s2_c32_n48_odd", (7, 9), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=32, goc=48, batch=3),
    DeconvCase("ds_3x3s2_c96_n16_nopad", (6, 5), (3, 3), subsampling=(2, 2), gic=96, goc=16, batch=2),
    DeconvCase("ds_3x3s2_c128_n32_asym_pad", (5, 6), (3, 3), (2, 0, 0, 1), subsampling=(2, 2), gic=128, goc=32),
    DeconvCase("ds_3x3s2_c32_n20_dword_stores", (9, 8), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=32, goc=20, batch=2),
    DeconvCase("ds_3x3s2_c32_n19_byte_stores", (9, 8), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=32, goc=19, batch=2),
    DeconvCase("ds_3x3s2_c64_n32_strides", (8, 9), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=64, goc=32, batch=2,
               input_pixel_stride=80, output_pixel_stride=48),
    DeconvCase("ds_3x3s2_c32_n32_zp_0_255", (7, 7), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=32, goc=32, izp=0, kzp=255),
    DeconvCase("ds_3x3s2_c32_n32_zp_255_0", (7, 7), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=32, goc=32, izp=255, kzp=0),
    DeconvCase("ds_3x3s2_c64_n32_qrange", (8, 8), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=64, goc=32, qmin=40, qmax=200,
               adjustment=(1, 0)),
    DeconvCase("ds_3x3s2_c32_n16_one_pixel", (1, 1), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=32, goc=16, batch=5),
    DeconvCase("ds_4x4s2p1_c64_n32", (13, 11), (4, 4), _pad(1, 1), subsampling=(2, 2), gic=64, goc=32, batch=2),
    DeconvCase("ds_4x4s2_c32_n48_nopad", (6, 7), (4, 4), subsampling=(2, 2), gic=32, goc=48, batch=3),
    DeconvCase("ds_4x4s2p1_c96_n24_adjust", (5, 9), (4, 4), _pad(1, 1), subsampling=(2, 2), gic=96, goc=24, adjustment=(1, 1)),
    DeconvCase("ds_3x3s2_c64_n96_many_units", (33, 35), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=64, goc=96, batch=4,
               adjustment=(1, 1)),
]
# the same family outside the streaming kernel's range: the phase GEMMs
STREAM_DECONV_FALLBACK_CASES = [
    DeconvCase("dsf_3x3s2_c160", (6, 6), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=160, goc=16),          # channels > 128
    DeconvCase("dsf_3x3s2_c64_unaligned_rows", (6, 6), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=64, goc=16,
               input_pixel_stride=72),
    DeconvCase("dsf_4x4s2_c128_n128_lds", (5, 5), (4, 4), _pad(1, 1), subsampling=(2, 2), gic=128, goc=128),  # 256 KiB of weights
    DeconvCase("dsf_3x3s2_grouped", (6, 6), (3, 3), _pad(1, 1), subsampling=(2, 2), groups=2, gic=32, goc=16),
    DeconvCase("dsf_3x4s2", (6, 6), (3, 4), _pad(1, 1), subsampling=(2, 2), gic=32, goc=16),
]


def deconv_tensors(case: DeconvCase):
    """Seeded input / kernel / bias; kernel in the deconvolution layout [g][ic][kh][kw][oc]
    (test/deconvolution-operator-tester.h:355, :411)."""
    rng = np.random.default_rng(seed_for(case.name))
    H, W = case.input_size
    pixels = case.batch * H * W
    in_len = max(pixels - 1, 0) * case.in_stride + case.groups * case.gic if pixels else 0
    inp = rng.integers(0, 256, size=in_len, dtype=np.uint8)
    kernel = rng.integers(0, 256, size=(case.groups, case.gic, case.kernel_size[0], case.kernel_size[1], case.goc),
                          dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=case.groups * case.goc, dtype=np.int32)
    return inp, kernel, bias
