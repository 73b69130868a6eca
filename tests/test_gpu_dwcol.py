"""GPU tier: the column-sliding depthwise kernel (q8_dwconv_col3x3_kernel in qnnpack_amd/csrc/hip/q8dwconv.hip),
forced with "dwconv_kernel" = 6, against the scalar oracle: strides 1 and 2, every padding combination the
reference's tests use (test/convolution.cc depthwise_3x3*), images smaller than a window, row segments (several
waves walking one image), rows whose dword count is not a multiple of the wave width, pixel strides, batch, zero
points, clamps, and every requantization flavour (shift 0, bounded / general shift >= 1, folded / late zero point).
Stride 1 has two tap arithmetics chosen per operator from the weights' range (pack.h qnnp_dwconv_weight_range): the
int8 dot-product walk when w - kzp (kzp = 128) or kzp - w (kzp = 127, the cases' default) fits int8, the int16 pair
walk otherwise; every case runs under all three."""
import dataclasses

import numpy as np
import pytest

from _cases import CONV_CASES, EXTRA_CONV_CASES, ConvCase, conv_tensors
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal, conv_expected, conv_run
from oracle import o1

CONV_BY_NAME = {c.name: c for c in list(CONV_CASES) + list(EXTRA_CONV_CASES)}

pytestmark = pytest.mark.gpu


def _dw(name, hw, c, **kw):
    kw.setdefault("padding", (1, 1, 1, 1))
    return ConvCase(name, hw, (3, 3), kw.pop("padding"), groups=c, gic=1, goc=1, **kw)


CASES = [
    _dw("g_c32_14", (14, 14), 32, batch=3),
    _dw("g_c4_1x1img", (1, 1), 4),
    _dw("g_c8_2x3img", (2, 3), 8, batch=2),
    _dw("g_c16_3x3img_nopad", (3, 3), 16, padding=(0, 0, 0, 0)),
    _dw("g_c16_5x4img_nopad", (5, 4), 16, padding=(0, 0, 0, 0), batch=2),
    _dw("g_c24_9x40_wide", (9, 40), 24),
    _dw("g_c20_40x9_tall", (40, 9), 20, batch=2),
    _dw("g_c96_s2", (29, 31), 96, subsampling=(2, 2)),
    _dw("g_c144_s2_even", (28, 28), 144, subsampling=(2, 2), batch=2),
    _dw("g_c32_s2_pad_tl_only", (15, 15), 32, subsampling=(2, 2), padding=(1, 0, 0, 1)),
    _dw("g_c32_s2_nopad", (15, 17), 32, subsampling=(2, 2), padding=(0, 0, 0, 0)),
    _dw("g_c32_pad_asym", (12, 13), 32, padding=(1, 0, 1, 0)),
    _dw("g_c32_pad_asym2", (12, 13), 32, padding=(0, 1, 0, 1)),
    _dw("g_c32_pad2", (10, 11), 32, padding=(2, 2, 2, 2)),
    _dw("g_c32_s2_pad2", (11, 10), 32, subsampling=(2, 2), padding=(2, 2, 2, 2)),
    _dw("g_c260_ragged_lanes", (7, 7), 260, batch=5),
    _dw("g_c960_7x7", (7, 7), 960, batch=2),
    _dw("g_c32_strided_pixels", (11, 12), 32, input_pixel_stride=40, output_pixel_stride=36),
    _dw("g_c64_zp", (9, 9), 64, izp=255, kzp=0),
    _dw("g_c64_zp2", (9, 9), 64, izp=0, kzp=255),
    _dw("g_c32_qmin_qmax", (9, 9), 32, qmin=100, qmax=150),
    _dw("g_c32_112", (112, 112), 32),                          # 16 row segments of 7 rows
    _dw("g_c16_56_segments", (56, 56), 16, batch=4),           # 8 row segments
    _dw("g_c16_57_s2_segments", (57, 57), 16, subsampling=(2, 2), batch=3),
    _dw("g_c144_56", (56, 56), 144, batch=2),                  # MobileNetV2 layer 8 shape
    _dw("g_c576_14_s2", (14, 14), 576, subsampling=(2, 2), batch=3),
]


@pytest.fixture()
def col(qnnp):
    qnnp.set_option("dwconv_kernel", 6)
    yield qnnp
    qnnp.set_option("dwconv_kernel", 0)


def _expected_name(case, kernel):
    x = kernel.astype(np.int32) - case.kzp
    fits = (x.min() >= -128 and x.max() <= 127) or (x.min() >= -127 and x.max() <= 128)
    return "q8_dwconv_col_3x3_dot4" if fits and case.subsampling == (1, 1) else "q8_dwconv_col_3x3"


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_col_kernel_matches_oracle(col, case):
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(col, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == _expected_name(case, kernel), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


STRIDE1 = [c for c in CASES if c.subsampling == (1, 1) and c.kzp == 127]


@pytest.mark.parametrize("kzp", [128, 100], ids=lambda v: f"kzp{v}")
@pytest.mark.parametrize("case", STRIDE1, ids=lambda c: c.name)
def test_col_kernel_other_weight_ranges(col, case, kzp):
    """kzp 128: w - kzp fits int8 as it is (dot-product walk, plain weights); kzp 100: neither sign fits (pair walk)"""
    case = dataclasses.replace(case, kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 2, 2, 0] = 0, 255          # the full range, whatever the seed drew
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(col, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == ("q8_dwconv_col_3x3_dot4" if kzp == 128 else "q8_dwconv_col_3x3"), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}, kzp {kzp}]")


@pytest.mark.parametrize("lo,hi,kzp,name", [
    (40, 200, 100, "q8_dwconv_col_3x3_dot4"),      # x in [-60, 100]
    (0, 129, 1, "q8_dwconv_col_3x3_dot4"),         # x in [-1, 128]: only the negated weights fit
    (0, 130, 1, "q8_dwconv_col_3x3"),              # x up to 129: neither
    (3, 3, 3, "q8_dwconv_col_3x3_dot4"),           # all-zero x
])
def test_col_kernel_flavour_follows_the_weights(col, lo, hi, kzp, name):
    case = dataclasses.replace(_dw("g_range", (17, 15), 40, batch=2), kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel = (lo + kernel.astype(np.int32) % (hi - lo + 1)).astype(np.uint8)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 2, 2, 0] = lo, hi
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(col, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == name, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [weights {lo}..{hi}, kzp {kzp}]")


@pytest.mark.parametrize("scale,zp,qmin,qmax", [
    (0.5, 127, 0, 255), (0.75, 3, 0, 255), (0.0125, 127, 0, 255), (0.0125, 0, 10, 240), (2.0 ** -9, 255, 0, 255),
    (0.3, 128, 128, 255), (float.fromhex("0x1.FFFFFEp-1"), 200, 0, 255), (2.0 ** -24, 17, 0, 255)],
    ids=lambda v: str(v))
@pytest.mark.parametrize("kzp", [127, 77], ids=lambda v: f"kzp{v}")
def test_col_kernel_requantization_flavours(col, scale, zp, qmin, qmax, kzp):
    """the fused epilogue is chosen per operator (requant_dispatch): drive each branch through both walks"""
    case = dataclasses.replace(_dw("g_rq", (19, 18), 48, batch=2), kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    shape = o1.conv_shape(case.batch, 19, 18, case.padding, (3, 3), (1, 1), (1, 1), 48, 1, 1, 48)
    acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
    expected = o1.requantize_rows(acc.reshape(-1, 48), np.float32(scale), zp, qmin, qmax).reshape(-1)
    op = col.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 48, 1, 1, case.izp, float(np.float32(scale)), case.kzp, 1.0,
                                          kernel, bias, zp, 1.0, qmin, qmax, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        col.setup_convolution2d_nhwc_q8(op, case.batch, 19, 18, d_in, 48, d_out, 48)
        col.run_operator(op)
        assert col.operator_kernel(op) == _expected_name(case, kernel)
        assert_bytes_equal(from_device(d_out), expected, f"col kernel (kzp {kzp}), requantization scale {scale} zp {zp} [{qmin}, {qmax}]")
    finally:
        col.delete_operator(op)


@pytest.mark.parametrize("name", ["depthwise_3x3d2", "depthwise_3x3s1x2", "depthwise_3x3", "depthwise_5x5"])
def test_unsupported_shapes_are_reported_not_silently_rerouted(col, name):
    from qnnpack_amd import QnnpackError
    # dilated with 27 channels, anisotropic stride, 27 channels (not a multiple of 4; 3x3 and 5x5). (Dilated windows
    # with channels % 4 == 0 take the walk since round 4: tests/test_gpu_dwcol_dilated.py.)
    case = CONV_BY_NAME[name]
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    with pytest.raises(QnnpackError):
        conv_run(col, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
