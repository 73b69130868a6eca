"""GPU tier: the 5x5 column-sliding depthwise kernel (q8_dwconv_col5x5_kernel, "kernel H" in
qnnpack_amd/csrc/hip/q8dwconv.hip) against the scalar oracle, bit for bit. The reference's 5x5 depthwise cases
(test/convolution.cc depthwise_5x5*, 27 channels) cannot take it (channels % 4) and stay on the LDS-tiled kernel;
these are the same windows with channel counts a 4-channel lane can own: strides 1 and 2, every padding from none
to four, images smaller than the window, row segments, ragged lane counts, pixel strides, batch, zero points,
clamps, every requantization flavour, both weight-range classes of the int8 dot-product walk (kzp 127: kzp - w
fits; kzp 128: w - kzp fits), and weights outside both (the operator must then keep the LDS-tiled kernel)."""
import dataclasses

import numpy as np
import pytest

from _cases import ConvCase, conv_tensors
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal, conv_expected, conv_run
from oracle import o1

pytestmark = pytest.mark.gpu

NAME = "q8_dwconv_col_5x5_dot4"


def _dw5(name, hw, c, **kw):
    kw.setdefault("padding", (2, 2, 2, 2))
    return ConvCase(name, hw, (5, 5), kw.pop("padding"), groups=c, gic=1, goc=1, **kw)


CASES = [
    _dw5("h_c32_14", (14, 14), 32, batch=3),
    _dw5("h_c4_1x1img", (1, 1), 4),
    _dw5("h_c8_2x3img", (2, 3), 8, batch=2),
    _dw5("h_c16_5x5img_nopad", (5, 5), 16, padding=(0, 0, 0, 0)),
    _dw5("h_c16_7x6img_nopad", (7, 6), 16, padding=(0, 0, 0, 0), batch=2),
    _dw5("h_c24_9x40_wide", (9, 40), 24),
    _dw5("h_c20_40x9_tall", (40, 9), 20, batch=2),
    _dw5("h_c72_s2", (29, 31), 72, subsampling=(2, 2)),
    _dw5("h_c72_s2_even", (28, 28), 72, subsampling=(2, 2), batch=2),
    _dw5("h_c32_s2_pad_tl_only", (15, 15), 32, subsampling=(2, 2), padding=(2, 0, 0, 2)),
    _dw5("h_c32_s2_nopad", (15, 17), 32, subsampling=(2, 2), padding=(0, 0, 0, 0)),
    _dw5("h_c32_s2_6x5img", (6, 5), 32, subsampling=(2, 2), batch=2),
    _dw5("h_c32_pad_asym", (12, 13), 32, padding=(2, 0, 1, 0)),
    _dw5("h_c32_pad_asym2", (12, 13), 32, padding=(0, 1, 0, 2)),
    _dw5("h_c32_pad1", (10, 11), 32, padding=(1, 1, 1, 1)),
    _dw5("h_c32_pad3", (10, 11), 32, padding=(3, 3, 3, 3)),
    _dw5("h_c32_pad4", (9, 8), 32, padding=(4, 4, 4, 4)),
    _dw5("h_c32_s2_pad4", (11, 10), 32, subsampling=(2, 2), padding=(4, 4, 4, 4)),
    _dw5("h_c260_ragged_lanes", (7, 7), 260, batch=5),
    _dw5("h_c672_14", (14, 14), 672, batch=2),                  # the bench shape's image
    _dw5("h_c240_28", (28, 28), 240, batch=2),
    _dw5("h_c32_strided_pixels", (11, 12), 32, input_pixel_stride=40, output_pixel_stride=36),
    _dw5("h_c64_qmin_qmax", (9, 9), 64, qmin=100, qmax=150),
    _dw5("h_c32_112_segments", (112, 112), 32),                 # row segments
    _dw5("h_c16_56_segments", (56, 56), 16, batch=4),
    _dw5("h_c16_57_s2_segments", (57, 57), 16, subsampling=(2, 2), batch=3),
    _dw5("h_c72_56_s2", (56, 56), 72, subsampling=(2, 2), batch=2),
    _dw5("h_c64_izp0", (9, 9), 64, izp=0),
    _dw5("h_c64_izp255", (9, 9), 64, izp=255),
]


@pytest.mark.parametrize("kzp", [127, 128], ids=lambda v: f"kzp{v}")
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_col5_kernel_matches_oracle(qnnp, case, kzp):
    case = dataclasses.replace(case, kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 4, 4, 0] = 0, 255          # the full range, whatever the seed drew
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == NAME, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}, kzp {kzp}]")


@pytest.mark.parametrize("case", [c for c in CASES if c.name in ("h_c32_14", "h_c72_s2", "h_c32_pad4", "h_c260_ragged_lanes")],
                         ids=lambda c: c.name)
def test_lds_and_col5_kernels_agree(qnnp, case):
    """"dwconv_kernel" = 2 keeps the LDS-tiled kernel, 6 forces the column walk: same bytes"""
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    for variant, want in ((2, "q8_dwconv_lds_5x5"), (6, NAME)):
        qnnp.set_option("dwconv_kernel", variant)
        try:
            out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
        finally:
            qnnp.set_option("dwconv_kernel", 0)
        assert kname == want, kname
        assert_bytes_equal(out, expected, f"dwconv_kernel={variant} {kname} [{case.name}]")


@pytest.mark.parametrize("lo,hi,kzp,walk", [
    (40, 200, 100, True),       # x in [-60, 100]
    (0, 129, 1, True),          # x in [-1, 128]: only the negated weights fit
    (0, 130, 1, False),         # x up to 129: neither -- the LDS-tiled kernel's int16 pairs
    (3, 3, 3, True),            # all-zero x
    (0, 255, 60, False),
])
def test_flavour_follows_the_weights(qnnp, lo, hi, kzp, walk):
    case = dataclasses.replace(_dw5("h_range", (17, 15), 40, batch=2), kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel = (lo + kernel.astype(np.int32) % (hi - lo + 1)).astype(np.uint8)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 4, 4, 0] = lo, hi
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == (NAME if walk else "q8_dwconv_lds_5x5"), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [weights {lo}..{hi}, kzp {kzp}]")
    if not walk:
        from qnnpack_amd import QnnpackError
        qnnp.set_option("dwconv_kernel", 6)
        try:
            with pytest.raises(QnnpackError):
                conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
        finally:
            qnnp.set_option("dwconv_kernel", 0)


@pytest.mark.parametrize("scale,zp,qmin,qmax", [
    (0.5, 127, 0, 255), (0.75, 3, 0, 255), (0.0125, 127, 0, 255), (0.0125, 0, 10, 240), (2.0 ** -9, 255, 0, 255),
    (0.3, 128, 128, 255), (float.fromhex("0x1.FFFFFEp-1"), 200, 0, 255), (2.0 ** -24, 17, 0, 255)],
    ids=lambda v: str(v))
@pytest.mark.parametrize("stride", [1, 2])
def test_requantization_flavours(qnnp, scale, zp, qmin, qmax, stride):
    """the fused epilogue is chosen per operator on the host (requant_dispatch_ofs): one kernel per flavour"""
    case = _dw5("h_rq", (19, 18), 48, batch=2, subsampling=(stride, stride))
    inp, kernel, bias = conv_tensors(case)
    shape = o1.conv_shape(case.batch, 19, 18, case.padding, (5, 5), (stride, stride), (1, 1), 48, 1, 1, 48)
    acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
    expected = o1.requantize_rows(acc.reshape(-1, 48), np.float32(scale), zp, qmin, qmax).reshape(-1)
    oh, ow = o1.conv_output_hw(shape)
    op = qnnp.create_convolution2d_nhwc_q8(2, 2, 2, 2, 5, 5, stride, stride, 1, 1, 48, 1, 1, case.izp, float(np.float32(scale)),
                                           case.kzp, 1.0, kernel, bias, zp, 1.0, qmin, qmax, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, case.batch, 19, 18, d_in, 48, d_out, 48)
        qnnp.run_operator(op)
        assert qnnp.operator_kernel(op) == NAME
        assert_bytes_equal(from_device(d_out), expected, f"kernel H stride {stride}, requantization scale {scale} zp {zp} [{qmin}, {qmax}]")
    finally:
        qnnp.delete_operator(op)


def test_misaligned_and_odd_channel_tensors_keep_the_other_kernels(qnnp):
    case = _dw5("h_c27", (15, 14), 27)                  # the reference's channel count: not a multiple of 4
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname != NAME, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [27 channels]")
