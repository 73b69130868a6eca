"""GPU tier: dilated 3x3 depthwise convolutions on the column walk (kernel G's int8 dot-product flavour in its
residue-class form, `q8_dwconv_col3x3_kernel<..., DIL = true>` in qnnpack_amd/csrc/hip/q8dwconv.hip) against the scalar
oracle, bit for bit. The reference routes dilated depthwise layers through the same q8dwconv family
(src/convolution.c:182-183, 213-230; test/convolution.cc depthwise_3x3d2 / d1x2 / d2x1 -- 27 channels, which a
4-channel lane cannot own: those stay on the LDS-tiled kernel and in tests/test_gpu_operators.py). Here: the same windows
with channel counts % 4 == 0 -- dilations 2, 3, 4 and mixed, every padding from none to twice the dilation, images
smaller than the dilated window, residue classes with no row at all, row segments, ragged lane counts, pixel strides,
batch, zero points, clamps, both weight-range classes, weights outside both (the operator keeps the LDS-tiled kernel),
stride 2 (likewise), and the bench's DeepLab-style shape at batch 128."""
import dataclasses

import numpy as np
import pytest

from _cases import ConvCase, conv_tensors
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal, conv_expected, conv_run
from oracle import o1

pytestmark = pytest.mark.gpu

NAME = "q8_dwconv_col_3x3_dot4_dilated"


def _dw(name, hw, c, d, **kw):
    dh, dw = d
    kw.setdefault("padding", (dh, dw, dh, dw))       # "same" (top, right, bottom, left)
    return ConvCase(name, hw, (3, 3), kw.pop("padding"), dilation=d, groups=c, gic=1, goc=1, **kw)


CASES = [
    _dw("g_d2_c32_14", (14, 14), 32, (2, 2), batch=3),
    _dw("g_d2_c192_28", (28, 28), 192, (2, 2), batch=2),                     # the bench shape's image
    _dw("g_d2_c8_5x5img_nopad", (5, 5), 8, (2, 2), padding=(0, 0, 0, 0)),     # exactly one window
    _dw("g_d2_c8_6x7img_nopad", (6, 7), 8, (2, 2), padding=(0, 0, 0, 0), batch=2),
    _dw("g_d2_c4_1x1img", (1, 1), 4, (2, 2)),
    _dw("g_d2_c8_2x3img", (2, 3), 8, (2, 2), batch=2),
    _dw("g_d2_c16_3x3img", (3, 3), 16, (2, 2)),
    _dw("g_d3_c16_13x12", (13, 12), 16, (3, 3), batch=2),
    _dw("g_d4_c16_9x9", (9, 9), 16, (4, 4)),
    _dw("g_d4_c16_3x3img", (3, 3), 16, (4, 4), batch=2),                     # one output row per residue class
    _dw("g_d4_c16_2x9img", (2, 9), 16, (4, 4)),                             # residue classes 2, 3 have no output row
    _dw("g_d1x2_c32", (15, 14), 32, (1, 2), batch=2),
    _dw("g_d2x1_c32", (15, 14), 32, (2, 1), batch=2),
    _dw("g_d2x3_c24", (17, 19), 24, (2, 3)),
    _dw("g_d3x2_c24", (19, 17), 24, (3, 2)),
    _dw("g_d2_pad_max", (10, 11), 32, (2, 2), padding=(4, 4, 4, 4)),
    _dw("g_d2_pad_asym", (12, 13), 32, (2, 2), padding=(4, 0, 1, 3)),
    _dw("g_d2_pad_asym2", (12, 13), 32, (2, 2), padding=(0, 3, 4, 0)),
    _dw("g_d2_pad1", (10, 11), 32, (2, 2), padding=(1, 1, 1, 1)),
    _dw("g_d3_pad_tl_only", (15, 15), 32, (3, 3), padding=(6, 0, 0, 6)),
    _dw("g_d2_c260_ragged_lanes", (7, 7), 260, (2, 2), batch=5),
    _dw("g_d2_c24_9x40_wide", (9, 40), 24, (2, 2)),
    _dw("g_d2_c20_40x9_tall", (40, 9), 20, (2, 2), batch=2),
    _dw("g_d2_strided_pixels", (11, 12), 32, (2, 2), input_pixel_stride=40, output_pixel_stride=36),
    _dw("g_d2_qmin_qmax", (9, 9), 64, (2, 2), qmin=100, qmax=150),
    _dw("g_d2_c32_112_segments", (112, 112), 32, (2, 2)),                    # row segments per residue class
    _dw("g_d3_c16_57_segments", (57, 57), 16, (3, 3), batch=3),
    _dw("g_d2_izp0", (9, 9), 64, (2, 2), izp=0),
    _dw("g_d2_izp255", (9, 9), 64, (2, 2), izp=255),
    _dw("g_d6_c32_33", (33, 33), 32, (6, 6)),                                # DeepLab's ASPP rate at a 33x33 feature map
    _dw("g_d12_c32_33", (33, 33), 32, (12, 12), batch=2),
]


@pytest.mark.parametrize("kzp", [127, 128], ids=lambda v: f"kzp{v}")
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_dilated_walk_matches_oracle(qnnp, case, kzp):
    case = dataclasses.replace(case, kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 2, 2, 0] = 0, 255          # the full range, whatever the seed drew
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == NAME, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}, kzp {kzp}]")


@pytest.mark.parametrize("case", [c for c in CASES if c.name in ("g_d2_c32_14", "g_d3_c16_13x12", "g_d2_pad_max", "g_d2_c260_ragged_lanes")],
                         ids=lambda c: c.name)
def test_lds_and_dilated_walk_agree(qnnp, case):
    """"dwconv_kernel" = 2 keeps the LDS-tiled kernel, 6 forces the column walk: same bytes"""
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    for variant, want in ((2, "q8_dwconv_lds_3x3"), (6, NAME)):
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
    (0, 130, 1, False),         # x up to 129: neither class -- no int8 walk, so no dilated walk
    (0, 255, 60, False),
])
def test_flavour_follows_the_weights(qnnp, lo, hi, kzp, walk):
    case = dataclasses.replace(_dw("g_range", (17, 15), 40, (2, 2), batch=2), kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel = (lo + kernel.astype(np.int32) % (hi - lo + 1)).astype(np.uint8)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 2, 2, 0] = lo, hi
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert (kname == NAME) == walk, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [weights {lo}..{hi}, kzp {kzp}]")


def test_stride_2_and_padding_beyond_the_window_keep_the_other_kernels(qnnp):
    for case in (_dw("g_d2_s2", (15, 14), 32, (2, 2), subsampling=(2, 2)),
                 _dw("g_d2_pad5", (15, 14), 32, (2, 2), padding=(5, 2, 2, 2))):
        inp, kernel, bias = conv_tensors(case)
        expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
        out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
        assert kname != NAME, kname
        assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("scale,zp,qmin,qmax", [
    (0.5, 127, 0, 255), (0.75, 3, 0, 255), (0.0125, 127, 0, 255), (0.0125, 0, 10, 240), (2.0 ** -9, 255, 0, 255),
    (0.3, 128, 128, 255), (float.fromhex("0x1.FFFFFEp-1"), 200, 0, 255), (2.0 ** -24, 17, 0, 255)],
    ids=lambda v: str(v))
def test_requantization_flavours(qnnp, scale, zp, qmin, qmax):
    """the fused epilogue is chosen per operator on the host (requant_dispatch_ofs): one kernel per flavour"""
    case = _dw("g_rq", (19, 18), 48, (2, 2), batch=2)
    inp, kernel, bias = conv_tensors(case)
    shape = o1.conv_shape(case.batch, 19, 18, case.padding, (3, 3), (1, 1), (2, 2), 48, 1, 1, 48)
    acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
    expected = o1.requantize_rows(acc.reshape(-1, 48), np.float32(scale), zp, qmin, qmax).reshape(-1)
    op = qnnp.create_convolution2d_nhwc_q8(2, 2, 2, 2, 3, 3, 1, 1, 2, 2, 48, 1, 1, case.izp, float(np.float32(scale)),
                                           case.kzp, 1.0, kernel, bias, zp, 1.0, qmin, qmax, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, case.batch, 19, 18, d_in, 48, d_out, 48)
        qnnp.run_operator(op)
        assert qnnp.operator_kernel(op) == NAME
        assert_bytes_equal(from_device(d_out), expected, f"dilated walk, requantization scale {scale} zp {zp} [{qmin}, {qmax}]")
    finally:
        qnnp.delete_operator(op)


def test_bench_shape_at_batch_128(qnnp):
    """bench.py's `dw3x3_dil2_28x28x192` (extra.q8dwconv_5x5_dilated_and_realistic_scale), every image"""
    case = _dw("g_d2_c192_28_b128", (28, 28), 192, (2, 2), batch=128)
    inp, kernel, bias = conv_tensors(case)
    o1.set_threads(16)
    try:
        expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    finally:
        o1.set_threads(1)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == NAME, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")
