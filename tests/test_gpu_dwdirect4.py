"""GPU tier: the generic depthwise kernel with four channels per thread (q8_dwconv_direct4_kernel in qnnpack_amd/csrc/hip/q8dwconv.hip,
round 6: what auto picks where no specialised depthwise kernel applies) against the scalar oracle: channel counts that are multiples of
nothing (ShuffleNet v2's 58 / 122: bench/convolution.cc:335-426), pixels that start at any byte, every window / stride / dilation /
padding, the last partial channel group, zero points and clamps; and the byte-per-thread kernel it replaced ("dwconv_kernel" 1) on the
same cases. Reference: q8dwconv under qnnp_run_operator (src/q8dwconv/up8x9-sse2.c:14-372, mp8x25-sse2.c:14-742)."""
import pytest

from _cases import ConvCase, conv_tensors
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu


def _dw(name, hw, c, k=(3, 3), **kw):
    kw.setdefault("padding", (k[0] // 2, k[1] // 2, k[0] // 2, k[1] // 2))
    return ConvCase(name, hw, k, kw.pop("padding"), groups=c, gic=1, goc=1, **kw)


CASES = [
    _dw("d4_c58_28", (28, 28), 58, batch=3),                                   # ShuffleNet v2 x1.0
    _dw("d4_c58_56_s2", (56, 56), 58, subsampling=(2, 2), batch=2),
    _dw("d4_c122_28", (28, 28), 122, batch=2),                                 # ShuffleNet v2 x2.0
    _dw("d4_c50_s2", (29, 31), 50, subsampling=(2, 2), batch=2),               # ShuffleNet v1 g2
    _dw("d4_c27", (9, 11), 27, batch=2),
    _dw("d4_c5", (7, 7), 5, batch=3),
    _dw("d4_c6_7x7_window", (12, 12), 6, k=(7, 7), batch=2),
    _dw("d4_c10_5x5_s2_dil", (17, 15), 10, k=(5, 5), subsampling=(2, 2), dilation=(2, 2), padding=(4, 4, 4, 4)),
    _dw("d4_c58_strided_pixels", (9, 9), 58, input_pixel_stride=61, output_pixel_stride=59, batch=2),
    _dw("d4_c22_zp_clamp", (9, 9), 22, izp=3, kzp=250, qmin=40, qmax=200, batch=2),
    _dw("d4_c58_1x1img", (1, 1), 58, batch=5),
]


def _auto_kernel(case):
    """3x3 windows (dilation 1, equal strides 1 | 2, padding <= 2) of >= 4 channels: the sliding-window kernel on unaligned dwords (round 6)"""
    k33 = case.kernel_size == (3, 3) and case.dilation == (1, 1) and case.subsampling in ((1, 1), (2, 2))
    return "q8_dwconv_row_3x3_any" if k33 and case.groups >= 4 and max(case.padding) <= 2 else "q8_dwconv_direct4"


@pytest.mark.parametrize("variant", [0, 9, 1], ids=["auto", "four_channels_per_thread", "byte_per_thread"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_generic_depthwise_matches_oracle(qnnp, case, variant):
    kernel = {0: _auto_kernel(case), 9: "q8_dwconv_direct4", 1: "q8_dwconv_direct"}[variant]
    inp, kern, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kern, bias)
    qnnp.set_option("dwconv_kernel", variant)
    try:
        out, kname = conv_run(qnnp, case, quant, out_hw, inp, kern, bias, to_device, from_device)
    finally:
        qnnp.set_option("dwconv_kernel", 0)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


# ---- the sliding-window kernel on unaligned dwords ("dwconv_kernel" 8; q8_dwconv_row3x3_kernel<SW, true>) ----
ANY_CASES = [
    _dw("ra_c58_28", (28, 28), 58, batch=3),
    _dw("ra_c58_56_s2", (56, 56), 58, subsampling=(2, 2), batch=2),
    _dw("ra_c122_28", (28, 28), 122, batch=2),
    _dw("ra_c244_14_s2", (14, 14), 244, subsampling=(2, 2), batch=3),
    _dw("ra_c488_7", (7, 7), 488, batch=2),
    _dw("ra_c50_odd_image_s2", (29, 31), 50, subsampling=(2, 2), batch=2),
    _dw("ra_c27", (9, 11), 27, batch=2),                                       # odd: groups start at odd bytes
    _dw("ra_c4", (7, 7), 4, batch=3),
    _dw("ra_c5", (7, 7), 5, batch=3),                                          # two groups, the second recomputes three channels
    _dw("ra_c7_s2", (10, 12), 7, subsampling=(2, 2), batch=2),
    _dw("ra_c58_strided_pixels", (9, 9), 58, input_pixel_stride=61, output_pixel_stride=59, batch=2),
    _dw("ra_c58_nopad", (9, 9), 58, padding=(0, 0, 0, 0), batch=2),
    _dw("ra_c58_pad2", (9, 9), 58, padding=(2, 2, 2, 2), batch=2),
    _dw("ra_c58_pad_asym", (9, 12), 58, padding=(0, 2, 1, 0), batch=2),
    _dw("ra_c22_zp_clamp", (9, 9), 22, izp=3, kzp=250, qmin=40, qmax=200, batch=2),
    _dw("ra_c22_zp_extremes", (9, 9), 22, izp=255, kzp=0, batch=2),
    _dw("ra_c58_1x1img", (1, 1), 58, batch=5),
    _dw("ra_c58_wide_row_segments", (5, 200), 58, batch=1),
    _dw("ra_c32_aligned_too", (14, 14), 32, batch=2),
]


@pytest.mark.parametrize("case", ANY_CASES, ids=lambda c: c.name)
def test_sliding_window_on_unaligned_dwords_matches_oracle(qnnp, case):
    inp, kern, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kern, bias)
    qnnp.set_option("dwconv_kernel", 8)
    try:
        out, kname = conv_run(qnnp, case, quant, out_hw, inp, kern, bias, to_device, from_device)
    finally:
        qnnp.set_option("dwconv_kernel", 0)
    assert kname == "q8_dwconv_row_3x3_any", kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    _dw("ra_bad_3_channels", (9, 9), 3, batch=2),
    _dw("ra_bad_5x5", (9, 9), 58, k=(5, 5), batch=2),
    _dw("ra_bad_dilated", (12, 12), 58, dilation=(2, 2), padding=(2, 2, 2, 2), batch=2),
    _dw("ra_bad_stride_3", (12, 12), 58, subsampling=(3, 3), batch=2),
], ids=lambda c: c.name)
def test_sliding_window_on_unaligned_dwords_refuses_what_it_cannot_take(qnnp, case):
    from qnnpack_amd import QnnpackError
    inp, kern, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kern, bias)
    qnnp.set_option("dwconv_kernel", 8)
    try:
        with pytest.raises(QnnpackError):
            conv_run(qnnp, case, quant, out_hw, inp, kern, bias, to_device, from_device)
    finally:
        qnnp.set_option("dwconv_kernel", 0)
