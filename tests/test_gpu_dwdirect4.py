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


@pytest.mark.parametrize("variant,kernel", [(0, "q8_dwconv_direct4"), (1, "q8_dwconv_direct")], ids=["auto", "byte_per_thread"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_generic_depthwise_matches_oracle(qnnp, case, variant, kernel):
    inp, kern, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kern, bias)
    qnnp.set_option("dwconv_kernel", variant)
    try:
        out, kname = conv_run(qnnp, case, quant, out_hw, inp, kern, bias, to_device, from_device)
    finally:
        qnnp.set_option("dwconv_kernel", 0)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")
