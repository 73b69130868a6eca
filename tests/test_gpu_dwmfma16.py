"""GPU tier: the depthwise 3x3 stride-1 walk on v_mfma_i32_16x16x64_i8 (q8_dwconv_mfma16_3x3_kernel in qnnpack_amd/csrc/hip/q8dwconv.hip,
round 6; forced with "dwconv_kernel" = 7) against the scalar oracle: every padding combination, images smaller than a window and
narrower than a strip, several strips / row segments / channel groups, one, two and three channel blocks per wave, pixel strides,
batch, zero points (both weight-range classes: kernel zero point 127 negates the weights, 128 takes them as they are), clamps and the
requantization flavours. Reference: q8dwconv_ukernel_up8x9__sse2 (src/q8dwconv/up8x9-sse2.c:14-372) under qnnp_run_operator."""
import dataclasses

import numpy as np
import pytest

from _cases import ConvCase, conv_tensors
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal, conv_expected, conv_run
from oracle import o1
from qnnpack_amd.binding import QnnpackError

pytestmark = pytest.mark.gpu
KERNEL = "q8_dwconv_mfma16_3x3"


def _dw(name, hw, c, **kw):
    kw.setdefault("padding", (1, 1, 1, 1))
    return ConvCase(name, hw, (3, 3), kw.pop("padding"), groups=c, gic=1, goc=1, **kw)


CASES = [
    _dw("m_c32_14", (14, 14), 32, batch=3),                        # two blocks per wave, one strip of 14 columns
    _dw("m_c16_1x1img", (1, 1), 16),
    _dw("m_c16_2x3img", (2, 3), 16, batch=2),
    _dw("m_c16_3x3img_nopad", (3, 3), 16, padding=(0, 0, 0, 0)),
    _dw("m_c48_5x4img_nopad", (5, 4), 48, padding=(0, 0, 0, 0), batch=2),      # three blocks per wave
    _dw("m_c16_9x40_wide", (9, 40), 16),                           # three strips, the last of 8 columns
    _dw("m_c32_40x9_tall", (40, 9), 32, batch=2),
    _dw("m_c32_pad_asym", (12, 13), 32, padding=(1, 0, 1, 0)),
    _dw("m_c32_pad_asym2", (12, 13), 32, padding=(0, 1, 0, 1)),
    _dw("m_c32_pad2", (10, 11), 32, padding=(2, 2, 2, 2)),
    _dw("m_c32_pad2_wide", (10, 37), 32, padding=(2, 2, 2, 2)),
    _dw("m_c80_17x17", (17, 17), 80, batch=2),                     # five blocks: one per wave
    _dw("m_c960_7x7", (7, 7), 960, batch=2),                       # MobileNetV2 layer 27 shape
    _dw("m_c32_strided_pixels", (11, 12), 32, input_pixel_stride=48, output_pixel_stride=64),
    _dw("m_c64_zp", (9, 9), 64, izp=255, kzp=128),
    _dw("m_c64_zp0", (9, 9), 64, izp=0, kzp=127),
    _dw("m_c32_qmin_qmax", (9, 9), 32, qmin=100, qmax=150),
    _dw("m_c32_112", (112, 112), 32),                              # MobileNetV2 layer 2 shape: 7 strips, row segments
    _dw("m_c144_56", (56, 56), 144, batch=2),                      # layer 8: three groups of three blocks
    _dw("m_c192_28", (28, 28), 192, batch=3),                      # layer 13
    _dw("m_c576_14", (14, 14), 576, batch=3),                      # layer 22
]


@pytest.fixture()
def m16(qnnp):
    qnnp.set_option("dwconv_kernel", 7)
    yield qnnp
    qnnp.set_option("dwconv_kernel", 0)


@pytest.mark.parametrize("kzp", [127, 128], ids=lambda v: f"kzp{v}")
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_mfma16_walk_matches_oracle(m16, case, kzp):
    if case.name in ("m_c64_zp", "m_c64_zp0") and kzp != case.kzp:
        pytest.skip("the case's own zero point")
    case = dataclasses.replace(case, kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 2, 2, 0] = (0, 255) if kzp == 128 else (0, 254)     # the whole int8 range of the class
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(m16, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}, kzp {kzp}]")


@pytest.mark.parametrize("case", [
    _dw("m_bad_s2", (15, 15), 32, subsampling=(2, 2)),
    _dw("m_bad_c24", (9, 9), 24),                                   # channels not a multiple of 16
    _dw("m_bad_kzp100", (9, 9), 32, kzp=100),                       # weights outside both int8 classes
    _dw("m_bad_pixel_stride", (9, 9), 32, input_pixel_stride=40),   # pixels not 16-byte aligned
], ids=lambda c: c.name)
def test_mfma16_refuses_what_it_cannot_take(m16, case):
    inp, kernel, bias = conv_tensors(case)
    if case.kzp == 100:
        kernel[0, 0, 0, 0, 0], kernel[-1, 0, 2, 2, 0] = 0, 255
    _, quant, out_hw = conv_expected(case, inp, kernel, bias)
    with pytest.raises(QnnpackError):
        conv_run(m16, case, quant, out_hw, inp, kernel, bias, to_device, from_device)


@pytest.mark.parametrize("scale,zp,qmin,qmax", [
    (0.5, 127, 0, 255), (0.75, 3, 0, 255), (0.0125, 127, 0, 255), (0.0125, 0, 10, 240), (2.0 ** -9, 255, 0, 255),
    (0.3, 128, 128, 255), (float.fromhex("0x1.FFFFFEp-1"), 200, 0, 255), (2.0 ** -24, 17, 0, 255)],
    ids=lambda v: str(v))
def test_mfma16_requantization_flavours(m16, scale, zp, qmin, qmax):
    case = _dw("m_rq", (19, 18), 48, batch=2)
    inp, kernel, bias = conv_tensors(case)
    shape = o1.conv_shape(case.batch, 19, 18, case.padding, (3, 3), (1, 1), (1, 1), 48, 1, 1, 48)
    acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
    expected = o1.requantize_rows(acc.reshape(-1, 48), np.float32(scale), zp, qmin, qmax).reshape(-1)
    op = m16.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 48, 1, 1, case.izp, float(np.float32(scale)), case.kzp, 1.0,
                                          kernel, bias, zp, 1.0, qmin, qmax, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        m16.setup_convolution2d_nhwc_q8(op, case.batch, 19, 18, d_in, 48, d_out, 48)
        m16.run_operator(op)
        assert m16.operator_kernel(op) == KERNEL
        out = from_device(d_out)
    finally:
        m16.delete_operator(op)
    assert_bytes_equal(out, expected, f"requantization scale {scale}, zero point {zp}, clamp [{qmin}, {qmax}]")
