"""GPU tier: the matrix-core depthwise kernel (q8_dwconv_mfma_kernel in qnnpack_amd/csrc/hip/q8dwconv.hip:
diagonal MFMA operands, int8 weight parts), forced with "dwconv_kernel" = 4, against the scalar oracle:
3x3, strides, dilation, every padding form, ragged pixel tiles and channel blocks, pixel strides,
batch, and the kernel zero points that need one (128), two (typical) and three (0 with a weight of 255)
weight parts."""
import numpy as np
import pytest

from _cases import CONV_CASES, EXTRA_CONV_CASES, ConvCase, conv_tensors
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu
CONV_BY_NAME = {c.name: c for c in list(CONV_CASES) + list(EXTRA_CONV_CASES)}


def _dw(name, hw, c, k=3, **kw):
    kw.setdefault("padding", (k // 2,) * 4)
    return ConvCase(name, hw, (k, k), kw.pop("padding"), groups=c, gic=1, goc=1, **kw)


CASES = [
    _dw("m_c32_14", (14, 14), 32, batch=3),
    _dw("m_c16_1x1img", (1, 1), 16),
    _dw("m_c16_2x3img", (2, 3), 16, batch=2),
    _dw("m_c16_3x3img_nopad", (3, 3), 16, padding=(0, 0, 0, 0)),
    _dw("m_c48_9x40_wide", (9, 40), 48),
    _dw("m_c80_40x9_tall", (40, 9), 80, batch=2),
    _dw("m_c96_s2", (29, 31), 96, subsampling=(2, 2)),
    _dw("m_c144_s2_even", (28, 28), 144, subsampling=(2, 2), batch=2),
    _dw("m_c32_s2_pad_tl_only", (15, 15), 32, subsampling=(2, 2), padding=(1, 0, 0, 1)),
    _dw("m_c32_pad_asym", (12, 13), 32, padding=(1, 0, 1, 0)),
    _dw("m_c32_pad2", (10, 11), 32, padding=(2, 2, 2, 2)),
    _dw("m_c272_ragged_block", (7, 7), 272, batch=5),
    _dw("m_c960_7x7", (7, 7), 960, batch=2),
    _dw("m_c32_strided_pixels", (11, 12), 32, input_pixel_stride=48, output_pixel_stride=36),
    _dw("m_c64_d2", (13, 14), 64, padding=(2, 2, 2, 2), dilation=(2, 2)),
    _dw("m_c64_kzp128_one_part", (9, 9), 64, kzp=128),
    _dw("m_c64_kzp0_three_parts", (9, 9), 64, izp=255, kzp=0),
    _dw("m_c64_kzp255", (9, 9), 64, izp=0, kzp=255),
    _dw("m_c32_qmin_qmax", (9, 9), 32, qmin=100, qmax=150),
    _dw("m_c32_112", (112, 112), 32),
]


@pytest.fixture(params=[4, 5], ids=["gather", "lds"])
def mf(qnnp, request):
    """4 = tap operands gathered from global memory, 5 = band staged in LDS first"""
    qnnp.set_option("dwconv_kernel", request.param)
    yield qnnp
    qnnp.set_option("dwconv_kernel", 0)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_mfma_kernel_matches_oracle(mf, case):
    inp, kernel, bias = conv_tensors(case)
    if "three_parts" in case.name:
        kernel = kernel.copy(); kernel.reshape(-1)[::7] = 255     # x = w - kzp = 255 needs the third part
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(mf, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname.startswith("q8_dwconv_mfma_"), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("name", ["x_dw3x3_c20_vec4", "x_dw5x5_c64"])
def test_unsupported_shapes_are_reported_not_silently_rerouted(mf, name):
    from qnnpack_amd import QnnpackError
    case = CONV_BY_NAME[name]
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    with pytest.raises(QnnpackError):
        conv_run(mf, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
