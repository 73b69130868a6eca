"""GPU tier: the LDS-tiled direct-convolution MFMA kernel (qnnpack_amd/csrc/hip/q8convlds.hip), forced with
"gemm_kernel" = 3, against the scalar oracle: channel counts 32..256, strides, dilations, asymmetric padding,
position counts that do not fill the 256-position workgroups, zero-point and clamp variants."""
import numpy as np
import pytest

from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ldsconv(qnnp):
    qnnp.set_option("gemm_kernel", 3)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


CASES = [
    ConvCase("l_3x3_c64_n64", (12, 10), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=3),
    ConvCase("l_3x3_c64_n64_56", (56, 56), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=2),
    ConvCase("l_3x3_c32_n96_s2", (17, 19), (3, 3), (1, 1, 1, 1), subsampling=(2, 2), gic=32, goc=96, batch=2),
    ConvCase("l_3x3_c128_n32", (9, 14), (3, 3), (1, 1, 1, 1), gic=128, goc=32, batch=2),
    ConvCase("l_3x3_c256_n32", (7, 6), (3, 3), (1, 1, 1, 1), gic=256, goc=32),
    ConvCase("l_3x3_c64_n64_many_items", (30, 31), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=70),
    ConvCase("l_5x5_c32_n64_d2", (15, 13), (5, 5), (4, 4, 4, 4), dilation=(2, 2), gic=32, goc=64),
    ConvCase("l_1x3_c64_n128", (8, 21), (1, 3), (0, 1, 0, 1), gic=64, goc=128, batch=2),
    ConvCase("l_3x1_c32_s1x2", (13, 13), (3, 1), (1, 0, 1, 0), subsampling=(1, 2), gic=32, goc=32),
    ConvCase("l_3x3_nopad", (11, 12), (3, 3), gic=64, goc=32, batch=2),
    ConvCase("l_3x3_asym_pad", (10, 9), (3, 3), (2, 0, 0, 1), gic=32, goc=64),
    ConvCase("l_3x3_zp", (9, 9), (3, 3), (1, 1, 1, 1), gic=64, goc=64, izp=3, kzp=250),
    ConvCase("l_3x3_zp0", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=32, izp=0, kzp=0),
    ConvCase("l_3x3_qmin_qmax", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=64, qmin=64, qmax=192),
    ConvCase("l_3x3_strided_pixels", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=32, input_pixel_stride=48,
             output_pixel_stride=48),
    ConvCase("l_3x3_tall", (70, 5), (3, 3), (1, 1, 1, 1), gic=32, goc=32),
    ConvCase("l_3x3_wide", (4, 300), (3, 3), (1, 1, 1, 1), gic=32, goc=32),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_lds_convolution_matches_oracle(ldsconv, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(ldsconv, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == "q8_conv_lds_mfma", kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def test_unsupported_shape_is_reported(ldsconv):
    from qnnpack_amd import QnnpackError
    case = ConvCase("l_bad_c48", (9, 9), (3, 3), (1, 1, 1, 1), gic=48, goc=32)
    _, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(ldsconv, case, quant, out_hw, to_device=to_device, from_device=from_device)
