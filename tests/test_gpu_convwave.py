"""GPU tier: the wave-per-8x8-block direct-convolution MFMA kernel (qnnpack_amd/csrc/hip/q8convwave.hip), forced
with "gemm_kernel" = 8, against the scalar oracle: 32 / 64 input channels, 32 / 64 output channels, windows whose
patch fits (3x3, 1x3, 3x1, 2x2, dilated 3x3 at 32 channels), asymmetric padding, image sizes that do not fill the
8x8 blocks, many units per workgroup, zero-point and clamp variants; every case with kernel zero points 127 and 128
(the zero-point-centred image of round 4: the weight-stationary kernel then runs without pixel sums and reports
`q8_conv_wave_ws_c_mfma`) and with one that has no centred image."""
import dataclasses

import pytest

from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu


# "gemm_kernel" = 8: the family with its default 3x3 / stride 1 flavour (weights in registers, round 3);
# "gemm_kernel" = 12: the same with the round-2 register-path kernel kept for A/B
@pytest.fixture(params=[8, 12], ids=["ws", "reg"])
def waveconv(qnnp, request):
    qnnp.set_option("gemm_kernel", request.param)
    qnnp._variant = request.param
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _is_k33(case):
    return case.kernel_size == (3, 3) and case.subsampling == (1, 1) and case.dilation == (1, 1)


CASES = [
    ConvCase("w_3x3_c64_n64", (12, 10), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=3),
    ConvCase("w_3x3_c64_n64_56", (56, 56), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=2),
    ConvCase("w_3x3_c64_n64_many_units", (30, 31), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=150),
    ConvCase("w_3x3_c32_n64", (17, 19), (3, 3), (1, 1, 1, 1), gic=32, goc=64, batch=2),
    ConvCase("w_3x3_c64_n32", (9, 14), (3, 3), (1, 1, 1, 1), gic=64, goc=32, batch=2),
    ConvCase("w_3x3_c32_n32_d2", (15, 13), (3, 3), (2, 2, 2, 2), dilation=(2, 2), gic=32, goc=32),
    ConvCase("w_1x3_c64_n64", (8, 21), (1, 3), (0, 1, 0, 1), gic=64, goc=64, batch=2),
    ConvCase("w_3x1_c32_n32", (13, 13), (3, 1), (1, 0, 1, 0), gic=32, goc=32),
    ConvCase("w_2x2_c64_n64", (9, 11), (2, 2), (0, 0, 1, 1), gic=64, goc=64),
    ConvCase("w_3x3_nopad", (11, 12), (3, 3), gic=64, goc=32, batch=2),
    ConvCase("w_3x3_asym_pad", (10, 9), (3, 3), (2, 0, 0, 1), gic=32, goc=64),
    ConvCase("w_3x3_zp", (9, 9), (3, 3), (1, 1, 1, 1), gic=64, goc=64, izp=3, kzp=250),
    ConvCase("w_3x3_zp0", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=32, izp=0, kzp=0),
    ConvCase("w_3x3_zp255", (9, 9), (3, 3), (1, 1, 1, 1), gic=64, goc=64, izp=255, kzp=255),
    ConvCase("w_3x3_qmin_qmax", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=64, qmin=64, qmax=192),
    ConvCase("w_3x3_strided_input", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=32, input_pixel_stride=48),
    ConvCase("w_3x3_tall", (70, 5), (3, 3), (1, 1, 1, 1), gic=32, goc=32),
    ConvCase("w_3x3_wide", (4, 300), (3, 3), (1, 1, 1, 1), gic=32, goc=32),
    ConvCase("w_3x3_one_pixel", (1, 1), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=5),
]


@pytest.mark.parametrize("kzp", [None, 127, 128, 77], ids=["kzp_of_case", "kzp127", "kzp128", "kzp77"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_wave_convolution_matches_oracle(waveconv, case, kzp):
    if kzp is not None:
        if case.kzp == kzp:
            pytest.skip("the case's own zero point")
        case = dataclasses.replace(case, kzp=kzp)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(waveconv, case, quant, out_hw, to_device=to_device, from_device=from_device)
    # (round 6: 64 input channels with a centred image run on the 16x16x64 flavour, q8_conv_wave_ws16_kernel)
    ws = ("q8_conv_wave_ws_c16_mfma" if case.gic == 64 else "q8_conv_wave_ws_c_mfma") if case.kzp in (127, 128) else "q8_conv_wave_ws_mfma"
    want = ws if (waveconv._variant == 8 and _is_k33(case)) else "q8_conv_wave_mfma"
    assert kname == want, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}, kzp {case.kzp}]")


@pytest.mark.parametrize("case", [
    ConvCase("w_bad_c128", (9, 9), (3, 3), (1, 1, 1, 1), gic=128, goc=32),
    ConvCase("w_bad_n96", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=96),
    ConvCase("w_bad_5x5_c64", (12, 12), (5, 5), (2, 2, 2, 2), gic=64, goc=64),
    ConvCase("w_bad_strided_output", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=32, output_pixel_stride=48),
], ids=lambda c: c.name)
def test_unsupported_shape_is_reported(waveconv, case):
    from qnnpack_amd import QnnpackError
    _, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(waveconv, case, quant, out_hw, to_device=to_device, from_device=from_device)
