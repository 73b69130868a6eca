"""GPU tier: the weight-stationary 3x3 kernel with 16 / 32 / 48 / 64 input channels (q8_conv_ws16s_kernel in
qnnpack_amd/csrc/hip/q8convws16s.hip, "gemm_kernel" = 32; round 6) against the scalar oracle: every channel pairing of SqueezeNet's fire
modules (bench/convolution.cc:543-640) and the ones between, images whose sides are multiples of nothing (units hang over the right and
bottom edges, patches over all four), every padding side, several images, pixel strides wider than the channel count, input zero points
and clamps, kernel zero points 127 (the centred image convolution.c builds) and 128 (the standard image is the centred one). Reference
path: q8conv under qnnp_run_operator (src/q8conv/4x4c2-sse2.c:14-273, src/operator-run.c:183-217)."""
import pytest

from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu
KERNEL = "q8_conv_ws16s_mfma"


def _c(name, hw, cin, cout, pad=(1, 1, 1, 1), **kw):
    return ConvCase(name, hw, (3, 3), pad, gic=cin, goc=cout, **kw)


CASES = [
    _c("s_16_64_fire2", (55, 55), 16, 64, batch=2),
    _c("s_32_128_fire4", (27, 27), 32, 128, batch=3),
    _c("s_48_192_fire6", (13, 13), 48, 192, batch=4),
    _c("s_64_256_fire8", (13, 13), 64, 256, batch=2),
    _c("s_16_16", (9, 11), 16, 16, batch=2),
    _c("s_16_32", (9, 11), 16, 32, batch=2),
    _c("s_16_48", (9, 11), 16, 48, batch=2),
    _c("s_32_64", (12, 9), 32, 64, batch=3),
    _c("s_48_48", (7, 17), 48, 48, batch=2),
    _c("s_48_128", (10, 10), 48, 128, batch=2),
    _c("s_64_64", (11, 13), 64, 64, batch=2),
    _c("s_64_192", (8, 8), 64, 192, batch=2),
    _c("s_16_256", (8, 8), 16, 256, batch=2),
    _c("s_one_pixel_images", (1, 1), 32, 64, batch=70),
    _c("s_one_row", (1, 37), 16, 64, batch=3),
    _c("s_one_column", (29, 1), 48, 64, batch=3),
    _c("s_nopad", (12, 14), 32, 64, pad=(0, 0, 0, 0), batch=2),
    _c("s_pad_right_bottom_only", (10, 12), 16, 64, pad=(0, 2, 2, 0), batch=2),
    _c("s_pad_left_top_2", (10, 12), 48, 64, pad=(2, 0, 0, 2), batch=2),
    _c("s_strided_pixels", (9, 9), 16, 64, input_pixel_stride=48, output_pixel_stride=96, batch=2),
    _c("s_izp_9", (9, 9), 48, 192, izp=9, batch=2),
    _c("s_izp_255", (9, 9), 16, 64, izp=255, batch=2),
    _c("s_kzp128", (9, 9), 16, 64, kzp=128, izp=3, batch=2),
    _c("s_kzp128_48", (9, 9), 48, 128, kzp=128, batch=2),
    _c("s_clamp", (9, 9), 32, 128, qmin=90, qmax=160, batch=2),
    _c("s_many_units", (40, 48), 16, 64, batch=24),
]


@pytest.fixture()
def ws16s(qnnp):
    qnnp.set_option("gemm_kernel", 32)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_small_channel_weight_stationary_kernel_matches_oracle(ws16s, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(ws16s, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    _c("s_bad_kzp126", (9, 9), 16, 64, kzp=126),                               # no centred image: the row term would be needed
    _c("s_bad_24_channels", (9, 9), 24, 64),
    _c("s_bad_80_outputs", (9, 9), 16, 80),
    _c("s_bad_stride_2", (9, 9), 16, 64, subsampling=(2, 2)),
    ConvCase("s_bad_5x5", (9, 9), (5, 5), (2, 2, 2, 2), gic=16, goc=64),
    _c("s_bad_unaligned_pixels", (9, 9), 16, 64, output_pixel_stride=72),
], ids=lambda c: c.name)
def test_unsupported_shapes_are_reported_not_silently_rerouted(ws16s, case):
    from qnnpack_amd import QnnpackError
    expected, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(ws16s, case, quant, out_hw, to_device=to_device, from_device=from_device)


def test_automatic_dispatch_takes_it_for_a_fire_module(qnnp):
    case = _c("s_auto_16_64", (55, 55), 16, 64, batch=6)                      # 18150 rows
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} (automatic) vs oracle")
