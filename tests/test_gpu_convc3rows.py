"""GPU tier: the first-layer kernel that fetches one 16-byte row slot per kernel row (q8_conv_c3rows_kernel in
qnnpack_amd/csrc/hip/q8convc3.hip), forced with "gemm_kernel" = 14, against the scalar oracle: first-layer shapes,
strides 1 / 2 / 3, every padding side (so that windows hang over each border and over corners), window heights 1..4 and
widths 1..5, 16 / 32 / 48 / 64 output channels, ragged last unit, zero points and clamps, batches whose images meet
inside a 32-pixel unit, the tensor's first and last pixel (the slot of the last pixel reads past the tensor's end: the
descriptor returns zeros there, and those bytes meet zero weights)."""
import pytest

from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu
KERNEL = "q8_conv_c3rows_mfma"


def _pad(h, w):
    return (h, w, h, w)


CASES = [
    ConvCase("r_3x3_s2_first_layer", (32, 32), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=2),
    ConvCase("r_3x3_s2_224", (224, 224), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32),
    ConvCase("r_3x3_s1_pad", (13, 11), (3, 3), _pad(1, 1), gic=3, goc=32, batch=3),
    ConvCase("r_3x3_nopad", (9, 9), (3, 3), gic=3, goc=32),
    ConvCase("r_3x3_s2_odd_images_meet_in_a_unit", (9, 7), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=5),
    ConvCase("r_3x3_pad_right_bottom_only", (10, 12), (3, 3), (0, 2, 2, 0), gic=3, goc=32, batch=2),
    ConvCase("r_3x3_pad_left_top_2", (10, 12), (3, 3), (2, 0, 0, 2), gic=3, goc=32, batch=2),
    ConvCase("r_3x3_s3", (17, 19), (3, 3), _pad(1, 1), subsampling=(3, 3), gic=3, goc=32),
    ConvCase("r_3x3_s2x1", (16, 9), (3, 3), _pad(1, 1), subsampling=(2, 1), gic=3, goc=32),
    ConvCase("r_1x1_s2", (9, 9), (1, 1), subsampling=(2, 2), gic=3, goc=32),
    ConvCase("r_2x2_s2", (10, 12), (2, 2), subsampling=(2, 2), gic=3, goc=16, batch=3),
    # round 6: channel counts in multiples of 8 (the last half piece leaves by an 8-byte store): ShuffleNet's 3 -> 24 first layer
    ConvCase("r_3x3_s2_24_channels", (40, 36), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=24, batch=3),
    ConvCase("r_3x3_s2_24_channels_flat_rows_odd_height", (41, 64), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=24, batch=3),
    ConvCase("r_3x3_s1_24_channels_strided_pixels", (12, 20), (3, 3), _pad(1, 1), gic=3, goc=24, output_pixel_stride=32, batch=2),
    ConvCase("r_3x3_s1_8_channels", (13, 11), (3, 3), _pad(1, 1), gic=3, goc=8, batch=2),
    ConvCase("r_3x3_s2_56_channels_zp", (17, 19), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=56, izp=9, kzp=200, batch=2),
    ConvCase("r_4x4_pad", (11, 10), (4, 4), (1, 2, 2, 1), gic=3, goc=64),
    ConvCase("r_4x5_wide_window", (12, 14), (4, 5), (1, 2, 2, 2), gic=3, goc=48, batch=2),
    ConvCase("r_1x5", (6, 20), (1, 5), (0, 2, 0, 2), gic=3, goc=32),
    ConvCase("r_3x1_tall", (20, 5), (3, 1), (1, 0, 1, 0), gic=3, goc=32),
    ConvCase("r_window_wider_than_image", (5, 2), (3, 3), _pad(1, 1), gic=3, goc=32, batch=4),
    ConvCase("r_one_pixel_image", (1, 1), (3, 3), _pad(1, 1), gic=3, goc=32, batch=70),
    ConvCase("r_3x3_n16", (9, 10), (3, 3), _pad(1, 1), gic=3, goc=16),
    ConvCase("r_3x3_n64", (9, 10), (3, 3), _pad(1, 1), gic=3, goc=64, batch=2),
    ConvCase("r_3x3_out_stride", (8, 8), (3, 3), _pad(1, 1), gic=3, goc=32, output_pixel_stride=48),
    ConvCase("r_3x3_zp", (10, 10), (3, 3), _pad(1, 1), gic=3, goc=32, izp=9, kzp=200),
    ConvCase("r_3x3_zp_extremes", (10, 10), (3, 3), _pad(1, 1), gic=3, goc=32, izp=255, kzp=0),
    ConvCase("r_3x3_zp_extremes2", (10, 10), (3, 3), _pad(1, 1), gic=3, goc=32, izp=0, kzp=255),
    ConvCase("r_3x3_clamp", (10, 10), (3, 3), _pad(1, 1), gic=3, goc=32, qmin=90, qmax=160),
    ConvCase("r_3x3_many_units", (64, 48), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=40),
]


@pytest.fixture()
def rows16(qnnp):
    qnnp.set_option("gemm_kernel", 14)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_row_slot_kernel_matches_oracle(rows16, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(rows16, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("r_bad_pixel_stride", (9, 10), (3, 3), _pad(1, 1), gic=3, goc=32, input_pixel_stride=5),
    ConvCase("r_bad_1x7", (6, 20), (1, 7), (0, 3, 0, 3), gic=3, goc=32),
    ConvCase("r_bad_6x3", (12, 12), (6, 3), (2, 1, 3, 1), gic=3, goc=32),
    ConvCase("r_bad_7x7_tensor_not_whole_dwords", (9, 9), (7, 7), _pad(3, 3), gic=3, goc=64),        # 243 bytes
    ConvCase("r_bad_7x11", (16, 16), (7, 11), _pad(3, 5), gic=3, goc=32),                              # 33-byte window rows
    ConvCase("r_bad_dilated", (12, 12), (3, 3), _pad(2, 2), dilation=(2, 2), gic=3, goc=32),
    ConvCase("r_bad_n20", (9, 9), (3, 3), _pad(1, 1), gic=3, goc=20),
    ConvCase("r_bad_n96", (9, 9), (3, 3), _pad(1, 1), gic=3, goc=96),
    ConvCase("r_bad_4_channels", (9, 9), (3, 3), _pad(1, 1), gic=4, goc=32),
], ids=lambda c: c.name)
def test_unsupported_shapes_are_reported_not_silently_rerouted(rows16, case):
    from qnnpack_amd import QnnpackError
    expected, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(rows16, case, quant, out_hw, to_device=to_device, from_device=from_device)


def test_automatic_dispatch_takes_the_row_slot_kernel_for_the_first_layer(qnnp):
    case = ConvCase("r_auto_112", (112, 112), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=2)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL16L, kname                           # (rows of whole chunks, kernel zero point 127: the LDS-staged flavour)
    assert_bytes_equal(out, expected, f"gfx950 {kname} (automatic) vs oracle")
    for case in (ConvCase("r_auto_100", (100, 100), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=2),        # rows of 300 bytes
                 ConvCase("r_auto_112_kzp9", (112, 112), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=2, kzp=9)):   # a row term
        expected, quant, out_hw = conv_expected(case)
        out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
        assert kname == KERNEL, kname
        assert_bytes_equal(out, expected, f"gfx950 {kname} (automatic) vs oracle")


# ---- the LDS-staged flavour of the 16-byte-slot kernel (q8_conv_c3rows_lds_kernel, round 6; "gemm_kernel" = 30) ----
KERNEL16L = "q8_conv_c3rows_lds_mfma"
CASES16L = [
    ConvCase("q_3x3_s2_224", (224, 224), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32),
    ConvCase("q_3x3_s2_first_layer", (32, 32), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=2),
    ConvCase("q_3x3_s1_pad", (13, 16), (3, 3), _pad(1, 1), gic=3, goc=32, batch=3),
    ConvCase("q_3x3_nopad", (9, 16), (3, 3), gic=3, goc=32),
    ConvCase("q_3x3_s2_odd_output_rows", (30, 32), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=5),
    ConvCase("q_3x3_pad_right_bottom_only", (10, 16), (3, 3), (0, 2, 2, 0), gic=3, goc=32, batch=2),
    ConvCase("q_3x3_pad_left_top_2", (10, 16), (3, 3), (2, 0, 0, 2), gic=3, goc=32, batch=2),
    ConvCase("q_3x3_s3", (17, 32), (3, 3), _pad(1, 1), subsampling=(3, 3), gic=3, goc=32),
    ConvCase("q_3x3_s2x1", (16, 16), (3, 3), _pad(1, 1), subsampling=(2, 1), gic=3, goc=32),
    ConvCase("q_1x1_s2", (9, 16), (1, 1), subsampling=(2, 2), gic=3, goc=32),
    ConvCase("q_2x2_s2", (10, 16), (2, 2), subsampling=(2, 2), gic=3, goc=16, batch=3),
    ConvCase("q_3x3_s2_24_channels", (40, 48), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=24, batch=3),
    ConvCase("q_3x3_s2_24_channels_flat_rows_odd_height", (41, 64), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=24, batch=3),
    ConvCase("q_3x3_s1_8_channels", (13, 16), (3, 3), _pad(1, 1), gic=3, goc=8, batch=2),
    ConvCase("q_3x3_s2_56_channels_izp", (17, 32), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=56, izp=9, batch=2),
    ConvCase("q_4x4_pad", (11, 16), (4, 4), (1, 2, 2, 1), gic=3, goc=64),
    ConvCase("q_4x5_wide_window", (12, 16), (4, 5), (1, 2, 2, 2), gic=3, goc=48, batch=2),
    ConvCase("q_1x5", (6, 32), (1, 5), (0, 2, 0, 2), gic=3, goc=32),
    ConvCase("q_3x1_tall", (20, 16), (3, 1), (1, 0, 1, 0), gic=3, goc=32),
    ConvCase("q_one_row_images", (1, 16), (3, 3), _pad(1, 1), gic=3, goc=32, batch=70),
    ConvCase("q_3x3_n64_whole_lines", (9, 16), (3, 3), _pad(1, 1), gic=3, goc=64, batch=2),
    ConvCase("q_3x3_s1_224_n64_vgg", (224, 224), (3, 3), _pad(1, 1), gic=3, goc=64),
    ConvCase("q_3x3_out_stride", (8, 16), (3, 3), _pad(1, 1), gic=3, goc=32, output_pixel_stride=48),
    ConvCase("q_3x3_kzp128", (10, 16), (3, 3), _pad(1, 1), gic=3, goc=32, kzp=128, izp=200),
    ConvCase("q_3x3_izp_extremes", (10, 16), (3, 3), _pad(1, 1), gic=3, goc=32, izp=255),
    ConvCase("q_3x3_izp_0", (10, 16), (3, 3), _pad(1, 1), gic=3, goc=32, izp=0),
    ConvCase("q_3x3_clamp", (10, 16), (3, 3), _pad(1, 1), gic=3, goc=32, qmin=90, qmax=160),
    ConvCase("q_3x3_many_bands", (96, 48), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=24),
]


@pytest.mark.parametrize("case", CASES16L, ids=lambda c: c.name)
def test_lds_staged_row_slot_kernel_matches_oracle(rows32lds, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(rows32lds, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL16L, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("q_bad_row_term", (10, 16), (3, 3), _pad(1, 1), gic=3, goc=32, kzp=200),        # the LDS flavour has no row term
    ConvCase("q_bad_w20", (12, 20), (3, 3), _pad(1, 1), gic=3, goc=32),                       # rows of 60 bytes
], ids=lambda c: c.name)
def test_lds_staged_row_slot_kernel_refuses_what_it_cannot_take(rows32lds, case):
    from qnnpack_amd import QnnpackError
    _, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(rows32lds, case, quant, out_hw, to_device=to_device, from_device=from_device)


# ---- the 32-byte-slot flavour (q8_conv_c3rows32_kernel, round 5): 5- and 7-row windows of up to 32 bytes per row ----
KERNEL32 = "q8_conv_c3rows32_mfma"
CASES32 = [
    ConvCase("w_7x7_s2_resnet_entry", (224, 224), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64),
    ConvCase("w_7x7_s2_small", (32, 32), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=2),
    ConvCase("w_7x7_s1", (12, 16), (7, 7), _pad(3, 3), gic=3, goc=64, batch=3),
    ConvCase("w_7x7_nopad", (12, 12), (7, 7), gic=3, goc=32),
    ConvCase("w_7x7_pad_right_bottom_only", (10, 12), (7, 7), (0, 6, 6, 0), gic=3, goc=64, batch=2),
    ConvCase("w_7x7_pad_left_top_only", (10, 12), (7, 7), (6, 0, 0, 6), gic=3, goc=64, batch=2),
    ConvCase("w_7x7_images_meet_in_a_unit", (6, 6), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=8),
    ConvCase("w_7x7_window_wider_than_image", (4, 2), (7, 7), _pad(3, 3), gic=3, goc=32, batch=4),
    ConvCase("w_7x7_one_pixel_images", (1, 1), (7, 7), _pad(3, 3), gic=3, goc=64, batch=72),
    ConvCase("w_7x7_s3", (20, 20), (7, 7), _pad(3, 3), subsampling=(3, 3), gic=3, goc=48),
    ConvCase("w_7x10_widest_row", (16, 16), (7, 10), (3, 5, 3, 4), gic=3, goc=32),                     # 30-byte window rows
    ConvCase("w_7x6_row_of_18", (16, 16), (7, 6), (3, 3, 3, 2), gic=3, goc=16, batch=2),
    ConvCase("w_5x5_s1", (14, 14), (5, 5), _pad(2, 2), gic=3, goc=32, batch=2),
    ConvCase("w_5x5_s2_n64", (28, 28), (5, 5), _pad(2, 2), subsampling=(2, 2), gic=3, goc=64),
    ConvCase("w_5x3", (12, 12), (5, 3), (2, 1, 2, 1), gic=3, goc=32),
    ConvCase("w_5x9", (12, 16), (5, 9), (2, 4, 2, 4), gic=3, goc=32),
    ConvCase("w_7x7_out_stride", (8, 8), (7, 7), _pad(3, 3), gic=3, goc=32, output_pixel_stride=48),
    ConvCase("w_7x7_zp", (10, 10), (7, 7), _pad(3, 3), gic=3, goc=64, izp=9, kzp=200, batch=2),
    ConvCase("w_7x7_zp_extremes", (10, 10), (7, 7), _pad(3, 3), gic=3, goc=64, izp=255, kzp=0, batch=2),
    ConvCase("w_7x7_zp_extremes2", (10, 10), (7, 7), _pad(3, 3), gic=3, goc=64, izp=0, kzp=255, batch=2),
    ConvCase("w_7x7_kzp128_no_row_term", (10, 10), (7, 7), _pad(3, 3), gic=3, goc=64, kzp=128, batch=2),
    ConvCase("w_7x7_clamp", (10, 10), (7, 7), _pad(3, 3), gic=3, goc=64, qmin=90, qmax=160, batch=2),
    ConvCase("w_7x7_many_units", (64, 48), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=40),
    # round 6: three channel blocks (SqueezeNet 1.0's 3 -> 96 entry layer, bench/convolution.cc:541)
    ConvCase("w_7x7_s2_96_channels", (48, 40), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=96, batch=3),
    ConvCase("w_7x7_80_channels_zp", (10, 12), (7, 7), _pad(3, 3), gic=3, goc=80, izp=9, kzp=200, batch=2),
    ConvCase("w_7x7_nopad_96", (64, 64), (7, 7), subsampling=(2, 2), gic=3, goc=96, batch=2),
]


@pytest.mark.parametrize("case", CASES32, ids=lambda c: c.name)
def test_wide_row_slot_kernel_matches_oracle(rows16, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(rows16, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL32, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


# ---- its LDS-staged flavour (q8_conv_c3rows32_lds_kernel, round 6; "gemm_kernel" = 30): image rows of whole 16-byte chunks
#      (W % 16 == 0), a band of output-row pairs per workgroup with its input rows -- padding materialised -- in LDS ----
KERNEL32L = "q8_conv_c3rows32_lds_mfma"
CASES32L = [
    ConvCase("l_7x7_s2_resnet_entry", (224, 224), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64),
    ConvCase("l_7x7_s2_small", (32, 32), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=2),
    ConvCase("l_7x7_s1", (12, 16), (7, 7), _pad(3, 3), gic=3, goc=64, batch=3),
    ConvCase("l_7x7_nopad", (12, 16), (7, 7), gic=3, goc=32),
    ConvCase("l_7x7_pad_right_bottom_only", (10, 16), (7, 7), (0, 6, 6, 0), gic=3, goc=64, batch=2),
    ConvCase("l_7x7_pad_left_top_only", (10, 16), (7, 7), (6, 0, 0, 6), gic=3, goc=64, batch=2),
    ConvCase("l_7x7_odd_output_rows", (30, 32), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=5),
    ConvCase("l_7x7_one_output_row", (2, 16), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=32, batch=9),
    ConvCase("l_7x7_s3", (20, 32), (7, 7), _pad(3, 3), subsampling=(3, 3), gic=3, goc=48),
    ConvCase("l_7x7_s2x1", (20, 32), (7, 7), _pad(3, 3), subsampling=(2, 1), gic=3, goc=32, batch=2),
    ConvCase("l_7x10_widest_row", (16, 16), (7, 10), (3, 5, 3, 4), gic=3, goc=32),                     # 30-byte window rows
    ConvCase("l_7x6_row_of_18", (16, 16), (7, 6), (3, 3, 3, 2), gic=3, goc=16, batch=2),
    ConvCase("l_7x7_wide_left_pad", (16, 16), (7, 7), (3, 1, 3, 6), gic=3, goc=32, batch=2),           # 18 bytes of left padding: two chunks
    ConvCase("l_5x5_s1", (14, 16), (5, 5), _pad(2, 2), gic=3, goc=32, batch=2),
    ConvCase("l_5x5_s2_n64", (28, 32), (5, 5), _pad(2, 2), subsampling=(2, 2), gic=3, goc=64),
    ConvCase("l_5x3", (12, 16), (5, 3), (2, 1, 2, 1), gic=3, goc=32),
    ConvCase("l_5x9", (12, 16), (5, 9), (2, 4, 2, 4), gic=3, goc=32),
    ConvCase("l_7x7_out_stride", (8, 16), (7, 7), _pad(3, 3), gic=3, goc=32, output_pixel_stride=48),
    ConvCase("l_7x7_zp", (10, 16), (7, 7), _pad(3, 3), gic=3, goc=64, izp=9, kzp=200, batch=2),
    ConvCase("l_7x7_zp_extremes", (10, 16), (7, 7), _pad(3, 3), gic=3, goc=64, izp=255, kzp=0, batch=2),
    ConvCase("l_7x7_zp_extremes2", (10, 16), (7, 7), _pad(3, 3), gic=3, goc=64, izp=0, kzp=255, batch=2),
    ConvCase("l_7x7_kzp128_no_row_term", (10, 16), (7, 7), _pad(3, 3), gic=3, goc=64, kzp=128, batch=2),
    ConvCase("l_7x7_clamp", (10, 16), (7, 7), _pad(3, 3), gic=3, goc=64, qmin=90, qmax=160, batch=2),
    ConvCase("l_7x7_many_bands", (96, 48), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=24),
    ConvCase("l_7x7_s1_tall_many_bands", (70, 16), (7, 7), _pad(3, 3), gic=3, goc=32, batch=3),
    ConvCase("l_7x7_s2_96_channels", (48, 48), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=96, batch=3),
    ConvCase("l_7x7_80_channels_zp", (10, 16), (7, 7), _pad(3, 3), gic=3, goc=80, izp=9, kzp=200, batch=2),
    ConvCase("l_7x7_nopad_96", (64, 64), (7, 7), subsampling=(2, 2), gic=3, goc=96, batch=2),
    ConvCase("l_7x7_widest_image_of_the_plan", (9, 512), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=32),
]


@pytest.fixture()
def rows32lds(qnnp):
    qnnp.set_option("gemm_kernel", 30)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("case", CASES32L, ids=lambda c: c.name)
def test_lds_staged_wide_row_slot_kernel_matches_oracle(rows32lds, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(rows32lds, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL32L, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def test_lds_staged_flavour_refuses_rows_that_are_not_whole_chunks(rows32lds):
    """W * 3 % 16 != 0: the staging's 16-byte loads would straddle rows -- forced, the flavour refuses; auto keeps the register-path kernel"""
    from qnnpack_amd import QnnpackError
    case = ConvCase("l_bad_w20", (12, 20), (7, 7), _pad(3, 3), gic=3, goc=32)
    _, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(rows32lds, case, quant, out_hw, to_device=to_device, from_device=from_device)


def test_automatic_dispatch_takes_the_wide_row_slot_kernel_for_a_7x7_entry_layer(qnnp):
    case = ConvCase("w_auto_7x7", (64, 64), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=2)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL32L, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} (automatic) vs oracle")
    case = ConvCase("w_auto_7x7_w40", (64, 40), (7, 7), _pad(3, 3), subsampling=(2, 2), gic=3, goc=64, batch=4)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL32, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} (automatic) vs oracle")
