"""GPU tier: the barrier-free streaming kernel for convolutions over 3-channel images
(q8_conv_stream_c3_kernel in qnnpack_amd/csrc/hip/q8pwconv.hip), forced with "gemm_kernel" = 7, against the
scalar oracle: first-layer shapes, strides, paddings, pixel strides, 1..16 taps, ragged pixel blocks and
channel counts, zero points; the very last pixel of the tensor (read bytewise) is covered by every case."""
import pytest

from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run

pytestmark = pytest.mark.gpu
KERNEL = "q8_conv_stream_c3_mfma"


def _pad(h, w):
    return (h, w, h, w)


CASES = [
    ConvCase("s_3x3_s2_first_layer", (32, 32), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32, batch=2),
    ConvCase("s_3x3_s1_pad", (13, 11), (3, 3), _pad(1, 1), gic=3, goc=24, batch=2),
    ConvCase("s_3x3_nopad", (9, 9), (3, 3), gic=3, goc=32),
    ConvCase("s_3x3_pixel_stride5", (9, 10), (3, 3), _pad(1, 1), gic=3, goc=40, input_pixel_stride=5),
    ConvCase("s_3x3_out_stride", (8, 8), (3, 3), _pad(1, 1), gic=3, goc=16, output_pixel_stride=20),
    ConvCase("s_1x1_s2", (9, 9), (1, 1), subsampling=(2, 2), gic=3, goc=33),
    ConvCase("s_2x2_s2", (10, 12), (2, 2), subsampling=(2, 2), gic=3, goc=8, batch=3),
    ConvCase("s_4x4_16taps", (11, 10), (4, 4), (1, 2, 2, 1), gic=3, goc=64),
    ConvCase("s_1x7", (6, 20), (1, 7), (0, 3, 0, 3), gic=3, goc=32),
    ConvCase("s_3x3_d2", (12, 12), (3, 3), _pad(2, 2), dilation=(2, 2), gic=3, goc=32),
    ConvCase("s_3x3_zp", (10, 10), (3, 3), _pad(1, 1), gic=3, goc=32, izp=9, kzp=200),
    ConvCase("s_3x3_zp_extremes", (10, 10), (3, 3), _pad(1, 1), gic=3, goc=32, izp=255, kzp=0),
    ConvCase("s_3x3_clamp", (10, 10), (3, 3), _pad(1, 1), gic=3, goc=32, qmin=90, qmax=160),
    ConvCase("s_3x3_n96", (7, 7), (3, 3), _pad(1, 1), gic=3, goc=96, batch=2),
    ConvCase("s_3x3_224", (224, 224), (3, 3), _pad(1, 1), subsampling=(2, 2), gic=3, goc=32),
]


@pytest.fixture()
def c3(qnnp):
    qnnp.set_option("gemm_kernel", 7)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c.name)
def test_stream_kernel_matches_oracle(c3, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(c3, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("s_bad_5x5_25taps", (17, 15), (5, 5), _pad(2, 2), gic=3, goc=16),
    ConvCase("s_bad_4_channels", (9, 9), (3, 3), _pad(1, 1), gic=4, goc=16),
], ids=lambda c: c.name)
def test_unsupported_shapes_are_reported_not_silently_rerouted(c3, case):
    from qnnpack_amd import QnnpackError
    expected, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(c3, case, quant, out_hw, to_device=to_device, from_device=from_device)
