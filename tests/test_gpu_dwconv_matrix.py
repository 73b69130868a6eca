"""GPU tier: the q8dwconv microkernel test matrix of the reference (test/q8dwconv.cc:731-963, the 19 up8x9__sse2 names:
single_output_channels_eq_8 [+ with_qmin, with_qmax, with_input_zero_point_only, with_kernel_zero_point_only],
multi_output_channels_eq_8 [+ with_subsampling, with_input_stride, with_output_stride], single/multi_output_channels_div_8
[+ with_output_stride], single_output_channels_gt_8 [+ qmin, qmax, input / kernel zero point only],
multi_output_channels_gt_8 [+ with_output_stride]; all ASSERT_EQ against the scalar q31 result,
test/dwconv-microkernel-tester.h) re-hosted on the whole-operator depthwise kernels through
qnnp_*_convolution2d_nhwc_q8 (groups = channels, one input / output channel per group).

The CPU kernel's channel tile cr = 8 becomes the device's channel granules: 4-channel dwords, 16-byte vectors, 32-channel
matrix-core blocks. "single output" = one output pixel, "multi" = an output row of 5 (and a 5 x 6 image). Every case
runs on the automatically selected kernel and forced onto each depthwise kernel that accepts the shape."""
import pytest

from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run
from qnnpack_amd import QnnpackError

pytestmark = pytest.mark.gpu

CR = 16   # device counterpart of the reference's cr = 8: one 16-byte channel vector


def dw(name, channels, width, height=1, **kw):
    """3x3 depthwise over an image whose 'valid' output is height x width (no padding, as the microkernel tester)"""
    s = kw.get("subsampling", (1, 1))[0]
    return ConvCase(name, ((height - 1) * s + 3, (width - 1) * s + 3), (3, 3), groups=channels, **kw)


def check(qnnp, case):
    expected, quant, out_hw = conv_expected(case)
    ran = []
    for variant in (0, 1, 2, 3, 4, 5, 6):
        qnnp.set_option("dwconv_kernel", variant)
        try:
            out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
        except QnnpackError:
            assert variant != 0, "the automatic choice must take every shape"
            continue
        finally:
            qnnp.set_option("dwconv_kernel", 0)
        assert_bytes_equal(out, expected, f"gfx950 {kname} (dwconv_kernel={variant}) vs oracle [{case.name}]")
        ran.append(kname)
    assert ran, case.name
    return ran


def test_single_output_channels_eq_cr(qnnp):
    ran = check(qnnp, dw("dm_1_c16", CR, 1))
    assert len(set(ran)) >= 3, ran          # the forced variants really are different kernels


@pytest.mark.parametrize("kw", [dict(qmin=128), dict(qmax=128), dict(izp=255, kzp=0), dict(izp=0, kzp=255)],
                         ids=["with_qmin", "with_qmax", "with_input_zero_point_only", "with_kernel_zero_point_only"])
def test_single_output_channels_eq_cr_variants(qnnp, kw):
    check(qnnp, dw("dm_1_c16_" + "_".join(f"{k}{v}" for k, v in kw.items()), CR, 1, **kw))


def test_multi_output_channels_eq_cr(qnnp):
    check(qnnp, dw("dm_5_c16", CR, 5))
    check(qnnp, dw("dm_5x6_c16", CR, 5, 6, batch=2))


def test_multi_output_channels_eq_cr_with_subsampling(qnnp):
    check(qnnp, dw("dm_5_c16_s2", CR, 5, subsampling=(2, 2)))
    check(qnnp, dw("dm_5x6_c16_s2", CR, 5, 6, subsampling=(2, 2), batch=2))


def test_multi_output_channels_eq_cr_with_input_stride(qnnp):
    check(qnnp, dw("dm_5_c16_in17", CR, 5, input_pixel_stride=17))
    check(qnnp, dw("dm_5_c16_in32", CR, 5, input_pixel_stride=32))


def test_multi_output_channels_eq_cr_with_output_stride(qnnp):
    check(qnnp, dw("dm_5_c16_out19", CR, 5, output_pixel_stride=19))
    check(qnnp, dw("dm_5_c16_out32", CR, 5, output_pixel_stride=32))


@pytest.mark.parametrize("channels", range(2 * CR, 16 * CR, 3 * CR))
def test_single_output_channels_div_cr(qnnp, channels):
    check(qnnp, dw(f"dm_1_c{channels}", channels, 1))


@pytest.mark.parametrize("channels", range(2 * CR, 16 * CR, 3 * CR))
def test_multi_output_channels_div_cr(qnnp, channels):
    check(qnnp, dw(f"dm_5_c{channels}", channels, 5))


@pytest.mark.parametrize("channels", range(2 * CR, 16 * CR, 3 * CR))
def test_multi_output_channels_div_cr_with_output_stride(qnnp, channels):
    check(qnnp, dw(f"dm_5_c{channels}_out", channels, 5, output_pixel_stride=channels + 171))


@pytest.mark.parametrize("channels", range(CR + 1, 2 * CR))
def test_single_output_channels_gt_cr(qnnp, channels):
    check(qnnp, dw(f"dm_1_c{channels}", channels, 1))


@pytest.mark.parametrize("channels", range(CR + 1, 2 * CR, 4))
@pytest.mark.parametrize("kw", [dict(qmin=128), dict(qmax=128), dict(izp=255, kzp=0), dict(izp=0, kzp=255)],
                         ids=["with_qmin", "with_qmax", "with_input_zero_point_only", "with_kernel_zero_point_only"])
def test_single_output_channels_gt_cr_variants(qnnp, channels, kw):
    check(qnnp, dw(f"dm_1_c{channels}_" + "_".join(f"{k}{v}" for k, v in kw.items()), channels, 1, **kw))


@pytest.mark.parametrize("channels", range(CR + 1, 2 * CR))
def test_multi_output_channels_gt_cr(qnnp, channels):
    check(qnnp, dw(f"dm_5_c{channels}", channels, 5))


@pytest.mark.parametrize("channels", range(CR + 1, 2 * CR, 3))
def test_multi_output_channels_gt_cr_with_output_stride(qnnp, channels):
    check(qnnp, dw(f"dm_5_c{channels}_out17", channels, 5, output_pixel_stride=channels + 17))
