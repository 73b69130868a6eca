"""GPU tier: the q8conv microkernel test matrix of the reference (test/q8conv.cc:935-1165, the 14 4x4c2__sse2 names:
k_eq_8 [+ strided_c, qmin128, qmax128, azp_only, bzp_only], k_gt_8 [+ strided_c, azp_only, bzp_only, subtile],
k_div_8 [+ strided_c, subtile]; all ASSERT_EQ against the scalar q31 result, test/gemm-microkernel-tester.h:257-274)
re-hosted on the whole-operator implicit-GEMM kernels through qnnp_*_convolution2d_nhwc_q8.

The CPU kernel's tile (mr = 4 pixels, nr = 4 channels, kr = 2, K unrolled by 8 per tap) becomes the device tiles:
32-wide MFMA blocks, 128-row workgroups, 64-byte K steps, 16/8/4/1-byte activation vectors per tap. So "k" sweeps the
per-tap channel count around those vector widths and K-step edges, "m" / "n" sweep output pixels / channels around
the 32 / 128 edges, aStride is the input pixel stride and cStride the output pixel stride. Each case runs on the
automatically selected kernel and, where they accept the shape, forced onto the generic offset-table kernel and the
LDS-tiled direct kernel."""
import pytest

from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run
from qnnpack_amd import QnnpackError

pytestmark = pytest.mark.gpu

STEP = 16   # per-tap channels of one 16-byte activation vector: the device counterpart of the reference's "8"


def conv3x3(name, k, m_hw=(4, 4), n=32, **kw):
    """3x3 'same' convolution: k input channels per tap, m = m_hw[0] * m_hw[1] output pixels, n output channels."""
    return ConvCase(name, m_hw, (3, 3), (1, 1, 1, 1), gic=k, goc=n, **kw)


def check(qnnp, case, variants=(0, 1, 3)):
    expected, quant, out_hw = conv_expected(case)
    ran = []
    for variant in variants:
        qnnp.set_option("gemm_kernel", variant)
        try:
            out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
        except QnnpackError:
            assert variant != 0, "the automatic choice must take every shape"
            continue          # a forced kernel may decline a shape (reported, never silently rerouted)
        finally:
            qnnp.set_option("gemm_kernel", 0)
        assert_bytes_equal(out, expected, f"gfx950 {kname} (gemm_kernel={variant}) vs oracle [{case.name}]")
        ran.append(kname)
    assert ran, case.name
    return ran


def test_k_eq_step(qnnp):
    check(qnnp, conv3x3("cm_k16", STEP, input_pixel_stride=37))


def test_k_eq_step_strided_c(qnnp):
    check(qnnp, conv3x3("cm_k16_strided_c", STEP, input_pixel_stride=37, output_pixel_stride=49))


def test_k_eq_step_qmin128(qnnp):
    check(qnnp, conv3x3("cm_k16_qmin128", STEP, qmin=128))


def test_k_eq_step_qmax128(qnnp):
    check(qnnp, conv3x3("cm_k16_qmax128", STEP, qmax=128))


def test_k_eq_step_azp_only(qnnp):
    check(qnnp, conv3x3("cm_k16_azp_only", STEP, izp=255, kzp=0))


def test_k_eq_step_bzp_only(qnnp):
    check(qnnp, conv3x3("cm_k16_bzp_only", STEP, izp=0, kzp=255))


@pytest.mark.parametrize("k", range(STEP + 1, 2 * STEP))
def test_k_gt_step(qnnp, k):
    check(qnnp, conv3x3(f"cm_k{k}", k, input_pixel_stride=37))


@pytest.mark.parametrize("k", range(STEP + 1, 2 * STEP, 3))
def test_k_gt_step_strided_c(qnnp, k):
    check(qnnp, conv3x3(f"cm_k{k}_strided_c", k, input_pixel_stride=37, output_pixel_stride=49))


@pytest.mark.parametrize("k", range(STEP + 1, 2 * STEP, 3))
def test_k_gt_step_azp_only(qnnp, k):
    check(qnnp, conv3x3(f"cm_k{k}_azp_only", k, input_pixel_stride=37, izp=255, kzp=0))


@pytest.mark.parametrize("k", range(STEP + 1, 2 * STEP, 3))
def test_k_gt_step_bzp_only(qnnp, k):
    check(qnnp, conv3x3(f"cm_k{k}_bzp_only", k, input_pixel_stride=37, izp=0, kzp=255))


@pytest.mark.parametrize("k", [17, 24, 31])
@pytest.mark.parametrize("m_hw", [(1, 1), (1, 31), (3, 11), (5, 7), (8, 16), (3, 43)], ids=lambda hw: f"m{hw[0] * hw[1]}")
@pytest.mark.parametrize("n", [1, 31, 33, 65])
def test_k_gt_step_subtile(qnnp, k, m_hw, n):
    check(qnnp, conv3x3(f"cm_k{k}_m{m_hw[0] * m_hw[1]}_n{n}", k, m_hw, n, input_pixel_stride=37), variants=(0, 1))


@pytest.mark.parametrize("k", range(2 * STEP, 16 * STEP, STEP))
def test_k_div_step(qnnp, k):
    check(qnnp, conv3x3(f"cm_kdiv{k}", k, input_pixel_stride=k + 43))


@pytest.mark.parametrize("k", range(2 * STEP, 16 * STEP, 3 * STEP))
def test_k_div_step_strided_c(qnnp, k):
    check(qnnp, conv3x3(f"cm_kdiv{k}_strided_c", k, input_pixel_stride=k + 43, output_pixel_stride=48))


@pytest.mark.parametrize("k", range(2 * STEP, 16 * STEP, 3 * STEP))
@pytest.mark.parametrize("m_hw", [(1, 1), (3, 11), (8, 16), (3, 43), (16, 17)], ids=lambda hw: f"m{hw[0] * hw[1]}")
@pytest.mark.parametrize("n", [1, 32, 33, 129])
def test_k_div_step_subtile(qnnp, k, m_hw, n):
    check(qnnp, conv3x3(f"cm_kdiv{k}_m{m_hw[0] * m_hw[1]}_n{n}", k, m_hw, n, input_pixel_stride=k + 43), variants=(0, 1))


def test_power_of_two_shapes_reach_the_lds_kernel(qnnp):
    """the sweep above forces gemm_kernel 3 wherever it accepts: make sure that is not 'nowhere'"""
    ran = check(qnnp, conv3x3("cm_lds_probe", 64, (12, 12), 64, batch=2))
    assert "q8_conv_lds_mfma" in ran, ran
