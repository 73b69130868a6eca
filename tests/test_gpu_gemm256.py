"""GPU tier: the 256x256 LDS-DMA MFMA kernel (qnnpack_amd/csrc/hip/q8gemm256.hip), forced with the
"gemm_kernel" = 2 option, against the scalar oracle -- tile-edge sweeps in M, N and K, grouped and
convolution (offset-table) forms, zero-point variants. The auto-selected path for BASELINE configs[1]
(4096^3) is this kernel's lean flavour ("gemm_kernel" = 15; picked by auto when K % 64 == 0 and N % 256 == 0);
tests/test_gpu_fullsize.py asserts that."""
import numpy as np
import pytest

from _cases import ConvCase, FcCase
from _gpu import from_device, to_device
from qnnpack_amd.binding import QnnpackError
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu


# gemm_kernel option -> kernel name. (The structures that lost their A/B in round 3 -- "gemm_kernel" 4 / 16: four waves of
# 128 x 128, 10: 128 x 256 tiles with two workgroups per CU, 11: the ping-pong schedule -- are compiled into measurement
# builds only since round 4: the product refuses them, test_losing_structures_are_not_in_the_product.)
_NAME = {2: "q8_gemm_mfma_256x256"}


@pytest.fixture(params=sorted(_NAME), ids=lambda v: _NAME[v].replace("q8_gemm_mfma_", ""))
def big(qnnp, request):
    qnnp.set_option("gemm_kernel", request.param)
    qnnp._kname = _NAME[request.param]
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc(big, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(big, case, quant, to_device=to_device, from_device=from_device)
    assert kname == big._kname, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("m", [1, 31, 255, 256, 257, 300, 511, 512, 1000])
def test_m_edges(big, m):
    _fc(big, FcCase(f"b_m{m}", m, 256, 256))


@pytest.mark.parametrize("n", [1, 4, 31, 32, 33, 100, 255, 256, 257, 300, 512, 520])
def test_n_edges(big, n):
    _fc(big, FcCase(f"b_n{n}", 130, 128, n))


@pytest.mark.parametrize("k", [16, 32, 48, 64, 80, 112, 128, 144, 256, 272, 384, 1024])
def test_k_edges(big, k):
    _fc(big, FcCase(f"b_k{k}", 270, k, 260))


# ---- the lean flavour ("gemm_kernel" = 15: saddr LDS-DMA, steady state unrolled over the ring): plain GEMMs whose K is a
# multiple of 64 and whose N is a multiple of 256; forced, so an unsupported shape is refused instead of rerouted ----
_LEAN = {15: "q8_gemm_mfma_256x256_lean"}


@pytest.fixture(params=sorted(_LEAN), ids=lambda v: _LEAN[v].replace("q8_gemm_mfma_256x256_", ""))
def lean(qnnp, request):
    qnnp.set_option("gemm_kernel", request.param)
    qnnp._kname = _LEAN[request.param]
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("m", [1, 31, 255, 256, 257, 300, 1000])
@pytest.mark.parametrize("k", [64, 128, 256, 320, 512, 576, 1088])      # 1 ... 17 K tiles: every prologue / drain shape
def test_lean_m_and_k(lean, m, k):
    _fc(lean, FcCase(f"lean_m{m}_k{k}", m, k, 256))


@pytest.mark.parametrize("n", [256, 512, 768])
@pytest.mark.parametrize("kw", [dict(), dict(izp=0, kzp=0), dict(izp=255, kzp=255), dict(izp=3, kzp=250), dict(qmin=128)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()) or "default")
def test_lean_n_and_quantization(lean, n, kw):
    _fc(lean, FcCase(f"lean_n{n}_" + "_".join(f"{k}{v}" for k, v in kw.items()), 520, 704, n, **kw))


def test_lean_strided_rows(lean):
    _fc(lean, FcCase("lean_strided", 300, 640, 256, input_stride=656, output_stride=264))


# (shapes the dispatcher hands to this kernel by itself: K beyond the streaming kernels' 1024, more than 400 generic tiles)
# (kernel zero point 126: 127 and 128 go to the centred flavour, tests/test_gpu_gemm256c.py)
# (round 6: where K % 64 == 0 and N % 256 == 0 the row-sum flavour of the 16x16x64 kernel, q8gemm256x.hip, takes what the lean one took)
@pytest.mark.parametrize("k,n,kernel", [(1088, 2048, "q8_gemm_mfma_256x256_r16"), (1104, 2048, "q8_gemm_mfma_128x256"),
                                        (1088, 2080, "q8_gemm_mfma_128x256")])      # (13 x 9 tiles of 256 rows on 256 CUs: the 128-row tiling, round 5)
def test_auto_takes_the_lean_flavour_where_it_applies(qnnp, k, n, kernel):
    case = FcCase(f"auto_k{k}_n{n}", 3328, k, n, kzp=126)
    expected, quant = fc_expected(case)
    out, kname = fc_run(qnnp, case, quant, to_device=to_device, from_device=from_device)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("k,n", [(80, 256), (640, 260), (48, 512)])
def test_lean_refuses_what_it_cannot_take(lean, k, n):
    case = FcCase(f"lean_refused_k{k}_n{n}", 300, k, n)
    expected, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(lean, case, quant, to_device=to_device, from_device=from_device)


@pytest.mark.parametrize("kw", [dict(izp=0, kzp=0), dict(izp=255, kzp=255), dict(izp=128, kzp=128),
                                dict(izp=3, kzp=250), dict(qmin=128), dict(qmax=128)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()))
def test_quantization_variants(big, kw):
    _fc(big, FcCase("b_q_" + "_".join(f"{k}{v}" for k, v in kw.items()), 300, 192, 288, **kw))


def test_strided_rows(big):
    _fc(big, FcCase("b_strided", 300, 128, 260, input_stride=160, output_stride=264))


def test_unaligned_output_uses_byte_stores(big):
    _fc(big, FcCase("b_out_unaligned", 260, 128, 258, output_stride=259))


@pytest.mark.parametrize("case", [
    ConvCase("b_conv3x3_c64", (12, 10), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=3),
    ConvCase("b_conv3x3_c16_s2", (15, 17), (3, 3), (1, 1, 1, 1), subsampling=(2, 2), gic=16, goc=48, batch=2),
    ConvCase("b_conv3x3_c48_d2", (13, 14), (3, 3), (2, 2, 2, 2), dilation=(2, 2), gic=48, goc=40),
    ConvCase("b_conv5x5_c32", (11, 12), (5, 5), (2, 2, 2, 2), gic=32, goc=300),
    ConvCase("b_grouped_conv3x3", (10, 11), (3, 3), (1, 1, 1, 1), groups=3, gic=16, goc=24, batch=2),
    ConvCase("b_grouped_1x1", (9, 9), groups=2, gic=32, goc=40, batch=4),
    ConvCase("b_conv3x3_strided_pixels", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=36, input_pixel_stride=48,
             output_pixel_stride=40),
    ConvCase("b_conv1x3_pad", (8, 19), (1, 3), (0, 1, 0, 1), gic=80, goc=33),
], ids=lambda c: c.name)
def test_convolution_forms(big, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(big, case, quant, out_hw, to_device=to_device, from_device=from_device)
    want = big._kname + ("_conv" if case.kernel_size != (1, 1) else "")
    assert kname == want, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def test_unsupported_alignment_is_reported_not_silently_rerouted(big):
    """Forcing the big kernel on a shape it cannot take (K % 16 != 0) must fail loudly."""
    from qnnpack_amd import QnnpackError
    case = FcCase("b_bad", 64, 23, 19)
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(big, case, quant, to_device=to_device, from_device=from_device)


@pytest.mark.parametrize("variant", [4, 10, 11, 16])
def test_losing_structures_are_not_in_the_product(qnnp, variant):
    case = FcCase(f"b_not_shipped_{variant}", 300, 640, 256)
    _, quant = fc_expected(case)
    qnnp.set_option("gemm_kernel", variant)
    try:
        with pytest.raises(QnnpackError):
            fc_run(qnnp, case, quant, to_device=to_device, from_device=from_device)
    finally:
        qnnp.set_option("gemm_kernel", 0)
