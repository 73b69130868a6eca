"""GPU tier: the zero-point-centred 256x256 GEMM kernel (qnnpack_amd/csrc/hip/q8gemm256c.hip; what auto picks for
BASELINE configs[1]) against the scalar oracle ("gemm_kernel" 20): every prologue / steady-state / tail length in
K, row edges, padded channel counts, both centring classes (kernel zero point 127 and 128), every requantization flavour
the launcher can pick (shift 0 / bounded shift >= 1 / general x saturating clamp / explicit clamp with and without a
folded zero point), strided rows, and what it must refuse."""
import numpy as np
import pytest

from _cases import ConvCase, FcCase
from _gpu import from_device, to_device
from oracle import o1
from qnnpack_amd.binding import QnnpackError
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu

# ("gemm_kernel" 21, the A/B structure with the fragment reads in one burst, lost its A/B and exists in measurement builds only)
# ("gemm_kernel" 23, round 6: the same GEMM on v_mfma_i32_16x16x64_i8, q8gemm256x.hip)
_NAME = {20: "q8_gemm_mfma_256x256_c", 23: "q8_gemm_mfma_256x256_c16"}
_MIN_K = {20: 512, 23: 512}


@pytest.fixture(params=sorted(_NAME), ids=lambda v: _NAME[v].replace("q8_gemm_mfma_256x256_", "").replace("c_", "") or "c")
def centred(qnnp, request):
    qnnp.set_option("gemm_kernel", request.param)
    qnnp._kname = _NAME[request.param]
    qnnp._min_k = _MIN_K[request.param]
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc(lib, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(lib, case, quant, to_device=to_device, from_device=from_device)
    assert kname == lib._kname, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


# 8 ... 21 K tiles: the unrolled steady state runs 0, 1 or 2 times, the run-time-slot steady state 0 ... 4 times
@pytest.mark.parametrize("k", [512, 576, 640, 704, 768, 832, 896, 960, 1024, 1088, 1152, 1344])
@pytest.mark.parametrize("m", [1, 255, 257, 1000])
def test_m_and_k(centred, m, k):
    if k < centred._min_k:
        pytest.skip("fewer K tiles than twice the ring")
    _fc(centred, FcCase(f"c_m{m}_k{k}", m, k, 256))


@pytest.mark.parametrize("n", [256, 1008, 768])
@pytest.mark.parametrize("kw", [dict(), dict(kzp=128), dict(izp=0, kzp=128), dict(izp=255), dict(izp=3, kzp=128), dict(qmin=128),
                                dict(qmax=128, kzp=128)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()) or "default")
def test_n_and_quantization(centred, n, kw):
    _fc(centred, FcCase(f"c_n{n}_" + "_".join(f"{k}{v}" for k, v in kw.items()), 520, 704, n, **kw))


def test_strided_rows(centred):
    _fc(centred, FcCase("c_strided", 300, 640, 256, input_stride=656, output_stride=272))


# ---- strided 1x1 convolutions (round 5): the rows' addresses come from the operator's offset table (ResNet-50's K = 512 / 1024
#      downsampling shortcuts; reference: indirection + q8conv, src/indirection.c:18-79) ----
@pytest.mark.parametrize("case", [
    ConvCase("c_1x1_s2_512_256", (9, 11), subsampling=(2, 2), gic=512, goc=256, batch=3),
    ConvCase("c_1x1_s2_odd_1024_512", (13, 7), subsampling=(2, 2), gic=1024, goc=512, batch=5),
    # ONE output pixel per image (rows_per_image == 1: no 32-bit division magic for a divisor of 1), several row tiles
    ConvCase("c_1x1_s2_one_pixel_per_image", (2, 2), subsampling=(2, 2), gic=512, goc=256, batch=600),
    ConvCase("c_1x1_s3x2_640_256_kzp128", (20, 17), subsampling=(3, 2), gic=640, goc=256, batch=4, kzp=128, izp=9),
    ConvCase("c_1x1_s2_strided_pixels", (8, 8), subsampling=(2, 2), gic=512, goc=256, batch=40, input_pixel_stride=528, output_pixel_stride=272),
], ids=lambda c: c.name)
def test_strided_pointwise_convolution(centred, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(centred, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == centred._kname, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def test_strided_pointwise_with_padding_taps_is_refused(centred):
    case = ConvCase("c_1x1_s2_padded", (14, 14), padding=(1, 1, 1, 1), subsampling=(2, 2), gic=512, goc=256)
    _, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(centred, case, quant, out_hw, to_device=to_device, from_device=from_device)


@pytest.mark.parametrize("kw", [dict(kzp=126), dict(kzp=0), dict(kzp=129)], ids=lambda d: f"kzp{d['kzp']}")
def test_other_zero_points_have_no_centred_image(centred, kw):
    case = FcCase("c_refused_kzp", 300, 640, 256, **kw)
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(centred, case, quant, to_device=to_device, from_device=from_device)


@pytest.mark.parametrize("k,n,stride", [(448, 256, 0), (640, 260, 0), (640, 1000, 0), (648, 256, 0), (640, 256, 260)])
def test_refuses_what_it_cannot_take(centred, k, n, stride):
    case = FcCase(f"c_refused_k{k}_n{n}", 300, k, n, output_stride=stride)
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(centred, case, quant, to_device=to_device, from_device=from_device)


SCALES = [float.fromhex("0x1.FFFFFEp-1"), float.fromhex("0x1.FFFFFCp-1"), 0.75, 0.5, 1 / 255.0, 0.0031, 2.0 ** -22,
          2.0 ** -24, 2.0 ** -32]
QUANT = [(0, 0, 255), (127, 0, 255), (255, 0, 255), (200, 1, 254), (100, 128, 255), (7, 5, 9)]


def _accumulators(n):
    rng = np.random.default_rng(5)
    acc = rng.integers(-2**31, 2**31, size=n).astype(np.int64)
    edge = [-2**31, 2**31 - 1, 0, -1, 1, -2**30, 2**30, -2**31 + 1, 2**31 - 2, 2**31 - 129, 2**31 - 257]
    acc[:len(edge)] = edge
    ties = []
    for s in range(1, 24):
        for k in (-3, -1, 0, 1, 2, 100):
            ties += [(k << s) + (1 << (s - 1)) + d for d in (-1, 0, 1)]
    acc[len(edge):len(edge) + len(ties)] = ties[:n - len(edge)]
    small = rng.integers(-70000, 70000, size=n // 4)
    acc[-small.size:] = small
    return np.clip(acc, -2**31, 2**31 - 1).astype(np.int32)


@pytest.mark.parametrize("code", sorted(_NAME))
@pytest.mark.parametrize("kzp", [127, 128])
@pytest.mark.parametrize("scale", SCALES, ids=lambda s: f"{s:.3e}")
def test_epilogue_corners(qnnp, scale, kzp, code):
    """Accumulators driven by the bias alone (activations on their zero point): +-2^31, ties of both roundings, the
    unfolded zero-point corners -- through the offset / general sequences of the centred kernel."""
    N, K, M = 1024, 640, 260
    acc = _accumulators(N)
    kernel = np.random.default_rng(9).integers(0, 256, size=(N, K), dtype=np.uint8)
    inp = np.full(M * K, 77, np.uint8)
    qnnp.set_option("gemm_kernel", code)
    try:
        for zp, qmin, qmax in QUANT:
            op = qnnp.create_fully_connected_nc_q8(K, N, 77, 1.0, kzp, float(scale), kernel, acc, zp, 1.0, qmin, qmax, 0)
            try:
                d_in, d_out = to_device(inp), to_device(np.zeros(M * N, np.uint8))
                qnnp.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
                qnnp.run_operator(op)
                assert qnnp.operator_kernel(op) == _NAME[code]
                out = from_device(d_out).reshape(M, N)
            finally:
                qnnp.delete_operator(op)
            exp = o1.q31_requantize(acc, np.float32(scale), zp, qmin, qmax)
            for m in (0, 131, M - 1):
                bad = np.flatnonzero(out[m] != exp)
                assert bad.size == 0, (scale, zp, qmin, qmax, acc[bad[:4]].tolist(), out[m][bad[:4]].tolist(), exp[bad[:4]].tolist())
    finally:
        qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("kzp,k,n,kernel", [(127, 1088, 2048, "q8_gemm_mfma_256x256_c16"),
                                            (128, 1088, 2048, "q8_gemm_mfma_256x256_c16"),
                                            (126, 1088, 2048, "q8_gemm_mfma_256x256_r16"),
                                            (127, 1088, 2080, "q8_gemm_mfma_128x256")])     # (N % 256 != 0: no centred flavour; underfilled: 128-row tiles)
def test_auto_takes_the_centred_flavour_where_it_applies(qnnp, kzp, k, n, kernel):
    case = FcCase(f"auto_kzp{kzp}_k{k}_n{n}", 3328, k, n, kzp=kzp)
    expected, quant = fc_expected(case)
    out, kname = fc_run(qnnp, case, quant, to_device=to_device, from_device=from_device)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("code", sorted(_NAME))
def test_repeated_launches_are_stable(qnnp, code):
    """The barrier-free tail and the image slots race with nothing: 20 launches of one operator, identical bytes."""
    case = FcCase("c_repeat", 2048, 1280, 1024)
    expected, quant = fc_expected(case)
    qnnp.set_option("gemm_kernel", code)
    try:
        for _ in range(20):
            out, kname = fc_run(qnnp, case, quant, to_device=to_device, from_device=from_device)
            assert kname == _NAME[code], kname
            assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")
    finally:
        qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("kw,kernel", [(dict(kzp=127), "q8_gemm_mfma_256x256_c16"), (dict(kzp=128), "q8_gemm_mfma_256x256_c16"),
                                       (dict(kzp=127, input_pixel_stride=1104, output_pixel_stride=528), "q8_gemm_mfma_256x256_c16"),
                                       (dict(kzp=126), "q8_gemm_mfma_256x256_r16")],
                         ids=["kzp127", "kzp128", "kzp127_strided_pixels", "kzp126"])
def test_pointwise_convolutions_take_it_too(qnnp, kw, kernel):
    """a 1x1 convolution is the same GEMM over pixels (src/convolution.c:196-199 routes it to qnnp_ukernel_type_gemm):
    convolution.c builds the centred image for single-group convolutions as fully-connected.c does"""
    from _cases import ConvCase
    from _runner import conv_expected, conv_run
    # (enough rows, and a reduction beyond the long-K streaming kernel's 1024, that the automatic choice is the 256x256 tiling)
    case = ConvCase("c_pw_1088_512", (81, 80), (1, 1), gic=1088, goc=512, batch=2, **kw)
    o1.set_threads(16)
    try:
        expected, quant, out_hw = conv_expected(case)
    finally:
        o1.set_threads(1)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}, {kw}]")


# ---- round 6: every other kernel zero point on the 16x16x64 kernel: the standard image + the kernel-zero-point row term
#      (ROWSUM flavour of q8gemm256x.hip, "gemm_kernel" 28; reference: the same q8gemm path, src/q8gemm/4x4c2-sse2.c:14-318) ----
R16 = "q8_gemm_mfma_256x256_r16"


@pytest.fixture
def rowsum(qnnp):
    qnnp.set_option("gemm_kernel", 28)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc_r16(lib, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(lib, case, quant, to_device=to_device, from_device=from_device)
    assert kname == R16, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("k", [512, 576, 704, 768, 1024, 1344])
@pytest.mark.parametrize("m", [1, 255, 257, 1000])
def test_rowsum_m_and_k(rowsum, m, k):
    _fc_r16(rowsum, FcCase(f"r_m{m}_k{k}", m, k, 256, kzp=126))


@pytest.mark.parametrize("n", [256, 1008])
@pytest.mark.parametrize("kw", [dict(kzp=126), dict(kzp=0), dict(kzp=255), dict(kzp=77, izp=3), dict(kzp=200, izp=255), dict(kzp=1, izp=0),
                                dict(kzp=126, qmin=128), dict(kzp=129, qmax=128)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()))
def test_rowsum_zero_points_and_clamps(rowsum, n, kw):
    _fc_r16(rowsum, FcCase(f"r_n{n}_" + "_".join(f"{k}{v}" for k, v in kw.items()), 520, 704, n, **kw))


def test_rowsum_strided_rows_and_pointwise_convolution(rowsum):
    _fc_r16(rowsum, FcCase("r_strided", 300, 640, 256, input_stride=656, output_stride=272, kzp=126))
    case = ConvCase("r_1x1_s2_512_256", (9, 11), subsampling=(2, 2), gic=512, goc=256, batch=3, kzp=90)
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(rowsum, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == R16, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("scale", [0.75, 0.5, 1 / 255.0, 0.0031, 2.0 ** -22], ids=lambda s: f"{s:.3e}")
def test_rowsum_requantization_flavours(rowsum, scale):
    """activations on their zero point: the accumulator is bias + the row term's and the bias fold's exact cancellation -- every
    rounding flavour of the launcher against the oracle, kernel zero point 100"""
    N, K, M = 1024, 640, 260
    acc = _accumulators(N)
    kernel = np.random.default_rng(9).integers(0, 256, size=(N, K), dtype=np.uint8)
    inp = np.full(M * K, 77, np.uint8)
    for zp, qmin, qmax in QUANT:
        op = rowsum.create_fully_connected_nc_q8(K, N, 77, 1.0, 100, float(scale), kernel, acc, zp, 1.0, qmin, qmax, 0)
        try:
            d_in, d_out = to_device(inp), to_device(np.zeros(M * N, np.uint8))
            rowsum.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
            rowsum.run_operator(op)
            assert rowsum.operator_kernel(op) == R16
            out = from_device(d_out).reshape(M, N)
        finally:
            rowsum.delete_operator(op)
        exp = o1.q31_requantize(acc, np.float32(scale), zp, qmin, qmax)
        for m in (0, 131, M - 1):
            bad = np.flatnonzero(out[m] != exp)
            assert bad.size == 0, (scale, zp, qmin, qmax, acc[bad[:4]].tolist(), out[m][bad[:4]].tolist(), exp[bad[:4]].tolist())
