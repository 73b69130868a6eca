"""GPU tier: the 128x128-tile zero-point-centred GEMM on v_mfma_i32_16x16x64_i8 (qnnpack_amd/csrc/hip/q8gemm128x.hip, "gemm_kernel" 24;
round 6) against the scalar oracle: every K tile count from 1 (ring fill and drain are run-time conditions), row and channel
edges (partial row tiles, channel counts that leave part of the last 128-wide tile -- or of its 32-channel blocks -- empty),
both centring classes (kernel zero point 127 and 128), every requantization flavour the launcher can pick, strided rows and
pixels, strided 1x1 convolutions through the dense offset table, and what it must refuse.
Reference path: qnnp_create/setup_fully_connected_nc_q8 / convolution2d_nhwc_q8 -> q8gemm (src/q8gemm/4x4c2-sse2.c:14-318)."""
import numpy as np
import pytest

from _cases import ConvCase, FcCase
from _gpu import from_device, to_device
from oracle import o1
from qnnpack_amd.binding import QnnpackError
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu
# "gemm_kernel" 24 = tile width by the channel count, 25 / 26 = 64- / 128-channel tiles forced
KERNELS = {24: ("q8_gemm_mfma_128x64_c16", "q8_gemm_mfma_128x128_c16"), 25: ("q8_gemm_mfma_128x64_c16",), 26: ("q8_gemm_mfma_128x128_c16",)}


class _Names:
    """kernel names the forced code may report"""
    def __init__(self, names): self.names = names
    def __eq__(self, other): return other in self.names
    def __repr__(self): return " | ".join(self.names)


KERNEL = _Names(KERNELS[24])


@pytest.fixture(params=[25, 26], ids=["n64", "n128"])
def mid(qnnp, request):
    qnnp.set_option("gemm_kernel", request.param)
    qnnp._names = KERNELS[request.param]
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc(lib, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(lib, case, quant, to_device=to_device, from_device=from_device)
    assert kname in lib._names, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


# 1 ... 11 K tiles: fewer tiles than ring slots, exactly the ring, and steady states of 1 ... 7 iterations
@pytest.mark.parametrize("k", [64, 128, 192, 256, 320, 384, 448, 512, 576, 704])
@pytest.mark.parametrize("m", [1, 127, 129, 700])
def test_m_and_k(mid, m, k):
    _fc(mid, FcCase(f"x_m{m}_k{k}", m, k, 128))


@pytest.mark.parametrize("n", [16, 64, 96, 160, 320, 1008])
@pytest.mark.parametrize("kw", [dict(), dict(kzp=128), dict(izp=0, kzp=128), dict(izp=255), dict(qmin=128), dict(qmax=128, kzp=128)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()) or "default")
def test_n_and_quantization(mid, n, kw):
    _fc(mid, FcCase(f"x_n{n}_" + "_".join(f"{k}{v}" for k, v in kw.items()), 300, 320, n, **kw))


def test_long_k(mid):
    _fc(mid, FcCase("x_k4096", 260, 4096, 256))


def test_strided_rows(mid):
    _fc(mid, FcCase("x_strided", 300, 320, 160, input_stride=336, output_stride=176))


@pytest.mark.parametrize("n,name", [(64, "q8_gemm_mfma_128x64_c16"), (96, "q8_gemm_mfma_128x128_c16"), (128, "q8_gemm_mfma_128x128_c16"),
                                    (160, "q8_gemm_mfma_128x64_c16"), (320, "q8_gemm_mfma_128x64_c16"), (1280, "q8_gemm_mfma_128x128_c16")])
def test_tile_width_follows_the_channel_count(qnnp, n, name):
    """code 24: 64-wide tiles when they cover the channels with fewer padded columns than 128-wide ones"""
    qnnp.set_option("gemm_kernel", 24)
    try:
        case = FcCase(f"x_auto_width_n{n}", 200, 192, n)
        expected, quant = fc_expected(case)
        out, kname = fc_run(qnnp, case, quant, to_device=to_device, from_device=from_device)
        assert kname == name, kname
        assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")
    finally:
        qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("case", [
    ConvCase("x_1x1_14x14_384_64", (14, 14), (1, 1), gic=384, goc=64, batch=3),            # MobileNetV2 layer 19
    ConvCase("x_1x1_7x7_960_320", (7, 7), (1, 1), gic=960, goc=320, batch=5),              # layer 29
    ConvCase("x_1x1_7x7_320_1280", (7, 7), (1, 1), gic=320, goc=1280, batch=3),            # layer 30
    ConvCase("x_1x1_strided_pixels", (9, 9), (1, 1), gic=192, goc=96, input_pixel_stride=208, output_pixel_stride=112, batch=2),
    ConvCase("x_1x1_kzp128", (13, 11), (1, 1), gic=576, goc=160, batch=2, kzp=128, izp=9),
], ids=lambda c: c.name)
def test_pointwise_convolutions(mid, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(mid, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("x_1x1_s2_512_256", (9, 11), (1, 1), subsampling=(2, 2), gic=512, goc=256, batch=3),
    ConvCase("x_1x1_s2_one_pixel_per_image", (2, 2), (1, 1), subsampling=(2, 2), gic=128, goc=64, batch=300),
    ConvCase("x_1x1_s3x2_strided_pixels", (20, 17), (1, 1), subsampling=(3, 2), gic=192, goc=64, batch=4, input_pixel_stride=208, output_pixel_stride=80),
], ids=lambda c: c.name)
def test_strided_pointwise_convolutions(mid, case):
    """rows' addresses from the operator's offset table (one valid entry per output pixel): src/indirection.c:18-79"""
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(mid, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("x_1x1_13x13_512_1000", (13, 13), (1, 1), gic=512, goc=1000, batch=2),        # SqueezeNet's classifier convolution
    ConvCase("x_1x1_n72", (9, 9), (1, 1), gic=128, goc=72, batch=3),                       # last piece 8 bytes
    ConvCase("x_1x1_n20", (9, 9), (1, 1), gic=64, goc=20, batch=3),                        # 4 bytes
    ConvCase("x_1x1_n44", (9, 9), (1, 1), gic=192, goc=44, batch=3, kzp=128),              # 12 bytes
    ConvCase("x_1x1_n128_rows_of_132", (9, 9), (1, 1), gic=128, goc=128, batch=3, output_pixel_stride=132),
], ids=lambda c: c.name)
def test_rows_of_whole_dwords_that_are_not_whole_16_byte_pieces(mid, case):
    """round 6: dword-aligned 16-byte stores, the group's last piece 4 / 8 / 12 bytes"""
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(mid, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("kw", [dict(kzp=126), dict(kzp=0)], ids=lambda d: f"kzp{d['kzp']}")
def test_other_zero_points_have_no_centred_image(mid, kw):
    case = FcCase("x_refused_kzp", 300, 320, 128, **kw)
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(mid, case, quant, to_device=to_device, from_device=from_device)


@pytest.mark.parametrize("k,n,stride", [(96, 128, 0), (320, 100, 0), (320, 128, 130)])
def test_refuses_what_it_cannot_take(mid, k, n, stride):
    case = FcCase(f"x_refused_k{k}_n{n}", 300, k, n, output_stride=stride)
    _, quant = fc_expected(case)
    with pytest.raises(QnnpackError):
        fc_run(mid, case, quant, to_device=to_device, from_device=from_device)


SCALES = [float.fromhex("0x1.FFFFFEp-1"), 0.75, 0.5, 1 / 255.0, 0.0031, 2.0 ** -22, 2.0 ** -32]
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


@pytest.mark.parametrize("kzp", [127, 128])
@pytest.mark.parametrize("scale", SCALES, ids=lambda s: f"{s:.3e}")
def test_epilogue_corners(qnnp, scale, kzp):
    """Accumulators driven by the bias alone (activations on their zero point): +-2^31, ties of both roundings, the unfolded
    zero-point corners -- through the offset / general sequences as this kernel's launcher picks them."""
    N, K, M = 1024, 192, 140
    acc = _accumulators(N)
    kernel = np.random.default_rng(9).integers(0, 256, size=(N, K), dtype=np.uint8)
    inp = np.full(M * K, 77, np.uint8)
    qnnp.set_option("gemm_kernel", 26 if kzp == 127 else 25)
    try:
        for zp, qmin, qmax in QUANT:
            op = qnnp.create_fully_connected_nc_q8(K, N, 77, 1.0, kzp, float(scale), kernel, acc, zp, 1.0, qmin, qmax, 0)
            try:
                d_in, d_out = to_device(inp), to_device(np.zeros(M * N, np.uint8))
                qnnp.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
                qnnp.run_operator(op)
                assert qnnp.operator_kernel(op) == KERNEL
                out = from_device(d_out).reshape(M, N)
            finally:
                qnnp.delete_operator(op)
            exp = o1.q31_requantize(acc, np.float32(scale), zp, qmin, qmax)
            for m in (0, 71, M - 1):
                bad = np.flatnonzero(out[m] != exp)
                assert bad.size == 0, (scale, zp, qmin, qmax, acc[bad[:4]].tolist(), out[m][bad[:4]].tolist(), exp[bad[:4]].tolist())
    finally:
        qnnp.set_option("gemm_kernel", 0)


def test_fine_scale(mid):
    """a scale that resolves single accumulator units (1e-4 of the accumulator range would hide errors of a few hundred)"""
    M, K, N = 300, 384, 160
    rng = np.random.default_rng(31)
    kernel = rng.integers(127 - 3, 127 + 4, size=(N, K)).astype(np.uint8)
    inp = rng.integers(127 - 3, 127 + 4, size=M * K).astype(np.uint8)
    bias = rng.integers(-40, 41, size=N, dtype=np.int32)
    acc = o1.gemm_acc(inp.reshape(M, K), kernel, bias, 127, 127)
    assert np.abs(acc).max() < 4000
    for scale, ozp in ((0.03, 127), (0.5, 120)):
        expected = o1.requantize_rows(acc, np.float32(scale), ozp, 0, 255).reshape(-1)
        op = mid.create_fully_connected_nc_q8(K, N, 127, 1.0, 127, float(scale), kernel, bias, ozp, 1.0, 0, 255, 0)
        try:
            d_in, d_out = to_device(inp), to_device(np.zeros(M * N, np.uint8))
            mid.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
            mid.run_operator(op)
            assert mid.operator_kernel(op) == KERNEL
            out = from_device(d_out)
        finally:
            mid.delete_operator(op)
        assert_bytes_equal(out, expected, f"fine scale {scale}")


def test_repeated_launches_are_stable(mid):
    case = FcCase("x_repeat", 1500, 960, 320)
    expected, quant = fc_expected(case)
    for _ in range(10):
        out, kname = fc_run(mid, case, quant, to_device=to_device, from_device=from_device)
        assert kname == KERNEL, kname
        assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")
