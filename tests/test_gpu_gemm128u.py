"""GPU tier: the 128x128-tile GEMM that stages its activation tile through registers (qnnpack_amd/csrc/hip/q8gemm128u.hip, "gemm_kernel"
29; round 6) against the scalar oracle: ANY channel counts (K and N from 1 up, multiples of nothing), any number of groups, any kernel
and input zero point, pixel strides and base addresses of any alignment -- the shapes of ShuffleNet v1's grouped 1x1 convolutions and
ShuffleNet v2's 58 / 116 / 122 / 232 / 244 / 488-channel pointwise layers (bench/convolution.cc:108-426), which ran on the generic tile
kernel until round 6. Reference path: q8gemm under qnnp_run_operator, one group after the other (src/operator-run.c:770-804)."""
import dataclasses

import numpy as np
import pytest

from _cases import ConvCase, FcCase
from _gpu import from_device, to_device
from oracle import o1
from qnnpack_amd.binding import QnnpackError
from _runner import FILL, assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu
class _Names:
    """kernel names the register-staged GEMM may report (tile width follows the channel count)"""
    names = ("q8_gemm_mfma_128x32_u16", "q8_gemm_mfma_128x64_u16", "q8_gemm_mfma_128x128_u16")
    def __eq__(self, other): return other in self.names
    def __repr__(self): return " | ".join(self.names)


KERNEL = _Names()


@pytest.fixture
def ugemm(qnnp):
    qnnp.set_option("gemm_kernel", 29)
    yield qnnp
    qnnp.set_option("gemm_kernel", 0)


def _fc(lib, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(lib, case, quant, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("k,n,name", [(58, 12, "q8_gemm_mfma_128x32_u16"), (58, 25, "q8_gemm_mfma_128x32_u16"), (58, 45, "q8_gemm_mfma_128x64_u16"),
                                      (58, 88, "q8_gemm_mfma_128x32_u16"), (100, 88, "q8_gemm_mfma_128x128_u16"),
                                      (58, 136, "q8_gemm_mfma_128x32_u16"), (100, 136, "q8_gemm_mfma_128x128_u16"),
                                      (100, 62, "q8_gemm_mfma_128x64_u16"), (100, 250, "q8_gemm_mfma_128x128_u16")])
def test_tile_width_follows_the_shape(ugemm, k, n, name):
    """short reductions: the width with the fewest padded columns; longer ones: the fewest channel tiles, narrowest width among those"""
    case = FcCase(f"u_width_k{k}_n{n}", 300, k, n)
    expected, quant = fc_expected(case)
    out, kname = fc_run(ugemm, case, quant, to_device=to_device, from_device=from_device)
    assert kname == name, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("k", [1, 3, 4, 25, 31, 32, 33, 58, 63, 64, 65, 100, 122, 127, 129, 200, 400, 577])
@pytest.mark.parametrize("m", [1, 127, 129, 300])
def test_m_and_k(ugemm, m, k):
    _fc(ugemm, FcCase(f"u_m{m}_k{k}", m, k, 45))


@pytest.mark.parametrize("n", [1, 3, 4, 25, 45, 58, 88, 122, 128, 129, 250, 488])
@pytest.mark.parametrize("kw", [dict(), dict(kzp=128), dict(kzp=0, izp=255), dict(kzp=255, izp=0), dict(kzp=77, izp=3), dict(qmin=100, qmax=150)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()) or "default")
def test_n_and_quantization(ugemm, n, kw):
    _fc(ugemm, FcCase(f"u_n{n}_" + "_".join(f"{k}{v}" for k, v in kw.items()), 200, 100, n, **kw))


@pytest.mark.parametrize("in_stride,out_stride", [(59, 61), (60, 64), (58, 58), (123, 77)])
def test_strided_and_unaligned_rows(ugemm, in_stride, out_stride):
    _fc(ugemm, FcCase(f"u_strides_{in_stride}_{out_stride}", 260, 58, 58, input_stride=in_stride, output_stride=out_stride))


@pytest.mark.parametrize("case", [
    ConvCase("u_g2_25_88", (28, 28), (1, 1), groups=2, gic=25, goc=88, batch=3),            # ShuffleNet v1 g2, bench/convolution.cc:152
    ConvCase("u_g2_100_25", (28, 28), (1, 1), groups=2, gic=100, goc=25, batch=3),
    ConvCase("u_g2_400_100", (7, 7), (1, 1), groups=2, gic=400, goc=100, batch=5),
    ConvCase("u_g3_40_80", (14, 14), (1, 1), groups=3, gic=40, goc=80, batch=3),
    ConvCase("u_g4_17_62", (28, 28), (1, 1), groups=4, gic=17, goc=62, batch=2),
    ConvCase("u_g8_12_45", (28, 28), (1, 1), groups=8, gic=12, goc=45, batch=2),
    ConvCase("u_g8_192_48", (7, 7), (1, 1), groups=8, gic=192, goc=48, batch=4),
    ConvCase("u_dense_58_58", (28, 28), (1, 1), gic=58, goc=58, batch=3),                   # ShuffleNet v2 x1.0
    ConvCase("u_dense_24_122", (14, 15), (1, 1), gic=24, goc=122, batch=3),
    ConvCase("u_dense_488_488", (7, 7), (1, 1), gic=488, goc=488, batch=5),
    ConvCase("u_g2_strided_pixels_zp", (9, 11), (1, 1), groups=2, gic=25, goc=50, batch=2, input_pixel_stride=53, output_pixel_stride=103,
             izp=9, kzp=200),
    ConvCase("u_g3_qrange", (6, 6), (1, 1), groups=3, gic=7, goc=5, batch=2, qmin=30, qmax=220),
], ids=lambda c: c.name)
def test_grouped_and_odd_pointwise_convolutions(ugemm, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(ugemm, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("u_auto_g2_25_88", (28, 28), (1, 1), groups=2, gic=25, goc=88, batch=8),
    ConvCase("u_auto_g4_68_34", (28, 28), (1, 1), groups=4, gic=68, goc=34, batch=8),
    ConvCase("u_auto_dense_116", (14, 14), (1, 1), gic=116, goc=116, batch=16),
], ids=lambda c: c.name)
def test_auto_takes_it_for_what_reached_the_generic_kernel(qnnp, case):
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname == KERNEL, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", [
    ConvCase("u_bad_3x3", (9, 9), (3, 3), (1, 1, 1, 1), gic=10, goc=10),
    ConvCase("u_bad_strided", (9, 9), (1, 1), subsampling=(2, 2), gic=10, goc=10),
], ids=lambda c: c.name)
def test_refuses_what_it_cannot_take(ugemm, case):
    _, quant, out_hw = conv_expected(case)
    with pytest.raises(QnnpackError):
        conv_run(ugemm, case, quant, out_hw, to_device=to_device, from_device=from_device)


SCALES = [float.fromhex("0x1.FFFFFEp-1"), 0.75, 0.5, 1 / 255.0, 0.0031, 2.0 ** -22, 2.0 ** -32]
QUANT = [(0, 0, 255), (127, 0, 255), (255, 0, 255), (200, 1, 254), (100, 128, 255), (7, 5, 9)]


@pytest.mark.parametrize("scale", SCALES, ids=lambda s: f"{s:.3e}")
def test_requantization_flavours(ugemm, scale):
    """activations on their zero point, kernel zero point 100: bias + row term + fold cancel exactly -- every rounding flavour"""
    N, K, M = 250, 58, 140
    rng = np.random.default_rng(5)
    acc = rng.integers(-2**31, 2**31, size=N).astype(np.int64)
    acc[:8] = [-2**31, 2**31 - 1, 0, -1, 1, -2**30, 2**30, 2**31 - 129]
    acc[-60:] = rng.integers(-70000, 70000, size=60)
    acc = acc.astype(np.int32)
    kernel = np.random.default_rng(9).integers(0, 256, size=(N, K), dtype=np.uint8)
    inp = np.full(M * K, 77, np.uint8)
    for zp, qmin, qmax in QUANT:
        op = ugemm.create_fully_connected_nc_q8(K, N, 77, 1.0, 100, float(scale), kernel, acc, zp, 1.0, qmin, qmax, 0)
        try:
            d_in, d_out = to_device(inp), to_device(np.zeros(M * N, np.uint8))
            ugemm.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
            ugemm.run_operator(op)
            assert ugemm.operator_kernel(op) == KERNEL
            out = from_device(d_out).reshape(M, N)
        finally:
            ugemm.delete_operator(op)
        exp = o1.q31_requantize(acc, np.float32(scale), zp, qmin, qmax)
        for m in (0, 71, M - 1):
            bad = np.flatnonzero(out[m] != exp)
            assert bad.size == 0, (scale, zp, qmin, qmax, acc[bad[:4]].tolist(), out[m][bad[:4]].tolist(), exp[bad[:4]].tolist())
