"""GPU tier: the q8gemm microkernel test matrix of the reference (test/q8gemm.cc:2365-2659 for
4x4c2__sse2: k_eq_8, strided a / c, qmin128, qmax128, azp0, bzp0, nozp, k_gt_8, k_div_8 and the
m/n sub-tile sweeps, all ASSERT_EQ against the scalar q31 result, gemm-microkernel-tester.h:257-274)
re-hosted on the whole-operator HIP kernel through qnnp_*_fully_connected_nc_q8. Tile sizes of the
CPU kernels (mr=4, nr=4, kr=2) become the device tile edges (32-wide MFMA tiles, 128-row workgroups,
64-byte K steps), so the sweeps straddle those instead."""
import numpy as np
import pytest

from _cases import FcCase
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, fc_expected, fc_run

pytestmark = pytest.mark.gpu


def _check(qnnp, case):
    expected, quant = fc_expected(case)
    out, kname = fc_run(qnnp, case, quant, to_device=to_device, from_device=from_device)
    assert_bytes_equal(out, expected, f"gfx950 vs oracle [{case.name}] kernel={kname}")


def test_k_eq_step(qnnp):
    _check(qnnp, FcCase("g_k64", 128, 64, 128))


def test_k_eq_step_strided_a(qnnp):
    _check(qnnp, FcCase("g_k64_strided_a", 128, 64, 128, input_stride=80))


def test_k_eq_step_strided_c(qnnp):
    _check(qnnp, FcCase("g_k64_strided_c", 128, 64, 128, output_stride=132))


@pytest.mark.parametrize("kw", [dict(qmin=128), dict(qmax=128), dict(izp=0), dict(kzp=0), dict(izp=0, kzp=0),
                                dict(izp=255, kzp=255), dict(izp=128, kzp=128), dict(izp=1, kzp=254)],
                         ids=lambda d: "_".join(f"{k}{v}" for k, v in d.items()))
def test_quantization_variants(qnnp, kw):
    name = "g_q_" + "_".join(f"{k}{v}" for k, v in kw.items())
    _check(qnnp, FcCase(name, 96, 72, 40, **kw))


@pytest.mark.parametrize("k", [1, 2, 3, 7, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 200])
def test_k_sweep(qnnp, k):
    _check(qnnp, FcCase(f"g_k{k}", 70, k, 45))


@pytest.mark.parametrize("k", [64, 128, 192, 512])
def test_k_div_step_strided(qnnp, k):
    _check(qnnp, FcCase(f"g_kdiv{k}", 33, k, 36, input_stride=k + 16, output_stride=40))


@pytest.mark.parametrize("m", [1, 2, 31, 32, 33, 127, 128, 129, 255, 256, 257])
def test_m_subtile_sweep(qnnp, m):
    _check(qnnp, FcCase(f"g_m{m}", m, 48, 36))


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 130, 255, 256, 257])
def test_n_subtile_sweep(qnnp, n):
    _check(qnnp, FcCase(f"g_n{n}", 40, 48, n))


def test_large_accumulators_wrap_like_int32(qnnp):
    """bias near INT32 limits: every path is exact int32 (mod 2^32) arithmetic, as the reference's
    packed bias + pmaddwd accumulation is (pack.h:24,43; 4x4c2-sse2.c:47-109)."""
    case = FcCase("g_bigbias", 64, 256, 64)
    from _cases import fc_tensors
    inp, kernel, bias = fc_tensors(case)
    bias = bias.copy()
    bias[::2] = 2**31 - 1 - np.arange(bias[::2].size)
    bias[1::2] = -2**31 + np.arange(bias[1::2].size)
    from oracle import o1
    from _cases import strided_view
    a = strided_view(inp, case.batch, case.input_channels, case.in_stride)
    acc = o1.gemm_acc(a, kernel, bias, case.izp, case.kzp)
    scale, zp = np.float32(2.0 ** -24), 128
    expected = o1.requantize_rows(acc, scale, zp, 0, 255).reshape(-1)
    out, _ = fc_run(qnnp, case, (np.float32(1.0) / scale, zp), inp, kernel, bias, to_device, from_device)
    assert_bytes_equal(out, expected, "gfx950 vs oracle [wrapping accumulators]")


@pytest.mark.parametrize("scale_hex,zp", [("0x1.FFFFFEp-1", 255), ("0x1.FFFFFEp-1", 0), ("0x1.0p-1", 128), ("0x1.8p-1", 3)])
def test_extreme_accumulators_at_scale_near_one(qnnp, scale_hex, zp):
    """|acc| close to 2^31 with requantization scale in [0.5, 1): the scaled value itself is ~+-2^31, so the
    zero-point add must not wrap before the clamp (the [0,255] path saturates to int16 first)."""
    from _cases import fc_tensors, strided_view
    from oracle import o1
    case = FcCase("g_extreme_" + scale_hex + str(zp), 70, 64, 48)
    inp, kernel, bias = fc_tensors(case)
    bias = bias.copy()
    bias[0::3] = 2**31 - 1 - 3000000
    bias[1::3] = -2**31 + 3000000
    a = strided_view(inp, case.batch, case.input_channels, case.in_stride)
    acc = o1.gemm_acc(a, kernel, bias, case.izp, case.kzp)
    scale = np.float32(float.fromhex(scale_hex))
    expected = o1.requantize_rows(acc, scale, zp, 0, 255).reshape(-1)
    # choose (input, kernel, output) scales whose float32 quotient is exactly `scale`
    op = qnnp.create_fully_connected_nc_q8(64, 48, case.izp, float(scale), case.kzp, 1.0, kernel, bias, zp, 1.0, 0, 255)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, 0xA5, np.uint8))
        qnnp.setup_fully_connected_nc_q8(op, case.batch, d_in, 64, d_out, 48)
        qnnp.run_operator(op)
        assert_bytes_equal(from_device(d_out), expected, "extreme accumulators")
    finally:
        qnnp.delete_operator(op)
