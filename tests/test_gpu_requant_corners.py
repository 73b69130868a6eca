"""GPU tier: the fused Q31 epilogue at the edges of its domain, through real operators. The accumulator is driven
to chosen values by a huge bias against all-zero-point inputs (every product is 0), so requantization alone decides
the output: accumulators at +-2^31, exact ties of both roundings, the largest and smallest legal scales, every
zero-point / clamp flavour (saturating fast path with the zero point folded into the rounding addend, generic path,
the unfolded corner at scale 1 - 2^-24 and below 2^-23). Fully connected (generic MFMA kernel), pointwise streaming
kernel and depthwise kernels all share the epilogue code but instantiate it separately."""
import numpy as np
import pytest

from _gpu import from_device, to_device
from oracle import o1

pytestmark = pytest.mark.gpu

SCALES = [float.fromhex("0x1.FFFFFEp-1"), float.fromhex("0x1.FFFFFCp-1"), 0.75, 0.5, 0.49999997, 1 / 255.0, 0.0031,
          2.0 ** -12, 2.0 ** -22, 2.0 ** -23, 2.0 ** -24, 2.0 ** -31, 2.0 ** -32]
QUANT = [(0, 0, 255), (127, 0, 255), (255, 0, 255), (200, 1, 254), (100, 128, 255), (100, 0, 128), (7, 5, 9)]


def _accumulators(n):
    rng = np.random.default_rng(5)
    acc = rng.integers(-2**31, 2**31, size=n).astype(np.int64)
    edge = [-2**31, 2**31 - 1, 0, -1, 1, -2**30, 2**30, -2**31 + 1, 2**31 - 2, 2**31 - 129, 2**31 - 257]
    acc[:len(edge)] = edge
    ties = []
    for s in range(1, 24):
        for k in (-3, -1, 0, 1, 2, 100):
            ties += [(k << s) + (1 << (s - 1)) + d for d in (-1, 0, 1)]
    room = max(0, n - len(edge) - n // 4)
    ties = ties[::max(1, len(ties) // room)][:room] if room else []
    acc[len(edge):len(edge) + len(ties)] = ties
    small = rng.integers(-70000, 70000, size=n // 4)
    acc[-small.size:] = small
    return np.clip(acc, -2**31, 2**31 - 1).astype(np.int32)


@pytest.mark.parametrize("scale", SCALES, ids=lambda s: f"{s:.3e}")
def test_fully_connected_epilogue_corners(qnnp, scale):
    N, K, M = 2048, 16, 3
    acc = _accumulators(N)
    kernel = np.full((N, K), 9, np.uint8)            # anything: the activations sit on their zero point
    inp = np.full(M * K, 77, np.uint8)
    for zp, qmin, qmax in QUANT:
        op = qnnp.create_fully_connected_nc_q8(K, N, 77, 1.0, 9, float(scale), kernel, acc, zp, 1.0, qmin, qmax, 0)
        try:
            d_in, d_out = to_device(inp), to_device(np.zeros(M * N, np.uint8))
            qnnp.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
            qnnp.run_operator(op)
            out = from_device(d_out).reshape(M, N)
        finally:
            qnnp.delete_operator(op)
        exp = o1.q31_requantize(acc, np.float32(scale), zp, qmin, qmax)
        for m in range(M):
            bad = np.flatnonzero(out[m] != exp)
            assert bad.size == 0, (scale, zp, qmin, qmax, acc[bad[:4]].tolist(), out[m][bad[:4]].tolist(), exp[bad[:4]].tolist())


@pytest.mark.parametrize("scale", [float.fromhex("0x1.FFFFFEp-1"), 0.5, 0.0031, 2.0 ** -24], ids=lambda s: f"{s:.3e}")
@pytest.mark.parametrize("variant", [5, 6])
def test_streaming_kernels_epilogue_corners(qnnp, scale, variant):
    N, K, M = 256, 32, 4096                           # many rows: the streaming kernels take it
    acc = _accumulators(N)
    kernel = np.full((N, K), 200, np.uint8)
    inp = np.full(M * K, 3, np.uint8)
    qnnp.set_option("gemm_kernel", variant)
    try:
        for zp, qmin, qmax in QUANT[:4]:
            op = qnnp.create_fully_connected_nc_q8(K, N, 3, 1.0, 200, float(scale), kernel, acc, zp, 1.0, qmin, qmax, 0)
            try:
                d_in, d_out = to_device(inp), to_device(np.zeros(M * N, np.uint8))
                qnnp.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
                qnnp.run_operator(op)
                out = from_device(d_out).reshape(M, N)
            finally:
                qnnp.delete_operator(op)
            exp = o1.q31_requantize(acc, np.float32(scale), zp, qmin, qmax)
            assert np.array_equal(out[0], exp) and np.array_equal(out[-1], exp) and np.array_equal(out[M // 2 + 5], exp), \
                (scale, zp, qmin, qmax)
    finally:
        qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("scale", [float.fromhex("0x1.FFFFFEp-1"), 0.5, 0.0031, 2.0 ** -24], ids=lambda s: f"{s:.3e}")
@pytest.mark.parametrize("variant", [1, 2, 3, 5])
def test_depthwise_epilogue_corners(qnnp, scale, variant):
    C, H, W = 64, 12, 60
    acc = _accumulators(C)
    kernel = np.full((C, 1, 3, 3, 1), 50, np.uint8)
    inp = np.full(H * W * C, 11, np.uint8)
    qnnp.set_option("dwconv_kernel", variant)
    try:
        for zp, qmin, qmax in QUANT[:4]:
            op = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, C, 1, 1,
                                                   11, 1.0, 50, float(scale), kernel, acc, zp, 1.0, qmin, qmax, 0)
            try:
                d_in, d_out = to_device(inp), to_device(np.zeros(H * W * C, np.uint8))
                qnnp.setup_convolution2d_nhwc_q8(op, 1, H, W, d_in, C, d_out, C)
                qnnp.run_operator(op)
                out = from_device(d_out).reshape(H * W, C)
            finally:
                qnnp.delete_operator(op)
            exp = o1.q31_requantize(acc, np.float32(scale), zp, qmin, qmax)
            assert np.array_equal(out[0], exp) and np.array_equal(out[H * W // 2], exp) and np.array_equal(out[-1], exp), \
                (scale, variant, zp, qmin, qmax)
    finally:
        qnnp.set_option("dwconv_kernel", 0)
