"""GPU tier: the channel-chunked flavour of the LDS-patch convolution kernel (hip/q8convpatch.hip, CHUNK: stride-2 3x3 with
256 / 512 input channels, the patch re-staged 128 channels at a time) held to the scalar oracle with a FINE requantization
scale (1e-4: an accumulator error of a few hundred changes the byte), so that a wrong fragment in any single (tap, chunk)
step shows -- the bench-batch test of the ResNet lists derives its scale from the accumulator range and would let an error
of that size through. Also: one (tap, chunk) slice of the weights at a time against all others on the kernel zero point,
the experiment that located the first build's stale fragment registers at the chunk boundaries."""
import numpy as np
import pytest

import bench
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal
from oracle import o1

pytestmark = pytest.mark.gpu

KERNEL = "q8_conv_patch_mfma"


def _run(qnnp, H, W, S, GIC, GOC, batch, kernel, bias, inp, scale, kzp=127):
    (pt, pr, pb, pl), oh, ow = bench.conv_geometry(H, W, 3, 3, S, 1)
    oshape = o1.conv_shape(batch, H, W, (pt, pr, pb, pl), (3, 3), (S, S), (1, 1), 1, GIC, GOC, GIC)
    o1.set_threads(16)
    try:
        acc = o1.conv2d_acc(oshape, inp, kernel, bias, 127, kzp).reshape(-1, GOC)
    finally:
        o1.set_threads(1)
    out_scale = 0.25 / scale
    req = np.float32(np.float32(0.25) / np.float32(out_scale))
    expected = o1.requantize_rows(acc, req, 127, 0, 255).reshape(-1)
    op = qnnp.create_convolution2d_nhwc_q8(pt, pr, pb, pl, 3, 3, S, S, 1, 1, 1, GIC, GOC,
                                           127, 0.5, kzp, 0.5, kernel, bias, 127, float(out_scale), 0, 255, 0)
    try:
        d_in = to_device(inp)
        d_out = to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, batch, H, W, d_in, GIC, d_out, GOC)
        qnnp.run_operator(op)
        kname = qnnp.operator_kernel(op)
        out = from_device(d_out)
    finally:
        qnnp.delete_operator(op)
    return out, expected, kname


@pytest.mark.parametrize("kzp", [127, 126])
@pytest.mark.parametrize("shape", [(14, 14, 2, 256, 512, 96), (14, 14, 2, 512, 512, 96), (28, 28, 2, 256, 256, 40), (28, 28, 2, 512, 128, 24)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_chunked_shapes_at_a_fine_scale(qnnp, shape, kzp):
    H, W, S, GIC, GOC, batch = shape
    rng = np.random.default_rng(GIC * 7 + GOC + kzp)
    kernel = rng.integers(0, 256, size=(1, GOC, 3, 3, GIC)).astype(np.uint8)
    inp = rng.integers(0, 256, size=batch * H * W * GIC).astype(np.uint8)
    bias = rng.integers(-3000, 3001, size=GOC, dtype=np.int32)
    out, expected, kname = _run(qnnp, H, W, S, GIC, GOC, batch, kernel, bias, inp, 1e-4, kzp)
    assert kname == KERNEL, kname
    assert float(np.mean((expected > 0) & (expected < 255))) > 0.95
    assert_bytes_equal(out, expected, f"{shape} kzp {kzp} ({kname}) vs oracle at requantization scale 1e-4")


@pytest.mark.parametrize("shape", [(28, 28, 1, 128, 128, 8), (14, 14, 1, 256, 256, 24), (7, 7, 1, 512, 512, 96), (56, 56, 2, 64, 128, 8),
                                   (56, 56, 2, 128, 128, 8), (28, 28, 2, 128, 256, 24), (28, 28, 1, 64, 128, 8)],
                         ids=lambda s: "x".join(str(v) for v in s))
def test_whole_patch_shapes_at_a_fine_scale(qnnp, shape):
    """the same bar for the flavours that keep the whole patch (their fragment reads use the same deferred waits)"""
    H, W, S, GIC, GOC, batch = shape
    rng = np.random.default_rng(GIC * 5 + GOC + S)
    kernel = rng.integers(0, 256, size=(1, GOC, 3, 3, GIC)).astype(np.uint8)
    inp = rng.integers(0, 256, size=batch * H * W * GIC).astype(np.uint8)
    bias = rng.integers(-3000, 3001, size=GOC, dtype=np.int32)
    out, expected, kname = _run(qnnp, H, W, S, GIC, GOC, batch, kernel, bias, inp, 1e-4)
    assert kname == KERNEL, kname
    assert float(np.mean((expected > 0) & (expected < 255))) > 0.95
    assert_bytes_equal(out, expected, f"{shape} ({kname}) vs oracle at requantization scale 1e-4")


def test_one_tap_and_chunk_of_the_weights_at_a_time(qnnp):
    H, W, S, GIC, GOC, batch = 14, 14, 2, 256, 512, 96
    rng = np.random.default_rng(11)
    inp = rng.integers(0, 256, size=batch * H * W * GIC).astype(np.uint8)
    full = rng.integers(0, 256, size=(1, GOC, 3, 3, GIC)).astype(np.uint8)
    bias = np.zeros(GOC, np.int32)
    for t in range(9):
        for c in range(GIC // 128):
            kernel = np.full((1, GOC, 3, 3, GIC), 127, np.uint8)
            kernel[0, :, t // 3, t % 3, c * 128:(c + 1) * 128] = full[0, :, t // 3, t % 3, c * 128:(c + 1) * 128]
            out, expected, kname = _run(qnnp, H, W, S, GIC, GOC, batch, kernel, bias, inp, 1e-3)
            assert kname == KERNEL, kname
            assert_bytes_equal(out, expected, f"only tap {t}, channels {c * 128}..{c * 128 + 127} off the kernel zero point")
