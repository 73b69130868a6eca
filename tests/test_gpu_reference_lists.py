"""GPU tier: every shape of the reference's convolution benchmark that no other test holds at the bench's batch
(bench/convolution.cc:108-942 -- ShuffleNet v1 g1 / g3 / g4 / g8 and v2, MobileNet v1, SqueezeNet 1.0 / 1.1, VGG, the three depthwise
lists; table: tests/golden/reference_bench_shapes.json, 318 distinct shapes beyond MobileNetV2 / ResNet-18 / ResNet-50 /
ShuffleNet v1 g2). Each is created and set up as bench.py's ConvLayer does, with whatever kernel auto picks, at batch 128 (fewer
images where 128 of them exceed 96 MB of input), and compared byte for byte with the compiled reference on ALL images where it
travelled (oracle/_ref) and with the scalar oracle on the first and last image otherwise (and always for light shapes). Two
quantizations: the output scale that spreads +-3 sigma of the accumulators over 0..255, and -- dense and grouped shapes -- the
requantization scale 1e-4, at which an accumulator error of a few hundred units moves bytes."""
import json
import os

import numpy as np
import pytest

import bench
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal
from oracle import o1, ref

pytestmark = pytest.mark.gpu

_TABLE = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_bench_shapes.json")))["lists"]
_HELD_ELSEWHERE = set(bench.RESNET18) | set(bench.RESNET50) | set(bench.SHUFFLENET_V1_G2) | set(bench.MOBILENETV2)
SHAPES = sorted({tuple(s) for rows in _TABLE.values() for s in rows} - _HELD_ELSEWHERE)
INDEX = {s: i for i, s in enumerate(SHAPES)}


def _shape_id(s):
    H, W, KH, KW, S, D, G, GIC, GOC = s
    return f"{H}x{W}_k{KH}s{S}d{D}_g{G}_{GIC}to{GOC}"


def _is_depthwise(s):
    return s[6] > 1 and s[7] == 1 and s[8] == 1


def _check(qnnp, shape, req_scale, flavour):
    H, W, KH, KW, S, D, G, GIC, GOC = shape
    (pt, pr, pb, pl), oh, ow = bench.conv_geometry(H, W, KH, KW, S, D)
    cin, cout = G * GIC, G * GOC
    in_img, out_img = H * W * cin, oh * ow * cout
    batch = int(min(128, max(4, (96 << 20) // in_img)))
    rng = np.random.default_rng(0x6C157 + 2 * INDEX[shape] + (flavour == "fine"))       # explicit seeds: the shape's index in the table
    kernel = rng.integers(0, 256, size=(G, GOC, KH, KW, GIC), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=G * GOC, dtype=np.int32)
    inp = rng.integers(0, 256, size=batch * in_img, dtype=np.uint8)
    izp = kzp = 127
    out_scale = 0.25 / req_scale                        # input scale 0.5 x kernel scale 0.5 / requantization scale
    ozp = 127
    op = qnnp.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                           izp, 0.5, kzp, 0.5, kernel, bias, ozp, float(out_scale), 0, 255, 0)
    try:
        d_in = to_device(inp)
        d_out = to_device(np.full(batch * out_img, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, batch, H, W, d_in, cin, d_out, cout)
        qnnp.run_operator(op)
        kname = qnnp.operator_kernel(op)
        out = from_device(d_out).reshape(batch, out_img)
    finally:
        qnnp.delete_operator(op)
    tag = f"{_shape_id(shape)} {flavour} ({kname})"
    ops_per_image = 2.0 * oh * ow * G * GIC * GOC * KH * KW
    checked = False
    if ref.available():
        rlib = ref.lib()
        want = np.full(batch * out_img, FILL, np.uint8)
        rop = rlib.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                                izp, 0.5, kzp, 0.5, kernel, bias, ozp, float(out_scale), 0, 255, 0)
        pool = rlib.threadpool(16)
        try:
            rlib.setup_convolution2d_nhwc_q8(rop, batch, H, W, inp, cin, want, cout)
            rlib.run_operator(rop, pool)
        finally:
            rlib.destroy_threadpool(pool)
            rlib.delete_operator(rop)
        mid = float(np.mean((want > 0) & (want < 255)))
        assert mid > 0.5, f"{tag}: only {mid:.1%} of the expected bytes lie inside (0, 255) -- not discriminating"
        assert_bytes_equal(out.reshape(-1), want, f"{tag}: all {batch} images vs the compiled reference")
        checked = True
    if not checked or ops_per_image < 2e8:
        sample = [0, batch - 1]
        o1.set_threads(16)
        try:
            oshape = o1.conv_shape(len(sample), H, W, (pt, pr, pb, pl), (KH, KW), (S, S), (D, D), G, GIC, GOC, cin)
            sub = np.concatenate([inp[i * in_img:(i + 1) * in_img] for i in sample])
            acc = o1.conv2d_acc(oshape, sub, kernel, bias, izp, kzp).reshape(-1, cout)
            req = np.float32(np.float32(0.5) * np.float32(0.5) / np.float32(out_scale))
            expected = o1.requantize_rows(acc, req, ozp, 0, 255).reshape(len(sample), out_img)
        finally:
            o1.set_threads(1)
        for j, i in enumerate(sample):
            assert_bytes_equal(out[i], expected[j], f"{tag} image {i} of {batch} vs oracle")


@pytest.mark.parametrize("shape", SHAPES, ids=_shape_id)
def test_reference_list_shape_spread_scale(qnnp, shape):
    """requantization scale that maps +-3 sigma of the accumulators (uniform bytes around zero point 127: sigma = 5461 sqrt(K)) to
    +-127 around the output zero point"""
    H, W, KH, KW, S, D, G, GIC, GOC = shape
    sigma = 5461.0 * np.sqrt(KH * KW * GIC)
    _check(qnnp, shape, min(0.9, 127.0 / (3.0 * sigma)), "spread")


@pytest.mark.parametrize("shape", [s for s in SHAPES if not _is_depthwise(s) and s[2] * s[3] * s[7] >= 64], ids=_shape_id)
def test_reference_list_shape_fine_scale(qnnp, shape):
    """requantization scale 1e-4 (10^4 accumulator units per output step; the accumulators' +-3 sigma stay inside 0..255 up to
    K = 4608): the dense and grouped shapes"""
    _check(qnnp, shape, 1e-4, "fine")
