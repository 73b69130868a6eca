"""GPU tier: the reference bench's OTHER convolution lists at the bench's batch (bench.py `extra.conv_lists`: ResNet-18,
ResNet-50, ShuffleNet v1 with 2 groups -- bench/convolution.cc:642-718, 147-184): the general implicit-GEMM path (7x7
stride 2 on 3 channels, 3x3 with 64..512 channels, stride-2 3x3 / 1x1, grouped 1x1 with 25..200 channels per group,
pointwise layers up to 2048 channels). Every distinct shape is created and set up exactly as bench.py's ConvLayer does,
at batch 128, with whatever kernel auto picks; two images (first, last) are held to the scalar oracle byte for byte and
ALL 128 images to the compiled reference where it travelled (oracle/_ref). Quantization as the reference's testers derive
it (test/convolution-operator-tester.h:407-412) so that the outputs span 0..255 instead of saturating, zero points
127 / 127 as the bench -- and a second flavour with kernel zero point 126, which has no zero-point-centred image."""
import numpy as np
import pytest

import bench
from _cases import output_quantization
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal
from oracle import o1, ref

pytestmark = pytest.mark.gpu

BATCH = 128
SAMPLE = [0, 127]
SHAPES = sorted(set(bench.RESNET18) | set(bench.RESNET50) | set(bench.SHUFFLENET_V1_G2))


def _shape_id(s):
    H, W, KH, KW, S, D, G, GIC, GOC = s
    return f"{H}x{W}_k{KH}s{S}_g{G}_{GIC}to{GOC}"


@pytest.mark.parametrize("kzp", [127, 126], ids=["kzp127", "kzp126"])
@pytest.mark.parametrize("shape", SHAPES, ids=_shape_id)
def test_conv_list_layer_at_bench_batch(qnnp, shape, kzp):
    H, W, KH, KW, S, D, G, GIC, GOC = shape
    if kzp == 126 and (G > 1 and GIC == 1):
        pytest.skip("depthwise rows are covered by the MobileNetV2 sweep tests")
    (pt, pr, pb, pl), oh, ow = bench.conv_geometry(H, W, KH, KW, S, D)
    rng = np.random.default_rng(0xC0117 + SHAPES.index(shape))        # explicit seed: the shape's index in the sorted table
    cin, cout = G * GIC, G * GOC
    in_img, out_img = H * W * cin, oh * ow * cout
    izp = 127
    kernel = rng.integers(0, 256, size=(G, GOC, KH, KW, GIC), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=G * GOC, dtype=np.int32)
    inp = rng.integers(0, 256, size=BATCH * in_img, dtype=np.uint8)

    o1.set_threads(16)
    try:
        oshape = o1.conv_shape(len(SAMPLE), H, W, (pt, pr, pb, pl), (KH, KW), (S, S), (D, D), G, GIC, GOC, cin)
        sub = np.concatenate([inp[i * in_img:(i + 1) * in_img] for i in SAMPLE])
        acc = o1.conv2d_acc(oshape, sub, kernel, bias, izp, kzp).reshape(-1, cout)
        oscale, ozp = output_quantization(acc)
        out_scale = 0.25 * float(oscale)                    # input scale 0.5 x kernel scale 0.5 / requantization scale
        req_scale = np.float32(np.float32(0.5) * np.float32(0.5) / np.float32(out_scale))
        expected = o1.requantize_rows(acc, req_scale, ozp, 0, 255).reshape(len(SAMPLE), out_img)
    finally:
        o1.set_threads(1)
    saturated = float(np.mean((expected == 0) | (expected == 255)))
    assert saturated < 0.10, f"{_shape_id(shape)}: {saturated:.1%} of the expected bytes are 0 / 255 -- not discriminating"

    op = qnnp.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                           izp, 0.5, kzp, 0.5, kernel, bias, ozp, float(out_scale), 0, 255, 0)
    try:
        d_in = to_device(inp)
        d_out = to_device(np.full(BATCH * out_img, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, BATCH, H, W, d_in, cin, d_out, cout)
        qnnp.run_operator(op)
        kname = qnnp.operator_kernel(op)
        out = from_device(d_out).reshape(BATCH, out_img)
    finally:
        qnnp.delete_operator(op)
    for j, i in enumerate(SAMPLE):
        assert_bytes_equal(out[i], expected[j], f"{_shape_id(shape)} kzp {kzp} ({kname}) image {i} of {BATCH} vs oracle")
    if ref.available():
        rlib = ref.lib()
        want = np.full(BATCH * out_img, FILL, np.uint8)
        rop = rlib.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                                izp, 0.5, kzp, 0.5, kernel, bias, ozp, float(out_scale), 0, 255, 0)
        pool = rlib.threadpool(16)
        try:
            rlib.setup_convolution2d_nhwc_q8(rop, BATCH, H, W, inp, cin, want, cout)
            rlib.run_operator(rop, pool)
        finally:
            rlib.destroy_threadpool(pool)
            rlib.delete_operator(rop)
        assert_bytes_equal(out.reshape(-1), want, f"{_shape_id(shape)} kzp {kzp} ({kname}): all {BATCH} images vs the compiled reference")
