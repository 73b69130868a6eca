"""GPU tier: the exact (shape, batch 128, automatically selected kernel) tuples bench.py times for BASELINE configs[4]
(the 31 MobileNetV2 layers of bench/convolution.cc:453-536). Kernel selection depends on the row count, so parity at
small batches does not cover what the bench runs: here every layer is created and set up exactly as bench.py's
ConvLayer does, at batch 128, the kernel name is compared with the committed dispatch table
(tests/golden/sweep_kernels.json, regenerated with QNNP_WRITE_SWEEP_KERNELS=1 -- bench.py prints the same names in its
per-layer rows), four images spread over the batch (first, second, middle, last) are held to the scalar oracle byte
for byte, and -- round 4 -- ALL 128 images of every layer and flavour to the compiled REFERENCE (oracle/_ref: the
reference's own SSE2 operators on the host threads, a fraction of a second per layer), because dispatch at batch 128
spreads row blocks over XCDs and segments and a fault confined to other images than the four would pass. Where the
prebuilt reference did not travel the four-image oracle check is what remains.

The bench's own quantization (bench/convolution.cc:71-74: scales 0.5 / 0.5 / 0.5 on full-range random data) saturates
98.8-99.9 % of the output bytes to 0 or 255, so with it a wrong accumulator is caught only if it changes sign. The
reference's testers derive the output scale and zero point from the accumulator range for exactly that reason
(test/convolution-operator-tester.h:407-412, test/gemm-microkernel-tester.h:228-241). Three flavours, each asserting
that fewer than 10 % of the EXPECTED bytes are 0 or 255:

  derived   full-range random tensors as the bench; output scale / zero point derived from the sampled images'
            accumulators as the testers do (zero points 127 / 127 as the bench) -> the shift >= 1 requantization
            sequences of the kernels, outputs spanning 0..255;
  realistic the same tensors at the bench's own "realistic" requantization scale 0.0125 (bench.py --out-scale 20,
            `extra.q8dwconv_5x5_dilated_and_realistic_scale`) -- kept when the accumulator range leaves < 10 % of the
            bytes saturated, otherwise the layer's derived scale is rounded DOWN to a power-of-two multiple of it so that
            the shift is at least 1 and the mantissa is 0.0125's;
  shift0    the bench's exact quantization (requantization scale 0.5 -> shift 0, the one-instruction epilogue the
            bench times): full-range random activations, but every output channel has just two non-zero centred weights
            of +-1 and a small bias, so the accumulators stay within +-256 and the outputs span 0..255 at scale 0.5.
"""
import json
import math
import os

import numpy as np
import pytest

import bench
from _cases import output_quantization
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal
from oracle import o1, ref

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "sweep_kernels.json")
BATCH = 128
SAMPLE = [0, 1, 63, 127]
MAX_SATURATED = 0.10


def _expected_kernels():
    if not os.path.exists(TABLE):
        return None
    return json.load(open(TABLE))


def sparse_unit_weights(rng, G, GOC, KH, KW, GIC, kzp):
    """Kernel whose centred weights (w - kzp) are zero except two +-1 entries per output channel."""
    per = KH * KW * GIC
    kernel = np.full((G * GOC, per), kzp, dtype=np.uint8)
    for n in range(G * GOC):
        pos = rng.choice(per, size=min(2, per), replace=False)
        kernel[n, pos] = (kzp + rng.choice([-1, 1], size=pos.size)).astype(np.uint8)
    return kernel.reshape(G, GOC, KH, KW, GIC)


def _sample_accumulators(index, inp, kernel, bias, izp, kzp):
    H, W, KH, KW, S, D, G, GIC, GOC = bench.MOBILENETV2[index]
    (pt, pr, pb, pl), oh, ow = bench.conv_geometry(H, W, KH, KW, S, D)
    cin, cout = G * GIC, G * GOC
    in_img = H * W * cin
    shape = o1.conv_shape(len(SAMPLE), H, W, (pt, pr, pb, pl), (KH, KW), (S, S), (D, D), G, GIC, GOC, cin)
    sub = np.concatenate([inp[i * in_img:(i + 1) * in_img] for i in SAMPLE])
    return o1.conv2d_acc(shape, sub, kernel, bias, izp, kzp).reshape(-1, cout)


@pytest.mark.parametrize("flavour", ["derived", "realistic", "shift0"])
@pytest.mark.parametrize("index", range(len(bench.MOBILENETV2)), ids=lambda i: f"layer{i + 1}")
def test_sweep_layer_at_bench_batch(qnnp, index, flavour):
    H, W, KH, KW, S, D, G, GIC, GOC = bench.MOBILENETV2[index]
    (pt, pr, pb, pl), oh, ow = bench.conv_geometry(H, W, KH, KW, S, D)
    rng = np.random.default_rng(100 + index)
    cin, cout = G * GIC, G * GOC
    in_img, out_img = H * W * cin, oh * ow * cout
    izp = kzp = 127                                  # bench/convolution.cc:71-72
    if flavour == "shift0":
        kernel = sparse_unit_weights(rng, G, GOC, KH, KW, GIC, kzp)
        bias = rng.integers(-20, 21, size=G * GOC, dtype=np.int32)
    else:
        kernel = rng.integers(0, 256, size=(G, GOC, KH, KW, GIC), dtype=np.uint8)
        bias = rng.integers(-10000, 10001, size=G * GOC, dtype=np.int32)
    inp = rng.integers(0, 256, size=BATCH * in_img, dtype=np.uint8)

    o1.set_threads(8)
    try:
        acc = _sample_accumulators(index, inp, kernel, bias, izp, kzp)
        if flavour == "shift0":
            req_scale, ozp = np.float32(0.5), 127           # the bench's quantization, exactly
            out_scale = 0.5
        else:
            oscale, ozp = output_quantization(acc)          # tester-style: outputs span 0..255
            req = 1.0 / float(oscale)
            if flavour == "realistic":
                # mantissa of the bench's realistic scale 0.0125, exponent from the data (never above the derived scale)
                m = 0.0125 / 2.0 ** math.floor(math.log2(0.0125))
                req = m * 2.0 ** math.floor(math.log2(req / m))
                ozp = int(max(0, min(255, round(127.5 - 0.5 * float(int(acc.min()) + int(acc.max())) * req))))
            out_scale = 0.25 / req                          # input scale 0.5 x kernel scale 0.5 / output scale
            req_scale = np.float32(np.float32(0.5) * np.float32(0.5) / np.float32(out_scale))
        expected = o1.requantize_rows(acc, req_scale, ozp, 0, 255).reshape(len(SAMPLE), out_img)
    finally:
        o1.set_threads(1)
    saturated = float(np.mean((expected == 0) | (expected == 255)))
    assert saturated < MAX_SATURATED, f"layer {index + 1} {flavour}: {saturated:.1%} of the expected bytes are 0 / 255 -- not discriminating"

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

    if os.environ.get("QNNP_WRITE_SWEEP_KERNELS"):
        if flavour == "shift0":
            table = _expected_kernels() or {}
            table[str(index + 1)] = kname
            json.dump(table, open(TABLE, "w"), indent=1, sort_keys=True)
    else:
        table = _expected_kernels()
        assert table is not None, f"{TABLE} missing: regenerate with QNNP_WRITE_SWEEP_KERNELS=1 on a GPU box"
        assert kname == table[str(index + 1)], f"layer {index + 1}: dispatch changed ({kname} vs committed {table[str(index + 1)]})"
    for j, i in enumerate(SAMPLE):
        assert_bytes_equal(out[i], expected[j], f"sweep layer {index + 1} {flavour} ({kname}) image {i} of {BATCH} vs oracle")
    if ref.available():
        want = _reference_output(pt, pr, pb, pl, KH, KW, S, D, G, GIC, GOC, izp, kzp, kernel, bias, ozp, float(out_scale),
                                 H, W, inp, cin, cout, out_img)
        assert_bytes_equal(out.reshape(-1), want, f"sweep layer {index + 1} {flavour} ({kname}): all {BATCH} images vs the compiled reference")


def _reference_output(pt, pr, pb, pl, KH, KW, S, D, G, GIC, GOC, izp, kzp, kernel, bias, ozp, out_scale, H, W, inp, cin, cout, out_img):
    """The same operator through the compiled reference (include/qnnpack.h ABI, its SSE2 microkernels), 16 host threads."""
    rlib = ref.lib()
    want = np.full(BATCH * out_img, FILL, np.uint8)
    rop = rlib.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                            izp, 0.5, kzp, 0.5, kernel, bias, ozp, out_scale, 0, 255, 0)
    pool = rlib.threadpool(16)
    try:
        rlib.setup_convolution2d_nhwc_q8(rop, BATCH, H, W, inp, cin, want, cout)
        rlib.run_operator(rop, pool)
    finally:
        rlib.destroy_threadpool(pool)
        rlib.delete_operator(rop)
    return want


@pytest.mark.parametrize("index", range(len(bench.MOBILENETV2)), ids=lambda i: f"layer{i + 1}")
def test_sweep_layer_bench_quantization_dispatch(qnnp, index):
    """The bench's literal create call (full-range weights AND scale 0.5): only the dispatch is pinned here -- its output is
    99 % saturated, the three flavours above are the parity checks -- plus a sign-level check of one image."""
    H, W, KH, KW, S, D, G, GIC, GOC = bench.MOBILENETV2[index]
    (pt, pr, pb, pl), oh, ow = bench.conv_geometry(H, W, KH, KW, S, D)
    rng = np.random.default_rng(100 + index)
    kernel = rng.integers(0, 256, size=(G, GOC, KH, KW, GIC), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=G * GOC, dtype=np.int32)
    cin, cout = G * GIC, G * GOC
    in_img, out_img = H * W * cin, oh * ow * cout
    inp = rng.integers(0, 256, size=BATCH * in_img, dtype=np.uint8)
    op = qnnp.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                           127, 0.5, 127, 0.5, kernel, bias, 127, 0.5, 0, 255, 0)
    try:
        d_in = to_device(inp)
        d_out = to_device(np.full(BATCH * out_img, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, BATCH, H, W, d_in, cin, d_out, cout)
        qnnp.run_operator(op)
        kname = qnnp.operator_kernel(op)
        out = from_device(d_out).reshape(BATCH, out_img)
    finally:
        qnnp.delete_operator(op)
    table = _expected_kernels()
    if table is not None and not os.environ.get("QNNP_WRITE_SWEEP_KERNELS"):
        assert kname == table[str(index + 1)], f"layer {index + 1}: dispatch changed ({kname} vs committed {table[str(index + 1)]})"
    o1.set_threads(8)
    try:
        shape = o1.conv_shape(1, H, W, (pt, pr, pb, pl), (KH, KW), (S, S), (D, D), G, GIC, GOC, cin)
        acc = o1.conv2d_acc(shape, inp[127 * in_img:], kernel, bias, 127, 127)
        expected = o1.requantize_rows(acc.reshape(-1, cout), np.float32(0.5), 127, 0, 255).reshape(out_img)
    finally:
        o1.set_threads(1)
    assert_bytes_equal(out[127], expected, f"sweep layer {index + 1} bench quantization ({kname}) image 127 vs oracle")
