"""GPU tier: the exact (shape, batch 128, automatically selected kernel) tuples bench.py times for BASELINE configs[4]
(the 31 MobileNetV2 layers of bench/convolution.cc:453-536 with the bench's quantization, bench/convolution.cc:71-74).
Kernel selection depends on the row count, so parity at small batches does not cover what the bench runs: here every
layer is created and set up exactly as bench.py's ConvLayer does, at batch 128, the kernel name is compared with the
committed dispatch table (tests/golden/sweep_kernels.json, regenerated with QNNP_WRITE_SWEEP_KERNELS=1 -- bench.py
prints the same names in its per-layer rows), and four images spread over the batch (first, second, middle, last) are
held to the scalar oracle byte for byte."""
import json
import os

import numpy as np
import pytest

import bench
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal
from oracle import o1

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "golden", "sweep_kernels.json")
BATCH = 128
SAMPLE = [0, 1, 63, 127]


def _expected_kernels():
    if not os.path.exists(TABLE):
        return None
    return json.load(open(TABLE))


@pytest.mark.parametrize("index", range(len(bench.MOBILENETV2)), ids=lambda i: f"layer{i + 1}")
def test_sweep_layer_at_bench_batch(qnnp, index):
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

    if os.environ.get("QNNP_WRITE_SWEEP_KERNELS"):
        table = _expected_kernels() or {}
        table[str(index + 1)] = kname
        json.dump(table, open(TABLE, "w"), indent=1, sort_keys=True)
    else:
        table = _expected_kernels()
        assert table is not None, f"{TABLE} missing: regenerate with QNNP_WRITE_SWEEP_KERNELS=1 on a GPU box"
        assert kname == table[str(index + 1)], f"layer {index + 1}: dispatch changed ({kname} vs committed {table[str(index + 1)]})"

    o1.set_threads(8)
    try:
        shape = o1.conv_shape(len(SAMPLE), H, W, (pt, pr, pb, pl), (KH, KW), (S, S), (D, D), G, GIC, GOC, cin)
        sub = np.concatenate([inp[i * in_img:(i + 1) * in_img] for i in SAMPLE])
        acc = o1.conv2d_acc(shape, sub, kernel, bias, 127, 127)
        # bench quantization: scales 0.5 * 0.5 / 0.5 -> requantization scale 0.5, zero point 127, clamp [0, 255]
        expected = o1.requantize_rows(acc.reshape(-1, cout), np.float32(0.5), 127, 0, 255).reshape(len(SAMPLE), out_img)
    finally:
        o1.set_threads(1)
    for j, i in enumerate(SAMPLE):
        assert_bytes_equal(out[i], expected[j], f"sweep layer {index + 1} ({kname}) image {i} of {BATCH} vs oracle")
