"""GPU tier: BASELINE.json's full-size configurations. The scalar oracle cannot run 137 GOP in
seconds, so full sizes are checked by (a) the oracle on a sample of rows / images that straddles
every tile edge, and (b) size-independent properties of the operator: outputs of a row / image do
not depend on where it sits in the batch (permutation equivariance), checked over the WHOLE output."""
import numpy as np
import pytest

from _cases import ConvCase, conv_tensors, output_quantization
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal
from oracle import o1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant,kernel", [(0, "q8_gemm_mfma_256x256_c16"), (20, "q8_gemm_mfma_256x256_c"), (28, "q8_gemm_mfma_256x256_r16"),
                                            (15, "q8_gemm_mfma_256x256_lean"), (2, "q8_gemm_mfma_256x256")],
                         ids=["auto", "centred_32x32x32", "rowsum_16x16x64", "lean", "general"])
def test_c2_q8gemm_4096_cubed(qnnp, variant, kernel):
    """configs[1]: q8gemm M=N=K=4096 through qnnp_fully_connected_nc_q8 -- the shipped kernel (round 6: the
    zero-point-centred flavour on v_mfma_i32_16x16x64_i8, q8gemm256x.hip, is what "auto" picks for this shape and these
    zero points), the 32x32x32 flavour it replaced, and the lean and general flavours of the kernel both came from (what
    other zero points run on)."""
    import torch
    qnnp.set_option("gemm_kernel", variant)
    M = N = K = 4096
    rng = np.random.default_rng(0x51A0 + 2)
    a = rng.integers(0, 256, size=(M, K), dtype=np.uint8)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    izp, kzp = 127, 127
    o1.set_threads(8)
    sample = np.unique(np.concatenate([
        [0, 1, 31, 32, 127, 128, 255, 256, 257, 2047, 2048, 4094, 4095], rng.integers(0, M, size=40)]))
    acc = o1.gemm_acc(np.ascontiguousarray(a[sample]), w, bias, izp, kzp)
    oscale, ozp = output_quantization(acc)
    expected = o1.requantize_rows(acc, np.float32(1.0) / oscale, ozp, 0, 255)
    o1.set_threads(1)

    op = qnnp.create_fully_connected_nc_q8(K, N, izp, 1.0, kzp, 1.0, w, bias, ozp, float(oscale), 0, 255)
    try:
        d_a = to_device(a.reshape(-1))
        d_c = to_device(np.full(M * N, FILL, np.uint8))
        qnnp.setup_fully_connected_nc_q8(op, M, d_a, K, d_c, N)
        qnnp.run_operator(op)
        assert qnnp.operator_kernel(op) == kernel, qnnp.operator_kernel(op)
        c = from_device(d_c).reshape(M, N)
        assert_bytes_equal(c[sample].reshape(-1), expected.reshape(-1), "4096^3 sampled rows vs oracle")
        assert c.min() < 64 and c.max() > 192, "outputs should span the uint8 range"
        # permutation equivariance over the full output: reversing the rows of A reverses the rows of C
        d_a2 = torch.flip(d_a.view(M, K), dims=[0]).contiguous().view(-1)
        d_c2 = to_device(np.full(M * N, FILL, np.uint8))
        qnnp.setup_fully_connected_nc_q8(op, M, d_a2, K, d_c2, N)
        qnnp.run_operator(op)
        c2 = from_device(d_c2).reshape(M, N)
        assert np.array_equal(c2[::-1], c), "row permutation equivariance violated at 4096^3"
    finally:
        qnnp.set_option("gemm_kernel", 0)
        qnnp.delete_operator(op)


@pytest.mark.parametrize("variant,kzp,kernel_name", [(0, 127, "q8_conv_wave_ws_c16_mfma"), (0, 128, "q8_conv_wave_ws_c16_mfma"),
                                                     (27, 127, "q8_conv_wave_ws_c_mfma"),
                                                     (0, 126, "q8_conv_wave_ws_mfma"), (1, 127, None),
                                                     (2, 127, "q8_gemm_mfma_256x256_conv")],
                         ids=["auto_kzp127", "auto_kzp128", "centred_32x32x32", "auto_kzp126", "offset_table_generic", "offset_table_256x256"])
def test_c3_q8conv_3x3_56x56x64_batch128(qnnp, variant, kzp, kernel_name):
    """configs[2]: 3x3 s1 pad1 conv, 56x56x64 -> 64, batch 128. "auto" is the weight-stationary wave kernel, which
    computes its patch addresses arithmetically (q8convwave.hip) -- with the zero-point-centred image for kernel zero
    points 127 (the bench's) and 128, with pixel sums for any other; "gemm_kernel" 1 and 2 force the two kernels that read the
    device-side OFFSET TABLE of csrc/indirection.c -- the path BASELINE configs[2] names literally -- at the full size."""
    import torch
    case = ConvCase("c3_fullsize", (56, 56), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=128, kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    img = 56 * 56 * 64
    sample = [0, 1, 63, 127]
    o1.set_threads(8)
    shape = o1.conv_shape(len(sample), 56, 56, case.padding, (3, 3), (1, 1), (1, 1), 1, 64, 64, 64)
    sub = np.concatenate([inp[i * img:(i + 1) * img] for i in sample])
    acc = o1.conv2d_acc(shape, sub, kernel, bias, case.izp, case.kzp)
    oscale, ozp = output_quantization(acc)
    expected = o1.requantize_rows(acc.reshape(-1, 64), np.float32(1.0) / oscale, ozp, 0, 255).reshape(len(sample), -1)
    o1.set_threads(1)

    qnnp.set_option("gemm_kernel", variant)
    op = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 1, 64, 64, case.izp, 1.0, case.kzp, 1.0,
                                           kernel, bias, ozp, float(oscale), 0, 255, 0)
    try:
        d_in = to_device(inp)
        d_out = to_device(np.full(128 * img, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, 128, 56, 56, d_in, 64, d_out, 64)
        qnnp.run_operator(op)
        if kernel_name is not None:
            assert qnnp.operator_kernel(op) == kernel_name, qnnp.operator_kernel(op)
        else:
            assert qnnp.operator_kernel(op).startswith("q8_igemm_mfma"), qnnp.operator_kernel(op)
        out = from_device(d_out).reshape(128, img)
        for j, i in enumerate(sample):
            assert_bytes_equal(out[i], expected[j], f"C3 image {i} vs oracle")
        # image-permutation equivariance over all 128 images
        d_in2 = torch.flip(d_in.view(128, img), dims=[0]).contiguous().view(-1)
        d_out2 = to_device(np.full(128 * img, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, 128, 56, 56, d_in2, 64, d_out2, 64)
        qnnp.run_operator(op)
        out2 = from_device(d_out2).reshape(128, img)
        assert np.array_equal(out2[::-1], out), "image permutation equivariance violated"
    finally:
        qnnp.set_option("gemm_kernel", 0)
        qnnp.delete_operator(op)


# configs[3]: every MobileNetV2 depthwise layer (bench/convolution.cc:461-518): (H, stride, C)
MOBILENETV2_DW = [(112, 1, 32), (112, 2, 96), (56, 1, 144), (56, 2, 144), (28, 1, 192), (28, 2, 192),
                  (14, 1, 384), (14, 1, 576), (14, 2, 576), (7, 1, 960)]


@pytest.mark.parametrize("kzp", [127, 100], ids=lambda v: f"kzp{v}")
@pytest.mark.parametrize("batch", [8, 128], ids=lambda v: f"batch{v}")
@pytest.mark.parametrize("h,s,c", MOBILENETV2_DW, ids=lambda v: str(v))
def test_c4_mobilenetv2_depthwise_layers(qnnp, h, s, c, batch, kzp):
    """Batch 128 is what bench.py times (segments per wave, XCD ranges and the walk all depend on the row count);
    kernel zero point 127 (the bench's) takes the int8 dot-product walk at stride 1, kernel zero point 100 leaves
    neither w - kzp nor kzp - w inside int8 and forces the int16 pair walk (`dw_wrange == 0`). Every image is checked."""
    if kzp != 127 and batch != 128:
        pytest.skip("the pair walk at small batches is covered by tests/test_gpu_dwcol.py")
    case = ConvCase(f"c4_dw_{h}_{s}_{c}", (h, h), (3, 3), (1, 1, 1, 1), subsampling=(s, s), groups=c, batch=batch, kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    o1.set_threads(8)
    shape = o1.conv_shape(batch, h, h, case.padding, (3, 3), (s, s), (1, 1), c, 1, 1, c)
    acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
    oscale, ozp = output_quantization(acc)
    expected = o1.requantize_rows(acc.reshape(-1, c), np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
    o1.set_threads(1)
    assert np.mean((expected == 0) | (expected == 255)) < 0.10
    op = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, s, s, 1, 1, c, 1, 1, case.izp, 1.0, case.kzp, 1.0,
                                           kernel, bias, ozp, float(oscale), 0, 255, 0)
    try:
        d_in = to_device(inp)
        d_out = to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, batch, h, h, d_in, c, d_out, c)
        qnnp.run_operator(op)
        kname = qnnp.operator_kernel(op)
        # the column-sliding window kernel (q8dwconv.hip make_plan); kzp 127: int8 walk at stride 1
        assert kname == ("q8_dwconv_col_3x3_dot4" if s == 1 and kzp == 127 else "q8_dwconv_col_3x3"), kname
        assert_bytes_equal(from_device(d_out), expected, f"C4 depthwise {h}x{h} s{s} C{c} batch {batch} kzp {kzp} vs oracle")
    finally:
        qnnp.delete_operator(op)


# configs[4] building blocks: every pointwise / first-layer shape of the MobileNetV2 sweep at batch 4
MOBILENETV2_GEMM = [(112, 32, 16), (112, 16, 96), (56, 96, 24), (56, 24, 144), (56, 144, 24), (28, 144, 32),
                    (28, 32, 192), (28, 192, 32), (14, 192, 64), (14, 64, 384), (14, 384, 64), (14, 384, 96),
                    (14, 96, 576), (14, 576, 96), (7, 576, 160), (7, 160, 960), (7, 960, 160), (7, 960, 320),
                    (7, 320, 1280), (1, 1280, 1000)]


@pytest.mark.parametrize("h,cin,cout", MOBILENETV2_GEMM, ids=lambda v: str(v))
def test_c5_mobilenetv2_pointwise_layers(qnnp, h, cin, cout):
    batch = 4 if h > 14 else 16
    case = ConvCase(f"c5_pw_{h}_{cin}_{cout}", (h, h), gic=cin, goc=cout, batch=batch)
    inp, kernel, bias = conv_tensors(case)
    o1.set_threads(8)
    a = inp.reshape(batch * h * h, cin)
    acc = o1.gemm_acc(a, kernel.reshape(cout, cin), bias, case.izp, case.kzp)
    oscale, ozp = output_quantization(acc)
    expected = o1.requantize_rows(acc, np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
    o1.set_threads(1)
    op = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, cin, cout, case.izp, 1.0, case.kzp, 1.0,
                                           kernel, bias, ozp, float(oscale), 0, 255, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, batch, h, h, d_in, cin, d_out, cout)
        qnnp.run_operator(op)
        if cin <= 256 and h >= 14 and cin * cout <= 60000:
            # short-K layers over many rows: the barrier-free streaming kernel must be the one that ran
            assert qnnp.operator_kernel(op) == "q8_pw_stream_mfma", qnnp.operator_kernel(op)
        assert_bytes_equal(from_device(d_out), expected,
                           f"C5 pointwise {h}x{h} {cin}->{cout} vs oracle kernel={qnnp.operator_kernel(op)}")
    finally:
        qnnp.delete_operator(op)


def test_c5_mobilenetv2_first_layer(qnnp):
    """3x3 s2 3->32 on 224x224 (bench/convolution.cc:457), padding as the bench computes it (:42-47)."""
    batch = 2
    case = ConvCase("c5_first", (224, 224), (3, 3), (1, 1, 1, 1), subsampling=(2, 2), gic=3, goc=32, batch=batch)
    inp, kernel, bias = conv_tensors(case)
    o1.set_threads(8)
    shape = o1.conv_shape(batch, 224, 224, case.padding, (3, 3), (2, 2), (1, 1), 1, 3, 32, 3)
    acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
    oscale, ozp = output_quantization(acc)
    expected = o1.requantize_rows(acc.reshape(-1, 32), np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
    o1.set_threads(1)
    op = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 2, 2, 1, 1, 1, 3, 32, case.izp, 1.0, case.kzp, 1.0,
                                           kernel, bias, ozp, float(oscale), 0, 255, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, batch, 224, 224, d_in, 3, d_out, 32)
        qnnp.run_operator(op)
        assert qnnp.operator_kernel(op) == "q8_conv_c3rows_lds_mfma", qnnp.operator_kernel(op)   # (round 6: W * 3 % 16 == 0, k_zp 127)
        assert_bytes_equal(from_device(d_out), expected, "C5 first layer vs oracle")
    finally:
        qnnp.delete_operator(op)


@pytest.mark.parametrize("kzp,kernel", [(127, "q8_gemm_mfma_256x256_c16"), (128, "q8_gemm_mfma_256x256_c16"),
                                        (126, "q8_gemm_mfma_256x256_r16")], ids=["kzp127", "kzp128", "kzp126"])
def test_c2_q8gemm_4096_cubed_full_output_vs_compiled_reference(qnnp, kzp, kernel):
    """(both centring classes of the shipped kernel, and a zero point that keeps the lean flavour)
    configs[1], every one of the 16.8 M output bytes: the compiled REFERENCE (oracle/_ref, its SSE2 q8gemm under
    qnnp_fully_connected_nc_q8, all host threads) computes the same 4096^3 problem with bench.py's quantization
    (bench/q8gemm.cc:103: zero points 127, scale 0.75, clamp [1, 254]) in well under a minute; the device output
    must equal it byte for byte. Where the prebuilt reference did not travel, the oracle restatement (O1, pinned to it)
    takes over on 512 rows so the test never silently does nothing."""
    from oracle import ref
    M = N = K = 4096
    rng = np.random.default_rng(0xC2F011)
    a = rng.integers(0, 256, size=(M, K), dtype=np.uint8)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    op = qnnp.create_fully_connected_nc_q8(K, N, 127, 0.75, kzp, 1.0, w, bias, 127, 1.0, 1, 254)
    try:
        d_a = to_device(a.reshape(-1))
        d_c = to_device(np.full(M * N, FILL, np.uint8))
        qnnp.setup_fully_connected_nc_q8(op, M, d_a, K, d_c, N)
        qnnp.run_operator(op)
        assert qnnp.operator_kernel(op) == kernel, qnnp.operator_kernel(op)
        got = from_device(d_c).reshape(M, N)
    finally:
        qnnp.delete_operator(op)
    if ref.available():
        rlib = ref.lib()
        want = np.full(M * N, FILL, np.uint8)
        rop = rlib.create_fully_connected_nc_q8(K, N, 127, 0.75, kzp, 1.0, w, bias, 127, 1.0, 1, 254)
        pool = rlib.threadpool(16)
        rlib.setup_fully_connected_nc_q8(rop, M, a.reshape(-1), K, want, N)
        rlib.run_operator(rop, pool)
        rlib.destroy_threadpool(pool)
        rlib.delete_operator(rop)
        assert_bytes_equal(got.reshape(-1), want, "4096^3 FULL output vs the compiled reference")
    else:
        rows = np.arange(0, M, 8)
        o1.set_threads(8)
        acc = o1.gemm_acc(np.ascontiguousarray(a[rows]), w, bias, 127, kzp)
        want = o1.requantize_rows(acc, np.float32(0.75), 127, 1, 254)
        o1.set_threads(1)
        assert_bytes_equal(got[rows].reshape(-1), want.reshape(-1), "4096^3 every 8th row vs oracle (no prebuilt reference here)")


@pytest.mark.parametrize("h,c,s", [(56, 72, 2), (28, 240, 1), (14, 672, 1)], ids=lambda v: str(v))
def test_next_row_depthwise_5x5_bench_shapes_batch128(qnnp, h, c, s):
    """bench.py's MobileNetV3-style 5x5 depthwise rows (extra.q8dwconv_5x5_...) at the benched batch, whole output
    against the oracle, with accumulator-derived quantization so that the bytes span their range."""
    batch = 128
    case = ConvCase(f"dw5_{h}_{c}_s{s}", (h, h), (5, 5), (2, 2, 2, 2), subsampling=(s, s), groups=c, gic=1, goc=1, batch=batch)
    inp, kernel, bias = conv_tensors(case)
    o1.set_threads(16)
    try:
        shape = o1.conv_shape(batch, h, h, case.padding, (5, 5), (s, s), (1, 1), c, 1, 1, c)
        acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
        oscale, ozp = output_quantization(acc)
        expected = o1.requantize_rows(acc.reshape(-1, c), np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
    finally:
        o1.set_threads(1)
    assert np.count_nonzero((expected == 0) | (expected == 255)) < 0.1 * expected.size
    op = qnnp.create_convolution2d_nhwc_q8(2, 2, 2, 2, 5, 5, s, s, 1, 1, c, 1, 1, case.izp, 1.0, case.kzp, 1.0, kernel, bias,
                                           ozp, float(oscale), 0, 255, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, batch, h, h, d_in, c, d_out, c)
        qnnp.run_operator(op)
        assert qnnp.operator_kernel(op) == "q8_dwconv_col_5x5_dot4"
        assert_bytes_equal(from_device(d_out), expected, f"kernel H, {h}x{h}x{c} stride {s}, batch 128")
    finally:
        qnnp.delete_operator(op)


def test_next_row_deconvolution_3x3_s2_bench_shape_batch128(qnnp):
    """bench.py's next_rows.q8deconv_3x3s2_28x28x64_32 at the benched batch: 28x28x64 -> 56x56x32, 3x3, stride 2, padding 1,
    adjustment 1 -- whole output against the oracle."""
    batch, H, cin, cout = 128, 28, 64, 32
    rng = np.random.default_rng(0xDEC0)
    inp = rng.integers(0, 256, size=batch * H * H * cin, dtype=np.uint8)
    kernel = rng.integers(0, 256, size=(1, cin, 3, 3, cout), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=cout, dtype=np.int32)
    izp, kzp = 127, 127
    o1.set_threads(16)
    try:
        shape = o1.conv_shape(batch, H, H, (1, 1, 1, 1), (3, 3), (2, 2), (1, 1), 1, cin, cout)
        acc = o1.deconv2d_acc(shape, (1, 1), inp, kernel, bias, izp, kzp)
        oscale, ozp = output_quantization(acc)
        expected = o1.requantize_rows(acc.reshape(-1, cout), np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
        oh, ow = o1.deconv_output_hw(shape, (1, 1))
    finally:
        o1.set_threads(1)
    assert (oh, ow) == (56, 56)
    op = qnnp.create_deconvolution2d_nhwc_q8(1, 1, 1, 1, 1, 1, 3, 3, 2, 2, 1, 1, 1, cin, cout, izp, 1.0, kzp, 1.0, kernel, bias,
                                             ozp, float(oscale), 0, 255, 0)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_deconvolution2d_nhwc_q8(op, batch, H, H, d_in, cin, d_out, cout)
        qnnp.run_operator(op)
        assert qnnp.operator_kernel(op) == "q8_deconv_s2_stream_3x3"
        assert_bytes_equal(from_device(d_out), expected, "stride-2 deconvolution, bench shape, batch 128")
    finally:
        qnnp.delete_operator(op)
