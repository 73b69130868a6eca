"""GPU tier: the residual add folded into a convolution (qnnp_gfx950_attach_residual_add, residual.c; epilogues in
hip/q8pwconv.hip). The reference runs a convolution operator and then an add operator (src/add.c + q8vadd); here
the same bytes must come out of ONE operator, whether the add rides in the convolution kernel's epilogue or is
launched in place behind it:

  * every MobileNetV2 project layer that has a residual, at the bench batch, on each kernel that can take it --
    against the scalar oracle (convolution accumulators -> Q31 requantization -> qnnp_add_quantize) AND against
    the library's own two-operator sequence, with the epilogue/fallback split asserted;
  * ragged row counts, strided tensors, misaligned residuals (fallback), depthwise and 3x3 convolutions (fallback);
  * a whole MobileNetV2 with all ten adds folded, against the oracle and as one hipGraph;
  * status codes and the detach-on-setup rule."""
import numpy as np
import pytest
import torch

from _cases import output_quantization
from _gpu import from_device, to_device
from examples import mobilenetv2 as mnv2
from oracle import o1
from test_gpu_network import oracle_forward

pytestmark = pytest.mark.gpu

ADD_Q = dict(a_zp=121, a_scale=0.75, b_scale=1.25, y_zp=133, y_scale=0.96875, y_min=3, y_max=250)


def _oracle(batch, hw, cin, cout, k, groups, x, kernel, bias, izp, kzp, res, res_stride, out_stride):
    pad = k // 2
    shape = o1.conv_shape(batch, hw, hw, (pad,) * 4, (k, k), (1, 1), (1, 1), groups, cin // groups, cout // groups)
    acc = o1.conv2d_acc(shape, x, kernel, bias, izp, kzp)
    oscale, ozp = output_quantization(acc)
    rows = batch * hw * hw
    conv = o1.requantize_rows(acc.reshape(-1, cout), np.float32(1.0) / oscale, ozp, 0, 255)
    y = np.full(rows * out_stride, 0xA5, np.uint8)
    o1.add_q8(rows, cout, ADD_Q["a_zp"], ADD_Q["a_scale"], ozp, ADD_Q["b_scale"], ADD_Q["y_zp"], ADD_Q["y_scale"],
              ADD_Q["y_min"], ADD_Q["y_max"], res, res_stride, np.ascontiguousarray(conv).reshape(-1), cout, y, out_stride)
    return y, float(oscale), int(ozp)


def _run_case(qnnp, batch, hw, cin, cout, k=1, groups=1, variant=0, res_stride=None, out_stride=None, res_misalign=0,
              seed=0):
    rng = np.random.default_rng(seed + 7 * hw + cin + cout)
    res_stride = res_stride or cout
    out_stride = out_stride or cout
    rows = batch * hw * hw
    izp, kzp = 119, 131
    x = rng.integers(0, 256, rows * cin, dtype=np.uint8)
    gic, goc = cin // groups, cout // groups
    kernel = rng.integers(0, 256, (groups, goc, k, k, gic), dtype=np.uint8)
    bias = rng.integers(-5000, 5000, cout).astype(np.int32)
    res = rng.integers(0, 256, rows * res_stride, dtype=np.uint8)
    expected, oscale, ozp = _oracle(batch, hw, cin, cout, k, groups, x, kernel, bias, izp, kzp, res, res_stride, out_stride)

    d_x, d_res = to_device(x), to_device(res, misalign=res_misalign)
    d_out = to_device(np.full(rows * out_stride, 0xA5, np.uint8))
    d_mid = to_device(np.zeros(rows * cout, np.uint8))
    d_two = to_device(np.full(rows * out_stride, 0xA5, np.uint8))
    pad = k // 2
    conv = qnnp.create_convolution2d_nhwc_q8(pad, pad, pad, pad, k, k, 1, 1, 1, 1, groups, gic, goc, izp, 1.0, kzp, 1.0,
                                             kernel, bias, ozp, oscale, 0, 255, 0)
    add = qnnp.create_add_nc_q8(cout, ADD_Q["a_zp"], ADD_Q["a_scale"], ozp, ADD_Q["b_scale"], ADD_Q["y_zp"],
                                ADD_Q["y_scale"], ADD_Q["y_min"], ADD_Q["y_max"], 0)
    try:
        qnnp.set_option("gemm_kernel", variant)
        # the reference's form: convolution, then add(a = residual, b = convolution output)
        qnnp.setup_convolution2d_nhwc_q8(conv, batch, hw, hw, d_x, cin, d_mid, cout)
        qnnp.run_operator(conv)
        assert qnnp.operator_residual_folded(conv) == -1
        qnnp.setup_add_nc_q8(add, rows, d_res, res_stride, d_mid, cout, d_two, out_stride)
        qnnp.run_operator(add)
        two = from_device(d_two)
        # one operator
        qnnp.setup_convolution2d_nhwc_q8(conv, batch, hw, hw, d_x, cin, d_out, out_stride)
        qnnp.attach_residual_add(conv, add, d_res, res_stride)
        qnnp.delete_operator(add)                      # its parameters were copied
        add = None
        qnnp.run_operator(conv)
        got = from_device(d_out)
        kernel_name, folded = qnnp.operator_kernel(conv), qnnp.operator_residual_folded(conv)
    finally:
        qnnp.set_option("gemm_kernel", 0)
        qnnp.delete_operator(conv)
        if add is not None:
            qnnp.delete_operator(add)
    bad = np.flatnonzero(got != expected)
    assert bad.size == 0, f"{kernel_name} folded={folded}: {bad.size} of {got.size} bytes differ from the oracle (first {bad[:4].tolist()})"
    assert np.array_equal(got, two), f"{kernel_name}: differs from the two-operator sequence"
    return kernel_name, folded


# (pixels per side, channels in, channels out): the MobileNetV2 project layers that feed a residual add
MOBILENET_RESIDUAL_PROJECTS = [(56, 144, 24), (28, 192, 32), (14, 384, 64), (14, 576, 96), (7, 960, 160)]


@pytest.mark.parametrize("hw,cin,cout", MOBILENET_RESIDUAL_PROJECTS)
def test_mobilenet_project_layers_fold_at_bench_batch(qnnp, hw, cin, cout):
    o1.set_threads(16)
    try:
        kernel_name, folded = _run_case(qnnp, 128, hw, cin, cout)
    finally:
        o1.set_threads(1)
    assert kernel_name.startswith("q8_pw_stream"), kernel_name
    assert folded == 1, f"{kernel_name} did not carry the add in its epilogue"


@pytest.mark.parametrize("variant,batch,hw,cin,cout,want", [
    (5, 9, 28, 64, 64, "q8_pw_stream_mfma"),          # staged flavour, whole dense blocks, ragged last unit (7056 rows)
    (5, 5, 28, 96, 192, "q8_pw_stream_mfma"),         # staged flavour with the channels split over workgroup columns
    (5, 3, 33, 40, 24, "q8_pw_stream_mfma"),          # direct-store flavour (24 channels), 3267 rows
    (5, 3, 33, 72, 32, "q8_pw_stream_mfma"),          # one channel block per row
    (5, 3, 33, 48, 24, "q8_pw_stream_mfma"),          # dense 24-byte rows, 16-byte loads: the staged flavour's contiguous copy-out, odd row count
    (5, 4, 56, 144, 24, "q8_pw_stream_mfma"),         # MobileNetV2's 56 x 56 x 144 -> 24 project layer
    (9, 7, 14, 384, 64, "q8_pw_stream_longk_mfma"),   # 1372 rows: ragged
    (9, 3, 14, 576, 96, "q8_pw_stream_longk_mfma"),
    (6, 3, 7, 960, 160, "q8_pw_stream_gwk_mfma"),     # 147 rows: ragged block, 5 channel blocks, K split over the waves
    (6, 16, 7, 192, 48, "q8_pw_stream_gw_mfma"),      # 784 rows, K < 256: one wave per block
])
def test_each_carrying_kernel(qnnp, variant, batch, hw, cin, cout, want):
    kernel_name, folded = _run_case(qnnp, batch, hw, cin, cout, variant=variant, seed=variant)
    assert kernel_name.startswith(want), kernel_name
    assert folded == 1


def test_gwk_kernel_carries_the_add(qnnp):
    # fully-connected sized problem: the K-split flavour (one workgroup per 32x32 block)
    kernel_name, folded = _run_case(qnnp, 2, 4, 1280, 64, seed=3)
    assert folded == 1 and kernel_name.startswith("q8_pw_stream_gw"), (kernel_name, folded)


@pytest.mark.parametrize("kwargs,why", [
    (dict(batch=4, hw=28, cin=64, cout=64, res_stride=80), "residual rows laid out differently from the output rows"),
    (dict(batch=4, hw=28, cin=64, cout=64, res_misalign=4), "residual not 16-byte aligned"),
    (dict(batch=4, hw=28, cin=64, cout=21, out_stride=24, res_stride=24), "byte stores"),
    (dict(batch=2, hw=12, cin=32, cout=32, k=3), "3x3 convolution"),
    (dict(batch=2, hw=12, cin=32, cout=32, k=3, groups=32), "depthwise"),
    (dict(batch=2, hw=10, cin=16, cout=16, groups=2), "grouped pointwise"),
    (dict(batch=1, hw=9, cin=24, cout=40, variant=1), "generic tile kernel"),
])
def test_fallback_is_the_add_kernel_behind_the_convolution(qnnp, kwargs, why):
    kernel_name, folded = _run_case(qnnp, seed=11, **kwargs)
    assert folded == 0, f"{why}: {kernel_name} claims to carry the add"


def test_strided_output_and_residual_fold(qnnp):
    # pixels wider than the channels on both sides (a slice of a concatenated tensor), same stride: still in the epilogue
    kernel_name, folded = _run_case(qnnp, 4, 28, 64, 64, res_stride=96, out_stride=96, seed=5)
    assert folded == 1, kernel_name


@pytest.mark.parametrize("input_hw,batch", [(96, 2), (224, 1), (128, 12)])
def test_network_with_folded_adds_matches_oracle(qnnp, input_hw, batch):
    plan = mnv2.build_plan(input_hw=input_hw, classes=1000, seed=0x51A0 + input_hw)
    rng = np.random.default_rng(77 + input_hw)
    image = rng.integers(0, 256, size=batch * input_hw * input_hw * 3, dtype=np.uint8)
    o1.set_threads(16)
    try:
        expected, quant = oracle_forward(plan, image, batch)
    finally:
        o1.set_threads(1)
    net = mnv2.DeviceNetwork(qnnp, torch, plan, batch, quant, fold_adds=True)
    try:
        assert len(net.folded) == 10 and len(net.schedule) == 54
        hidden = {plan.ops[c].dst for c, _ in net.folded.values()}          # project outputs: never written
        net.buffers[0].copy_(torch.from_numpy(image))
        net.run()
        for op in plan.ops:
            if op.dst in hidden:
                continue
            got = from_device(net.buffers[op.dst])
            bad = np.flatnonzero(got != expected[op.dst])
            assert bad.size == 0, f"{op.name}: {bad.size} of {got.size} bytes differ"
        assert "q8_vadd_flat" not in set(net.kernels.values())
        net.capture()
        for t in net.buffers:
            if t != 0:
                net.buffers[t].zero_()
        torch.cuda.synchronize()
        net.replay()
        for op in plan.ops:
            if op.dst not in hidden:
                assert np.array_equal(from_device(net.buffers[op.dst]), expected[op.dst]), f"graph replay: {op.name}"
    finally:
        net.close()


def test_status_codes_and_detach(qnnp):
    from qnnpack_amd import Status
    k1 = np.zeros((1, 16, 1, 1, 32), np.uint8)
    b16 = np.zeros(16, np.int32)
    conv = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 32, 16, 4, 1.0, 2, 1.0, k1, b16, 5, 2.0, 0, 255, 0)
    add = qnnp.create_add_nc_q8(16, 1, 1.0, 5, 1.0, 6, 2.0, 0, 255, 0)
    add8 = qnnp.create_add_nc_q8(8, 1, 1.0, 5, 1.0, 6, 2.0, 0, 255, 0)
    buf = to_device(np.zeros(1 << 16, np.uint8))
    out = to_device(np.zeros(1 << 12, np.uint8))
    res = to_device(np.zeros(1 << 12, np.uint8))
    host = np.zeros(1 << 16, np.uint8)
    try:
        assert qnnp.attach_residual_add_status(conv, add, res, 16) == Status.invalid_parameter       # before any setup
        qnnp.setup_convolution2d_nhwc_q8(conv, 1, 8, 8, buf, 32, out, 16)
        assert qnnp.attach_residual_add_status(None, add, res, 16) == Status.invalid_parameter
        assert qnnp.attach_residual_add_status(conv, None, res, 16) == Status.invalid_parameter
        assert qnnp.attach_residual_add_status(conv, conv, res, 16) == Status.invalid_parameter      # not an add operator
        assert qnnp.attach_residual_add_status(add, add, res, 16) == Status.invalid_parameter        # not a convolution
        assert qnnp.attach_residual_add_status(conv, add8, res, 16) == Status.invalid_parameter      # channel mismatch
        assert qnnp.attach_residual_add_status(conv, add, res, 15) == Status.invalid_parameter       # stride < channels
        assert qnnp.attach_residual_add_status(conv, add, None, 16) == Status.invalid_parameter
        assert qnnp.attach_residual_add_status(conv, add, host, 16) == Status.unsupported_parameter  # host residual
        assert qnnp.attach_residual_add_status(conv, add, out, 16) == Status.invalid_parameter       # residual overlaps the output
        assert qnnp.operator_residual_folded(conv) == -1
        assert qnnp.attach_residual_add_status(conv, add, res, 16) == Status.success
        assert qnnp.operator_residual_folded(conv) == 0
        qnnp.run_operator(conv)
        # the next setup detaches
        qnnp.setup_convolution2d_nhwc_q8(conv, 1, 8, 8, buf, 32, out, 16)
        assert qnnp.operator_residual_folded(conv) == -1
        # host endpoints keep the two-operator form
        qnnp.setup_convolution2d_nhwc_q8(conv, 1, 8, 8, host, 32, out, 16)
        assert qnnp.attach_residual_add_status(conv, add, res, 16) == Status.unsupported_parameter
        # refused while a graph is being captured
        qnnp.setup_convolution2d_nhwc_q8(conv, 1, 8, 8, buf, 32, out, 16)
        qnnp.graph_begin()
        try:
            assert qnnp.attach_residual_add_status(conv, add, res, 16) == Status.invalid_parameter
        finally:
            qnnp.graph_destroy(qnnp.graph_end())
    finally:
        for h in (conv, add, add8):
            qnnp.delete_operator(h)


def test_fully_connected_setup_detaches_a_stale_residual(qnnp):
    """qnnp_gfx950_attach_residual_add accepts fully connected operators; like the convolution's, the next setup of the
    operator detaches the residual (it belongs to the previous binding: a new batch would read past its end)."""
    from qnnpack_amd import Status
    rng = np.random.default_rng(11)
    K, N = 64, 32
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    b = rng.integers(-1000, 1000, size=N).astype(np.int32)
    fc = qnnp.create_fully_connected_nc_q8(K, N, 127, 0.5, 127, 0.5, w, b, 127, 40.0, 0, 255, 0)
    add = qnnp.create_add_nc_q8(N, 3, 1.0, 5, 1.0, 6, 2.0, 0, 255, 0)
    try:
        a4 = to_device(rng.integers(0, 256, size=4 * K, dtype=np.uint8))
        out4 = to_device(np.zeros(4 * N, np.uint8))
        res4 = to_device(rng.integers(0, 256, size=4 * N, dtype=np.uint8))
        qnnp.setup_fully_connected_nc_q8(fc, 4, a4, K, out4, N)
        assert qnnp.attach_residual_add_status(fc, add, res4, N) == Status.success
        qnnp.run_operator(fc)
        with_add = from_device(out4).copy()
        # a new binding with MORE rows than the residual has: the stale residual must be gone
        a64 = to_device(np.tile(from_device(a4), 16))
        out64 = to_device(np.zeros(64 * N, np.uint8))
        qnnp.setup_fully_connected_nc_q8(fc, 64, a64, K, out64, N)
        assert qnnp.operator_residual_folded(fc) == -1
        qnnp.run_operator(fc)
        plain = from_device(out64).reshape(16, 4 * N)
        assert (plain == plain[0]).all()                       # 16 repeats of the 4 rows, no add anywhere
        assert not np.array_equal(plain[0], with_add)          # ... and different from the run that had the add
    finally:
        qnnp.delete_operator(fc)
        qnnp.delete_operator(add)
