"""GPU tier: API behaviour of the drop-in boundary on a live device -- status codes
(reference src/convolution.c:69-168, 391-407; src/fully-connected.c:44-78; src/operator-run.c:642-644;
src/operator-delete.c:17-19), ownership (kernel/bias copied at create, input/output pointers only
borrowed at setup) and the gfx950 extensions."""
import numpy as np
import pytest

from _cases import ConvCase, FcCase, conv_tensors, fc_tensors
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal, conv_expected, fc_expected
from qnnpack_amd import Status

pytestmark = pytest.mark.gpu


def _conv_args(**kw):
    a = dict(pad=(0, 0, 0, 0), k=(1, 1), s=(1, 1), d=(1, 1), groups=1, gic=4, goc=4,
             iscale=1.0, kscale=1.0, oscale=2.0)
    a.update(kw)
    kernel = np.zeros((a["groups"], a["goc"], a["k"][0], a["k"][1], a["gic"]), np.uint8)
    bias = np.zeros(a["groups"] * a["goc"], np.int32)
    return (*a["pad"], a["k"][0], a["k"][1], a["s"][0], a["s"][1], a["d"][0], a["d"][1],
            a["groups"], a["gic"], a["goc"], 127, a["iscale"], 127, a["kscale"], kernel, bias,
            127, a["oscale"], 0, 255, 0)


def test_create_convolution_invalid_parameters(qnnp):
    for kw in [dict(k=(0, 1)), dict(k=(1, 0)), dict(s=(0, 1)), dict(s=(1, 0)), dict(d=(0, 1)), dict(d=(1, 0)),
               dict(iscale=0.0), dict(iscale=-1.0), dict(iscale=float("inf")), dict(iscale=float("nan")),
               dict(kscale=0.0), dict(oscale=0.0), dict(oscale=float("nan"))]:
        st, handle = qnnp.create_convolution2d_nhwc_q8_status(*_conv_args(**kw))
        assert st == Status.invalid_parameter and not handle, kw


def test_create_convolution_scale_ge_one_is_unsupported(qnnp):
    st, handle = qnnp.create_convolution2d_nhwc_q8_status(*_conv_args(oscale=1.0))
    assert st == Status.unsupported_parameter and not handle     # convolution.c:161-168
    st, handle = qnnp.create_convolution2d_nhwc_q8_status(*_conv_args(oscale=0.5))
    assert st == Status.unsupported_parameter and not handle


def test_create_fully_connected_status_codes(qnnp):
    k, b = np.zeros((4, 4), np.uint8), np.zeros(4, np.int32)
    st, h = qnnp.create_fully_connected_nc_q8_status(4, 4, 0, 0.0, 0, 1.0, k, b, 0, 2.0, 0, 255)
    assert st == Status.invalid_parameter and not h
    st, h = qnnp.create_fully_connected_nc_q8_status(4, 4, 0, 1.0, 0, 1.0, k, b, 0, 1.0, 0, 255)
    assert st == Status.unsupported_parameter and not h          # fully-connected.c:71-78


def test_setup_zero_input_dims_invalid(qnnp):
    op = qnnp.create_convolution2d_nhwc_q8(*_conv_args())
    try:
        buf = to_device(np.zeros(64, np.uint8))
        assert qnnp.setup_convolution2d_nhwc_q8_status(op, 1, 0, 3, buf, 4, buf, 4) == Status.invalid_parameter
        assert qnnp.setup_convolution2d_nhwc_q8_status(op, 1, 3, 0, buf, 4, buf, 4) == Status.invalid_parameter
    finally:
        qnnp.delete_operator(op)


def test_zero_batch_is_a_successful_noop(qnnp):
    # convolution.c:396-399 + operator-run.c:642-644
    op = qnnp.create_convolution2d_nhwc_q8(*_conv_args())
    try:
        out = to_device(np.full(64, FILL, np.uint8))
        assert qnnp.setup_convolution2d_nhwc_q8_status(op, 0, 5, 5, out, 4, out, 4) == Status.success
        assert qnnp.run_operator_status(op) == Status.success
        assert np.all(from_device(out) == FILL)
    finally:
        qnnp.delete_operator(op)
    k, b = np.zeros((2, 2), np.uint8), np.zeros(2, np.int32)
    op = qnnp.create_fully_connected_nc_q8(2, 2, 0, 1.0, 0, 1.0, k, b, 0, 2.0, 0, 255)
    try:
        assert qnnp.setup_fully_connected_nc_q8_status(op, 0, None, 2, None, 2) == Status.success
        assert qnnp.run_operator_status(op) == Status.success
    finally:
        qnnp.delete_operator(op)


def test_kernel_and_bias_are_copied_at_create(qnnp):
    case = FcCase("api_copy", 17, 40, 24)
    inp, kernel, bias = fc_tensors(case)
    expected, (oscale, ozp) = fc_expected(case, inp, kernel, bias)
    k2, b2 = kernel.copy(), bias.copy()
    op = qnnp.create_fully_connected_nc_q8(40, 24, case.izp, 1.0, case.kzp, 1.0, k2, b2, ozp, float(oscale), 0, 255)
    k2[:] = 0
    b2[:] = 0          # caller may free / reuse its buffers after create (SURVEY 8b ownership)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_fully_connected_nc_q8(op, case.batch, d_in, case.in_stride, d_out, case.out_stride)
        qnnp.run_operator(op)
        assert_bytes_equal(from_device(d_out), expected, "weights must have been copied at create")
    finally:
        qnnp.delete_operator(op)


def test_resetup_with_new_pointers_batch_and_geometry(qnnp):
    """setup may be repeated with other pointers / batch / spatial size (convolution.c:380-492);
    the device offset table is pointer- and batch-invariant and is rebuilt only when geometry changes."""
    base = ConvCase("api_resetup_a", (9, 8), (3, 3), (1, 1, 1, 1), gic=16, goc=24, batch=2)
    inp, kernel, bias = conv_tensors(base)
    exp_a, (oscale, ozp), (oh, ow) = conv_expected(base, inp, kernel, bias)
    op = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 1, 16, 24, base.izp, 1.0, base.kzp, 1.0,
                                           kernel, bias, ozp, float(oscale), 0, 255, 0)
    try:
        d_in = to_device(inp)
        d_out = to_device(np.full(exp_a.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, 2, 9, 8, d_in, 16, d_out, 24)
        qnnp.run_operator(op)
        assert_bytes_equal(from_device(d_out), exp_a, "first setup")
        # same geometry, different pointers and batch 1 (second image only)
        img_bytes = 9 * 8 * 16
        d_in2 = to_device(inp[img_bytes:])
        d_out2 = to_device(np.full(exp_a.size // 2, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, 1, 9, 8, d_in2, 16, d_out2, 24)
        qnnp.run_operator(op)
        assert_bytes_equal(from_device(d_out2), exp_a[exp_a.size // 2:], "re-setup, new pointers, batch 1")
        # different spatial size with the same operator
        other = ConvCase("api_resetup_b", (6, 11), (3, 3), (1, 1, 1, 1), gic=16, goc=24, batch=1)
        inp_b = np.random.default_rng(5).integers(0, 256, size=6 * 11 * 16, dtype=np.uint8)
        from oracle import o1
        shape = o1.conv_shape(1, 6, 11, other.padding, (3, 3), (1, 1), (1, 1), 1, 16, 24, 16)
        acc = o1.conv2d_acc(shape, inp_b, kernel, bias, base.izp, base.kzp)
        exp_b = o1.requantize_rows(acc.reshape(-1, 24), np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
        d_in3, d_out3 = to_device(inp_b), to_device(np.full(exp_b.size, FILL, np.uint8))
        qnnp.setup_convolution2d_nhwc_q8(op, 1, 6, 11, d_in3, 16, d_out3, 24)
        qnnp.run_operator(op)
        assert_bytes_equal(from_device(d_out3), exp_b, "re-setup, new geometry")
    finally:
        qnnp.delete_operator(op)


def test_async_mode_and_event_timer(qnnp):
    case = FcCase("api_async", 256, 128, 128)
    inp, kernel, bias = fc_tensors(case)
    expected, (oscale, ozp) = fc_expected(case, inp, kernel, bias)
    op = qnnp.create_fully_connected_nc_q8(128, 128, case.izp, 1.0, case.kzp, 1.0, kernel, bias,
                                           ozp, float(oscale), 0, 255)
    try:
        d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
        qnnp.setup_fully_connected_nc_q8(op, case.batch, d_in, 128, d_out, 128)
        qnnp.set_async(True)
        try:
            qnnp.run_operator(op)
            qnnp.synchronize()
        finally:
            qnnp.set_async(False)
        assert_bytes_equal(from_device(d_out), expected, "async run + explicit synchronize")
        ms = qnnp.time_operator(op, 2, 5)
        assert 0.0 < ms < 100.0
        assert_bytes_equal(from_device(d_out), expected, "timed launches are real launches")
    finally:
        qnnp.delete_operator(op)


def test_device_info_reports_gfx950(qnnp):
    info = qnnp.device_info()
    assert info["arch"].startswith("gfx950") and info["compute_units"] >= 200
    assert qnnp.get_device() == 0
