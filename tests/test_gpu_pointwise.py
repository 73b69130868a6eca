"""GPU tier: qnnp add and global average pooling of the product against the scalar oracle, bit-exact, over the
reference's operator-test sweeps (test/add.cc, test/global-average-pooling.cc as restated in tests/_pointwise.py)
plus MobileNetV2-sized and vector-path cases; device and host (staged) tensors; kernel selection; error behaviour
(reference src/add.c:36-89, :128-136; src/global-average-pooling.c:34-70, :118-131)."""
import numpy as np
import pytest

import _pointwise as pw
from _gpu import from_device, to_device

pytestmark = pytest.mark.gpu


def test_add_matches_oracle_over_the_reference_sweep(qnnp):
    bad = []
    kernels = set()
    for case in pw.add_cases():
        a, b, _ = pw.add_tensors(case)
        got, kname = pw.add_run(qnnp, case, a, b, to_device=to_device, from_device=from_device)
        if case.batch:
            kernels.add(kname)
        if not np.array_equal(got, pw.add_expected(case, a, b)):
            bad.append(case.name)
    assert not bad, (len(bad), bad[:10])
    assert kernels == {"q8_vadd_flat", "q8_vadd_strided"}, kernels


@pytest.mark.parametrize("name", ["a_zero_batch", "a_small_c46", "a_strided_c91_qmin", "ax_dense_vec16"])
def test_add_host_tensors(qnnp, name):
    case = {c.name: c for c in pw.add_cases()}[name]
    a, b, _ = pw.add_tensors(case)
    got, _ = pw.add_run(qnnp, case, a, b)
    assert np.array_equal(got, pw.add_expected(case, a, b))


def test_global_average_pooling_matches_oracle_over_the_reference_sweep(qnnp):
    bad = []
    kernels = set()
    for case in pw.gap_cases(full=False):
        inp = pw.gap_tensors(case)
        got, kname = pw.gap_run(qnnp, case, inp, to_device=to_device, from_device=from_device)
        if case.batch:
            kernels.add(kname)
        if not np.array_equal(got, pw.gap_expected(case, inp)):
            bad.append(case.name)
    assert not bad, (len(bad), bad[:10])
    assert kernels == {"q8_gavgpool_x4", "q8_gavgpool_x1"}, kernels


@pytest.mark.parametrize("name", ["g_zero_batch", "gx_mobilenetv2_7x7x1280", "gx_c6_w300_unaligned", "g_few_c7_w16_b3_ostride"])
def test_global_average_pooling_host_tensors(qnnp, name):
    case = {c.name: c for c in pw.gap_cases()}[name]
    inp = pw.gap_tensors(case)
    got, _ = pw.gap_run(qnnp, case, inp)
    assert np.array_equal(got, pw.gap_expected(case, inp))


def test_resetup_global_average_pooling_with_another_width(qnnp):
    # the quantization parameters depend on the width and are rebuilt at setup (reference global-average-pooling.c:138-145)
    op = qnnp.create_global_average_pooling_nwc_q8(32, 121, 1.0, 133, 1.0, 0, 255, 0)
    try:
        for width in (49, 7, 196):
            case = pw.GapCase(f"resetup_w{width}", 2, width, 32)
            inp = pw.gap_tensors(case)
            d_in, d_out = to_device(inp), to_device(np.zeros(64, np.uint8))
            qnnp.setup_global_average_pooling_nwc_q8(op, 2, width, d_in, 32, d_out, 32)
            qnnp.run_operator(op)
            assert np.array_equal(from_device(d_out), pw.gap_expected(case, inp)), width
    finally:
        qnnp.delete_operator(op)


def test_error_statuses(qnnp):
    from qnnpack_amd import Status
    S = Status

    def add(channels=8, a_scale=0.75, b_scale=1.25, y_scale=1.0, qmin=0, qmax=255):
        st, h = qnnp.create_add_nc_q8_status(channels, 1, a_scale, 2, b_scale, 3, y_scale, qmin, qmax, 0)
        if h:
            qnnp.delete_operator(h)
        return st

    assert add() == S.success
    assert add(channels=0) == S.invalid_parameter
    assert add(a_scale=0.0) == S.invalid_parameter
    assert add(b_scale=float("nan")) == S.invalid_parameter
    assert add(y_scale=-1.0) == S.invalid_parameter
    assert add(qmin=10, qmax=10) == S.invalid_parameter          # min must be BELOW max (add.c:66-71)
    assert add(a_scale=1.0, y_scale=1.0e5) == S.unsupported_parameter    # ratio < 2^-14
    assert add(b_scale=300.0, y_scale=1.0) == S.unsupported_parameter    # ratio >= 2^8

    def gap(channels=8, in_scale=1.0, out_scale=1.0):
        st, h = qnnp.create_global_average_pooling_nwc_q8_status(channels, 1, in_scale, 2, out_scale, 0, 255, 0)
        if h:
            qnnp.delete_operator(h)
        return st

    assert gap() == S.success
    assert gap(channels=0) == S.invalid_parameter
    assert gap(in_scale=0.0) == S.invalid_parameter
    assert gap(out_scale=float("inf")) == S.invalid_parameter
    assert gap(in_scale=1.0, out_scale=300.0) == S.unsupported_parameter   # ratio < 2^-8
    assert gap(in_scale=256.0, out_scale=1.0) == S.unsupported_parameter   # ratio >= 2^8

    buf = to_device(np.zeros(1024, np.uint8))
    st, op = qnnp.create_global_average_pooling_nwc_q8_status(8, 1, 1.0, 2, 1.0, 0, 255, 0)
    st2, op2 = qnnp.create_add_nc_q8_status(8, 1, 1.0, 2, 1.0, 3, 1.0, 0, 255, 0)
    try:
        assert qnnp.setup_global_average_pooling_nwc_q8_status(op, 0, 0, None, 8, None, 8) == S.success    # batch 0 first
        assert qnnp.run_operator_status(op) == S.success
        assert qnnp.setup_global_average_pooling_nwc_q8_status(op, 1, 0, buf, 8, buf, 8) == S.invalid_parameter
        assert qnnp.setup_global_average_pooling_nwc_q8_status(op, 1, 4, buf, 7, buf, 8) == S.invalid_parameter
        assert qnnp.setup_add_nc_q8_status(op2, 0, None, 8, None, 8, None, 8) == S.success
        assert qnnp.run_operator_status(op2) == S.success
        assert qnnp.setup_add_nc_q8_status(op2, 2, buf, 8, buf, 7, buf, 8) == S.invalid_parameter
        # handles are typed
        assert qnnp.setup_add_nc_q8_status(op, 1, buf, 8, buf, 8, buf, 8) == S.invalid_parameter
        assert qnnp.setup_global_average_pooling_nwc_q8_status(op2, 1, 4, buf, 8, buf, 8) == S.invalid_parameter
    finally:
        qnnp.delete_operator(op)
        qnnp.delete_operator(op2)
