"""GPU tier: hipGraph capture of operator launches (qnnp_gfx950_graph_*, include/qnnpack_gfx950.h). A graph
of several operators must produce exactly the bytes the same operators produce when run one by one (and the
oracle's), replays must be repeatable, and what cannot be captured must be refused, not silently run."""
import numpy as np
import pytest

from _cases import ConvCase, FcCase, conv_tensors, fc_tensors
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal, conv_expected, fc_expected

pytestmark = pytest.mark.gpu


def _make_conv(qnnp, case):
    inp, kernel, bias = conv_tensors(case)
    expected, (oscale, ozp), (oh, ow) = conv_expected(case, inp, kernel, bias)
    op = qnnp.create_convolution2d_nhwc_q8(
        case.padding[0], case.padding[1], case.padding[2], case.padding[3], case.kernel_size[0], case.kernel_size[1],
        case.subsampling[0], case.subsampling[1], case.dilation[0], case.dilation[1], case.groups, case.gic, case.goc,
        case.izp, 1.0, case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
    d_in = to_device(inp)
    d_out = to_device(np.full(expected.size, FILL, np.uint8))
    qnnp.setup_convolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1], d_in, case.in_stride,
                                     d_out, case.out_stride)
    return op, d_in, d_out, expected


def test_graph_of_three_operators_matches_oracle_and_replays(qnnp):
    import torch
    cases = [ConvCase("g_pw", (14, 14), gic=32, goc=64, batch=2),
             ConvCase("g_dw", (14, 14), (3, 3), (1, 1, 1, 1), groups=64, batch=2),
             ConvCase("g_3x3", (14, 14), (3, 3), (1, 1, 1, 1), gic=16, goc=32, batch=2)]
    built = [_make_conv(qnnp, c) for c in cases]
    try:
        qnnp.graph_begin()
        for op, *_ in built:
            qnnp.run_operator(op)
        graph = qnnp.graph_end()
        # nothing ran during the capture
        for _, _, d_out, expected in built:
            assert np.all(from_device(d_out) == FILL)
        for rep in range(3):
            for _, _, d_out, _ in built:
                d_out.fill_(FILL)
            torch.cuda.synchronize()
            qnnp.graph_launch(graph)               # synchronous (async is off): outputs complete on return
            for case, (_, _, d_out, expected) in zip(cases, built):
                assert_bytes_equal(from_device(d_out), expected, f"graph replay {rep} vs oracle [{case.name}]")
        ms = qnnp.graph_time(graph, 1, 5)
        assert ms > 0.0
        qnnp.graph_destroy(graph)
    finally:
        for op, *_ in built:
            qnnp.delete_operator(op)


def test_host_pointer_operator_cannot_be_captured(qnnp):
    from qnnpack_amd import QnnpackError
    case = FcCase("g_fc_host", 8, 32, 16)
    inp, kernel, bias = fc_tensors(case)
    expected, (oscale, ozp) = fc_expected(case, inp, kernel, bias)
    op = qnnp.create_fully_connected_nc_q8(case.input_channels, case.output_channels, case.izp, 1.0, case.kzp, 1.0,
                                           kernel, bias, ozp, float(oscale), case.qmin, case.qmax)
    out = np.full(expected.size, FILL, np.uint8)
    try:
        qnnp.setup_fully_connected_nc_q8(op, case.batch, inp, case.in_stride, out, case.out_stride)
        qnnp.graph_begin()
        try:
            with pytest.raises(QnnpackError):
                qnnp.run_operator(op)
        finally:
            graph = qnnp.graph_end()
            qnnp.graph_destroy(graph)
        qnnp.run_operator(op)                      # outside a capture the staged path still works
        assert_bytes_equal(out, expected, "staged run after a refused capture")
    finally:
        qnnp.delete_operator(op)


def test_timing_with_and_without_graph_agree_roughly(qnnp):
    case = ConvCase("g_time", (56, 56), gic=64, goc=64, batch=16)
    op, d_in, d_out, expected = _make_conv(qnnp, case)
    try:
        qnnp.set_option("timing_graph", 1)
        with_graph = qnnp.time_operator(op, 2, 20)
        qnnp.set_option("timing_graph", 0)
        without = qnnp.time_operator(op, 2, 20)
        # (a sanity bound, not a benchmark: a replay should not cost more than the launch loop; 3x leaves room for a
        #  preempted 20-launch sample on a shared box)
        assert 0.0 < with_graph <= without * 3.0, (with_graph, without)
        assert_bytes_equal(from_device(d_out), expected, "outputs after timed runs")
    finally:
        qnnp.set_option("timing_graph", 1)
        qnnp.delete_operator(op)


def test_create_setup_and_copies_are_refused_during_capture(qnnp):
    """An upload recorded into a graph would read host memory that create / setup free right afterwards, and the
    synchronizing copies would invalidate the capture: everything but operator launches answers
    invalid_parameter between graph_begin and graph_end, the capture stays valid, and the operator keeps working."""
    from qnnpack_amd import QnnpackError
    case = ConvCase("g_refuse", (12, 12), (3, 3), (1, 1, 1, 1), gic=16, goc=32, batch=2)
    op, d_in, d_out, expected = _make_conv(qnnp, case)
    inp, kernel, bias = conv_tensors(case)
    scratch = qnnp.malloc(256)
    host = np.zeros(256, np.uint8)
    try:
        qnnp.graph_begin()
        try:
            with pytest.raises(QnnpackError):
                qnnp.setup_convolution2d_nhwc_q8(op, case.batch, 12, 12, d_in, case.in_stride, d_out, case.out_stride)
            with pytest.raises(QnnpackError):
                qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 1, 16, 32, 127, 1.0, 127, 1.0,
                                                  kernel, bias, 127, 1000.0, 0, 255, 0)
            with pytest.raises(QnnpackError):
                qnnp.memcpy_h2d(scratch, host)
            with pytest.raises(QnnpackError):
                qnnp.memcpy_d2h(host, scratch)
            with pytest.raises(QnnpackError):
                qnnp.memset(scratch, 0, 256)
            # the refused setup left the operator unrunnable by design? no: it was refused before it touched anything
            qnnp.run_operator(op)
        finally:
            graph = qnnp.graph_end()
        qnnp.graph_launch(graph)
        assert_bytes_equal(from_device(d_out), expected, "graph recorded around refused calls vs oracle")
        qnnp.graph_destroy(graph)
    finally:
        qnnp.free(scratch)
        qnnp.delete_operator(op)
