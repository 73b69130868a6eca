"""GPU tier: the "streaming_stores" option (include/qnnpack_gfx950.h). The kernels that write whole lines exactly once --
the staged and long-K pointwise kernels' copy-out (whole dense blocks and row chunks, with and without a channel
split), the lean GEMM's copy-out, the depthwise column walks (3x3 and 5x5), the element-wise add -- take a different store instruction with the option on (the
default) and off; the bytes must be the oracle's either way. The option is read at launch time."""
import numpy as np
import pytest

import _pointwise as pw
from _cases import ConvCase, FcCase, conv_tensors
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0], ids=["streaming", "plain"])
def stores(qnnp, request):
    qnnp.set_option("streaming_stores", request.param)
    yield qnnp
    qnnp.set_option("streaming_stores", 1)
    qnnp.set_option("gemm_kernel", 0)
    qnnp.set_option("dwconv_kernel", 0)


@pytest.mark.parametrize("case,variant,kernel", [
    (FcCase("ss_dense_whole", 4100, 32, 96), 5, "q8_pw_stream_mfma"),                 # whole dense 32-row blocks
    (FcCase("ss_dense_tail", 1000, 64, 144), 5, "q8_pw_stream_mfma"),                 # a partly filled last block
    (FcCase("ss_strided_chunks", 2050, 48, 80, output_stride=96), 5, "q8_pw_stream_mfma"),   # row chunks
    (FcCase("ss_split_columns", 300, 96, 576), 5, "q8_pw_stream_mfma"),               # channel columns: chunks of a row
    (FcCase("ss_longk", 900, 384, 96), 9, "q8_pw_stream_longk_mfma"),
    (FcCase("ss_lean_gemm", 520, 704, 512), 15, "q8_gemm_mfma_256x256_lean"),
], ids=lambda v: v.name if isinstance(v, FcCase) else None)
def test_outputs_do_not_depend_on_the_store_flavour(stores, case, variant, kernel):
    stores.set_option("gemm_kernel", variant)
    expected, quant = fc_expected(case)
    out, kname = fc_run(stores, case, quant, to_device=to_device, from_device=from_device)
    assert kname == kernel, kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def _dw(name, hw, c, k=3, **kw):
    pad = kw.pop("padding", (k // 2,) * 4)
    return ConvCase(name, hw, (k, k), pad, groups=c, gic=1, goc=1, **kw)


@pytest.mark.parametrize("case,prefix", [
    (_dw("ss_dw_c32_56", (56, 56), 32, batch=2), "q8_dwconv_col_3x3"),                      # several row segments
    (_dw("ss_dw_c144_s2", (29, 31), 144, subsampling=(2, 2)), "q8_dwconv_col_3x3"),
    (_dw("ss_dw_c260_ragged", (7, 7), 260, batch=3), "q8_dwconv_col_3x3"),                  # partly filled last wave
    (_dw("ss_dw_strided_pixels", (11, 12), 32, input_pixel_stride=40, output_pixel_stride=36), "q8_dwconv_col_3x3"),
    (_dw("ss_dw5_c72_s2", (28, 28), 72, k=5, subsampling=(2, 2), batch=2), "q8_dwconv_col_5x5"),
], ids=lambda v: v.name if isinstance(v, ConvCase) else None)
def test_depthwise_column_walks_do_not_depend_on_the_store_flavour(stores, case, prefix):
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(stores, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname.startswith(prefix), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


def test_add_does_not_depend_on_the_store_flavour(stores):
    flat = 0
    for case in pw.add_cases():
        a, b, _ = pw.add_tensors(case)
        got, kname = pw.add_run(stores, case, a, b, to_device=to_device, from_device=from_device)
        flat += kname == "q8_vadd_flat"
        assert np.array_equal(got, pw.add_expected(case, a, b)), case.name
    assert flat > 0


def test_option_values(qnnp):
    from qnnpack_amd.binding import QnnpackError
    with pytest.raises(QnnpackError):
        qnnp.set_option("streaming_stores", 2)
    qnnp.set_option("streaming_stores", 1)


def test_the_hint_is_scoped_per_operator(qnnp):
    """qnnp_gfx950_operator_set_streaming_stores: one operator with the hint off and one with it on in the same process,
    launched alternately, with the process-wide option left alone; both produce the oracle's bytes, -1 restores the default."""
    from qnnpack_amd import QnnpackError
    case = FcCase("ss_scoped", 4100, 32, 96)
    expected, quant = fc_expected(case)
    inp, kernel, bias = __import__("_cases").fc_tensors(case)
    oscale, ozp = quant
    ops, outs = [], []
    qnnp.set_option("gemm_kernel", 5)
    try:
        d_in = to_device(inp)
        for value in (0, 1):
            op = qnnp.create_fully_connected_nc_q8(case.input_channels, case.output_channels, case.izp, 1.0, case.kzp, 1.0,
                                                   kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
            d_out = to_device(np.zeros(expected.size, np.uint8))
            qnnp.setup_fully_connected_nc_q8(op, case.batch, d_in, case.in_stride, d_out, case.out_stride)
            qnnp.operator_set_streaming_stores(op, value)
            ops.append(op); outs.append(d_out)
        for _ in range(3):
            for op in ops:
                qnnp.run_operator(op)
        for d_out in outs:
            assert_bytes_equal(from_device(d_out), expected, "per-operator streaming hint")
        qnnp.operator_set_streaming_stores(ops[0], -1)
        qnnp.run_operator(ops[0])
        assert_bytes_equal(from_device(outs[0]), expected, "per-operator streaming hint, back to the default")
        with pytest.raises(QnnpackError):
            qnnp.operator_set_streaming_stores(ops[0], 2)
    finally:
        qnnp.set_option("gemm_kernel", 0)
        for op in ops:
            qnnp.delete_operator(op)
