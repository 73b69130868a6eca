"""GPU tier: qnnp_create/setup_deconvolution2d_nhwc_q8 of the product against the scalar oracle, bit-exact, on the
reference's own case list (test/deconvolution.cc, 23 cases) and the extras (adjustment, 2x upsampling shapes,
aligned channels, zero-point and clamp corners); device-resident and host (staged) tensors; re-setup; error
behaviour (reference src/deconvolution.c:69-129, :225-243)."""
import numpy as np
import pytest

from _cases import (DECONV_CASES, EXTRA_DECONV_CASES, STREAM_DECONV_CASES, STREAM_DECONV_FALLBACK_CASES,
                    deconv_tensors)
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, deconv_expected, deconv_run

pytestmark = pytest.mark.gpu


def _expected_kernel(case):
    if case.name.startswith("dx_d2s_") or case.name == "dx_2x2s2_c64_n32":
        return "q8_pw_stream_d2s_mfma"
    if case.name.startswith("ds_") or case.name == "dx_4x4s2p1_c32_n16":
        return "q8_deconv_s2_stream"
    return "q8_igemm_mfma"


@pytest.mark.parametrize("case", DECONV_CASES + EXTRA_DECONV_CASES + STREAM_DECONV_CASES + STREAM_DECONV_FALLBACK_CASES,
                         ids=lambda c: c.name)
def test_deconvolution_matches_oracle_device_tensors(qnnp, case):
    expected, quant, out_hw = deconv_expected(case)
    out, kname = deconv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    if case.batch:
        assert kname is not None and kname.startswith(_expected_kernel(case)), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{case.name}]")


@pytest.mark.parametrize("case", STREAM_DECONV_CASES, ids=lambda c: c.name)
def test_streaming_and_phase_table_kernels_agree(qnnp, case):
    """"gemm_kernel" = 1 keeps the phase-table GEMMs, 13 forces the streaming kernel: same bytes, both against the oracle."""
    expected, quant, out_hw = deconv_expected(case)
    for variant, want in ((1, "q8_igemm_mfma"), (13, "q8_deconv_s2_stream")):
        qnnp.set_option("gemm_kernel", variant)
        try:
            out, kname = deconv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
        finally:
            qnnp.set_option("gemm_kernel", 0)
        assert kname.startswith(want), kname
        assert_bytes_equal(out, expected, f"gemm_kernel={variant} {kname} [{case.name}]")


def test_forced_streaming_kernel_outside_its_range_is_refused(qnnp):
    from qnnpack_amd import QnnpackError, Status
    case = next(c for c in STREAM_DECONV_FALLBACK_CASES if c.name == "dsf_3x3s2_c160")
    expected, quant, out_hw = deconv_expected(case)
    qnnp.set_option("gemm_kernel", 13)
    try:
        with pytest.raises(QnnpackError) as err:
            deconv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
        assert err.value.status == Status.unsupported_parameter
    finally:
        qnnp.set_option("gemm_kernel", 0)


@pytest.mark.parametrize("case", [c for c in DECONV_CASES if c.name in
                                  ("d_zero_batch", "d_3x3s2", "d_grouped_3x3", "d_3x3_with_output_stride")],
                         ids=lambda c: c.name)
def test_deconvolution_matches_oracle_host_tensors(qnnp, case):
    expected, quant, out_hw = deconv_expected(case)
    out, _ = deconv_run(qnnp, case, quant, out_hw)
    assert_bytes_equal(out, expected, f"gfx950 (staged host tensors) vs oracle [{case.name}]")


def test_gemm_kernel_option_does_not_reroute_a_deconvolution(qnnp):
    # the geometry-derived convolution kernels would compute a CONVOLUTION; the operator pins the table kernel
    case = next(c for c in EXTRA_DECONV_CASES if c.name == "dx_3x3_c128_n64")
    expected, quant, out_hw = deconv_expected(case)
    for variant in (0, 3, 5):
        qnnp.set_option("gemm_kernel", variant)
        try:
            out, kname = deconv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
        finally:
            qnnp.set_option("gemm_kernel", 0)
        assert kname.startswith("q8_igemm_mfma"), kname
        assert_bytes_equal(out, expected, f"gemm_kernel={variant} [{case.name}]")


def test_resetup_with_new_geometry_and_pointers(qnnp):
    case_a = next(c for c in DECONV_CASES if c.name == "d_3x3s2")
    inp_a, kernel, bias = deconv_tensors(case_a)
    exp_a, quant, hw_a = deconv_expected(case_a, inp_a, kernel, bias)
    oscale, ozp = quant
    op = qnnp.create_deconvolution2d_nhwc_q8(
        *case_a.padding, 0, 0, 3, 3, 2, 2, 1, 1, 1, case_a.gic, case_a.goc,
        case_a.izp, 1.0, case_a.kzp, 1.0, kernel, bias, ozp, float(oscale), 0, 255, 0)
    try:
        d_in = to_device(inp_a)
        d_out = to_device(np.full(exp_a.size, 0xA5, np.uint8))
        qnnp.setup_deconvolution2d_nhwc_q8(op, 1, 19, 21, d_in, case_a.in_stride, d_out, case_a.out_stride)
        qnnp.run_operator(op)
        assert_bytes_equal(from_device(d_out), exp_a, "first setup")
        # smaller image, same operator: the table is rebuilt
        from dataclasses import replace
        case_b = replace(case_a, input_size=(7, 5))
        rng = np.random.default_rng(7)
        inp_b = rng.integers(0, 256, size=7 * 5 * case_b.in_stride, dtype=np.uint8)
        from oracle import o1
        shape = o1.conv_shape(1, 7, 5, case_b.padding, (3, 3), (2, 2), (1, 1), 1, case_b.gic, case_b.goc, case_b.in_stride)
        acc = o1.deconv2d_acc(shape, (0, 0), inp_b, kernel, bias, case_b.izp, case_b.kzp)
        exp_b = o1.requantize_rows(acc.reshape(-1, case_b.goc), np.float32(1.0) / oscale, ozp, 0, 255)
        d_in_b = to_device(inp_b)
        d_out_b = to_device(np.zeros(exp_b.size, np.uint8))
        qnnp.setup_deconvolution2d_nhwc_q8(op, 1, 7, 5, d_in_b, case_b.in_stride, d_out_b, case_b.goc)
        qnnp.run_operator(op)
        assert_bytes_equal(from_device(d_out_b), exp_b.reshape(-1), "after re-setup")
    finally:
        qnnp.delete_operator(op)


def test_error_statuses(qnnp):
    from qnnpack_amd import Status
    k = np.zeros((1, 4, 3, 3, 4), np.uint8)
    b = np.zeros(4, np.int32)
    ok = dict(pad=(1, 1, 1, 1), adj=(0, 0), k=(3, 3), s=(1, 1), d=(1, 1))

    def create(pad=ok["pad"], adj=ok["adj"], kk=ok["k"], s=ok["s"], d=ok["d"], in_scale=1.0, k_scale=1.0,
               out_scale=2.0, kernel=k, bias=b):
        st, h = qnnp.create_deconvolution2d_nhwc_q8_status(
            *pad, *adj, *kk, *s, *d, 1, 4, 4, 127, in_scale, 127, k_scale, kernel, bias, 127, out_scale, 0, 255, 0)
        if h:
            qnnp.delete_operator(h)
        return st

    assert create() == Status.success
    assert create(kk=(0, 3)) == Status.invalid_parameter
    assert create(s=(1, 0)) == Status.invalid_parameter
    assert create(d=(0, 1)) == Status.invalid_parameter
    assert create(in_scale=0.0) == Status.invalid_parameter
    assert create(k_scale=float("inf")) == Status.invalid_parameter
    assert create(out_scale=-1.0) == Status.invalid_parameter
    assert create(out_scale=0.5) == Status.unsupported_parameter      # scale 2.0 >= 1
    st, op = qnnp.create_deconvolution2d_nhwc_q8_status(
        1, 1, 1, 1, 0, 0, 3, 3, 1, 1, 1, 1, 1, 4, 4, 127, 1.0, 127, 1.0, k, b, 127, 2.0, 0, 255, 0)
    assert st == Status.success
    try:
        buf = to_device(np.zeros(4096, np.uint8))
        assert qnnp.setup_deconvolution2d_nhwc_q8_status(op, 0, 0, 0, None, 4, None, 4) == Status.success   # batch 0 first
        assert qnnp.run_operator_status(op) == Status.success                                               # ... is a no-op
        assert qnnp.setup_deconvolution2d_nhwc_q8_status(op, 1, 0, 5, buf, 4, buf, 4) == Status.invalid_parameter
        assert qnnp.setup_deconvolution2d_nhwc_q8_status(op, 1, 5, 5, buf, 3, buf, 4) == Status.invalid_parameter
        # a deconvolution handle is not a convolution handle
        assert qnnp.setup_convolution2d_nhwc_q8_status(op, 1, 5, 5, buf, 4, buf, 4) == Status.invalid_parameter
    finally:
        qnnp.delete_operator(op)
