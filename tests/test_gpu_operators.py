"""GPU tier: the HIP path, called through the C ABI (include/qnnpack.h), must equal the scalar oracle
BYTE FOR BYTE on the reference's own operator test lists (test/convolution.cc, test/fully-connected.cc;
the reference itself only asks for +-0.9 LSB there, convolution-operator-tester.h:461-464) and on the
committed outputs of the compiled reference (tests/golden/)."""
import numpy as np
import pytest

import _golden
from _cases import CONV_CASES, EXTRA_CONV_CASES, EXTRA_FC_CASES, FC_CASES, conv_tensors, fc_tensors
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu

CONV_BY_NAME = {c.name: c for c in CONV_CASES + EXTRA_CONV_CASES}
FC_BY_NAME = {c.name: c for c in FC_CASES + EXTRA_FC_CASES}


def _is_depthwise(case):
    return case.gic == 1 and case.goc == 1 and case.groups > 1


@pytest.mark.parametrize("case", CONV_CASES + EXTRA_CONV_CASES, ids=lambda c: c.name)
def test_convolution_matches_oracle(qnnp, case):
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert_bytes_equal(out, expected, f"gfx950 vs oracle [{case.name}] kernel={kname}")
    if case.batch:
        assert kname is not None
        assert kname.startswith("q8_dwconv") == _is_depthwise(case), kname


@pytest.mark.parametrize("case", FC_CASES + EXTRA_FC_CASES, ids=lambda c: c.name)
def test_fully_connected_matches_oracle(qnnp, case):
    inp, kernel, bias = fc_tensors(case)
    expected, quant = fc_expected(case, inp, kernel, bias)
    out, kname = fc_run(qnnp, case, quant, inp, kernel, bias, to_device, from_device)
    assert_bytes_equal(out, expected, f"gfx950 vs oracle [{case.name}] kernel={kname}")
    if case.batch:
        assert kname.startswith(("q8_igemm", "q8_pw_stream", "q8_gemm_mfma")), kname   # a GEMM-family kernel


@pytest.mark.parametrize("name", _golden.names("conv"))
def test_convolution_matches_reference_golden(qnnp, name):
    case = CONV_BY_NAME[name]
    inp, kernel, bias, quant, ref_out = _golden.entry("conv", name)
    H, W = case.input_size
    # output dims as the reference computes them (src/convolution.c:29-37)
    eh = (case.kernel_size[0] - 1) * case.dilation[0] + 1
    ew = (case.kernel_size[1] - 1) * case.dilation[1] + 1
    oh = (H + case.padding[0] + case.padding[2] - eh) // case.subsampling[0] + 1
    ow = (W + case.padding[1] + case.padding[3] - ew) // case.subsampling[1] + 1
    out, _ = conv_run(qnnp, case, quant, (oh, ow), inp, kernel, bias, to_device, from_device)
    assert_bytes_equal(out, ref_out, f"gfx950 vs compiled-reference golden [{name}]")


@pytest.mark.parametrize("name", _golden.names("fc"))
def test_fully_connected_matches_reference_golden(qnnp, name):
    case = FC_BY_NAME[name]
    inp, kernel, bias, quant, ref_out = _golden.entry("fc", name)
    out, _ = fc_run(qnnp, case, quant, inp, kernel, bias, to_device, from_device)
    assert_bytes_equal(out, ref_out, f"gfx950 vs compiled-reference golden [{name}]")


HOST_POINTER_CASES = ["1x1", "1x1_with_output_stride", "3x3_with_input_stride", "3x3_with_batch",
                      "depthwise_3x3", "x_dw3x3_c32_strided", "x_3x3_c64_vec16"]


@pytest.mark.parametrize("name", HOST_POINTER_CASES)
def test_convolution_with_host_pointers_is_staged(qnnp, name):
    """Reference-style callers pass host memory (convolution-operator-tester.h:430-440): staged path.
    Bytes between output pixels (output_pixel_stride > channels) must survive untouched."""
    case = CONV_BY_NAME[name]
    expected, quant, out_hw = conv_expected(case)
    out, _ = conv_run(qnnp, case, quant, out_hw)
    assert_bytes_equal(out, expected, f"gfx950 staged host pointers vs oracle [{name}]")


def test_fully_connected_with_host_pointers_is_staged(qnnp):
    case = FC_BY_NAME["small_batch_with_output_stride"]
    expected, quant = fc_expected(case)
    out, _ = fc_run(qnnp, case, quant)
    assert_bytes_equal(out, expected, "gfx950 staged host pointers vs oracle [fc]")


@pytest.mark.parametrize("variant,prefix", [(1, "q8_dwconv_direct"), (2, "q8_dwconv_lds")])
@pytest.mark.parametrize("name", ["x_dw3x3_c32", "x_dw3x3_c96_s2", "x_dw3x3_c144", "x_dw3x3_c20_vec4",
                                  "x_dw3x3_c960_7x7", "x_dw3x3_c32_strided", "x_dw5x5_c64", "x_dw3x3_c64_d2",
                                  "x_dw3x3_c32_qmin_qmax", "x_dw3x3_c64_zp"])
def test_depthwise_kernel_variants_agree_with_oracle(qnnp, name, variant, prefix):
    case = CONV_BY_NAME[name]
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    qnnp.set_option("dwconv_kernel", variant)
    try:
        out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    finally:
        qnnp.set_option("dwconv_kernel", 0)
    assert kname.startswith(prefix), kname
    assert_bytes_equal(out, expected, f"gfx950 {kname} vs oracle [{name}]")


@pytest.mark.parametrize("misalign", [1, 2, 4, 8])
def test_misaligned_device_pointers(qnnp, misalign):
    """Activation vector width / dword stores are chosen from the ACTUAL pointer alignment."""
    for name in ["x_1x1_k64_n64_vec16", "x_3x3_c64_vec16", "x_dw3x3_c32"]:
        case = CONV_BY_NAME[name]
        inp, kernel, bias = conv_tensors(case)
        expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
        out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias,
                              lambda a: to_device(a, misalign), from_device)
        assert_bytes_equal(out, expected, f"gfx950 misaligned by {misalign} vs oracle [{name}] kernel={kname}")
