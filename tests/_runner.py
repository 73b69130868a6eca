"""Drive a QNNPACK-ABI library (product or compiled reference) through
create -> setup -> run -> delete for one case, and compute the oracle's answer.

Mirrors the flow of the reference operator testers
(test/convolution-operator-tester.h:415-449, test/fully-connected-operator-tester.h:158-185),
but asserts BIT-EXACT equality with the scalar oracle instead of the testers'
+-0.9 LSB tolerance (north_star requirement).
"""
from __future__ import annotations

import numpy as np

from _cases import (ConvCase, DeconvCase, FcCase, conv_tensors, deconv_tensors, fc_tensors, output_quantization,
                    strided_view)
from oracle import o1

FILL = 0xA5  # the testers pre-fill outputs with 0xA5 (convolution-operator-tester.h:361)


def conv_expected(case: ConvCase, inp=None, kernel=None, bias=None):
    """Oracle output buffer (flat, strided, gaps = FILL) + the quantization it used."""
    if inp is None:
        inp, kernel, bias = conv_tensors(case)
    H, W = case.input_size
    shape = o1.conv_shape(case.batch, H, W, case.padding, case.kernel_size, case.subsampling, case.dilation,
                          case.groups, case.gic, case.goc, case.in_stride)
    oh, ow = o1.conv_output_hw(shape)
    cout = case.groups * case.goc
    if case.batch == 0:
        return np.zeros(0, np.uint8), (np.float32(900 / 255.0), 0), (oh, ow)
    acc = o1.conv2d_acc(shape, inp, kernel, bias, case.izp, case.kzp)
    oscale, ozp = output_quantization(acc)
    rows = case.batch * oh * ow
    out = np.full((rows - 1) * case.out_stride + cout, FILL, dtype=np.uint8)
    # scales 1.0 / 1.0 / oscale as in the tester (:426-427) -> requantization scale 1/oscale
    req_scale = np.float32(np.float32(1.0) * np.float32(1.0) / oscale)
    o1.requantize_rows(acc.reshape(rows, cout), req_scale, ozp, case.qmin, case.qmax, out, case.out_stride)
    return out, (oscale, ozp), (oh, ow)


def conv_run(lib, case: ConvCase, quant, out_hw, inp=None, kernel=None, bias=None,
             to_device=None, from_device=None, threadpool=None):
    """Run the case through `lib`. With to_device/from_device the tensors live in device
    memory (zero-copy path); otherwise host numpy buffers are passed (staged path)."""
    if inp is None:
        inp, kernel, bias = conv_tensors(case)
    oscale, ozp = quant
    oh, ow = out_hw
    cout = case.groups * case.goc
    rows = case.batch * oh * ow
    out = np.full(max(rows - 1, 0) * case.out_stride + cout if rows else 0, FILL, dtype=np.uint8)
    op = lib.create_convolution2d_nhwc_q8(
        case.padding[0], case.padding[1], case.padding[2], case.padding[3],
        case.kernel_size[0], case.kernel_size[1], case.subsampling[0], case.subsampling[1],
        case.dilation[0], case.dilation[1], case.groups, case.gic, case.goc,
        case.izp, 1.0, case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
    try:
        if to_device is not None and rows:
            d_in, d_out = to_device(inp), to_device(out)
            lib.setup_convolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1],
                                            d_in, case.in_stride, d_out, case.out_stride)
            lib.run_operator(op, threadpool)
            out = from_device(d_out)
        else:
            host_in = inp if inp.size else np.zeros(1, np.uint8)
            host_out = out if out.size else np.zeros(1, np.uint8)
            lib.setup_convolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1],
                                            host_in, case.in_stride, host_out, case.out_stride)
            lib.run_operator(op, threadpool)
        kernel_name = lib.operator_kernel(op) if hasattr(lib, "operator_kernel") else None
    finally:
        lib.delete_operator(op)
    return out, kernel_name


def deconv_expected(case: DeconvCase, inp=None, kernel=None, bias=None):
    """Oracle output buffer for a deconvolution case (test/deconvolution-operator-tester.h:383-446 flow)."""
    if inp is None:
        inp, kernel, bias = deconv_tensors(case)
    H, W = case.input_size
    shape = o1.conv_shape(case.batch, H, W, case.padding, case.kernel_size, case.subsampling, case.dilation,
                          case.groups, case.gic, case.goc, case.in_stride)
    oh, ow = o1.deconv_output_hw(shape, case.adjustment)
    cout = case.groups * case.goc
    if case.batch == 0:
        return np.zeros(0, np.uint8), (np.float32(900 / 255.0), 0), (oh, ow)
    acc = o1.deconv2d_acc(shape, case.adjustment, inp, kernel, bias, case.izp, case.kzp)
    oscale, ozp = output_quantization(acc)
    rows = case.batch * oh * ow
    out = np.full((rows - 1) * case.out_stride + cout, FILL, dtype=np.uint8)
    req_scale = np.float32(np.float32(1.0) * np.float32(1.0) / oscale)
    o1.requantize_rows(acc.reshape(rows, cout), req_scale, ozp, case.qmin, case.qmax, out, case.out_stride)
    return out, (oscale, ozp), (oh, ow)


def deconv_run(lib, case: DeconvCase, quant, out_hw, inp=None, kernel=None, bias=None,
               to_device=None, from_device=None, threadpool=None):
    if inp is None:
        inp, kernel, bias = deconv_tensors(case)
    oscale, ozp = quant
    oh, ow = out_hw
    cout = case.groups * case.goc
    rows = case.batch * oh * ow
    out = np.full(max(rows - 1, 0) * case.out_stride + cout if rows else 0, FILL, dtype=np.uint8)
    op = lib.create_deconvolution2d_nhwc_q8(
        case.padding[0], case.padding[1], case.padding[2], case.padding[3],
        case.adjustment[0], case.adjustment[1],
        case.kernel_size[0], case.kernel_size[1], case.subsampling[0], case.subsampling[1],
        case.dilation[0], case.dilation[1], case.groups, case.gic, case.goc,
        case.izp, 1.0, case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
    try:
        if to_device is not None and rows:
            d_in, d_out = to_device(inp), to_device(out)
            lib.setup_deconvolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1],
                                              d_in, case.in_stride, d_out, case.out_stride)
            lib.run_operator(op, threadpool)
            out = from_device(d_out)
        else:
            host_in = inp if inp.size else np.zeros(1, np.uint8)
            host_out = out if out.size else np.zeros(1, np.uint8)
            lib.setup_deconvolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1],
                                              host_in, case.in_stride, host_out, case.out_stride)
            lib.run_operator(op, threadpool)
        kernel_name = lib.operator_kernel(op) if hasattr(lib, "operator_kernel") else None
    finally:
        lib.delete_operator(op)
    return out, kernel_name


def fc_expected(case: FcCase, inp=None, kernel=None, bias=None):
    if inp is None:
        inp, kernel, bias = fc_tensors(case)
    if case.batch == 0:
        return np.zeros(0, np.uint8), (np.float32(900 / 255.0), 0)
    a = strided_view(inp, case.batch, case.input_channels, case.in_stride)
    acc = o1.gemm_acc(a, kernel, bias, case.izp, case.kzp)
    oscale, ozp = output_quantization(acc)
    out = np.full((case.batch - 1) * case.out_stride + case.output_channels, FILL, dtype=np.uint8)
    req_scale = np.float32(np.float32(1.0) * np.float32(1.0) / oscale)
    o1.requantize_rows(acc, req_scale, ozp, case.qmin, case.qmax, out, case.out_stride)
    return out, (oscale, ozp)


def fc_run(lib, case: FcCase, quant, inp=None, kernel=None, bias=None, to_device=None, from_device=None):
    if inp is None:
        inp, kernel, bias = fc_tensors(case)
    oscale, ozp = quant
    out = np.full(max(case.batch - 1, 0) * case.out_stride + case.output_channels if case.batch else 0,
                  FILL, dtype=np.uint8)
    op = lib.create_fully_connected_nc_q8(
        case.input_channels, case.output_channels, case.izp, 1.0, case.kzp, 1.0, kernel, bias,
        ozp, float(oscale), case.qmin, case.qmax, 0)
    try:
        if to_device is not None and case.batch:
            d_in, d_out = to_device(inp), to_device(out)
            lib.setup_fully_connected_nc_q8(op, case.batch, d_in, case.in_stride, d_out, case.out_stride)
            lib.run_operator(op)
            out = from_device(d_out)
        else:
            host_in = inp if inp.size else np.zeros(1, np.uint8)
            host_out = out if out.size else np.zeros(1, np.uint8)
            lib.setup_fully_connected_nc_q8(op, case.batch, host_in, case.in_stride, host_out, case.out_stride)
            lib.run_operator(op)
        kernel_name = lib.operator_kernel(op) if hasattr(lib, "operator_kernel") else None
    finally:
        lib.delete_operator(op)
    return out, kernel_name


def assert_bytes_equal(actual: np.ndarray, expected: np.ndarray, what: str):
    assert actual.shape == expected.shape, f"{what}: shape {actual.shape} vs {expected.shape}"
    if not np.array_equal(actual, expected):
        bad = np.nonzero(actual != expected)[0]
        i = int(bad[0])
        raise AssertionError(
            f"{what}: {bad.size} of {actual.size} bytes differ; first at flat index {i}: "
            f"got {int(actual[i])}, oracle {int(expected[i])}")
