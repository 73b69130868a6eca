"""ctypes front end of oracle_q8.c (the scalar restatement "O1"). TEST INFRASTRUCTURE ONLY.

Each function cites the reference lines its C implementation follows; see
oracle_q8.h for the full list.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, Structure, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_void_p

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_DIR, "liboracle_q8.so")
_lib = None


class Q31Params(Structure):
    """struct oracle_q31_params <- src/qnnpack/params.h:94-104 (scalar member)."""

    _fields_ = [("multiplier", c_int32), ("remainder_mask", c_int32), ("remainder_threshold", c_int32),
                ("shift", c_uint32), ("min_less_zero_point", c_int32), ("max_less_zero_point", c_int32),
                ("zero_point", c_int32)]


class ConvShape(Structure):
    _fields_ = [("batch", c_size_t), ("input_height", c_size_t), ("input_width", c_size_t),
                ("pad_top", c_uint32), ("pad_right", c_uint32), ("pad_bottom", c_uint32), ("pad_left", c_uint32),
                ("kernel_height", c_uint32), ("kernel_width", c_uint32),
                ("stride_height", c_uint32), ("stride_width", c_uint32),
                ("dilation_height", c_uint32), ("dilation_width", c_uint32),
                ("groups", c_uint32),
                ("group_input_channels", c_size_t), ("group_output_channels", c_size_t),
                ("input_pixel_stride", c_size_t)]


def build() -> str:
    subprocess.run(["make", "-C", _DIR, "oracle"], check=True, stdout=subprocess.DEVNULL)
    return _PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        L = ctypes.CDLL(_PATH)
        L.oracle_q31_params_init.restype = c_int
        L.oracle_q31_params_init.argtypes = [c_float, c_uint8, c_uint8, c_uint8, POINTER(Q31Params)]
        L.oracle_q31_requantize.restype = c_uint8
        L.oracle_q31_requantize.argtypes = [c_int32, POINTER(Q31Params)]
        L.oracle_q31_requantize_array.restype = c_int
        L.oracle_q31_requantize_array.argtypes = [c_size_t, c_void_p, c_float, c_uint8, c_uint8, c_uint8, c_void_p]
        L.oracle_gemm_acc.restype = None
        L.oracle_gemm_acc.argtypes = [c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p,
                                      c_uint8, c_uint8, c_void_p]
        L.oracle_conv_output_dim.restype = c_size_t
        L.oracle_conv_output_dim.argtypes = [c_size_t] * 4
        L.oracle_conv2d_acc.restype = None
        L.oracle_conv2d_acc.argtypes = [POINTER(ConvShape), c_void_p, c_void_p, c_void_p, c_uint8, c_uint8, c_void_p]
        L.oracle_deconv_output_dim.restype = c_size_t
        L.oracle_deconv_output_dim.argtypes = [c_size_t] * 6
        L.oracle_deconv2d_acc.restype = None
        L.oracle_deconv2d_acc.argtypes = [POINTER(ConvShape), c_uint32, c_uint32, c_void_p, c_void_p, c_void_p,
                                          c_uint8, c_uint8, c_void_p]
        L.oracle_requantize_rows.restype = c_int
        L.oracle_requantize_rows.argtypes = [c_size_t, c_size_t, c_void_p, c_float, c_uint8, c_uint8, c_uint8,
                                             c_void_p, c_size_t]
        L.oracle_fully_connected_q8.restype = c_int
        L.oracle_fully_connected_q8.argtypes = [c_size_t, c_size_t, c_size_t, c_uint8, c_float, c_uint8, c_float,
                                                c_void_p, c_void_p, c_uint8, c_float, c_uint8, c_uint8,
                                                c_void_p, c_size_t, c_void_p, c_size_t]
        L.oracle_convolution2d_q8.restype = c_int
        L.oracle_convolution2d_q8.argtypes = [POINTER(ConvShape), c_uint8, c_float, c_uint8, c_float,
                                              c_void_p, c_void_p, c_uint8, c_float, c_uint8, c_uint8,
                                              c_void_p, c_void_p, c_size_t]
        L.oracle_add_q8.restype = c_int
        L.oracle_add_q8.argtypes = [c_size_t, c_size_t, c_uint8, c_float, c_uint8, c_float, c_uint8, c_float,
                                    c_uint8, c_uint8, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]
        L.oracle_global_average_pooling_q8.restype = c_int
        L.oracle_global_average_pooling_q8.argtypes = [c_size_t, c_size_t, c_size_t, c_uint8, c_float, c_uint8,
                                                       c_float, c_uint8, c_uint8, c_void_p, c_size_t, c_void_p,
                                                       c_size_t]
        L.oracle_set_threads.restype = None
        L.oracle_set_threads.argtypes = [c_int]
        L.oracle_get_threads.restype = c_int
        _lib = L
    return _lib


def set_threads(n: int) -> None:
    lib().oracle_set_threads(n)


def q31_params(scale: float, zero_point: int, qmin: int, qmax: int) -> Q31Params:
    """src/qnnpack/requantization.h:22-54"""
    p = Q31Params()
    if lib().oracle_q31_params_init(np.float32(scale), zero_point, qmin, qmax, ctypes.byref(p)) != 0:
        raise ValueError(f"scale {scale} outside [2**-32, 1)")
    return p


def q31_requantize(acc: np.ndarray, scale: float, zero_point: int, qmin: int = 0, qmax: int = 255) -> np.ndarray:
    """src/requantization/q31-scalar.c:17-138 == src/qnnpack/requantization.h:464-480, element-wise."""
    acc = np.ascontiguousarray(acc, dtype=np.int32)
    out = np.empty(acc.shape, dtype=np.uint8)
    if lib().oracle_q31_requantize_array(acc.size, acc.ctypes.data, np.float32(scale), zero_point, qmin, qmax,
                                         out.ctypes.data) != 0:
        raise ValueError(f"scale {scale} outside [2**-32, 1)")
    return out


def gemm_acc(a: np.ndarray, w: np.ndarray, bias: np.ndarray, izp: int, kzp: int) -> np.ndarray:
    """test/gemm-microkernel-tester.h:213-226. a: [M, >=K] uint8 (row stride = a.strides[0]); w: [N, K]."""
    assert a.dtype == np.uint8 and w.dtype == np.uint8 and a.ndim == 2 and w.ndim == 2
    N, K = w.shape
    M = a.shape[0]
    assert a.strides[1] == 1 and a.shape[1] >= K
    w = np.ascontiguousarray(w)
    bias = np.ascontiguousarray(bias, dtype=np.int32)
    acc = np.empty((M, N), dtype=np.int32)
    lib().oracle_gemm_acc(M, N, K, a.ctypes.data, a.strides[0], w.ctypes.data, bias.ctypes.data, izp, kzp,
                          acc.ctypes.data)
    return acc


def conv_shape(batch, input_height, input_width, pads, kernel, stride, dilation, groups, gic, goc,
               input_pixel_stride=None) -> ConvShape:
    """pads = (top, right, bottom, left); kernel/stride/dilation = (height, width)."""
    return ConvShape(batch, input_height, input_width, pads[0], pads[1], pads[2], pads[3],
                     kernel[0], kernel[1], stride[0], stride[1], dilation[0], dilation[1],
                     groups, gic, goc, input_pixel_stride or groups * gic)


def conv_output_hw(s: ConvShape):
    """src/convolution.c:29-37, :415-424"""
    L = lib()
    oh = L.oracle_conv_output_dim(s.pad_top + s.input_height + s.pad_bottom, s.kernel_height,
                                  s.dilation_height, s.stride_height)
    ow = L.oracle_conv_output_dim(s.pad_left + s.input_width + s.pad_right, s.kernel_width,
                                  s.dilation_width, s.stride_width)
    return int(oh), int(ow)


def conv2d_acc(s: ConvShape, input: np.ndarray, kernel: np.ndarray, bias: np.ndarray, izp: int, kzp: int) -> np.ndarray:
    """test/convolution-operator-tester.h:367-403. Returns acc [N, OH, OW, G*GOC] int32."""
    oh, ow = conv_output_hw(s)
    input = np.ascontiguousarray(input, dtype=np.uint8)
    kernel = np.ascontiguousarray(kernel, dtype=np.uint8)
    bias = np.ascontiguousarray(bias, dtype=np.int32)
    acc = np.empty((s.batch, oh, ow, s.groups * s.group_output_channels), dtype=np.int32)
    lib().oracle_conv2d_acc(ctypes.byref(s), input.ctypes.data, kernel.ctypes.data, bias.ctypes.data, izp, kzp,
                            acc.ctypes.data)
    return acc


def deconv_output_hw(s: ConvShape, adjustment=(0, 0)):
    """src/deconvolution.c:25-37, :253-258"""
    L = lib()
    oh = L.oracle_deconv_output_dim(s.input_height, s.pad_top + s.pad_bottom, adjustment[0], s.kernel_height,
                                    s.dilation_height, s.stride_height)
    ow = L.oracle_deconv_output_dim(s.input_width, s.pad_left + s.pad_right, adjustment[1], s.kernel_width,
                                    s.dilation_width, s.stride_width)
    return int(oh), int(ow)


def deconv2d_acc(s: ConvShape, adjustment, input: np.ndarray, kernel: np.ndarray, bias: np.ndarray,
                 izp: int, kzp: int) -> np.ndarray:
    """test/deconvolution-operator-tester.h:383-419. kernel [g][ic][ky][kx][oc]; returns acc [N, OH, OW, G*GOC]."""
    oh, ow = deconv_output_hw(s, adjustment)
    input = np.ascontiguousarray(input, dtype=np.uint8)
    kernel = np.ascontiguousarray(kernel, dtype=np.uint8)
    bias = np.ascontiguousarray(bias, dtype=np.int32)
    acc = np.empty((s.batch, oh, ow, s.groups * s.group_output_channels), dtype=np.int32)
    lib().oracle_deconv2d_acc(ctypes.byref(s), adjustment[0], adjustment[1], input.ctypes.data, kernel.ctypes.data,
                              bias.ctypes.data, izp, kzp, acc.ctypes.data)
    return acc


def requantize_rows(acc: np.ndarray, scale: float, ozp: int, omin: int, omax: int,
                    out: np.ndarray = None, out_stride: int = None) -> np.ndarray:
    acc = np.ascontiguousarray(acc, dtype=np.int32)
    rows = int(np.prod(acc.shape[:-1]))
    cols = acc.shape[-1]
    if out is None:
        out = np.empty((rows, cols), dtype=np.uint8)
        out_stride = cols
    if lib().oracle_requantize_rows(rows, cols, acc.ctypes.data, np.float32(scale), ozp, omin, omax,
                                    out.ctypes.data, out_stride) != 0:
        raise ValueError(f"scale {scale} outside [2**-32, 1)")
    return out


def add_q8(batch, channels, a_zp, a_scale, b_zp, b_scale, y_zp, y_scale, y_min, y_max,
           a: np.ndarray, a_stride: int, b: np.ndarray, b_stride: int, y: np.ndarray, y_stride: int) -> np.ndarray:
    """src/qnnpack/requantization.h:327-360, :400-413, :500-522 -- writes into the flat strided buffer `y`."""
    if lib().oracle_add_q8(batch, channels, a_zp, np.float32(a_scale), b_zp, np.float32(b_scale), y_zp,
                           np.float32(y_scale), y_min, y_max, a.ctypes.data, a_stride, b.ctypes.data, b_stride,
                           y.ctypes.data, y_stride) != 0:
        raise ValueError("scale ratio outside [2**-14, 2**8)")
    return y


def global_average_pooling_q8(batch, width, channels, izp, iscale, ozp, oscale, omin, omax,
                              inp: np.ndarray, in_stride: int, out: np.ndarray, out_stride: int) -> np.ndarray:
    """src/global-average-pooling.c:138-145 + src/qnnpack/requantization.h:200-222, :482-498."""
    if lib().oracle_global_average_pooling_q8(batch, width, channels, izp, np.float32(iscale), ozp, np.float32(oscale),
                                              omin, omax, inp.ctypes.data, in_stride, out.ctypes.data, out_stride) != 0:
        raise ValueError("scale outside [2**-32, 256)")
    return out
