"""ctypes binding of the QNNPACK q8 conv/GEMM C API (include/qnnpack.h).

One class binds any shared library that exports the reference's C ABI for the
hot path: the product (``libqnnpack_gfx950.so``) and, in the test
infrastructure, the compiled reference (``oracle/_ref/libqnnpack_ref.so``).
That both load through the same signatures IS the drop-in claim.

Reference interface mirrored: include/qnnpack.h:24-76, 118-140, 327-332.
Pointers are passed as raw addresses (``int``), numpy arrays (host memory) or
anything with a ``data_ptr()`` method (e.g. a torch tensor on the bound GPU).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_size_t, c_uint8, c_uint32, c_void_p
from enum import IntEnum
from typing import Optional

import numpy as np


class Status(IntEnum):
    """enum qnnp_status, include/qnnpack.h:24-32"""

    success = 0
    uninitialized = 1
    invalid_parameter = 2
    unsupported_parameter = 3
    unsupported_hardware = 4
    out_of_memory = 5


class QnnpackError(RuntimeError):
    def __init__(self, call: str, status: int):
        self.status = Status(status)
        super().__init__(f"{call} -> qnnp_status_{self.status.name}")


def address_of(buf) -> Optional[int]:
    """Raw address of a buffer argument (None -> NULL)."""
    if buf is None:
        return None
    if isinstance(buf, int):
        return buf
    if isinstance(buf, np.ndarray):
        return buf.ctypes.data
    if hasattr(buf, "data_ptr"):
        return int(buf.data_ptr())
    raise TypeError(f"cannot take the address of {type(buf)!r}")


class QnnpackLibrary:
    """The C API of include/qnnpack.h bound over ``path``."""

    def __init__(self, path: str):
        self.path = path
        self.lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        L = self.lib
        L.qnnp_initialize.restype = c_int
        L.qnnp_initialize.argtypes = []
        L.qnnp_deinitialize.restype = c_int
        L.qnnp_deinitialize.argtypes = []
        L.qnnp_create_convolution2d_nhwc_q8.restype = c_int
        L.qnnp_create_convolution2d_nhwc_q8.argtypes = (
            [c_uint32] * 11 + [c_size_t, c_size_t, c_uint8, c_float, c_uint8, c_float, c_void_p, c_void_p,
                               c_uint8, c_float, c_uint8, c_uint8, c_uint32, POINTER(c_void_p)])
        L.qnnp_setup_convolution2d_nhwc_q8.restype = c_int
        L.qnnp_setup_convolution2d_nhwc_q8.argtypes = [
            c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]
        L.qnnp_create_deconvolution2d_nhwc_q8.restype = c_int
        L.qnnp_create_deconvolution2d_nhwc_q8.argtypes = (
            [c_uint32] * 13 + [c_size_t, c_size_t, c_uint8, c_float, c_uint8, c_float, c_void_p, c_void_p,
                               c_uint8, c_float, c_uint8, c_uint8, c_uint32, POINTER(c_void_p)])
        L.qnnp_setup_deconvolution2d_nhwc_q8.restype = c_int
        L.qnnp_setup_deconvolution2d_nhwc_q8.argtypes = [
            c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]
        L.qnnp_create_global_average_pooling_nwc_q8.restype = c_int
        L.qnnp_create_global_average_pooling_nwc_q8.argtypes = [
            c_size_t, c_uint8, c_float, c_uint8, c_float, c_uint8, c_uint8, c_uint32, POINTER(c_void_p)]
        L.qnnp_setup_global_average_pooling_nwc_q8.restype = c_int
        L.qnnp_setup_global_average_pooling_nwc_q8.argtypes = [
            c_void_p, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]
        L.qnnp_create_add_nc_q8.restype = c_int
        L.qnnp_create_add_nc_q8.argtypes = [
            c_size_t, c_uint8, c_float, c_uint8, c_float, c_uint8, c_float, c_uint8, c_uint8, c_uint32, POINTER(c_void_p)]
        L.qnnp_setup_add_nc_q8.restype = c_int
        L.qnnp_setup_add_nc_q8.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]
        L.qnnp_create_fully_connected_nc_q8.restype = c_int
        L.qnnp_create_fully_connected_nc_q8.argtypes = [
            c_size_t, c_size_t, c_uint8, c_float, c_uint8, c_float, c_void_p, c_void_p,
            c_uint8, c_float, c_uint8, c_uint8, c_uint32, POINTER(c_void_p)]
        L.qnnp_setup_fully_connected_nc_q8.restype = c_int
        L.qnnp_setup_fully_connected_nc_q8.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]
        L.qnnp_run_operator.restype = c_int
        L.qnnp_run_operator.argtypes = [c_void_p, c_void_p]
        L.qnnp_delete_operator.restype = c_int
        L.qnnp_delete_operator.argtypes = [c_void_p]

    # -- status-returning raw calls (tests of error behaviour use these) ----
    def initialize_status(self) -> Status:
        return Status(self.lib.qnnp_initialize())

    def initialize(self) -> None:
        st = self.lib.qnnp_initialize()
        if st != 0:
            raise QnnpackError("qnnp_initialize", st)

    def deinitialize(self) -> Status:
        return Status(self.lib.qnnp_deinitialize())

    def create_convolution2d_nhwc_q8_status(
            self, pad_top, pad_right, pad_bottom, pad_left, kernel_height, kernel_width,
            subsampling_height, subsampling_width, dilation_height, dilation_width,
            groups, group_input_channels, group_output_channels,
            input_zero_point, input_scale, kernel_zero_point, kernel_scale,
            kernel, bias, output_zero_point, output_scale, output_min, output_max, flags=0):
        kernel = None if kernel is None else np.ascontiguousarray(kernel, dtype=np.uint8)
        bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.int32)
        handle = c_void_p(None)
        st = self.lib.qnnp_create_convolution2d_nhwc_q8(
            pad_top, pad_right, pad_bottom, pad_left, kernel_height, kernel_width,
            subsampling_height, subsampling_width, dilation_height, dilation_width,
            groups, group_input_channels, group_output_channels,
            input_zero_point, input_scale, kernel_zero_point, kernel_scale,
            address_of(kernel), address_of(bias),
            output_zero_point, output_scale, output_min, output_max, flags, ctypes.byref(handle))
        return Status(st), handle.value

    def create_convolution2d_nhwc_q8(self, *args, **kwargs) -> int:
        st, handle = self.create_convolution2d_nhwc_q8_status(*args, **kwargs)
        if st != Status.success:
            raise QnnpackError("qnnp_create_convolution2d_nhwc_q8", st)
        return handle

    def setup_convolution2d_nhwc_q8_status(
            self, op, batch_size, input_height, input_width, input, input_stride, output, output_stride) -> Status:
        return Status(self.lib.qnnp_setup_convolution2d_nhwc_q8(
            op, batch_size, input_height, input_width, address_of(input), input_stride,
            address_of(output), output_stride, None))

    def setup_convolution2d_nhwc_q8(self, *args) -> None:
        st = self.setup_convolution2d_nhwc_q8_status(*args)
        if st != Status.success:
            raise QnnpackError("qnnp_setup_convolution2d_nhwc_q8", st)

    def create_deconvolution2d_nhwc_q8_status(
            self, pad_top, pad_right, pad_bottom, pad_left, adjustment_height, adjustment_width,
            kernel_height, kernel_width, stride_height, stride_width, dilation_height, dilation_width,
            groups, group_input_channels, group_output_channels,
            input_zero_point, input_scale, kernel_zero_point, kernel_scale,
            kernel, bias, output_zero_point, output_scale, output_min, output_max, flags=0):
        """kernel: [groups][group_input_channels][kh][kw][group_output_channels] (reference include/qnnpack.h:78-105)."""
        kernel = None if kernel is None else np.ascontiguousarray(kernel, dtype=np.uint8)
        bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.int32)
        handle = c_void_p(None)
        st = self.lib.qnnp_create_deconvolution2d_nhwc_q8(
            pad_top, pad_right, pad_bottom, pad_left, adjustment_height, adjustment_width,
            kernel_height, kernel_width, stride_height, stride_width, dilation_height, dilation_width,
            groups, group_input_channels, group_output_channels,
            input_zero_point, input_scale, kernel_zero_point, kernel_scale,
            address_of(kernel), address_of(bias),
            output_zero_point, output_scale, output_min, output_max, flags, ctypes.byref(handle))
        return Status(st), handle.value

    def create_deconvolution2d_nhwc_q8(self, *args, **kwargs) -> int:
        st, handle = self.create_deconvolution2d_nhwc_q8_status(*args, **kwargs)
        if st != Status.success:
            raise QnnpackError("qnnp_create_deconvolution2d_nhwc_q8", st)
        return handle

    def setup_deconvolution2d_nhwc_q8_status(
            self, op, batch_size, input_height, input_width, input, input_stride, output, output_stride) -> Status:
        return Status(self.lib.qnnp_setup_deconvolution2d_nhwc_q8(
            op, batch_size, input_height, input_width, address_of(input), input_stride,
            address_of(output), output_stride, None))

    def setup_deconvolution2d_nhwc_q8(self, *args) -> None:
        st = self.setup_deconvolution2d_nhwc_q8_status(*args)
        if st != Status.success:
            raise QnnpackError("qnnp_setup_deconvolution2d_nhwc_q8", st)

    def create_global_average_pooling_nwc_q8_status(
            self, channels, input_zero_point, input_scale, output_zero_point, output_scale,
            output_min, output_max, flags=0):
        handle = c_void_p(None)
        st = self.lib.qnnp_create_global_average_pooling_nwc_q8(
            channels, input_zero_point, input_scale, output_zero_point, output_scale, output_min, output_max, flags,
            ctypes.byref(handle))
        return Status(st), handle.value

    def create_global_average_pooling_nwc_q8(self, *args, **kwargs) -> int:
        st, handle = self.create_global_average_pooling_nwc_q8_status(*args, **kwargs)
        if st != Status.success:
            raise QnnpackError("qnnp_create_global_average_pooling_nwc_q8", st)
        return handle

    def setup_global_average_pooling_nwc_q8_status(self, op, batch_size, width, input, input_stride,
                                                   output, output_stride) -> Status:
        return Status(self.lib.qnnp_setup_global_average_pooling_nwc_q8(
            op, batch_size, width, address_of(input), input_stride, address_of(output), output_stride))

    def setup_global_average_pooling_nwc_q8(self, *args) -> None:
        st = self.setup_global_average_pooling_nwc_q8_status(*args)
        if st != Status.success:
            raise QnnpackError("qnnp_setup_global_average_pooling_nwc_q8", st)

    def create_add_nc_q8_status(self, channels, a_zero_point, a_scale, b_zero_point, b_scale,
                                sum_zero_point, sum_scale, sum_min, sum_max, flags=0):
        handle = c_void_p(None)
        st = self.lib.qnnp_create_add_nc_q8(channels, a_zero_point, a_scale, b_zero_point, b_scale,
                                            sum_zero_point, sum_scale, sum_min, sum_max, flags, ctypes.byref(handle))
        return Status(st), handle.value

    def create_add_nc_q8(self, *args, **kwargs) -> int:
        st, handle = self.create_add_nc_q8_status(*args, **kwargs)
        if st != Status.success:
            raise QnnpackError("qnnp_create_add_nc_q8", st)
        return handle

    def setup_add_nc_q8_status(self, op, batch_size, a, a_stride, b, b_stride, sum, sum_stride) -> Status:
        return Status(self.lib.qnnp_setup_add_nc_q8(
            op, batch_size, address_of(a), a_stride, address_of(b), b_stride, address_of(sum), sum_stride))

    def setup_add_nc_q8(self, *args) -> None:
        st = self.setup_add_nc_q8_status(*args)
        if st != Status.success:
            raise QnnpackError("qnnp_setup_add_nc_q8", st)

    def create_fully_connected_nc_q8_status(
            self, input_channels, output_channels, input_zero_point, input_scale,
            kernel_zero_point, kernel_scale, kernel, bias,
            output_zero_point, output_scale, output_min, output_max, flags=0):
        kernel = None if kernel is None else np.ascontiguousarray(kernel, dtype=np.uint8)
        bias = None if bias is None else np.ascontiguousarray(bias, dtype=np.int32)
        handle = c_void_p(None)
        st = self.lib.qnnp_create_fully_connected_nc_q8(
            input_channels, output_channels, input_zero_point, input_scale,
            kernel_zero_point, kernel_scale, address_of(kernel), address_of(bias),
            output_zero_point, output_scale, output_min, output_max, flags, ctypes.byref(handle))
        return Status(st), handle.value

    def create_fully_connected_nc_q8(self, *args, **kwargs) -> int:
        st, handle = self.create_fully_connected_nc_q8_status(*args, **kwargs)
        if st != Status.success:
            raise QnnpackError("qnnp_create_fully_connected_nc_q8", st)
        return handle

    def setup_fully_connected_nc_q8_status(self, op, batch_size, input, input_stride, output, output_stride) -> Status:
        return Status(self.lib.qnnp_setup_fully_connected_nc_q8(
            op, batch_size, address_of(input), input_stride, address_of(output), output_stride))

    def setup_fully_connected_nc_q8(self, *args) -> None:
        st = self.setup_fully_connected_nc_q8_status(*args)
        if st != Status.success:
            raise QnnpackError("qnnp_setup_fully_connected_nc_q8", st)

    def run_operator_status(self, op, threadpool=None) -> Status:
        return Status(self.lib.qnnp_run_operator(op, threadpool))

    def run_operator(self, op, threadpool=None) -> None:
        st = self.lib.qnnp_run_operator(op, threadpool)
        if st != 0:
            raise QnnpackError("qnnp_run_operator", st)

    def delete_operator_status(self, op) -> Status:
        return Status(self.lib.qnnp_delete_operator(op))

    def delete_operator(self, op) -> None:
        st = self.lib.qnnp_delete_operator(op)
        if st != 0:
            raise QnnpackError("qnnp_delete_operator", st)


class Gfx950Library(QnnpackLibrary):
    """Product library: qnnpack.h plus the extensions of include/qnnpack_gfx950.h."""

    def __init__(self, path: str):
        super().__init__(path)
        L = self.lib
        L.qnnp_gfx950_set_device.restype = c_int
        L.qnnp_gfx950_set_device.argtypes = [c_int]
        L.qnnp_gfx950_get_device.restype = c_int
        L.qnnp_gfx950_device_count.restype = c_int
        L.qnnp_gfx950_device_count.argtypes = []
        L.qnnp_gfx950_set_stream.restype = c_int
        L.qnnp_gfx950_set_stream.argtypes = [c_void_p]
        L.qnnp_gfx950_set_async.restype = c_int
        L.qnnp_gfx950_set_async.argtypes = [c_int]
        L.qnnp_gfx950_synchronize.restype = c_int
        L.qnnp_gfx950_malloc.restype = c_void_p
        L.qnnp_gfx950_malloc.argtypes = [c_size_t]
        L.qnnp_gfx950_free.restype = None
        L.qnnp_gfx950_free.argtypes = [c_void_p]
        for name in ("qnnp_gfx950_memcpy_h2d", "qnnp_gfx950_memcpy_d2h"):
            getattr(L, name).restype = c_int
            getattr(L, name).argtypes = [c_void_p, c_void_p, c_size_t]
        L.qnnp_gfx950_memset.restype = c_int
        L.qnnp_gfx950_memset.argtypes = [c_void_p, c_int, c_size_t]
        L.qnnp_gfx950_time_operator.restype = c_int
        L.qnnp_gfx950_time_operator.argtypes = [c_void_p, c_int, c_int, POINTER(c_float)]
        L.qnnp_gfx950_time_operator_rotating.restype = c_int
        L.qnnp_gfx950_time_operator_rotating.argtypes = [
            c_void_p, c_size_t, POINTER(c_void_p), POINTER(c_void_p), c_int, c_int, POINTER(c_float)]
        L.qnnp_gfx950_graph_begin.restype = c_int
        L.qnnp_gfx950_graph_begin.argtypes = []
        L.qnnp_gfx950_graph_end.restype = c_int
        L.qnnp_gfx950_graph_end.argtypes = [POINTER(c_void_p)]
        L.qnnp_gfx950_graph_launch.restype = c_int
        L.qnnp_gfx950_graph_launch.argtypes = [c_void_p]
        L.qnnp_gfx950_graph_synchronize.restype = c_int
        L.qnnp_gfx950_graph_synchronize.argtypes = [c_void_p]
        L.qnnp_gfx950_graph_time.restype = c_int
        L.qnnp_gfx950_graph_time.argtypes = [c_void_p, c_int, c_int, POINTER(c_float)]
        L.qnnp_gfx950_graph_destroy.restype = None
        L.qnnp_gfx950_graph_destroy.argtypes = [c_void_p]
        L.qnnp_gfx950_create_fused_block.restype = c_int
        L.qnnp_gfx950_create_fused_block.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]
        L.qnnp_gfx950_setup_fused_block.restype = c_int
        L.qnnp_gfx950_setup_fused_block.argtypes = [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t]
        # (QNNP_GFX950_LIBRARY may point at an OLDER build for a same-box A/B: entry points younger than it are then
        #  simply not bound; the product library must have them all -- tests/test_abi.py)
        if hasattr(L, "qnnp_gfx950_attach_residual_add") or not os.environ.get("QNNP_GFX950_LIBRARY"):
            L.qnnp_gfx950_attach_residual_add.restype = c_int
            L.qnnp_gfx950_attach_residual_add.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
            L.qnnp_gfx950_operator_residual_folded.restype = c_int
            L.qnnp_gfx950_operator_residual_folded.argtypes = [c_void_p]
        L.qnnp_gfx950_set_option.restype = c_int
        L.qnnp_gfx950_set_option.argtypes = [c_char_p, c_int]
        L.qnnp_gfx950_test_force_kernel.restype = c_int           # include/qnnpack_gfx950_test.h
        L.qnnp_gfx950_test_force_kernel.argtypes = [c_char_p, c_int]
        L.qnnp_gfx950_test_operator_ran_dense.restype = c_int
        L.qnnp_gfx950_test_operator_ran_dense.argtypes = [c_void_p]
        L.qnnp_gfx950_operator_set_streaming_stores.restype = c_int
        L.qnnp_gfx950_operator_set_streaming_stores.argtypes = [c_void_p, c_int]
        L.qnnp_gfx950_operator_kernel.restype = c_char_p
        L.qnnp_gfx950_operator_kernel.argtypes = [c_void_p]
        L.qnnp_gfx950_device_info.restype = c_int
        L.qnnp_gfx950_device_info.argtypes = [
            ctypes.c_char_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]

    def _check(self, call: str, st: int) -> None:
        if st != 0:
            raise QnnpackError(call, st)

    def set_device(self, device: int) -> None:
        self._check("qnnp_gfx950_set_device", self.lib.qnnp_gfx950_set_device(device))

    def get_device(self) -> int:
        return self.lib.qnnp_gfx950_get_device()

    def device_count(self) -> int:
        return self.lib.qnnp_gfx950_device_count()

    def set_stream(self, stream: Optional[int]) -> None:
        self._check("qnnp_gfx950_set_stream", self.lib.qnnp_gfx950_set_stream(stream))

    def set_async(self, enabled: bool) -> None:
        self._check("qnnp_gfx950_set_async", self.lib.qnnp_gfx950_set_async(1 if enabled else 0))

    def synchronize(self) -> None:
        self._check("qnnp_gfx950_synchronize", self.lib.qnnp_gfx950_synchronize())

    def malloc(self, nbytes: int) -> int:
        p = self.lib.qnnp_gfx950_malloc(nbytes)
        if not p:
            raise MemoryError(f"qnnp_gfx950_malloc({nbytes})")
        return p

    def free(self, ptr: int) -> None:
        self.lib.qnnp_gfx950_free(ptr)

    def memcpy_h2d(self, dst: int, src: np.ndarray) -> None:
        src = np.ascontiguousarray(src)
        self._check("qnnp_gfx950_memcpy_h2d", self.lib.qnnp_gfx950_memcpy_h2d(dst, src.ctypes.data, src.nbytes))

    def memcpy_d2h(self, dst: np.ndarray, src: int) -> None:
        assert dst.flags["C_CONTIGUOUS"]
        self._check("qnnp_gfx950_memcpy_d2h", self.lib.qnnp_gfx950_memcpy_d2h(dst.ctypes.data, src, dst.nbytes))

    def memset(self, dst: int, value: int, nbytes: int) -> None:
        self._check("qnnp_gfx950_memset", self.lib.qnnp_gfx950_memset(dst, value, nbytes))

    def time_operator(self, op, warmup: int, iters: int) -> float:
        ms = c_float(0.0)
        self._check("qnnp_gfx950_time_operator",
                    self.lib.qnnp_gfx950_time_operator(op, warmup, iters, ctypes.byref(ms)))
        return float(ms.value)

    def time_operator_rotating(self, op, inputs, outputs, warmup: int, iters: int) -> float:
        n = len(inputs)
        assert n == len(outputs) and n > 0
        ins = (c_void_p * n)(*[address_of(x) for x in inputs])
        outs = (c_void_p * n)(*[address_of(x) for x in outputs])
        ms = c_float(0.0)
        self._check("qnnp_gfx950_time_operator_rotating",
                    self.lib.qnnp_gfx950_time_operator_rotating(op, n, ins, outs, warmup, iters, ctypes.byref(ms)))
        return float(ms.value)

    # ---- fused inverted-residual block (qnnpack_gfx950.h) ----
    def create_fused_block_status(self, expand, depthwise, project, residual_add=None):
        handle = c_void_p(None)
        st = self.lib.qnnp_gfx950_create_fused_block(expand, depthwise, project, residual_add, ctypes.byref(handle))
        return Status(st), handle.value

    def create_fused_block(self, expand, depthwise, project, residual_add=None) -> int:
        st, handle = self.create_fused_block_status(expand, depthwise, project, residual_add)
        if st != Status.success:
            raise QnnpackError("qnnp_gfx950_create_fused_block", st)
        return handle

    def setup_fused_block_status(self, op, batch_size, input_height, input_width, input, input_stride,
                                 output, output_stride) -> Status:
        return Status(self.lib.qnnp_gfx950_setup_fused_block(
            op, batch_size, input_height, input_width, address_of(input), input_stride, address_of(output), output_stride))

    def setup_fused_block(self, *args) -> None:
        st = self.setup_fused_block_status(*args)
        if st != Status.success:
            raise QnnpackError("qnnp_gfx950_setup_fused_block", st)

    # ---- residual add folded into a convolution (qnnpack_gfx950.h) ----
    def attach_residual_add_status(self, convolution, add, residual, residual_stride) -> Status:
        return Status(self.lib.qnnp_gfx950_attach_residual_add(convolution, add, address_of(residual), residual_stride))

    def attach_residual_add(self, convolution, add, residual, residual_stride) -> None:
        st = self.attach_residual_add_status(convolution, add, residual, residual_stride)
        if st != Status.success:
            raise QnnpackError("qnnp_gfx950_attach_residual_add", st)

    def operator_residual_folded(self, op) -> int:
        """After a run: 1 = the add rode in the convolution kernel's epilogue, 0 = separate launch, -1 = none attached."""
        return int(self.lib.qnnp_gfx950_operator_residual_folded(op))

    # ---- hipGraph capture of operator launches (qnnpack_gfx950.h) ----
    def graph_begin(self) -> None:
        self._check("qnnp_gfx950_graph_begin", self.lib.qnnp_gfx950_graph_begin())

    def graph_end(self) -> int:
        g = c_void_p()
        self._check("qnnp_gfx950_graph_end", self.lib.qnnp_gfx950_graph_end(ctypes.byref(g)))
        return g.value

    def graph_launch(self, graph: int) -> None:
        self._check("qnnp_gfx950_graph_launch", self.lib.qnnp_gfx950_graph_launch(graph))

    def graph_synchronize(self, graph: int) -> None:
        self._check("qnnp_gfx950_graph_synchronize", self.lib.qnnp_gfx950_graph_synchronize(graph))

    def graph_time(self, graph: int, warmup: int, iters: int) -> float:
        ms = c_float(0.0)
        self._check("qnnp_gfx950_graph_time", self.lib.qnnp_gfx950_graph_time(graph, warmup, iters, ctypes.byref(ms)))
        return float(ms.value)

    def graph_destroy(self, graph: int) -> None:
        self.lib.qnnp_gfx950_graph_destroy(graph)

    def set_option(self, key: str, value: int) -> None:
        if key in ("gemm_kernel", "dwconv_kernel", "fused_kernel", "fused_rows", "fused_weights"):
            # kernel-forcing codes are test / measurement hooks (include/qnnpack_gfx950_test.h), not product options
            self._check("qnnp_gfx950_test_force_kernel", self.lib.qnnp_gfx950_test_force_kernel(key.encode(), value))
            return
        self._check("qnnp_gfx950_set_option", self.lib.qnnp_gfx950_set_option(key.encode(), value))

    def operator_set_streaming_stores(self, op, value: int) -> None:
        """1 / 0: the streaming-store hint of this operator's launches; -1: follow the process-wide option again."""
        self._check("qnnp_gfx950_operator_set_streaming_stores", self.lib.qnnp_gfx950_operator_set_streaming_stores(op, value))

    def operator_ran_dense(self, op) -> bool:
        """include/qnnpack_gfx950_test.h: the operator's last run took the dense image of a grouped 1x1 convolution"""
        return bool(self.lib.qnnp_gfx950_test_operator_ran_dense(op))

    def operator_kernel(self, op) -> Optional[str]:
        name = self.lib.qnnp_gfx950_operator_kernel(op)
        return name.decode() if name else None

    def device_info(self) -> dict:
        arch = ctypes.create_string_buffer(64)
        cus, clk, mem = c_int(0), c_int(0), c_size_t(0)
        self._check("qnnp_gfx950_device_info",
                    self.lib.qnnp_gfx950_device_info(arch, 64, ctypes.byref(cus), ctypes.byref(clk), ctypes.byref(mem)))
        return {"arch": arch.value.decode(), "compute_units": cus.value, "clock_khz": clk.value,
                "hbm_bytes": mem.value}
