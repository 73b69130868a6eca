"""The compiled reference library "O2" (oracle/_ref/libqnnpack_ref.so). TEST INFRASTRUCTURE ONLY.

Built by oracle/Makefile from the unmodified sources under /root/reference
(present in the build container only; the .so travels to the GPU box with the
gpurun snapshot). Bound with the SAME ctypes class as the product, because it
exports the same include/qnnpack.h ABI.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_float, c_size_t, c_uint8, c_void_p

import numpy as np

from qnnpack_amd.binding import QnnpackLibrary

_DIR = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_DIR, "_ref", "libqnnpack_ref.so")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def build() -> bool:
    """Compile O2 if the reference tree is present; returns whether the .so exists afterwards."""
    subprocess.run(["make", "-C", _DIR, "-j8", "ref"], check=True, stdout=subprocess.DEVNULL)
    return available()


class ReferenceLibrary(QnnpackLibrary):
    def __init__(self, path: str):
        super().__init__(path)
        L = self.lib
        L.pthreadpool_create.restype = c_void_p
        L.pthreadpool_create.argtypes = [c_size_t]
        L.pthreadpool_destroy.restype = None
        L.pthreadpool_destroy.argtypes = [c_void_p]
        # void qnnp_requantize_q31__scalar(size_t n, const int32_t*, float scale, uint8_t zp, uint8_t qmin,
        #                                  uint8_t qmax, uint8_t* out)  -- src/requantization/q31-scalar.c:17
        L.qnnp_requantize_q31__scalar.restype = None
        L.qnnp_requantize_q31__scalar.argtypes = [c_size_t, c_void_p, c_float, c_uint8, c_uint8, c_uint8, c_void_p]

    def threadpool(self, threads: int) -> int:
        return self.lib.pthreadpool_create(threads)

    def destroy_threadpool(self, pool: int) -> None:
        self.lib.pthreadpool_destroy(pool)

    def requantize_q31_scalar(self, acc: np.ndarray, scale: float, zero_point: int, qmin: int, qmax: int) -> np.ndarray:
        acc = np.ascontiguousarray(acc, dtype=np.int32)
        assert acc.size % 4 == 0  # the reference asserts n % 4 == 0 (q31-scalar.c:26)
        out = np.empty(acc.shape, dtype=np.uint8)
        self.lib.qnnp_requantize_q31__scalar(acc.size, acc.ctypes.data, np.float32(scale), zero_point, qmin, qmax,
                                             out.ctypes.data)
        return out


def lib() -> ReferenceLibrary:
    global _lib
    if _lib is None:
        if not available():
            raise RuntimeError(f"{_PATH} not built (needs /root/reference; run `make -C oracle ref`)")
        _lib = ReferenceLibrary(_PATH)
        _lib.initialize()
    return _lib
