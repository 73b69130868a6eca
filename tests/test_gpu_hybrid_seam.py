"""GPU tier: the reference tree with its hot path re-routed at the documented seam, as a library that really links and
runs (SURVEY section 8b "internal seam"). oracle/_ref/libqnnpack_hybrid.so = the unmodified
reference objects + the reference's src/operator-run.c with ONE inserted statement at the top of qnnp_run_operator
(oracle/make_seam.py) + the product's host code and HIP kernels behind it (oracle/hybrid_seam.c, oracle/Makefile target
`hybrid`). Through the reference's one public API:
  * a convolution and a fully connected operator run on the MI355X (bit-exact with the scalar oracle),
  * a max-pooling operator -- not part of the hot path -- runs the reference's own SSE2 kernel on the host,
  * both are run by the same qnnp_run_operator and deleted by the same qnnp_delete_operator."""
import ctypes
import os
from ctypes import POINTER, c_size_t, c_uint8, c_uint32, c_void_p

import numpy as np
import pytest

from _cases import ConvCase, FcCase
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HYBRID = os.path.join(ROOT, "oracle", "_ref", "libqnnpack_hybrid.so")


@pytest.fixture(scope="module")
def hybrid():
    if not os.path.exists(HYBRID):
        pytest.skip("oracle/_ref/libqnnpack_hybrid.so was not built (needs /root/reference at build time)")
    import torch  # noqa: F401 -- the HIP runtime the library binds to
    from qnnpack_amd.binding import QnnpackLibrary
    lib = QnnpackLibrary(HYBRID)
    lib.initialize()
    L = lib.lib
    # reference include/qnnpack.h:162-189
    L.qnnp_create_max_pooling2d_nhwc_u8.restype = ctypes.c_int
    L.qnnp_create_max_pooling2d_nhwc_u8.argtypes = [c_uint32] * 10 + [c_size_t, c_uint8, c_uint8, c_uint32, POINTER(c_void_p)]
    L.qnnp_setup_max_pooling2d_nhwc_u8.restype = ctypes.c_int
    L.qnnp_setup_max_pooling2d_nhwc_u8.argtypes = [c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_size_t, c_void_p,
                                                   c_size_t, c_void_p]
    L.qnnp_hybrid_owns.restype = ctypes.c_int
    L.qnnp_hybrid_owns.argtypes = [c_void_p]
    return lib


def test_convolution_and_fully_connected_run_on_the_device_through_the_reference_dispatch(hybrid):
    conv = ConvCase("seam_conv3x3", (14, 13), (3, 3), (1, 1, 1, 1), gic=32, goc=48, batch=3)
    expected, quant, out_hw = conv_expected(conv)
    out, _ = conv_run(hybrid, conv, quant, out_hw)                 # host pointers, as a reference caller passes them
    assert_bytes_equal(out, expected, "hybrid library: 3x3 convolution vs oracle")
    dw = ConvCase("seam_dw3x3", (12, 12), (3, 3), (1, 1, 1, 1), groups=40, batch=2)
    expected, quant, out_hw = conv_expected(dw)
    out, _ = conv_run(hybrid, dw, quant, out_hw)
    assert_bytes_equal(out, expected, "hybrid library: depthwise convolution vs oracle")
    fc = FcCase("seam_fc", 33, 256, 100)
    expected, quant = fc_expected(fc)
    out, _ = fc_run(hybrid, fc, quant)
    assert_bytes_equal(out, expected, "hybrid library: fully connected vs oracle")


def test_max_pooling_keeps_the_reference_cpu_kernel_and_shares_run_and_delete(hybrid):
    L = hybrid.lib
    rng = np.random.default_rng(5)
    n, h, w, c = 2, 9, 10, 24
    x = rng.integers(0, 256, size=(n, h, w, c), dtype=np.uint8)
    y = np.zeros((n, (h - 3) // 2 + 1, (w - 3) // 2 + 1, c), dtype=np.uint8)
    op = c_void_p()
    # padding 0, 3x3 window, stride 2, dilation 1 (reference include/qnnpack.h:162-175)
    assert L.qnnp_create_max_pooling2d_nhwc_u8(0, 0, 0, 0, 3, 3, 2, 2, 1, 1, c, 0, 255, 0, ctypes.byref(op)) == 0
    assert L.qnnp_hybrid_owns(op) == 0                             # a reference (CPU) operator
    assert L.qnnp_setup_max_pooling2d_nhwc_u8(op, n, h, w, x.ctypes.data, c, y.ctypes.data, c, None) == 0
    hybrid.run_operator(op.value)                                  # the SAME entry point the device operators go through
    want = np.zeros_like(y)
    for oy in range(y.shape[1]):
        for ox in range(y.shape[2]):
            want[:, oy, ox, :] = x[:, 2 * oy:2 * oy + 3, 2 * ox:2 * ox + 3, :].max(axis=(1, 2))
    assert np.array_equal(y, want), "reference max pooling through the hybrid library"
    hybrid.delete_operator(op.value)

    # and a device operator created next to it is owned by the seam, runs and is deleted through the same calls
    fc = FcCase("seam_fc2", 8, 64, 32)
    inp_expected, quant = fc_expected(fc)
    out, _ = fc_run(hybrid, fc, quant)
    assert_bytes_equal(out, inp_expected, "hybrid library: fully connected after the pooling operator")
