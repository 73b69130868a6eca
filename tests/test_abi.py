"""CPU tier: the C-ABI library loads without a GPU, exports every symbol the public headers
declare, and fails LOUDLY (no CPU fallback) when no gfx950 device is present."""
import ctypes
import os
import re

import numpy as np
import pytest

import qnnpack_amd
from qnnpack_amd import Status

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qnnp_[a-z0-9_]+)\s*\(", text)))


def test_headers_declare_the_reference_entry_points():
    names = declared_functions("qnnpack.h")
    # reference include/qnnpack.h:34-36, 40-76, 78-116, 118-140, 142-160, 234-255, 327-332
    # (hot-path subset + the "next" rows of SURVEY 8f: deconvolution, global average pooling, add)
    assert names == sorted([
        "qnnp_initialize", "qnnp_deinitialize",
        "qnnp_create_convolution2d_nhwc_q8", "qnnp_setup_convolution2d_nhwc_q8",
        "qnnp_create_deconvolution2d_nhwc_q8", "qnnp_setup_deconvolution2d_nhwc_q8",
        "qnnp_create_fully_connected_nc_q8", "qnnp_setup_fully_connected_nc_q8",
        "qnnp_create_global_average_pooling_nwc_q8", "qnnp_setup_global_average_pooling_nwc_q8",
        "qnnp_create_add_nc_q8", "qnnp_setup_add_nc_q8",
        "qnnp_run_operator", "qnnp_delete_operator"])


@pytest.mark.parametrize("header", ["qnnpack.h", "qnnpack_gfx950.h", "qnnpack_gfx950_test.h"])
def test_library_exports_every_declared_symbol(product, header):
    for name in declared_functions(header):
        assert hasattr(product.lib, name), f"{name} declared in include/{header} but not exported"


def test_kernel_forcing_codes_are_not_product_options(product):
    """round 6: "gemm_kernel" & co. moved out of qnnp_gfx950_set_option into the test header's entry point"""
    assert product.lib.qnnp_gfx950_set_option(b"gemm_kernel", 0) == Status.invalid_parameter
    assert product.lib.qnnp_gfx950_set_option(b"dwconv_kernel", 0) == Status.invalid_parameter
    assert product.lib.qnnp_gfx950_test_force_kernel(b"gemm_kernel", 23) == 0
    assert product.lib.qnnp_gfx950_test_force_kernel(b"gemm_kernel", 0) == 0
    assert product.lib.qnnp_gfx950_test_force_kernel(b"gemm_kernel", 18) == Status.invalid_parameter
    assert product.lib.qnnp_gfx950_test_force_kernel(b"no_such_family", 0) == Status.invalid_parameter
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "qnnpack_gfx950.h")).read()
    assert '"gemm_kernel":' not in text and "qnnp_gfx950_test_force_kernel" in text


def test_status_enum_values_match_reference():
    # reference include/qnnpack.h:24-32
    assert [s.value for s in Status] == [0, 1, 2, 3, 4, 5]
    assert Status.unsupported_hardware == 4 and Status.out_of_memory == 5


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_has_gpu(), reason="CPU-only behaviour")
def test_no_gpu_means_unsupported_hardware_not_a_fallback(product):
    assert product.initialize_status() == Status.unsupported_hardware
    kernel = np.zeros((4, 4), np.uint8)
    bias = np.zeros(4, np.int32)
    st, handle = product.create_fully_connected_nc_q8_status(4, 4, 0, 1.0, 0, 1.0, kernel, bias, 0, 2.0, 0, 255)
    assert st == Status.uninitialized and not handle       # reference fully-connected.c:44-47
    st, handle = product.create_add_nc_q8_status(4, 0, 1.0, 0, 1.0, 0, 1.0, 0, 255)
    assert st == Status.uninitialized and not handle       # reference add.c:36-39
    st, handle = product.create_global_average_pooling_nwc_q8_status(4, 0, 1.0, 0, 1.0, 0, 255)
    assert st == Status.uninitialized and not handle       # reference global-average-pooling.c:34-37
    st, handle = product.create_deconvolution2d_nhwc_q8_status(
        0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 4, 4, 0, 1.0, 0, 1.0, kernel, bias, 0, 2.0, 0, 255)
    assert st == Status.uninitialized and not handle       # reference deconvolution.c:69-72
    st, handle = product.create_convolution2d_nhwc_q8_status(
        0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 4, 4, 0, 1.0, 0, 1.0, kernel, bias, 0, 2.0, 0, 255)
    assert st == Status.uninitialized and not handle       # reference convolution.c:69-72
    assert product.malloc.__self__ is product
    with pytest.raises(MemoryError):
        product.malloc(16)


def test_delete_null_is_invalid_parameter(product):
    # reference operator-delete.c:17-19
    assert product.delete_operator_status(None) == Status.invalid_parameter


def test_run_null_is_invalid_parameter(product):
    assert product.run_operator_status(None) == Status.invalid_parameter


def test_product_does_not_link_the_oracle(product):
    # the product path must not route through the checker
    with open(qnnpack_amd.library_path(), "rb") as f:
        blob = f.read()
    assert b"oracle_" not in blob and b"liboracle" not in blob and b"libqnnpack_ref" not in blob
    src = os.path.join(ROOT, "qnnpack_amd")
    for dirpath, _, files in os.walk(src):
        for fn in files:
            if fn.endswith((".py", ".c", ".h", ".hip", ".cuh")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "import oracle" not in text and "from oracle" not in text and "oracle_q8" not in text, fn
