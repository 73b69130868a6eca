"""CPU tier: the INTEGRATION section 2 seam library (oracle/_ref/libqnnpack_hybrid.so: the reference objects + the
product's objects + one patched statement, oracle/Makefile) must resolve every symbol at load time. Its object lists
are written out in oracle/Makefile; when a product source file is added and the list is not, the link used to succeed
(shared libraries may carry undefined symbols) and the first dlopen on the GPU box failed -- the link now runs with
--no-undefined, and this test loads the library the way tests/test_gpu_hybrid_seam.py will."""
import ctypes
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HYBRID = os.path.join(ROOT, "oracle", "_ref", "libqnnpack_hybrid.so")


def test_hybrid_library_resolves_every_symbol():
    if not os.path.exists(HYBRID):
        pytest.skip("oracle/_ref/libqnnpack_hybrid.so not built (needs /root/reference)")
    try:
        import torch  # noqa: F401  -- the HIP runtime the product binds to (qnnpack_amd.load does the same)
    except Exception:
        pass
    lib = ctypes.CDLL(HYBRID, mode=os.RTLD_NOW | os.RTLD_LOCAL)
    for sym in ("qnnp_initialize", "qnnp_run_operator", "qnnp_create_convolution2d_nhwc_q8",
                "qnnp_create_fully_connected_nc_q8", "qnnp_create_max_pooling2d_nhwc_u8", "qnnp_delete_operator"):
        assert hasattr(lib, sym), sym


def test_oracle_makefile_lists_every_product_object():
    """the seam's object lists against the product's source lists (qnnpack_amd/csrc/Makefile)"""
    import re
    prod = open(os.path.join(ROOT, "qnnpack_amd", "csrc", "Makefile")).read()
    orac = open(os.path.join(ROOT, "oracle", "Makefile")).read()

    def words(text, var):
        m = re.search(rf"^{var}\s*:=\s*(.*?)(?<!\\)$", text, re.M | re.S)
        assert m, var
        return m.group(1).replace("\\\n", " ").split()

    c_srcs = {os.path.splitext(os.path.basename(w))[0] for w in words(prod, "C_SRCS")}
    hip_srcs = {os.path.splitext(os.path.basename(w))[0] for w in words(prod, "HIP_SRCS")}
    assert set(words(orac, "PRODUCT_C_OBJS")) == c_srcs, (sorted(c_srcs), words(orac, "PRODUCT_C_OBJS"))
    assert set(words(orac, "PRODUCT_HIP_OBJS")) == hip_srcs, (sorted(hip_srcs), words(orac, "PRODUCT_HIP_OBJS"))
