"""Pins the scalar oracle's transposed convolution (oracle_deconv2d_acc, the gather form of
test/deconvolution-operator-tester.h:383-419) to the reference: it must reproduce, byte for byte, the outputs
the COMPILED REFERENCE produced through qnnp_create/setup_deconvolution2d_nhwc_q8 for the seeded cases
(tests/golden/reference_deconv_outputs.npz, made by tests/golden/generate_golden_deconv.py from the reference's
own test/deconvolution.cc list plus extras). With oracle/_ref present the comparison is also done live."""
import os

import numpy as np
import pytest

from _cases import DECONV_CASES, EXTRA_DECONV_CASES, deconv_tensors
from _runner import assert_bytes_equal, deconv_expected, deconv_run
from oracle import ref

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_deconv_outputs.npz")
_GOLDEN = dict(np.load(_PATH))
_NAMES = sorted({k.split("/")[1] for k in _GOLDEN})
BY_NAME = {c.name: c for c in DECONV_CASES + EXTRA_DECONV_CASES}


@pytest.mark.parametrize("name", _NAMES)
def test_oracle_reproduces_reference_deconvolution(name):
    case = BY_NAME[name]
    key = "deconv/" + name
    inp, kernel, bias = _GOLDEN[key + "/input"], _GOLDEN[key + "/kernel"], _GOLDEN[key + "/bias"]
    scale, zp = _GOLDEN[key + "/quant"]
    g_inp, g_kernel, g_bias = deconv_tensors(case)
    assert np.array_equal(inp, g_inp) and np.array_equal(kernel, g_kernel) and np.array_equal(bias, g_bias)
    out, o_quant, _ = deconv_expected(case, inp, kernel, bias)
    assert (float(o_quant[0]), o_quant[1]) == (float(np.float32(scale)), int(zp))
    assert_bytes_equal(out, _GOLDEN[key + "/output"], f"oracle vs reference golden [{name}]")


def test_golden_covers_the_reference_case_list():
    assert {c.name for c in DECONV_CASES if c.batch > 0} <= set(_NAMES)


needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def _padded(buf):
    p = np.concatenate([np.zeros(8, np.uint8), buf, np.zeros(8, np.uint8)])
    return p[8:8 + buf.size] if buf.size else p[8:9]


@needs_ref
@pytest.mark.parametrize("case", DECONV_CASES + EXTRA_DECONV_CASES, ids=lambda c: c.name)
def test_oracle_equals_compiled_reference_deconvolution_live(case):
    inp, kernel, bias = deconv_tensors(case)
    expected, quant, out_hw = deconv_expected(case, inp, kernel, bias)
    out, _ = deconv_run(ref.lib(), case, quant, out_hw, _padded(inp), kernel, bias)
    assert_bytes_equal(out, expected, f"compiled reference vs oracle [{case.name}]")
