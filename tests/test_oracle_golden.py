"""Pins the scalar oracle (O1) to the reference: it must reproduce, byte for byte,
the outputs the COMPILED REFERENCE produced for the seeded operator cases
(tests/golden/reference_outputs.npz, made by tests/golden/generate_golden.py from
the reference's own test/convolution.cc and test/fully-connected.cc case lists).
When oracle/_ref is present (build container) the comparison is also done live,
single- and multi-threaded."""
import numpy as np
import pytest

import _golden
from _cases import CONV_CASES, EXTRA_CONV_CASES, EXTRA_FC_CASES, FC_CASES, conv_tensors, fc_tensors
from _runner import assert_bytes_equal, conv_expected, conv_run, fc_expected, fc_run
from oracle import o1, ref

CONV_BY_NAME = {c.name: c for c in CONV_CASES + EXTRA_CONV_CASES}
FC_BY_NAME = {c.name: c for c in FC_CASES + EXTRA_FC_CASES}


@pytest.mark.parametrize("name", _golden.names("conv"))
def test_oracle_reproduces_reference_convolution(name):
    case = CONV_BY_NAME[name]
    inp, kernel, bias, quant, ref_out = _golden.entry("conv", name)
    # fixture inputs must be what the seeded generator still produces (guards generator drift)
    g_inp, g_kernel, g_bias = conv_tensors(case)
    assert np.array_equal(inp, g_inp) and np.array_equal(kernel, g_kernel) and np.array_equal(bias, g_bias)
    out, o_quant, _ = conv_expected(case, inp, kernel, bias)
    assert (float(o_quant[0]), o_quant[1]) == (float(quant[0]), quant[1])
    assert_bytes_equal(out, ref_out, f"oracle vs reference golden [{name}]")


@pytest.mark.parametrize("name", _golden.names("fc"))
def test_oracle_reproduces_reference_fully_connected(name):
    case = FC_BY_NAME[name]
    inp, kernel, bias, quant, ref_out = _golden.entry("fc", name)
    g_inp, g_kernel, g_bias = fc_tensors(case)
    assert np.array_equal(inp, g_inp) and np.array_equal(kernel, g_kernel) and np.array_equal(bias, g_bias)
    out, o_quant = fc_expected(case, inp, kernel, bias)
    assert (float(o_quant[0]), o_quant[1]) == (float(quant[0]), quant[1])
    assert_bytes_equal(out, ref_out, f"oracle vs reference golden [{name}]")


needs_ref = pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")


def _padded(buf):
    # the reference's SSE2 kernels may read up to 7 bytes before a row (4x4c2-sse2.c:111-114)
    p = np.concatenate([np.zeros(8, np.uint8), buf, np.zeros(8, np.uint8)])
    return p[8:8 + buf.size] if buf.size else p[8:9]


@needs_ref
@pytest.mark.parametrize("case", CONV_CASES + EXTRA_CONV_CASES, ids=lambda c: c.name)
def test_oracle_equals_compiled_reference_convolution_live(case):
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, _ = conv_run(ref.lib(), case, quant, out_hw, _padded(inp), kernel, bias)
    assert_bytes_equal(out, expected, f"compiled reference vs oracle [{case.name}]")


@needs_ref
@pytest.mark.parametrize("case", FC_CASES + EXTRA_FC_CASES, ids=lambda c: c.name)
def test_oracle_equals_compiled_reference_fully_connected_live(case):
    inp, kernel, bias = fc_tensors(case)
    expected, quant = fc_expected(case, inp, kernel, bias)
    out, _ = fc_run(ref.lib(), case, quant, _padded(inp), kernel, bias)
    assert_bytes_equal(out, expected, f"compiled reference vs oracle [{case.name}]")


@needs_ref
def test_compiled_reference_threadpool_matches_single_thread():
    # cpu_baseline uses the OpenMP pthreadpool shim (oracle/ref_stubs.c); results must not depend on it
    case = CONV_BY_NAME["x_3x3_c64_vec16"]
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    lib = ref.lib()
    pool = lib.threadpool(4)
    try:
        out, _ = conv_run(lib, case, quant, out_hw, _padded(inp), kernel, bias, threadpool=pool)
    finally:
        lib.destroy_threadpool(pool)
    assert_bytes_equal(out, expected, "compiled reference with 4-thread pool vs oracle")


def test_oracle_threads_do_not_change_results():
    case = CONV_BY_NAME["3x3s2"]
    o1.set_threads(1)
    a, _, _ = conv_expected(case)
    o1.set_threads(4)
    b, _, _ = conv_expected(case)
    o1.set_threads(1)
    assert np.array_equal(a, b)
