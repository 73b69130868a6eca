"""Pins the oracle's Q31 requantization against the reference's deterministic
known-answer tests: test/requantization-tester.h as driven for Q31__SCALAR in
test/requantization.cc:250-308 (exact_divide_by_po2_with_zero_point,
divide_by_po2_with_rounding_up, ..._rounding_away, special_cases, random_cases).
"No rounding down test - it fails because of upward bias in multiplication"
(test/requantization.cc:280) -- reproduced below as a documented property.
"""
import numpy as np
import pytest

from oracle import o1, ref

INT32_MIN, INT32_MAX = -2**31, 2**31 - 1


def _req(inputs, scale, zp, qmin=0, qmax=255):
    return o1.q31_requantize(np.asarray(inputs, dtype=np.int64).astype(np.int32), scale, zp, qmin, qmax)


@pytest.mark.parametrize("s", range(1, 32))
def test_exact_divide_by_po2_with_zero_point(s):
    # requantization-tester.h:84-109, zero points 0..255 (requantization.cc:250-260)
    for zp in range(0, 256, 3 if s > 4 else 1):
        max_i = (INT32_MAX >> s) + zp
        min_i = -((2**31) >> s) + zp
        i = np.arange(256)
        clamped = np.clip(i, min_i, max_i)
        inputs = ((clamped - zp).astype(np.int64) << s)
        inputs = ((inputs + 2**31) % 2**32 - 2**31)           # int32(uint32(x) << s)
        out = _req(inputs, 2.0 ** -s, zp)
        assert np.array_equal(out, clamped.astype(np.uint8)), (s, zp)


@pytest.mark.parametrize("s", range(1, 32))
def test_divide_by_po2_with_rounding_up(s):
    # requantization-tester.h:118-144
    for zp in range(0, 256, 5):
        i = np.arange(256, dtype=np.int64)
        inputs = ((i - zp) << s) - (1 << (s - 1)) + (i <= zp)
        ok = (inputs >= INT32_MIN) & (inputs <= INT32_MAX)
        out = _req(np.where(ok, inputs, 0), 2.0 ** -s, zp)
        assert np.array_equal(out[ok], i[ok].astype(np.uint8)), (s, zp)


@pytest.mark.parametrize("s", range(2, 32))
def test_divide_by_po2_with_rounding_away(s):
    # requantization-tester.h:181-215. The reference driver loops s from 1, but at s == 1
    # (scale 0.5 -> shift 0, remainder mask 0) the reference implementation itself
    # rounds negative halves UP (measured on the compiled qnnp_requantize_q31__scalar:
    # -9 * 0.5 -> -4), so that iteration of its own test cannot pass for zero points >= 1.
    # The oracle follows the implementation; s == 1 is pinned by the test below.
    for zp in range(0, 256, 5):
        i = np.arange(256, dtype=np.int64)
        inputs = (i - zp) << s
        inputs = np.where(inputs > 0, inputs - (1 << (s - 1)), np.where(inputs < 0, inputs + (1 << (s - 1)), inputs))
        ok = (inputs >= INT32_MIN) & (inputs <= INT32_MAX)
        out = _req(np.where(ok, inputs, 0), 2.0 ** -s, zp)
        assert np.array_equal(out[ok], i[ok].astype(np.uint8)), (s, zp)


def test_shift_zero_rounds_half_up_like_the_reference_implementation():
    # src/qnnpack/requantization.h:469-471 with shift == 0: q31 product is round-half-up and the
    # remainder correction is inert (mask 0, threshold 0).
    out = _req([-9, -7, -5, -3, -1, 1, 3, 5], 0.5, 5)
    assert out.tolist() == [1, 2, 3, 4, 5, 6, 7, 8]


def test_rounding_down_is_not_exact_for_q31():
    # test/requantization.cc:280: the rounding-down property (tester.h:146-179) does NOT hold for
    # q31 because of the upward bias of the first rounding; the oracle must reproduce that.
    s, zp = 8, 128
    i = np.arange(256, dtype=np.int64)
    inputs = ((i - zp) << s) + (1 << (s - 1)) - (i >= zp)
    out = _req(inputs, 2.0 ** -s, zp)
    assert not np.array_equal(out, i.astype(np.uint8))


def test_special_cases():
    # requantization-tester.h:217-246
    for zp in range(256):
        out = _req([INT32_MIN] * 256, 2.0 ** -32, zp)
        assert int(out.min()) == max(0, zp - 1)
    out = _req([INT32_MAX] * 256, float.fromhex("0x1.FFFFFEp-1"), 255)
    assert np.all(out == 255)


def test_random_cases_against_precise_rounding():
    # requantization-tester.h:288-328: |q31 - exact| <= 0.55 on random accumulators
    rng = np.random.default_rng(0x51A0)
    for _ in range(50):
        zp = int(rng.integers(0, 256))
        scale = np.float32(rng.uniform(2.0 ** -20, 0.99))
        acc = rng.integers(-2**26, 2**26, size=4096)
        out = _req(acc, scale, zp).astype(np.float64)
        exact = np.clip(acc.astype(np.float64) * float(scale) + zp, 0, 255)
        assert np.max(np.abs(out - exact)) <= 0.55


def test_params_match_reference_ranges():
    # requantization.h:31-38: multiplier in [0x40000000, 0x7FFFFF80], shift in [0, 31]
    for scale in [2.0 ** -32, 2.0 ** -31, 0.25, 0.5, 0.75, float.fromhex("0x1.FFFFFEp-1"), 1 / 255.0]:
        p = o1.q31_params(scale, 1, 2, 250)
        assert 0x40000000 <= p.multiplier <= 0x7FFFFF80
        assert 0 <= p.shift <= 31
        assert p.remainder_mask == (1 << p.shift) - 1 and p.remainder_threshold == p.remainder_mask >> 1
        assert (p.min_less_zero_point, p.max_less_zero_point, p.zero_point) == (1, 249, 1)
    with pytest.raises(ValueError):
        o1.q31_params(1.0, 0, 0, 255)
    with pytest.raises(ValueError):
        o1.q31_params(2.0 ** -33, 0, 0, 255)


@pytest.mark.skipif(not ref.available(), reason="oracle/_ref not built (needs /root/reference)")
def test_oracle_equals_compiled_reference_q31_scalar():
    # O1 == qnnp_requantize_q31__scalar (src/requantization/q31-scalar.c:17-138) on random + edge inputs
    r = ref.lib()
    rng = np.random.default_rng(7)
    acc = rng.integers(INT32_MIN, INT32_MAX + 1, size=1 << 16).astype(np.int32)
    acc[:8] = [INT32_MIN, INT32_MAX, 0, -1, 1, -2**30, 2**30, -2**30 - 1]
    for scale in [2.0 ** -32, 2.0 ** -17, 0.003, 1 / 255.0, 0.49999, 0.5, 0.75, float.fromhex("0x1.FFFFFEp-1")]:
        for zp, qmin, qmax in [(0, 0, 255), (127, 0, 255), (255, 0, 255), (100, 128, 255), (100, 0, 128), (7, 5, 9)]:
            a = o1.q31_requantize(acc, scale, zp, qmin, qmax)
            b = r.requantize_q31_scalar(acc, scale, zp, qmin, qmax)
            assert np.array_equal(a, b), (scale, zp, qmin, qmax)
