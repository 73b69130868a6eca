"""CPU tier: the create/setup-time host logic of the product library (weight packing into MFMA
fragment panels with folded zero points, the device offset table, Q31 parameter builder) is
replayed through a numpy model of the device kernels' index arithmetic and must reproduce the
oracle byte for byte. No GPU and no compute entry point of the library is used here."""
import numpy as np
import pytest

import _emulate as em
from _cases import CONV_CASES, EXTRA_CONV_CASES, EXTRA_FC_CASES, FC_CASES, conv_tensors, fc_tensors, strided_view
from _runner import assert_bytes_equal, conv_expected, fc_expected
from oracle import o1

SMALL_CONV = [c for c in CONV_CASES + EXTRA_CONV_CASES
              if c.batch > 0 and c.batch * c.input_size[0] * c.input_size[1] * c.groups * c.gic <= 30000]
SMALL_FC = [c for c in FC_CASES + EXTRA_FC_CASES if c.batch > 0 and c.batch * c.input_channels <= 40000]


@pytest.fixture(scope="module")
def hooks(debug_hooks):
    return em.bind_debug_hooks(debug_hooks)


def test_requant_params_match_oracle(hooks):
    for scale in [2.0 ** -32, 2.0 ** -20, 0.0031, 1 / 255.0, 0.5, 0.75, float.fromhex("0x1.FFFFFEp-1")]:
        for zp, qmin, qmax in [(0, 0, 255), (127, 0, 255), (255, 10, 200), (5, 128, 255)]:
            rq = em.host_requant(hooks, scale, zp, qmin, qmax)
            p = o1.q31_params(scale, zp, qmin, qmax)
            assert (rq.multiplier, rq.remainder_mask, rq.remainder_threshold, rq.shift,
                    rq.output_min_less_zero_point, rq.output_max_less_zero_point, rq.output_zero_point) == \
                   (p.multiplier, p.remainder_mask, p.remainder_threshold, p.shift,
                    p.min_less_zero_point, p.max_less_zero_point, p.zero_point)


def test_numpy_requant_model_matches_oracle(hooks):
    rng = np.random.default_rng(3)
    acc = rng.integers(-2**31, 2**31, size=20000).astype(np.int32)
    for scale in [2.0 ** -32, 0.003, 0.5, 0.9999]:
        rq = em.host_requant(hooks, scale, 77, 3, 250)
        assert np.array_equal(em.q31_requantize_np(acc, rq), o1.q31_requantize(acc, scale, 77, 3, 250))


@pytest.mark.parametrize("case", SMALL_FC, ids=lambda c: c.name)
def test_fully_connected_host_images(hooks, case):
    inp, kernel, bias = fc_tensors(case)
    expected, (oscale, ozp) = fc_expected(case, inp, kernel, bias)
    rq = em.host_requant(hooks, np.float32(1.0) / oscale, ozp, case.qmin, case.qmax)
    a = strided_view(inp, case.batch, case.input_channels, case.in_stride)
    out = em.emulate_igemm(hooks, 1, case.output_channels, case.input_channels, 1, case.izp, case.kzp,
                           kernel, bias, lambda g: a, case.batch, rq, case.out_stride)
    assert_bytes_equal(out, expected, f"host images replay vs oracle [{case.name}]")


@pytest.mark.parametrize("izp", [0, 3, 127, 128, 255])
@pytest.mark.parametrize("n,k", [(256, 512), (1000, 640), (300, 64)])
def test_centred_gemm_image_replays_the_oracle(debug_hooks, izp, n, k):
    """pack.h qnnp_pack_igemm_w_centred127 + the arithmetic of hip/q8gemm256c.hip for kernel zero point 127: both
    operands recentred with ^ 0x7F (a'' = 127 - a, w'' = 127 - w), no row term, bias folded for both zero points.
    (Kernel zero point 128 is the standard image with a zero row coefficient: test_fully_connected_host_images.)"""
    import ctypes
    L = debug_hooks.lib
    L.qnnp_debug_pack_igemm_w_centred127.restype = None
    L.qnnp_debug_pack_igemm_w_centred127.argtypes = [ctypes.c_uint32] * 3 + [ctypes.c_uint8] + [ctypes.c_void_p] * 4
    rng = np.random.default_rng(n * 7 + k + izp)
    m = 37
    a = rng.integers(0, 256, size=(m, k), dtype=np.uint8)
    w = rng.integers(0, 256, size=(n, k), dtype=np.uint8)
    w[0, :] = 255; w[1, :] = 0; a[0, :] = 255; a[1, :] = 0           # the corners of both ranges
    bias = rng.integers(-10000, 10001, size=n).astype(np.int32)
    bias[2] = 2**31 - 1; bias[3] = -2**31                               # folding wraps like the reference's int32
    n_pad = em.round_up(n, 32)
    packed = np.empty(n_pad * k, dtype=np.int8)
    biasc = np.empty(n_pad, dtype=np.int32)
    L.qnnp_debug_pack_igemm_w_centred127(n, k, n_pad, izp, w.ctypes.data, bias.ctypes.data, packed.ctypes.data, biasc.ctypes.data)
    wc = em.unpack_fragments(packed, 1, n_pad, k)[0]                   # what the MFMA sees
    assert np.array_equal(wc[:n], 127 - w.astype(np.int64)) and not wc[n:].any() and not biasc[n:].any()
    assert np.array_equal(wc[:n].astype(np.int8), (w ^ 0x7F).view(np.int8))
    a_c = (a ^ 0x7F).view(np.int8).astype(np.int64)                    # the kernel's v_xor with 0x7F7F7F7F
    assert np.array_equal(a_c, 127 - a.astype(np.int64))
    acc = em.wrap32(a_c @ wc[:n].T + biasc[None, :n].astype(np.int64))
    assert np.array_equal(acc, o1.gemm_acc(a, w, bias, izp, 127).astype(np.int64))


@pytest.mark.parametrize("case", SMALL_CONV, ids=lambda c: c.name)
def test_convolution_host_images(hooks, case):
    inp, kernel, bias = conv_tensors(case)
    expected, (oscale, ozp), (oh, ow) = conv_expected(case, inp, kernel, bias)
    rq = em.host_requant(hooks, np.float32(1.0) / oscale, ozp, case.qmin, case.qmax)
    depthwise = case.gic == 1 and case.goc == 1 and case.groups > 1
    if depthwise:
        out = em.emulate_dwconv(hooks, case, inp, kernel, bias, rq, oh, ow)
    elif case.groups == 1 and case.gic == 3 and not (
            case.kernel_size == (1, 1) and case.subsampling == (1, 1) and case.padding == (0, 0, 0, 0)):
        # 3-channel slot mode (first layers): one dword fetch per tap, 4-wide K slots
        offsets = em.host_offsets(hooks, case, oh, ow)
        taps = case.kernel_size[0] * case.kernel_size[1]
        rows = case.batch * oh * ow
        a_rows = em.gather_conv_rows(case, inp, offsets, 0, oh, ow)
        out = em.emulate_igemm_c3(hooks, case.goc, taps, case.izp, case.kzp,
                                  kernel.reshape(case.goc, taps * 3), bias, a_rows, rows, rq, case.out_stride)
    else:
        offsets = em.host_offsets(hooks, case, oh, ow)
        taps = case.kernel_size[0] * case.kernel_size[1]
        rows = case.batch * oh * ow
        out = em.emulate_igemm(
            hooks, case.groups, case.goc, case.gic, taps, case.izp, case.kzp,
            kernel.reshape(case.groups * case.goc, taps * case.gic), bias,
            lambda g: em.gather_conv_rows(case, inp, offsets, g, oh, ow), rows, rq, case.out_stride)
    assert_bytes_equal(out, expected, f"host images replay vs oracle [{case.name}]")


def test_offset_table_semantics(hooks):
    # src/indirection.c:56-60 index arithmetic: 3x3, pad 1, stride 2 on a 5x4 image with pixel stride 7
    from _cases import ConvCase
    case = ConvCase("t", (5, 4), (3, 3), (1, 1, 1, 1), subsampling=(2, 2), gic=7, goc=1)
    table = em.host_offsets(hooks, case, 3, 2)
    assert table.shape == (6, 9)
    # output (0,0): taps with ky=0 or kx=0 fall in the padding
    assert table[0].tolist() == [-1, -1, -1, -1, 0, 7, -1, 28, 35]
    # output (2,1): oy*2-1 = 3, ox*2-1 = 1 -> rows 3,4,(5 out), cols 1,2,3
    assert table[5].tolist() == [(3 * 4 + 1) * 7, (3 * 4 + 2) * 7, (3 * 4 + 3) * 7,
                                 (4 * 4 + 1) * 7, (4 * 4 + 2) * 7, (4 * 4 + 3) * 7, -1, -1, -1]


def test_device_requantization_arithmetic_matches_oracle(debug_hooks):
    """hip/requant_math.h (the single-shift form the kernels evaluate) compiled for the host."""
    import ctypes
    L = debug_hooks.lib
    L.qnnp_debug_requant_fast.restype = None
    L.qnnp_debug_requant_fast.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float, ctypes.c_uint8,
                                          ctypes.c_uint8, ctypes.c_uint8, ctypes.c_void_p]
    rng = np.random.default_rng(11)
    acc = rng.integers(-2**31, 2**31, size=1 << 20).astype(np.int32)
    acc[:8] = [-2**31, 2**31 - 1, 0, -1, 1, -2**30, 2**30, -2**31 + 1]
    # exact ties of the second rounding for a few shifts
    ties = []
    for s in range(1, 12):
        for k in range(-20, 21):
            ties += [(k << s) + (1 << (s - 1)) + d for d in (-1, 0, 1)]
    acc[8:8 + len(ties)] = np.array(ties, dtype=np.int64).astype(np.int32)
    for scale in [2.0 ** -32, 2.0 ** -31, 2.0 ** -12, 2.0 ** -5, 0.0031, 1 / 255.0, 0.25, 0.49999997, 0.5, 0.75,
                  float.fromhex("0x1.FFFFFEp-1")]:
        for zp, qmin, qmax in [(0, 0, 255), (127, 0, 255), (255, 0, 255), (100, 128, 255), (100, 0, 128), (7, 5, 9)]:
            out = np.empty(acc.size, np.uint8)
            L.qnnp_debug_requant_fast(acc.size, acc.ctypes.data, np.float32(scale), zp, qmin, qmax, out.ctypes.data)
            assert np.array_equal(out, o1.q31_requantize(acc, scale, zp, qmin, qmax)), (scale, zp, qmin, qmax)


def test_bounded_requantization_sequence_matches_oracle(debug_hooks):
    """hip/requant_math.h, qnnp_requant_scale_sn_bounded: the four-instruction rounding sequence the kernels use when
    the operator's accumulators are bounded at create time (|acc| < 2^bits, bits <= 30, 1 <= shift <= 20)."""
    import ctypes
    L = debug_hooks.lib
    fn = L.qnnp_debug_requant_fast_bits
    fn.restype = None
    fn.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float, ctypes.c_uint8, ctypes.c_uint8, ctypes.c_uint8,
                   ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(12)
    for bits in (30, 27, 17):
        lim = 2 ** bits
        acc = rng.integers(-lim + 1, lim, size=1 << 19).astype(np.int32)
        acc[:6] = [-lim + 1, lim - 1, 0, -1, 1, -(lim // 2)]
        ties = []
        for s in range(1, 21):
            for k in range(-12, 13):
                ties += [v for v in ((k << s) + (1 << (s - 1)) + d for d in (-1, 0, 1)) if -lim < v < lim]
        acc[6:6 + len(ties)] = np.array(ties, dtype=np.int64).astype(np.int32)
        # (shift 1..7 under the [0, 255] clamp answer through the packed tail, qnnp_requant_lane_sn_pk: carry-out sign, int16 pairs)
        for scale in [0.49999997, 0.25, 0.3, 0.12, 0.05, 0.02, 0.0125, 2.0 ** -7, 1 / 255.0, 0.0031, 2.0 ** -12, 1.7e-5, 2.0 ** -20, 1.9e-6]:
            for zp, qmin, qmax in [(0, 0, 255), (127, 0, 255), (255, 0, 255), (100, 128, 255), (7, 5, 9)]:
                out = np.empty(acc.size, np.uint8)
                bounded = ctypes.c_int(-1)
                fn(acc.size, acc.ctypes.data, np.float32(scale), zp, qmin, qmax, bits, out.ctypes.data, ctypes.byref(bounded))
                assert bounded.value == 1, (scale, bits)          # all of these qualify (shift 1..20)
                assert np.array_equal(out, o1.q31_requantize(acc, scale, zp, qmin, qmax)), (bits, scale, zp, qmin, qmax)
    # and it is refused where its preconditions fail: unknown / too wide a bound, shift 0, shift > 20
    acc = rng.integers(-2**31, 2**31, size=1 << 12).astype(np.int32)
    out = np.empty(acc.size, np.uint8)
    for scale, bits in [(0.25, 0), (0.25, 31), (0.75, 20), (2.0 ** -22, 20), (2.0 ** -31, 20)]:
        bounded = ctypes.c_int(-1)
        fn(acc.size, acc.ctypes.data, np.float32(scale), 9, 0, 255, bits, out.ctypes.data, ctypes.byref(bounded))
        assert bounded.value == 0, (scale, bits)
        assert np.array_equal(out, o1.q31_requantize(acc, scale, 9, 0, 255)), (scale, bits)


def test_offset_requantization_forms_match_oracle(debug_hooks):
    """hip/requant_math.h, qnnp_requant_scale_{s0,sn_bounded}_ofs: the unsigned multiply-add forms the streaming kernels
    evaluate on accumulators that carry + 2^31 (one instruction for shift 0, four for the bounded shift >= 1 form)."""
    import ctypes
    L = debug_hooks.lib
    fn = L.qnnp_debug_requant_fast_offset
    fn.restype = None
    fn.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float, ctypes.c_uint8, ctypes.c_uint8, ctypes.c_uint8,
                   ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(13)
    clamps = [(0, 0, 255), (127, 0, 255), (255, 0, 255), (100, 128, 255), (7, 5, 9)]
    # shift 0 (scales in [0.5, 1)): every accumulator, bound irrelevant
    acc = rng.integers(-2**31, 2**31, size=1 << 20).astype(np.int32)
    acc[:8] = [-2**31, 2**31 - 1, 0, -1, 1, -2**30, 2**30, -2**31 + 1]
    for scale in [0.5, 0.50000006, 0.6180339, 0.75, 0.99, float.fromhex("0x1.FFFFFCp-1"), float.fromhex("0x1.FFFFFEp-1")]:
        for zp, qmin, qmax in clamps:
            out = np.empty(acc.size, np.uint8)
            kind = ctypes.c_int(-1)
            fn(acc.size, acc.ctypes.data, np.float32(scale), zp, qmin, qmax, 0, out.ctypes.data, ctypes.byref(kind))
            assert kind.value == 1, scale
            assert np.array_equal(out, o1.q31_requantize(acc, scale, zp, qmin, qmax)), (scale, zp, qmin, qmax)
    # bounded shift >= 1
    for bits in (30, 27, 20, 12):
        lim = 2 ** bits
        acc = rng.integers(-lim + 1, lim, size=1 << 19).astype(np.int32)
        acc[:6] = [-lim + 1, lim - 1, 0, -1, 1, -(lim // 2)]
        ties = []
        for s in range(1, 21):
            for k in range(-12, 13):
                ties += [v for v in ((k << s) + (1 << (s - 1)) + d for d in (-1, 0, 1)) if -lim < v < lim]
        acc[6:6 + len(ties)] = np.array(ties, dtype=np.int64).astype(np.int32)
        for scale in [0.49999997, 0.25, 0.3, 1 / 255.0, 0.0031, 2.0 ** -12, 1.7e-5, 2.0 ** -20, 1.9e-6]:
            for zp, qmin, qmax in clamps:
                out = np.empty(acc.size, np.uint8)
                kind = ctypes.c_int(-1)
                fn(acc.size, acc.ctypes.data, np.float32(scale), zp, qmin, qmax, bits, out.ctypes.data, ctypes.byref(kind))
                assert kind.value == 2, (scale, bits)
                assert np.array_equal(out, o1.q31_requantize(acc, scale, zp, qmin, qmax)), (bits, scale, zp, qmin, qmax)
    # no offset form: shift >= 1 without a bound -> the general sequence answers, still exact
    acc = rng.integers(-2**31, 2**31, size=1 << 14).astype(np.int32)
    out = np.empty(acc.size, np.uint8)
    for scale, bits in [(0.25, 0), (0.25, 31), (2.0 ** -22, 20)]:
        kind = ctypes.c_int(-1)
        fn(acc.size, acc.ctypes.data, np.float32(scale), 9, 0, 255, bits, out.ctypes.data, ctypes.byref(kind))
        assert kind.value == 0, (scale, bits)
        assert np.array_equal(out, o1.q31_requantize(acc, scale, 9, 0, 255)), (scale, bits)


def test_lane_requantization_forms_match_oracle(debug_hooks):
    """hip/requant_math.h, qnnp_requant_lane_*: the forms in which the row term enters through a per-lane addend of the
    multiply-add and the sign of the rounding correction comes from the Q31 product -- no add per output value.
    acc = a + rowterm is what the oracle sees; the form sees a + 2^31 and the row term separately."""
    import ctypes
    L = debug_hooks.lib
    fn = L.qnnp_debug_requant_lane
    fn.restype = None
    fn.argtypes = [ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_uint8, ctypes.c_uint8,
                   ctypes.c_uint8, ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    rng = np.random.default_rng(14)
    clamps = [(0, 0, 255), (127, 0, 255), (255, 0, 255), (100, 128, 255), (7, 5, 9)]

    def rowterms(acc, lim):
        rt = rng.integers(-lim, lim, size=acc.size).astype(np.int64)
        rt[::7] = 0
        rt[1::11] = acc[1::11]                       # a == 0
        # keep a = acc - rowterm inside int32 (it is a bias plus a dot product)
        a = acc.astype(np.int64) - rt
        rt = np.where((a >= -2**31) & (a < 2**31), rt, 0)
        return rt.astype(np.int32)

    # shift 0: accumulators bounded at create time (a = acc - rowterm must fit int32 beside acc itself)
    for bits in (30, 24):
        lim = 2 ** bits
        acc = rng.integers(-lim + 1, lim, size=1 << 19).astype(np.int32)
        acc[:10] = [-lim + 1, lim - 1, 0, -1, 1, -(lim // 2), lim // 2, -lim + 2, -2, 2]
        rt = rowterms(acc, 2**28)
        for scale in [0.5, 0.50000006, 0.6180339, 0.75, 0.99, float.fromhex("0x1.FFFFFCp-1"), float.fromhex("0x1.FFFFFEp-1")]:
            for zp, qmin, qmax in clamps:
                out = np.empty(acc.size, np.uint8)
                kind = ctypes.c_int(-1)
                fn(acc.size, acc.ctypes.data, rt.ctypes.data, np.float32(scale), zp, qmin, qmax, bits, out.ctypes.data, ctypes.byref(kind))
                # (the single largest multiplier with a zero point that cannot be folded keeps the other sequence)
                assert kind.value in (0, 1) and (kind.value == 1 or scale == float.fromhex("0x1.FFFFFEp-1")), scale
                assert np.array_equal(out, o1.q31_requantize(acc, scale, zp, qmin, qmax)), (scale, zp, qmin, qmax)
    # shift 0 without a bound: not applicable (the offset form answers, for every int32 accumulator)
    acc = rng.integers(-2**31, 2**31, size=1 << 16).astype(np.int32)
    acc[:8] = [-2**31, 2**31 - 1, 0, -1, 1, -2**30, 2**30, -2**31 + 1]
    rt = rowterms(acc, 2**31)
    for scale in [0.5, 0.75]:
        out = np.empty(acc.size, np.uint8)
        kind = ctypes.c_int(-1)
        fn(acc.size, acc.ctypes.data, rt.ctypes.data, np.float32(scale), 0, 0, 255, 0, out.ctypes.data, ctypes.byref(kind))
        assert kind.value == 0
        assert np.array_equal(out, o1.q31_requantize(acc, scale, 0, 0, 255)), scale
    # bounded shift >= 1: ties of the second rounding on both sides of zero, and the n < 0 with q == 0 band
    for bits in (30, 27, 20, 12):
        lim = 2 ** bits
        acc = rng.integers(-lim + 1, lim, size=1 << 19).astype(np.int32)
        special = [-lim + 1, lim - 1, 0, -1, 1, -2, -3, 2, -(lim // 2)]
        for s in range(1, 21):
            for k in range(-12, 13):
                special += [v for v in ((k << s) + (1 << (s - 1)) + d for d in (-1, 0, 1)) if -lim < v < lim]
        acc[:len(special)] = np.array(special, dtype=np.int64).astype(np.int32)
        rt = rowterms(acc, lim)
        for scale in [0.49999997, 0.25, 0.3, 1 / 255.0, 0.0031, 2.0 ** -12, 1.7e-5, 2.0 ** -20, 1.9e-6]:
            for zp, qmin, qmax in clamps:
                out = np.empty(acc.size, np.uint8)
                kind = ctypes.c_int(-1)
                fn(acc.size, acc.ctypes.data, rt.ctypes.data, np.float32(scale), zp, qmin, qmax, bits, out.ctypes.data, ctypes.byref(kind))
                assert kind.value == 2, (scale, bits)
                assert np.array_equal(out, o1.q31_requantize(acc, scale, zp, qmin, qmax)), (bits, scale, zp, qmin, qmax)
    # exhaustive around zero for small multipliers' neighbourhoods: every n in [-70000, 70000]
    acc = np.arange(-70000, 70001, dtype=np.int32)
    rt = rowterms(acc, 2**20)
    for scale in [0.49999997, 0.4, 0.26, 0.25, 0.1, 1 / 255.0, 2.0 ** -9, 3.1e-4]:
        for zp in (0, 3, 128, 255):
            out = np.empty(acc.size, np.uint8)
            kind = ctypes.c_int(-1)
            fn(acc.size, acc.ctypes.data, rt.ctypes.data, np.float32(scale), zp, 0, 255, 24, out.ctypes.data, ctypes.byref(kind))
            assert kind.value == 2
            assert np.array_equal(out, o1.q31_requantize(acc, scale, zp, 0, 255)), (scale, zp)
    # not applicable: shift >= 1 without a bound
    acc = rng.integers(-2**31, 2**31, size=1 << 14).astype(np.int32)
    rt = rowterms(acc, 2**31)
    out = np.empty(acc.size, np.uint8)
    for scale, bits in [(0.25, 0), (0.25, 31), (2.0 ** -22, 20)]:
        kind = ctypes.c_int(-1)
        fn(acc.size, acc.ctypes.data, rt.ctypes.data, np.float32(scale), 9, 0, 255, bits, out.ctypes.data, ctypes.byref(kind))
        assert kind.value == 0, (scale, bits)
        assert np.array_equal(out, o1.q31_requantize(acc, scale, 9, 0, 255)), (scale, bits)


def test_accumulator_bound(debug_hooks):
    """requantization.h, qnnp_accumulator_bits: |bias| + K * 255^2 < 2^bits, 0 when it does not fit 31 bits."""
    import ctypes
    fn = debug_hooks.lib.qnnp_debug_accumulator_bits
    fn.restype = ctypes.c_uint32
    fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]

    def bits(bias, k):
        b = np.asarray(bias, dtype=np.int32)
        return fn(b.ctypes.data, b.size, k)

    for bias, k in [([0], 1), ([10000, -10000], 1280), ([-2**31], 9), ([2**30], 4096), ([5, -7, 3], 27), ([0], 33000)]:
        bound = max(abs(int(v)) for v in bias) + k * 65025 + 1
        expect = next(b for b in range(64) if (1 << b) >= bound)
        assert bits(bias, k) == (expect if expect <= 31 else 0), (bias, k)
    assert bits([10000], 1280) == 27 and bits([0], 9) == 20


@pytest.mark.parametrize("izp,kzp", [(127, 127), (0, 255), (255, 0), (3, 128), (128, 1)])
def test_depthwise_matrix_core_weight_parts(debug_hooks, izp, kzp):
    """qnnp_pack_dwconv_mfma (pack.h): int8 parts sum to w - kzp, the part count is minimal, and the folded bias
    makes  biasm + sum_t (a_t - 128) * x_t  ==  bias + sum_t (a_t - izp) * (w_t - kzp)  for every activation."""
    import ctypes
    L = debug_hooks.lib
    fn = L.qnnp_debug_pack_dwconv_mfma
    fn.restype = ctypes.c_uint32
    fn.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint8, ctypes.c_uint8,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(izp * 256 + kzp)
    C, KH, KW = 40, 3, 3
    taps, c_pad32 = KH * KW, 64
    kernel = rng.integers(0, 256, size=(C, taps), dtype=np.uint8)
    kernel[0, :3] = [0, 255, 128]                       # the extremes of x = w - kzp
    bias = rng.integers(-2**31, 2**31, size=C).astype(np.int32)
    xparts = np.full((3, taps, c_pad32), 99, np.int8)
    biasm = np.full(c_pad32, 99, np.int32)
    parts = fn(C, c_pad32, KH, KW, izp, kzp, kernel.ctypes.data, bias.ctypes.data, xparts.ctypes.data, biasm.ctypes.data)
    x = kernel.astype(np.int64).T - kzp                 # [taps][C]
    assert np.array_equal(xparts[:, :, :C].astype(np.int64).sum(axis=0), x)
    assert not xparts[:, :, C:].any() and not biasm[C:].any(), "padding channels must multiply to zero"
    need = 1 if x.min() >= -128 and x.max() <= 127 else (3 if x.max() == 255 else 2)
    assert parts == need, (parts, need)
    assert not xparts[parts:].any()
    a = rng.integers(0, 256, size=(taps, C)).astype(np.int64)
    lhs = (biasm[:C].astype(np.int64) + ((a - 128) * x).sum(axis=0)) & 0xFFFFFFFF
    rhs = (bias.astype(np.int64) + ((a - izp) * x).sum(axis=0)) & 0xFFFFFFFF
    assert np.array_equal(lhs, rhs)


@pytest.mark.parametrize("izp,kzp,lo,hi,expect", [
    (127, 127, 0, 255, 2), (3, 128, 0, 255, 1), (255, 100, 40, 200, 1), (0, 1, 0, 129, 2), (9, 1, 0, 130, 0),
    (127, 126, 0, 255, 0), (127, 129, 0, 255, 0), (200, 77, 77, 77, 1)])
def test_depthwise_dot4_register_image(debug_hooks, izp, kzp, lo, hi, expect):
    """qnnp_dwconv_weight_range / qnnp_pack_dwconv_dot4 (pack.h): the class is 1 when every w - kzp fits int8, 2 when
    every kzp - w does, else 0; for classes 1 / 2 the image makes
        image[3][c] + sum_rows dot4(int8(a ^ kx), image[r][c])  ==  bias + sum_t (a_t - izp) * (w_t - kzp)
    with kx = 0x80 (class 1) or 0x7f (class 2), for every activation -- the identity the kernel's walk relies on."""
    import ctypes
    L = debug_hooks.lib
    fn = L.qnnp_debug_pack_dwconv_dot4
    fn.restype = ctypes.c_uint32
    fn.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint8, ctypes.c_uint8] + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(izp * 256 + kzp + hi)
    C, c_pad = 24, 32
    kernel = rng.integers(lo, hi + 1, size=(C, 9)).astype(np.uint8)
    kernel[0, 0], kernel[C - 1, 8] = lo, hi
    bias = rng.integers(-2**31, 2**31, size=C).astype(np.int32)
    wadj = np.empty(9 * c_pad, np.int16); bias1 = np.empty(c_pad, np.int32)
    image = np.full(4 * c_pad, 0x5A5A5A5A, np.uint32)
    got = fn(C, c_pad, izp, kzp, kernel.ctypes.data, bias.ctypes.data, wadj.ctypes.data, bias1.ctypes.data, image.ctypes.data)
    assert got == expect
    if expect == 0:
        assert (image == 0x5A5A5A5A).all()
        return
    image = image.reshape(4, c_pad)
    w8 = image[:3].view(np.int8).reshape(3, c_pad, 4).astype(np.int64)       # [row][channel][col0, col1, col2, 0]
    assert not w8[:, :, 3].any()
    assert not image[:3, C:].any(), "padding channels multiply to zero"
    x = kernel.astype(np.int64) - kzp
    sign = 1 if expect == 1 else -1
    assert np.array_equal(w8[:, :C, :3].transpose(1, 0, 2).reshape(C, 9), sign * x)
    a = rng.integers(0, 256, size=(C, 9), dtype=np.uint8)
    a[0], a[1] = 0, 255
    kx = 0x80 if expect == 1 else 0x7F
    a8 = (a ^ kx).view(np.int8).astype(np.int64)
    lhs = (image[3, :C].astype(np.int64) + (a8 * (sign * x)).sum(axis=1)) & 0xFFFFFFFF
    rhs = (bias.astype(np.int64) + ((a.astype(np.int64) - izp) * x).sum(axis=1)) & 0xFFFFFFFF
    assert np.array_equal(lhs, rhs)


@pytest.mark.parametrize("izp,kzp,lo,hi,expect", [
    (127, 127, 0, 255, 2), (3, 128, 0, 255, 1), (255, 100, 40, 200, 1), (0, 1, 0, 129, 2), (9, 1, 0, 130, 0),
    (200, 77, 77, 77, 1)])
def test_depthwise_dot4_5x5_register_image(debug_hooks, izp, kzp, lo, hi, expect):
    """qnnp_pack_dwconv_dot4_5x5 (pack.h), the register image of the 5x5 column walk (q8dwconv.hip, kernel H): replay
    the walk's arithmetic for one output -- five quads (columns 0..3 of a kernel row), the sliding column-4 dword of kernel
    rows 0..3, the one-hot weight against the newest row's raw column-4 dword (whose other three bytes are OTHER
    channels' activations) -- and compare with bias + sum_t (a_t - izp) * (w_t - kzp)."""
    import ctypes
    L = debug_hooks.lib
    fn = L.qnnp_debug_pack_dwconv_dot4_5x5
    fn.restype = ctypes.c_uint32
    fn.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint8, ctypes.c_uint8] + [ctypes.c_void_p] * 5
    rng = np.random.default_rng(izp * 256 + kzp + hi + 5)
    C, c_pad = 24, 32
    kernel = rng.integers(lo, hi + 1, size=(C, 25)).astype(np.uint8)
    kernel[0, 0], kernel[C - 1, 24] = lo, hi
    bias = rng.integers(-2**31, 2**31, size=C).astype(np.int32)
    wadj = np.empty(25 * c_pad, np.int16); bias1 = np.empty(c_pad, np.int32)
    image = np.full(8 * c_pad, 0x5A5A5A5A, np.uint32)
    got = fn(C, c_pad, izp, kzp, kernel.ctypes.data, bias.ctypes.data, wadj.ctypes.data, bias1.ctypes.data, image.ctypes.data)
    assert got == expect
    if expect == 0:
        assert (image == 0x5A5A5A5A).all()
        return
    image = image.reshape(8, c_pad)
    assert not image[:7, C:].any(), "padding channels multiply to zero"
    w8 = image[:7].view(np.int8).reshape(7, c_pad, 4).astype(np.int64)
    sign = 1 if expect == 1 else -1
    x = (kernel.astype(np.int64) - kzp).reshape(C, 5, 5)
    assert np.array_equal(w8[:5, :C].transpose(1, 0, 2), sign * x[:, :, :4])               # quads
    assert np.array_equal(w8[5, :C], sign * x[:, :4, 4])                                    # column 4, kernel rows 0..3
    for c in range(C):
        onehot = np.zeros(4, np.int64); onehot[c & 3] = sign * x[c, 4, 4]
        assert np.array_equal(w8[6, c], onehot)
    # the walk on random activations: a[ky][kx][channel]
    a = rng.integers(0, 256, size=(5, 5, C), dtype=np.uint8)
    a[0, 0], a[4, 4] = 0, 255
    kx = 0x80 if expect == 1 else 0x7F
    a8 = (a ^ kx).view(np.int8).astype(np.int64)
    acc = image[7, :C].astype(np.int64).copy()
    for c in range(C):
        for ky in range(5):
            acc[c] += (a8[ky, :4, c] * w8[ky, c]).sum()                                    # T[ky].q[c] . WQ[ky][c]
        acc[c] += (a8[:4, 4, c] * w8[5, c]).sum()                                          # V[c] . WV[c]
        group = (c // 4) * 4
        acc[c] += (a8[4, 4, group:group + 4] * w8[6, c]).sum()                             # raw column-4 dword . W5[c]
    rhs = (bias.astype(np.int64) + ((a.astype(np.int64) - izp).transpose(2, 0, 1) * x).sum(axis=(1, 2))) & 0xFFFFFFFF
    assert np.array_equal(acc & 0xFFFFFFFF, rhs)


@pytest.mark.parametrize("kzp", [127, 128, 100])
@pytest.mark.parametrize("shape", [((6, 7), 8, (1, 1, 1, 1)), ((5, 5), 20, (0, 0, 0, 0)), ((4, 9), 12, (2, 1, 0, 2))],
                         ids=["6x7c8", "5x5c20_nopad", "4x9c12_asym"])
def test_depthwise_dot4_walk_replay(hooks, shape, kzp):
    """the arithmetic of the int8 dot-product depthwise walk, replayed from the host image, equals the oracle --
    padding rows and columns included; kzp 100 with full-range weights has no int8 form and must say so"""
    import dataclasses
    from _cases import ConvCase
    (hw, c, padding) = shape
    case = ConvCase(f"dot4_{hw[0]}x{hw[1]}_c{c}_k{kzp}", hw, (3, 3), padding, groups=c, gic=1, goc=1, batch=2, izp=93, kzp=kzp)
    inp, kernel, bias = conv_tensors(case)
    kernel[0, 0, 0, 0, 0], kernel[-1, 0, 2, 2, 0] = 0, 255
    expected, (oscale, ozp), (oh, ow) = conv_expected(case, inp, kernel, bias)
    rq = em.host_requant(hooks, np.float32(1.0) / oscale, ozp, case.qmin, case.qmax)
    out = em.emulate_dwconv_dot4(hooks, case, inp, kernel, bias, rq, oh, ow)
    if kzp == 100:
        assert out is None
    else:
        assert_bytes_equal(out, expected, f"dot-product walk replay vs oracle [{case.name}]")


@pytest.mark.parametrize("kzps", [(127, 127, 127), (128, 128, 128), (127, 128, 127)], ids=lambda k: "kzp" + "_".join(map(str, k)))
@pytest.mark.parametrize("cin,ch,cout,stride", [(16, 96, 24, 2), (24, 144, 24, 1), (64, 384, 64, 1)])
def test_fused_strip_images_replay_the_oracle(hooks, debug_hooks, kzps, cin, ch, cout, stride):
    """pack.h qnnp_strip_pointwise_images / qnnp_strip_depthwise_images (what fused-block.c derives from the stand-alone
    operators' device images) + the arithmetic of hip/q8fusedstrip.hip, replayed in numpy for one small image: expand,
    depthwise and project accumulators -- operands recentred with the stage's flip, padding taps = (input zero point ^ flip),
    no row terms -- must equal the oracle's of the three stand-alone operators, every intermediate requantized by the oracle."""
    import ctypes
    L = debug_hooks.lib
    L.qnnp_debug_strip_pointwise_images.restype = None
    L.qnnp_debug_strip_pointwise_images.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_uint32] * 3 + \
        [ctypes.c_uint8, ctypes.c_uint8, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.qnnp_debug_strip_depthwise_images.restype = None
    L.qnnp_debug_strip_depthwise_images.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_uint32] * 3 + \
        [ctypes.c_uint8, ctypes.c_uint8, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(cin * 1000 + ch + stride)
    H = W = 6
    OH = OW = (H + 2 - 3) // stride + 1
    izp1, izp2, izp3 = 3, 200, 127
    x = rng.integers(0, 256, size=(H * W, cin), dtype=np.uint8)
    w1 = rng.integers(0, 256, size=(ch, cin), dtype=np.uint8); b1 = rng.integers(-5000, 5000, size=ch).astype(np.int32)
    w2 = rng.integers(0, 256, size=(ch, 3, 3), dtype=np.uint8); b2 = rng.integers(-5000, 5000, size=ch).astype(np.int32)
    w3 = rng.integers(0, 256, size=(cout, ch), dtype=np.uint8); b3 = rng.integers(-5000, 5000, size=cout).astype(np.int32)
    w1[0, :] = 255; w1[1, :] = 0; w2[0] = 255; w2[1] = 0; w3[0, :] = 0; w3[1, :] = 255; x[0, :] = 255; x[1, :] = 0

    def flip(k): return 0x80 if k == 128 else 0x7F
    def s8(a, f): return (a ^ f).astype(np.uint8).view(np.int8).astype(np.int64)

    def pointwise(a_u8, w, b, izp, kzp, ofs):
        n, k = w.shape
        packed, bias2, n_pad, k_pad = em.host_pack_igemm(hooks, 1, n, k, izp, kzp, w.reshape(1, n, k), b)
        nb, kb = (n + 31) // 32, (k + 31) // 32
        frags = np.empty(nb * kb * 1024, np.int8); biasc = np.empty(nb * 32, np.int32)
        L.qnnp_debug_strip_pointwise_images(packed.ctypes.data, bias2.ctypes.data, k_pad, n, k, izp, kzp, ofs,
                                            frags.ctypes.data, biasc.ctypes.data)
        wc = em.unpack_fragments(frags, 1, nb * 32, kb * 32)[0]               # what the MFMA sees, [n_pad][kb * 32]
        assert np.array_equal(wc[:n, :k], s8(w, flip(kzp))) and not wc[n:].any() and not wc[:, k:].any()
        acc = em.wrap32(s8(a_u8, flip(kzp)) @ wc[:n, :k].T + biasc[None, :n].astype(np.int64) - (ofs << 31))
        assert np.array_equal(acc, o1.gemm_acc(a_u8, w, b, izp, kzp).astype(np.int64))
        return acc

    kz1, kz2, kz3 = kzps
    acc1 = pointwise(x, w1, b1, izp1, kz1, 1)
    hid = o1.requantize_rows(acc1.astype(np.int32), np.float32(0.002), izp2, 0, 255).reshape(H, W, ch)
    # depthwise: padded hidden image holding (pixel ^ flip2), padding cells = izp2 ^ flip2
    c_pad = em.round_up(ch, 16)
    wadj = np.zeros(9 * c_pad, np.int16); bias1 = np.zeros(c_pad, np.int32)
    hooks.qnnp_debug_pack_dwconv_w(ch, c_pad, 3, 3, izp2, kz2, np.ascontiguousarray(w2).ctypes.data, b2.ctypes.data,
                                   wadj.ctypes.data, bias1.ctypes.data)
    hp = em.round_up(ch, 32)
    w2c = np.empty(9 * hp, np.int8); biasc2 = np.empty(hp, np.int32)
    L.qnnp_debug_strip_depthwise_images(wadj.ctypes.data, bias1.ctypes.data, c_pad, ch, hp, izp2, kz2, 0,
                                        w2c.ctypes.data, biasc2.ctypes.data)
    w2c = w2c.reshape(9, hp).astype(np.int64)
    assert np.array_equal(w2c[:, :ch], s8(w2.reshape(ch, 9).T.copy(), flip(kz2))) and not w2c[:, ch:].any()
    padded = np.full((H + 2, W + 2, ch), izp2, np.uint8)
    padded[1:-1, 1:-1] = hid
    pc = s8(padded, flip(kz2))
    acc2 = np.zeros((OH, OW, ch), np.int64)
    for oy in range(OH):
        for ox in range(OW):
            for t in range(9):
                acc2[oy, ox] += pc[oy * stride + t // 3, ox * stride + t % 3] * w2c[t, :ch]
    acc2 = em.wrap32(acc2 + biasc2[None, None, :ch].astype(np.int64)).reshape(OH * OW, ch)
    shape = o1.conv_shape(1, H, W, (1, 1, 1, 1), (3, 3), (stride, stride), (1, 1), ch, 1, 1, ch)
    want2 = o1.conv2d_acc(shape, hid.reshape(-1), w2.reshape(ch, 1, 3, 3, 1), b2, izp2, kz2).reshape(OH * OW, ch)
    assert np.array_equal(acc2, want2.astype(np.int64))
    dwo = o1.requantize_rows(want2.astype(np.int32), np.float32(0.001), izp3, 0, 255)
    pointwise(dwo, w3, b3, izp3, kz3, 0)
