"""GPU tier: the packed shift >= 1 tail of the lane requantization (hip/requant.hip.h kRqBoundedLanePk; host model
hip/requant_math.h qnnp_requant_lane_sn_pk, held to the oracle in tests/test_host_logic.py) through the kernels that
instantiate it: streaming pointwise, wave-per-block 3x3, LDS-patch 3x3, the two 3-channel first-layer kernels. The sign
of the rounding correction comes from the multiply-add's carry out and the second shift works on saturated int16 pairs,
so the cases that matter are ties of the second rounding on both sides of zero (one value in 2^shift is one) next to
the clamp at both ends: every shift 1..7 (and 8, which keeps the 32-bit tail), zero points 0 / 127 / 255, accumulators
spread over +-2^shift * 400 by the choice of the kernel/input zero points. Bit-exact against the scalar oracle
(reference semantics: src/qnnpack/requantization.h:464-480)."""
import numpy as np
import pytest

import bench
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal
from oracle import o1

pytestmark = pytest.mark.gpu

#        name              H   W   KH KW S  G  GIC  GOC  batch  kernel expected
CASES = [("pw_16_96",      56, 56, 1, 1, 1, 1, 16,  96,  4,  "q8_pw_stream_mfma"),
         ("pw_96_24",      28, 28, 1, 1, 1, 1, 96,  24,  8,  "q8_pw_stream_mfma"),
         ("ws3x3_64",      56, 56, 3, 3, 1, 1, 64,  64,  8,  "q8_conv_wave_ws"),
         ("patch3x3_128",  28, 28, 3, 3, 1, 1, 128, 128, 8,  "q8_conv_patch_mfma"),
         ("c3rows_3x3s2",  64, 64, 3, 3, 2, 1, 3,   32,  4,  "q8_conv_c3rows_lds_mfma"),
         ("c3rows32_7x7",  64, 64, 7, 7, 2, 1, 3,   64,  4,  "q8_conv_c3rows32_lds_mfma")]
# requantization scale -> shift: 0.3 -> 1, 0.12 -> 3, 0.05 -> 4, 0.02 -> 5, 0.0125 -> 6, 0.006 -> 7, 0.0031 -> 8
SCALES = [0.3, 0.12, 0.05, 0.02, 0.0125, 0.006, 0.0031]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c[0])
def test_packed_tail_matches_oracle(qnnp, case):
    name, H, W, KH, KW, S, G, GIC, GOC, batch, kname_want = case
    (pt, pr, pb, pl), oh, ow = bench.conv_geometry(H, W, KH, KW, S, 1)
    rng = np.random.default_rng(abs(hash(name)) % (1 << 31))
    cin, cout = G * GIC, G * GOC
    # inputs and weights close to their zero points: accumulators of a few thousand, so that scales down to 0.003 leave
    # outputs inside 0..255 around every zero point instead of saturating
    spread = np.sqrt(187.0 / np.sqrt(KH * KW * GIC))       # accumulator standard deviation ~ 3000 whatever the reduction length
    ak, ax = max(1, int(6 * spread)), max(1, int(8 * spread))
    kernel = rng.integers(127 - ak, 127 + ak + 1, size=(G, GOC, KH, KW, GIC)).astype(np.uint8)
    inp = rng.integers(127 - ax, 127 + ax + 1, size=batch * H * W * cin).astype(np.uint8)
    bias = rng.integers(-3000, 3001, size=cout, dtype=np.int32)
    oshape = o1.conv_shape(batch, H, W, (pt, pr, pb, pl), (KH, KW), (S, S), (1, 1), G, GIC, GOC, cin)
    o1.set_threads(16)
    try:
        acc = o1.conv2d_acc(oshape, inp, kernel, bias, 127, 127).reshape(-1, cout)
    finally:
        o1.set_threads(1)
    d_in = to_device(inp)
    seen = set()
    for scale in SCALES:
        for ozp in (0, 127, 255):
            out_scale = 0.25 / scale
            req = np.float32(np.float32(0.5) * np.float32(0.5) / np.float32(out_scale))
            expected = o1.requantize_rows(acc, req, ozp, 0, 255).reshape(-1)
            if ozp == 127 and scale <= 0.05:
                inside = float(np.mean((expected > 0) & (expected < 255)))
                assert inside > 0.25, (name, scale, inside)      # (not a saturated image: the rounding decides the bytes)
            op = qnnp.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, 1, 1, G, GIC, GOC,
                                                   127, 0.5, 127, 0.5, kernel, bias, ozp, float(out_scale), 0, 255, 0)
            try:
                d_out = to_device(np.full(expected.size, FILL, np.uint8))
                qnnp.setup_convolution2d_nhwc_q8(op, batch, H, W, d_in, cin, d_out, cout)
                qnnp.run_operator(op)
                seen.add(qnnp.operator_kernel(op))
                out = from_device(d_out)
            finally:
                qnnp.delete_operator(op)
            assert_bytes_equal(out, expected, f"{name} scale {scale} zero point {ozp}")
    assert all(k.startswith(kname_want) for k in seen), (name, seen)


@pytest.mark.parametrize("kzp,kname_want,code", [(126, "q8_gemm_mfma_256x256_lean", 15), (126, "q8_gemm_mfma_256x256_r16", 0),
                                                 (127, "q8_gemm_mfma_128x128_c16", 0)])
def test_packed_tail_in_the_gemm_epilogues(qnnp, kzp, kname_want, code):
    """the lean 256 x 256 GEMM (kernel zero point without a centred image) takes the packed tail in its lane-form epilogue; the
    centred one keeps the bounded offset form -- both against the oracle at the same scales and zero points"""
    M, K, N = 2048, 1024, 256
    rng = np.random.default_rng(77 + kzp)
    kernel = rng.integers(kzp - 9, kzp + 10, size=(N, K)).astype(np.uint8)
    inp = rng.integers(127 - 10, 127 + 11, size=M * K).astype(np.uint8)
    bias = rng.integers(-3000, 3001, size=N, dtype=np.int32)
    acc = o1.gemm_acc(inp.reshape(M, K), kernel, bias, 127, kzp)
    d_in = to_device(inp)
    for scale in SCALES:
        for ozp in (0, 127, 255):
            req = np.float32(scale)
            expected = o1.requantize_rows(acc, req, ozp, 0, 255).reshape(-1)
            if ozp == 127 and scale <= 0.05:
                assert float(np.mean((expected > 0) & (expected < 255))) > 0.25, (scale,)
            qnnp.set_option("gemm_kernel", code)      # (15: the lean 32x32x32 kernel, what auto took before round 6)
            try:
                op = qnnp.create_fully_connected_nc_q8(K, N, 127, 1.0, kzp, float(scale), kernel, bias, ozp, 1.0, 0, 255, 0)
                try:
                    d_out = to_device(np.full(M * N, FILL, np.uint8))
                    qnnp.setup_fully_connected_nc_q8(op, M, d_in, K, d_out, N)
                    qnnp.run_operator(op)
                    kname = qnnp.operator_kernel(op)
                    out = from_device(d_out)
                finally:
                    qnnp.delete_operator(op)
            finally:
                qnnp.set_option("gemm_kernel", 0)
            assert kname == kname_want, kname
            assert_bytes_equal(out, expected, f"4096-style GEMM, kernel zero point {kzp}, scale {scale}, zero point {ozp}")
