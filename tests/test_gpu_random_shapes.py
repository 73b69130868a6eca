"""GPU tier: seeded RANDOM shapes through the kernels added in rounds 3 and 4, against the scalar oracle (the fused block:
against its stand-alone operators), bit for bit. The
hand-picked matrices (test_gpu_dwcol5, test_gpu_dwcol, test_gpu_convc3rows, test_gpu_deconvolution, test_gpu_residual)
aim at the edges their authors thought of; this file draws image sizes, channel counts, strides, paddings, pixel
strides, batch, zero points and clamps at random inside each kernel's range (the seed is the case id, so a failure names
a reproducible case) and asserts the kernel that ran, so a draw that silently left the kernel under test fails too."""
import dataclasses

import numpy as np
import pytest

from _cases import ConvCase, DeconvCase, conv_tensors, deconv_tensors
from _gpu import from_device, to_device
from _runner import assert_bytes_equal, conv_expected, conv_run, deconv_expected, deconv_run

pytestmark = pytest.mark.gpu

N = 24   # draws per kernel


def _common(rng, cin, cout):
    kw = {}
    if rng.random() < 0.3:
        kw["input_pixel_stride"] = cin + 4 * int(rng.integers(1, 5))          # (multiples of 4 keep dword alignment)
    if rng.random() < 0.3:
        kw["output_pixel_stride"] = cout + 4 * int(rng.integers(1, 5))
    if rng.random() < 0.25:
        kw["qmin"], kw["qmax"] = int(rng.integers(0, 100)), int(rng.integers(156, 256))
    kw["izp"] = int(rng.choice([0, 255, 127, int(rng.integers(0, 256))]))
    return kw


@pytest.mark.parametrize("seed", range(N))
def test_random_depthwise_3x3_kernel_g(qnnp, seed):
    rng = np.random.default_rng(0x3300 + seed)
    s = int(rng.choice([1, 2]))
    c = 4 * int(rng.integers(1, 70))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    pad = tuple(int(x) for x in rng.integers(0, 3, size=4))
    if (h + pad[0] + pad[2] < 3) or (w + pad[1] + pad[3] < 3):
        pad = (1, 1, 1, 1)
        h, w = max(h, 1), max(w, 1)
    case = ConvCase(f"rand_g_{seed}", (h, w), (3, 3), pad, subsampling=(s, s), groups=c, gic=1, goc=1,
                    batch=int(rng.integers(1, 5)), kzp=int(rng.choice([127, 128, 100])), **_common(rng, c, c))
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname.startswith("q8_dwconv_col_3x3"), (kname, case)
    assert_bytes_equal(out, expected, f"{kname} [{case}]")


@pytest.mark.parametrize("seed", range(N))
def test_random_depthwise_5x5_kernel_h(qnnp, seed):
    rng = np.random.default_rng(0x5500 + seed)
    s = int(rng.choice([1, 2]))
    c = 4 * int(rng.integers(1, 50))
    h, w = int(rng.integers(1, 36)), int(rng.integers(1, 36))
    pad = tuple(int(x) for x in rng.integers(0, 5, size=4))
    if (h + pad[0] + pad[2] < 5) or (w + pad[1] + pad[3] < 5):
        pad = (2, 2, 2, 2)
    case = ConvCase(f"rand_h_{seed}", (h, w), (5, 5), pad, subsampling=(s, s), groups=c, gic=1, goc=1,
                    batch=int(rng.integers(1, 4)), kzp=int(rng.choice([127, 128])), **_common(rng, c, c))
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == "q8_dwconv_col_5x5_dot4", (kname, case)
    assert_bytes_equal(out, expected, f"{kname} [{case}]")


@pytest.mark.parametrize("seed", range(N))
def test_random_first_layer_c3rows(qnnp, seed):
    rng = np.random.default_rng(0xC300 + seed)
    kh, kw = [(3, 3), (3, 3), (4, 5), (4, 4), (3, 5), (2, 3)][int(rng.integers(0, 6))]   # <= 4 kernel rows, <= 5 columns
    s = int(rng.choice([1, 2]))
    h, w = int(rng.integers(kh, 70)), int(rng.integers(kw, 70))
    pad = (int(rng.integers(0, kh // 2 + 1)), int(rng.integers(0, kw // 2 + 1)), int(rng.integers(0, kh // 2 + 1)), int(rng.integers(0, kw // 2 + 1)))
    cout = 16 * int(rng.integers(1, 5))
    batch = int(rng.integers(1, 4))
    # (the kernel is chosen from 2048 output pixels on; smaller draws are scaled up through the batch)
    oh = (h + pad[0] + pad[2] - kh) // s + 1
    ow = (w + pad[1] + pad[3] - kw) // s + 1
    while batch * oh * ow < 2048:
        batch += 1
    kwargs = _common(rng, 3, cout)
    kwargs.pop("input_pixel_stride", None)               # dense 3-byte pixels are what the kernel is for
    if "output_pixel_stride" in kwargs:
        kwargs["output_pixel_stride"] = cout + 16 * int(rng.integers(1, 3))
    case = ConvCase(f"rand_c3_{seed}", (h, w), (kh, kw), pad, subsampling=(s, s), gic=3, goc=cout, batch=batch,
                    kzp=int(rng.choice([127, 128, 3, 250])), **kwargs)
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    # (round 6: image rows of whole 16-byte chunks without a row term -- kernel zero point 127 / 128 -- take the LDS-staged flavour)
    lds = w % 16 == 0 and case.kzp in (127, 128)
    assert kname == ("q8_conv_c3rows_lds_mfma" if lds else "q8_conv_c3rows_mfma"), (kname, case)
    assert_bytes_equal(out, expected, f"{kname} [{case}]")


@pytest.mark.parametrize("seed", range(N))
def test_random_deconvolution_stride2_stream(qnnp, seed):
    rng = np.random.default_rng(0xDE00 + seed)
    k = int(rng.choice([3, 4]))
    taps = 9 if k == 3 else 16
    for _ in range(64):                                  # draw until the four sub-kernels fit the kernel's 64 KiB of LDS
        cin = 32 * int(rng.integers(1, 5))
        cout = int(rng.choice([8, 16, 19, 20, 32, 48, 64]))
        n_pad = (cout + 31) // 32 * 32
        if taps * cin * n_pad + 16 * n_pad <= 64 * 1024:
            break
    else:
        cin, cout = 32, 16
    h, w = int(rng.integers(1, 20)), int(rng.integers(1, 20))
    pad = tuple(int(x) for x in rng.integers(0, 3, size=4))
    adj = (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
    if 2 * (h - 1) + adj[0] + k <= pad[0] + pad[2] or 2 * (w - 1) + adj[1] + k <= pad[1] + pad[3]:
        pad = (0, 0, 0, 0)
    kwargs = _common(rng, cin, cout)
    if "input_pixel_stride" in kwargs:
        kwargs["input_pixel_stride"] = cin + 16 * int(rng.integers(1, 3))          # rows stay 16-byte aligned
    case = DeconvCase(f"rand_ds_{seed}", (h, w), (k, k), pad, subsampling=(2, 2), gic=cin, goc=cout,
                      batch=int(rng.integers(1, 4)), adjustment=adj, kzp=int(rng.choice([127, 128, 7, 249])), **kwargs)
    expected, quant, out_hw = deconv_expected(case)
    out, kname = deconv_run(qnnp, case, quant, out_hw, to_device=to_device, from_device=from_device)
    assert kname.startswith("q8_deconv_s2_stream"), (kname, case)
    assert_bytes_equal(out, expected, f"{kname} [{case}]")


# ---- round 4 ----
@pytest.mark.parametrize("seed", range(N))
def test_random_dilated_depthwise_walk(qnnp, seed):
    """kernel G's residue-class form (tests/test_gpu_dwcol_dilated.py holds the hand-picked edges)"""
    rng = np.random.default_rng(0xD100 + seed)
    dh, dw = int(rng.integers(1, 7)), int(rng.integers(1, 7))
    if dh == 1 and dw == 1:
        dh = 2
    c = 4 * int(rng.integers(1, 60))
    h, w = int(rng.integers(1, 45)), int(rng.integers(1, 45))
    pad = (int(rng.integers(0, 2 * dh + 1)), int(rng.integers(0, 2 * dw + 1)), int(rng.integers(0, 2 * dh + 1)), int(rng.integers(0, 2 * dw + 1)))
    if h + pad[0] + pad[2] < 2 * dh + 1 or w + pad[1] + pad[3] < 2 * dw + 1:      # at least one window
        pad = (dh, dw, dh, dw)
    case = ConvCase(f"rand_dil_{seed}", (h, w), (3, 3), pad, dilation=(dh, dw), groups=c, gic=1, goc=1,
                    batch=int(rng.integers(1, 5)), kzp=int(rng.choice([127, 128])), **_common(rng, c, c))
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    assert kname == "q8_dwconv_col_3x3_dot4_dilated", (kname, case)
    assert_bytes_equal(out, expected, f"{kname} [{case}]")


@pytest.mark.parametrize("seed", range(N))
def test_random_3x3_convolution_weight_stationary(qnnp, seed):
    """the weight-stationary 3x3 kernel, centred (kernel zero point 127 / 128) and with pixel sums (any other)"""
    rng = np.random.default_rng(0x3C00 + seed)
    cin, cout = int(rng.choice([32, 64])), int(rng.choice([32, 64]))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    pad = tuple(int(x) for x in rng.integers(0, 3, size=4))
    if h + pad[0] + pad[2] < 3 or w + pad[1] + pad[3] < 3:
        pad = (1, 1, 1, 1)
    kzp = int(rng.choice([127, 128, int(rng.integers(0, 256))]))
    kw = _common(rng, cin, cout)
    kw.pop("output_pixel_stride", None)                       # (the wave kernels want dense output pixels)
    if "input_pixel_stride" in kw:
        kw["input_pixel_stride"] = cin + 16 * int(rng.integers(1, 3))           # 16-byte aligned pixels
    case = ConvCase(f"rand_ws_{seed}", (h, w), (3, 3), pad, gic=cin, goc=cout, batch=int(rng.integers(1, 6)), kzp=kzp, **kw)
    inp, kernel, bias = conv_tensors(case)
    expected, quant, out_hw = conv_expected(case, inp, kernel, bias)
    qnnp.set_option("gemm_kernel", 8)
    try:
        out, kname = conv_run(qnnp, case, quant, out_hw, inp, kernel, bias, to_device, from_device)
    finally:
        qnnp.set_option("gemm_kernel", 0)
    centred = "q8_conv_wave_ws_c16_mfma" if cin == 64 else "q8_conv_wave_ws_c_mfma"      # (64 channels in: the 16x16x64 flavour)
    assert kname == (centred if kzp in (127, 128) else "q8_conv_wave_ws_mfma"), (kname, case)
    assert_bytes_equal(out, expected, f"{kname} [{case}]")


@pytest.mark.parametrize("seed", range(N))
def test_random_fused_block_strip_kernel(qnnp, seed):
    """one inverted-residual block on the strip kernel against its three (four) stand-alone operators run in sequence:
    random image sizes, channel counts, stride, residual, strip height, zero points 127 / 128 per member, batch"""
    import torch
    rng = np.random.default_rng(0xF500 + seed)
    stride = int(rng.choice([1, 2]))
    cin = 4 * int(rng.integers(2, 41))                          # <= 160
    hidden = 4 * int(rng.integers(2, 120))
    has_res = stride == 1 and rng.random() < 0.5
    cout = cin if has_res else 4 * int(rng.integers(2, 60))
    h, w = int(rng.integers(3, 30)), int(rng.integers(3, 30))
    batch = int(rng.integers(1, 4))
    oh, ow = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    zp = lambda: int(rng.choice([127, 128]))
    k1 = rng.integers(0, 256, size=(1, hidden, 1, 1, cin), dtype=np.uint8)
    kd = rng.integers(0, 256, size=(hidden, 1, 3, 3, 1), dtype=np.uint8)
    k3 = rng.integers(0, 256, size=(1, cout, 1, 1, hidden), dtype=np.uint8)
    b1 = rng.integers(-5000, 5000, size=hidden, dtype=np.int32)
    bd = rng.integers(-5000, 5000, size=hidden, dtype=np.int32)
    b3 = rng.integers(-5000, 5000, size=cout, dtype=np.int32)
    s1, sd, s3 = float(2.0 ** -int(rng.integers(8, 12))), float(2.0 ** -int(rng.integers(6, 10))), float(2.0 ** -int(rng.integers(9, 13)))
    z = [int(rng.integers(0, 256)) for _ in range(5)]           # tensor zero points: input, hidden, dw out, project out, sum
    ex = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, cin, hidden, z[0], 1.0, zp(), 1.0, k1, b1, z[1], 1.0 / s1, 0, 255, 0)
    dw = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, stride, stride, 1, 1, hidden, 1, 1, z[1], 1.0, zp(), 1.0, kd, bd, z[2], 1.0 / sd, 0, 255, 0)
    pr = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, hidden, cout, z[2], 1.0, zp(), 1.0, k3, b3, z[3], 1.0 / s3, 0, 255, 0)
    add = qnnp.create_add_nc_q8(cout, z[0], 1.0, z[3], 0.75, z[4], 1.25, 0, 255, 0) if has_res else None
    fused = None
    try:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        x = torch.randint(0, 256, (batch * h * w * cin,), dtype=torch.uint8, device="cuda", generator=gen)
        t1 = torch.empty(batch * h * w * hidden, dtype=torch.uint8, device="cuda")
        t2 = torch.empty(batch * oh * ow * hidden, dtype=torch.uint8, device="cuda")
        t3 = torch.empty(batch * oh * ow * cout, dtype=torch.uint8, device="cuda")
        want = torch.empty(batch * oh * ow * cout, dtype=torch.uint8, device="cuda")
        got = torch.full((batch * oh * ow * cout,), 0xA5, dtype=torch.uint8, device="cuda")
        qnnp.setup_convolution2d_nhwc_q8(ex, batch, h, w, x, cin, t1, hidden)
        qnnp.setup_convolution2d_nhwc_q8(dw, batch, h, w, t1, hidden, t2, hidden)
        qnnp.setup_convolution2d_nhwc_q8(pr, batch, oh, ow, t2, hidden, t3 if has_res else want, cout)
        for op in (ex, dw, pr):
            qnnp.run_operator(op)
        if has_res:
            qnnp.setup_add_nc_q8(add, batch * oh * ow, x, cin, t3, cout, want, cout)
            qnnp.run_operator(add)
        qnnp.set_option("fused_kernel", 2)                      # the strip kernel or nothing
        qnnp.set_option("fused_rows", int(rng.choice([0, 0, 1, 2, 5])))
        try:
            fused = qnnp.create_fused_block(ex, dw, pr, add)
            qnnp.setup_fused_block(fused, batch, h, w, x, cin, got, cout)
        finally:
            qnnp.set_option("fused_kernel", 0)
            qnnp.set_option("fused_rows", 0)
        qnnp.run_operator(fused)
        torch.cuda.synchronize()
        assert qnnp.operator_kernel(fused) == "q8_fused_strip"
        assert torch.equal(got, want), f"strip kernel differs from the stand-alone chain: {h}x{w}x{cin} -> {hidden} -> {cout}, stride {stride}, residual {has_res}, batch {batch}"
    finally:
        for op in (fused, ex, dw, pr, add):
            if op is not None:
                qnnp.delete_operator(op)
