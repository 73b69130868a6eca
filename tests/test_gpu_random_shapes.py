"""GPU tier: seeded RANDOM shapes through the kernels added in round 3, against the scalar oracle, bit for bit. The
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
    assert kname == "q8_conv_c3rows_mfma", (kname, case)
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
