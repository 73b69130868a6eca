#!/usr/bin/env python3
"""Generate tests/golden/reference_deconv_outputs.npz by running the COMPILED REFERENCE
(oracle/_ref/libqnnpack_ref.so) through qnnp_create/setup_deconvolution2d_nhwc_q8 on the seeded cases of
tests/_cases.py (the reference's own test/deconvolution.cc list + the extras named below).

Run in the build container:  python tests/golden/generate_golden_deconv.py
Per case the fixture stores the exact input / kernel / bias bytes, the output quantization and the reference's
output bytes. The quantization is derived from int64 accumulators computed here with numpy (scatter form of the
transposed convolution: every input pixel adds its kernel-weighted patch into the full output, which is then
cropped by the padding) -- independent of the oracle under test, which uses the gather form.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _cases import DECONV_CASES, EXTRA_DECONV_CASES, deconv_tensors, output_quantization, strided_view  # noqa: E402
from oracle import ref  # noqa: E402

FILL = 0xA5
GOLDEN_EXTRA = {"dx_3x3s2_adjust", "dx_2x2s2_c64_n32", "dx_4x4s2p1_c32_n16", "dx_3x3s2_zp_0_255",
                "dx_3x3s2_zp_255_0", "dx_3x3_asym_pad", "dx_grouped_2x2s2", "dx_1x1s2"}


def numpy_deconv_acc(case, inp, kernel, bias):
    H, W = case.input_size
    G, GIC, GOC = case.groups, case.gic, case.goc
    KH, KW = case.kernel_size
    pt, pr, pb, pl = case.padding
    sh, sw = case.subsampling
    dh, dw = case.dilation
    full_h = sh * (H - 1) + case.adjustment[0] + (KH - 1) * dh + 1
    full_w = sw * (W - 1) + case.adjustment[1] + (KW - 1) * dw + 1
    x = strided_view(inp, case.batch * H * W, G * GIC, case.in_stride).astype(np.int64)
    x = x.reshape(case.batch, H, W, G, GIC) - case.izp
    w = kernel.astype(np.int64) - case.kzp          # [G, GIC, KH, KW, GOC]
    full = np.zeros((case.batch, full_h, full_w, G, GOC), dtype=np.int64)
    for ky in range(KH):
        for kx in range(KW):
            contrib = np.einsum("nyxgi,gio->nyxgo", x, w[:, :, ky, kx, :])
            full[:, ky * dh:ky * dh + sh * (H - 1) + 1:sh, kx * dw:kx * dw + sw * (W - 1) + 1:sw] += contrib
    OH, OW = full_h - pt - pb, full_w - pl - pr
    acc = full[:, pt:pt + OH, pl:pl + OW] + bias.astype(np.int64).reshape(G, GOC)
    return acc.reshape(case.batch, OH, OW, G * GOC), OH, OW


def main():
    lib = ref.lib()
    blobs = {}
    cases = [c for c in DECONV_CASES if c.batch > 0] + [c for c in EXTRA_DECONV_CASES if c.name in GOLDEN_EXTRA]
    for case in cases:
        inp, kernel, bias = deconv_tensors(case)
        acc, OH, OW = numpy_deconv_acc(case, inp, kernel, bias)
        assert np.abs(acc).max() < 2**31
        oscale, ozp = output_quantization(acc)
        cout = case.groups * case.goc
        rows = case.batch * OH * OW
        out = np.full((rows - 1) * case.out_stride + cout, FILL, dtype=np.uint8)
        op = lib.create_deconvolution2d_nhwc_q8(
            *case.padding, case.adjustment[0], case.adjustment[1], case.kernel_size[0], case.kernel_size[1],
            case.subsampling[0], case.subsampling[1], case.dilation[0], case.dilation[1],
            case.groups, case.gic, case.goc,
            case.izp, 1.0, case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
        padded = np.concatenate([np.zeros(8, np.uint8), inp, np.zeros(8, np.uint8)])
        lib.setup_deconvolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1],
                                          padded[8:], case.in_stride, out, case.out_stride)
        lib.run_operator(op)
        lib.delete_operator(op)
        key = "deconv/" + case.name
        blobs[key + "/input"] = inp
        blobs[key + "/kernel"] = kernel
        blobs[key + "/bias"] = bias
        blobs[key + "/quant"] = np.array([float(oscale), float(ozp)], dtype=np.float64)
        blobs[key + "/output"] = out
        print(f"{key}: {OH}x{OW}, {out.size} bytes, scale {float(oscale):.4f} zp {ozp}")
    path = os.path.join(HERE, "reference_deconv_outputs.npz")
    np.savez_compressed(path, **blobs)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
