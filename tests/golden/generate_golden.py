#!/usr/bin/env python3
"""Generate tests/golden/reference_outputs.npz by running the COMPILED REFERENCE
(oracle/_ref/libqnnpack_ref.so, built from /root/reference by oracle/Makefile)
through its public API on the seeded cases of tests/_cases.py.

Run in the build container (the reference tree is not available on the GPU box):
    python tests/golden/generate_golden.py
The fixture stores, per case, the exact input / kernel / bias bytes fed to the
reference, the output quantization and the reference's output bytes, so neither
numpy's generators nor the oracle are trusted when the fixture is replayed.
The quantization (output scale / zero point) is derived from int32 accumulators
computed here with plain numpy int64 loops-free arithmetic (im2col + matmul), not
with the oracle under test.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from _cases import (CONV_CASES, EXTRA_CONV_CASES, EXTRA_FC_CASES, FC_CASES, conv_tensors, fc_tensors,  # noqa: E402
                    output_quantization, strided_view)
from oracle import ref  # noqa: E402

FILL = 0xA5
GOLDEN_EXTRA_CONV = {"x_1x1_k64_n64_vec16", "x_3x3_c64_vec16", "x_3x3_c3_first_layer", "x_dw3x3_c32",
                     "x_dw3x3_c96_s2", "x_dw5x5_c64", "x_1x1_zp_0_255", "x_1x1_zp_255_0", "x_dw3x3_c64_zp"}
GOLDEN_EXTRA_FC = {"x_c1_plumbing_1x1024x1000", "x_m129_k72_n33"}


def numpy_conv_acc(case, inp, kernel, bias):
    """Independent int64 accumulators: explicit padding with the zero point + einsum."""
    H, W = case.input_size
    G, GIC, GOC = case.groups, case.gic, case.goc
    KH, KW = case.kernel_size
    pt, pr, pb, pl = case.padding
    x = strided_view(inp, case.batch * H * W, G * GIC, case.in_stride).astype(np.int64)
    x = x.reshape(case.batch, H, W, G, GIC) - case.izp
    xp = np.zeros((case.batch, H + pt + pb, W + pl + pr, G, GIC), dtype=np.int64)
    xp[:, pt:pt + H, pl:pl + W] = x
    eh = (KH - 1) * case.dilation[0] + 1
    ew = (KW - 1) * case.dilation[1] + 1
    OH = (H + pt + pb - eh) // case.subsampling[0] + 1
    OW = (W + pl + pr - ew) // case.subsampling[1] + 1
    w = kernel.astype(np.int64) - case.kzp          # [G, GOC, KH, KW, GIC]
    acc = np.zeros((case.batch, OH, OW, G, GOC), dtype=np.int64)
    for ky in range(KH):
        for kx in range(KW):
            ys = ky * case.dilation[0]
            xs = kx * case.dilation[1]
            patch = xp[:, ys:ys + (OH - 1) * case.subsampling[0] + 1:case.subsampling[0],
                       xs:xs + (OW - 1) * case.subsampling[1] + 1:case.subsampling[1]]
            acc += np.einsum("nyxgi,goi->nyxgo", patch, w[:, :, ky, kx, :])
    acc += bias.astype(np.int64).reshape(G, GOC)
    return acc.reshape(case.batch, OH, OW, G * GOC), OH, OW


def main():
    lib = ref.lib()
    blobs = {}
    conv_cases = [c for c in CONV_CASES if c.batch > 0] + [c for c in EXTRA_CONV_CASES if c.name in GOLDEN_EXTRA_CONV]
    for case in conv_cases:
        inp, kernel, bias = conv_tensors(case)
        acc, OH, OW = numpy_conv_acc(case, inp, kernel, bias)
        assert np.abs(acc).max() < 2**31
        oscale, ozp = output_quantization(acc)
        cout = case.groups * case.goc
        rows = case.batch * OH * OW
        out = np.full((rows - 1) * case.out_stride + cout, FILL, dtype=np.uint8)
        op = lib.create_convolution2d_nhwc_q8(
            *case.padding, case.kernel_size[0], case.kernel_size[1], case.subsampling[0], case.subsampling[1],
            case.dilation[0], case.dilation[1], case.groups, case.gic, case.goc,
            case.izp, 1.0, case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
        # the SSE2 kernels may read up to 7 bytes before a row: give them slack
        padded = np.concatenate([np.zeros(8, np.uint8), inp, np.zeros(8, np.uint8)])
        lib.setup_convolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1],
                                        padded[8:], case.in_stride, out, case.out_stride)
        lib.run_operator(op)
        lib.delete_operator(op)
        key = "conv/" + case.name
        blobs[key + "/input"] = inp
        blobs[key + "/kernel"] = kernel
        blobs[key + "/bias"] = bias
        blobs[key + "/quant"] = np.array([float(oscale), float(ozp)], dtype=np.float64)
        blobs[key + "/output"] = out
        print(f"{key}: {out.size} bytes, scale {float(oscale):.4f} zp {ozp}")
    fc_cases = [c for c in FC_CASES if c.batch > 0] + [c for c in EXTRA_FC_CASES if c.name in GOLDEN_EXTRA_FC]
    for case in fc_cases:
        inp, kernel, bias = fc_tensors(case)
        a = strided_view(inp, case.batch, case.input_channels, case.in_stride).astype(np.int64) - case.izp
        acc = a @ (kernel.astype(np.int64) - case.kzp).T + bias.astype(np.int64)
        oscale, ozp = output_quantization(acc)
        out = np.full((case.batch - 1) * case.out_stride + case.output_channels, FILL, dtype=np.uint8)
        op = lib.create_fully_connected_nc_q8(case.input_channels, case.output_channels, case.izp, 1.0,
                                              case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
        padded = np.concatenate([np.zeros(8, np.uint8), inp, np.zeros(8, np.uint8)])
        lib.setup_fully_connected_nc_q8(op, case.batch, padded[8:], case.in_stride, out, case.out_stride)
        lib.run_operator(op)
        lib.delete_operator(op)
        key = "fc/" + case.name
        blobs[key + "/input"] = inp
        blobs[key + "/kernel"] = kernel
        blobs[key + "/bias"] = bias
        blobs[key + "/quant"] = np.array([float(oscale), float(ozp)], dtype=np.float64)
        blobs[key + "/output"] = out
        print(f"{key}: {out.size} bytes")
    path = os.path.join(HERE, "reference_outputs.npz")
    np.savez_compressed(path, **blobs)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
