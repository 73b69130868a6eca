"""Cases and drivers for the two byte-streaming operators (add, global average pooling).

Case lists restate the reference's operator tests: test/add.cc (31 tests: batch 1 / 3 / strided, channels
1..91 step 15, qmin / qmax 128, scales 1e-2..1e1, zero points 0..255 step 51; defaults of
test/add-operator-tester.h:251-258) and test/global-average-pooling.cc (34 tests over the SSE2 tile nr = 8,
mr = 7: channels 8..24 x width 1..7 and 7..28, channels 1..7 x width 1..16, strides 5*nr, scales 0.01*pi^k,
zero points step 51, min / max 128; defaults of test/global-average-pooling-operator-tester.h:221-226).
The reference testers compare against a float model with a 0.5-0.6 LSB tolerance; here the scalar oracle
(qnnp_add_quantize / qnnp_avgpool_quantize restated) is the bit-exact expectation.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass
from typing import List

import numpy as np

from oracle import o1

FILL = 0xA5


def _seed(name: str) -> int:
    return 0x51A0 ^ (zlib.crc32(name.encode()) & 0x7FFFFFFF)


@dataclass(frozen=True)
class AddCase:
    name: str
    batch: int
    channels: int
    a_stride: int = 0
    b_stride: int = 0
    y_stride: int = 0
    a_scale: float = 0.75
    b_scale: float = 1.25
    y_scale: float = 0.96875
    a_zp: int = 121
    b_zp: int = 127
    y_zp: int = 133
    qmin: int = 0
    qmax: int = 255

    @property
    def strides(self):
        return (self.a_stride or self.channels, self.b_stride or self.channels, self.y_stride or self.channels)


def add_cases() -> List[AddCase]:
    out = [AddCase("a_zero_batch", 0, 2)]
    scales = [1.0e-2, 1.0e-1, 1.0, 1.0e+1]
    for prefix, batch, strides in (("unit", 1, {}), ("small", 3, {}),
                                   ("strided", 3, dict(a_stride=129, b_stride=123, y_stride=117))):
        for ch in range(1, 100, 15):
            base = dict(batch=batch, channels=ch, **strides)
            out.append(AddCase(f"a_{prefix}_c{ch}", **base))
            out.append(AddCase(f"a_{prefix}_c{ch}_qmin", qmin=128, **base))
            out.append(AddCase(f"a_{prefix}_c{ch}_qmax", qmax=128, **base))
            for s in scales:
                out.append(AddCase(f"a_{prefix}_c{ch}_as{s:g}", a_scale=s, **base))
                out.append(AddCase(f"a_{prefix}_c{ch}_bs{s:g}", b_scale=s, **base))
                out.append(AddCase(f"a_{prefix}_c{ch}_ys{s:g}", y_scale=s, **base))
            for zp in range(0, 256, 51):
                out.append(AddCase(f"a_{prefix}_c{ch}_azp{zp}", a_zp=zp, **base))
                out.append(AddCase(f"a_{prefix}_c{ch}_bzp{zp}", b_zp=zp, **base))
                out.append(AddCase(f"a_{prefix}_c{ch}_yzp{zp}", y_zp=zp, **base))
    # beyond the reference's list: the dense 16-byte-vector path, ragged tails, a residual-sized tensor
    out += [
        AddCase("ax_dense_vec16", 4, 64),
        AddCase("ax_dense_c24_b6", 6, 24),                # 144 bytes: 9 vectors
        AddCase("ax_dense_odd_total", 3, 7),
        AddCase("ax_residual_56x56x24", 56 * 56, 24),
        AddCase("ax_residual_strided", 14 * 14, 96, a_stride=128, b_stride=96, y_stride=112),
        AddCase("ax_clamp_tight", 5, 33, qmin=100, qmax=101),
    ]
    return out


def add_tensors(case: AddCase):
    rng = np.random.default_rng(_seed(case.name))
    sa, sb, sy = case.strides
    n = case.batch
    a = rng.integers(0, 256, size=max(n - 1, 0) * sa + case.channels if n else 0, dtype=np.uint8)
    b = rng.integers(0, 256, size=max(n - 1, 0) * sb + case.channels if n else 0, dtype=np.uint8)
    y = np.full(max(n - 1, 0) * sy + case.channels if n else 0, FILL, dtype=np.uint8)
    return a, b, y


def add_expected(case: AddCase, a, b):
    sa, sb, sy = case.strides
    y = np.full(max(case.batch - 1, 0) * sy + case.channels if case.batch else 0, FILL, dtype=np.uint8)
    if case.batch:
        o1.add_q8(case.batch, case.channels, case.a_zp, case.a_scale, case.b_zp, case.b_scale, case.y_zp, case.y_scale,
                  case.qmin, case.qmax, a, sa, b, sb, y, sy)
    return y


def add_run(lib, case: AddCase, a, b, to_device=None, from_device=None):
    sa, sb, sy = case.strides
    y = np.full(max(case.batch - 1, 0) * sy + case.channels if case.batch else 0, FILL, dtype=np.uint8)
    op = lib.create_add_nc_q8(case.channels, case.a_zp, case.a_scale, case.b_zp, case.b_scale,
                              case.y_zp, case.y_scale, case.qmin, case.qmax, 0)
    try:
        if to_device is not None and case.batch:
            d_a, d_b, d_y = to_device(a), to_device(b), to_device(y)
            lib.setup_add_nc_q8(op, case.batch, d_a, sa, d_b, sb, d_y, sy)
            lib.run_operator(op)
            y = from_device(d_y)
        else:
            one = np.zeros(1, np.uint8)
            lib.setup_add_nc_q8(op, case.batch, a if a.size else one, sa, b if b.size else one, sb,
                                y if y.size else one, sy)
            lib.run_operator(op)
        kname = lib.operator_kernel(op) if hasattr(lib, "operator_kernel") else None
    finally:
        lib.delete_operator(op)
    return y, kname


@dataclass(frozen=True)
class GapCase:
    name: str
    batch: int
    width: int
    channels: int
    in_stride: int = 0
    out_stride: int = 0
    in_scale: float = 1.0
    out_scale: float = 1.0
    in_zp: int = 121
    out_zp: int = 133
    qmin: int = 0
    qmax: int = 255

    @property
    def strides(self):
        return (self.in_stride or self.channels, self.out_stride or self.channels)


def gap_cases(full: bool = True) -> List[GapCase]:
    nr, mr = 8, 7
    out = [GapCase("g_zero_batch", 0, 1, 8)]
    pis = []
    s = 0.01
    while s < 100.0:
        pis.append(float(np.float32(s)))
        s *= 3.14159265
    regimes = [("many_small", range(nr, 3 * nr + 1), range(1, mr + 1)),
               ("many_large", range(nr, 3 * nr + 1), range(mr, 4 * mr + 1)),
               ("few", range(1, nr), range(1, 2 * nr + 1))]
    for rname, chans, widths in regimes:
        chans, widths = list(chans), list(widths)
        if not full:
            chans, widths = chans[::3] + chans[-1:], widths[::3] + widths[-1:]
        for ch in chans:
            for w in widths:
                tag = f"g_{rname}_c{ch}_w{w}"
                out.append(GapCase(tag, 1, w, ch))
                out.append(GapCase(tag + "_istride", 1, w, ch, in_stride=5 * nr))
                out.append(GapCase(tag + "_omin", 1, w, ch, qmin=128))
                out.append(GapCase(tag + "_omax", 1, w, ch, qmax=128))
                out.append(GapCase(tag + "_b3", 3, w, ch))
                out.append(GapCase(tag + "_b3_istride", 3, w, ch, in_stride=5 * nr))
                out.append(GapCase(tag + "_b3_ostride", 3, w, ch, out_stride=5 * nr))
                if (ch + w) % 4 == 0 or not full:      # the scale / zero-point sweeps on a quarter of the grid
                    for sc in pis:
                        out.append(GapCase(tag + f"_is{sc:.4g}", 1, w, ch, in_scale=sc))
                        out.append(GapCase(tag + f"_os{sc:.4g}", 1, w, ch, out_scale=sc))
                    for zp in range(0, 256, 51):
                        out.append(GapCase(tag + f"_izp{zp}", 1, w, ch, in_zp=zp))
                        out.append(GapCase(tag + f"_ozp{zp}", 1, w, ch, out_zp=zp))
    # beyond the reference's list: MobileNetV2's own pooling (7x7x1280), the dword path with strides, big widths
    out += [
        GapCase("gx_mobilenetv2_7x7x1280", 4, 49, 1280),
        GapCase("gx_mobilenetv2_strided", 3, 49, 1280, in_stride=1296, out_stride=1284),
        GapCase("gx_c320_w196", 2, 196, 320),
        GapCase("gx_c4_w1000", 2, 1000, 4),
        GapCase("gx_c6_w300_unaligned", 2, 300, 6, in_stride=7),
        GapCase("gx_c1024_w2", 5, 2, 1024),
        GapCase("gx_scale_ratio_low", 2, 9, 16, in_scale=0.004, out_scale=1.0),
        GapCase("gx_scale_ratio_high", 2, 9, 16, in_scale=200.0, out_scale=1.0),
    ]
    return out


def gap_tensors(case: GapCase):
    rng = np.random.default_rng(_seed(case.name))
    si, so = case.strides
    px = case.batch * case.width
    inp = rng.integers(0, 256, size=max(px - 1, 0) * si + case.channels if px else 0, dtype=np.uint8)
    return inp


def _gap_out(case):
    si, so = case.strides
    return np.full(max(case.batch - 1, 0) * so + case.channels if case.batch else 0, FILL, dtype=np.uint8)


def gap_expected(case: GapCase, inp):
    si, so = case.strides
    out = _gap_out(case)
    if case.batch:
        o1.global_average_pooling_q8(case.batch, case.width, case.channels, case.in_zp, case.in_scale, case.out_zp,
                                     case.out_scale, case.qmin, case.qmax, inp, si, out, so)
    return out


def gap_run(lib, case: GapCase, inp, to_device=None, from_device=None):
    si, so = case.strides
    out = _gap_out(case)
    op = lib.create_global_average_pooling_nwc_q8(case.channels, case.in_zp, case.in_scale, case.out_zp, case.out_scale,
                                                  case.qmin, case.qmax, 0)
    try:
        if to_device is not None and case.batch:
            d_in, d_out = to_device(inp), to_device(out)
            lib.setup_global_average_pooling_nwc_q8(op, case.batch, case.width, d_in, si, d_out, so)
            lib.run_operator(op)
            out = from_device(d_out)
        else:
            one = np.zeros(1, np.uint8)
            lib.setup_global_average_pooling_nwc_q8(op, case.batch, case.width, inp if inp.size else one, si,
                                                    out if out.size else one, so)
            lib.run_operator(op)
        kname = lib.operator_kernel(op) if hasattr(lib, "operator_kernel") else None
    finally:
        lib.delete_operator(op)
    return out, kname
