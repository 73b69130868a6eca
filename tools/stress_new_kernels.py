"""One-off stress run (round 6): random shapes through the kernels added in the second half of the round, each forced, against the scalar
oracle.  python tools/stress_new_kernels.py [draws per kernel]   -- prints a line per family, exits non-zero on the first mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import qnnpack_amd
from qnnpack_amd.binding import QnnpackError
from _cases import ConvCase, conv_tensors
from _gpu import from_device, to_device
from _runner import conv_expected, conv_run

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
lib = qnnpack_amd.load(); lib.initialize()


def run(family, key, code, cases, want=None):
    bad = 0
    for case in cases:
        inp, kern, bias = conv_tensors(case)
        expected, quant, out_hw = conv_expected(case, inp, kern, bias)
        lib.set_option(key, code)
        try:
            out, kname = conv_run(lib, case, quant, out_hw, inp, kern, bias, to_device, from_device)
        except QnnpackError as e:
            print("REFUSED", family, case, e); bad += 1; continue
        finally:
            lib.set_option(key, 0)
        if want is not None and not kname.startswith(want):
            print("KERNEL", family, kname, case); bad += 1
        if not np.array_equal(out, expected):
            print("MISMATCH", family, kname, case, int(np.count_nonzero(out != expected)), "bytes"); bad += 1
    print(f"{family}: {len(cases)} cases, {bad} bad", flush=True)
    return bad


def zp(rng, centred=False):
    kz = int(rng.choice([127, 128])) if centred else int(rng.choice([127, 128, 0, 255, 9, 200]))
    return dict(izp=int(rng.choice([127, 0, 255, 3, 250])), kzp=kz)


def clamp(rng):
    return dict(qmin=int(rng.choice([0, 0, 40])), qmax=int(rng.choice([255, 255, 200])))


bad = 0
rng = np.random.default_rng(20260930)
# small-channel weight-stationary 3x3 ("gemm_kernel" 32)
cases = []
for i in range(N):
    cin = int(rng.choice([16, 32, 48, 64])); cout = int(rng.choice([16, 32, 48, 64, 128, 192, 256]))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    pad = tuple(int(x) for x in rng.integers(0, 3, size=4))
    if h + pad[0] + pad[2] < 3 or w + pad[1] + pad[3] < 3: pad = (1, 1, 1, 1)
    cases.append(ConvCase(f"st_ws_{i}", (h, w), (3, 3), pad, gic=cin, goc=cout, batch=int(rng.integers(1, 5)), **zp(rng, True), **clamp(rng)))
bad += run("ws16s", "gemm_kernel", 32, cases, "q8_conv_ws16s")
# grouped 1x1 as a dense GEMM ("gemm_kernel" 31)
cases = []
for i in range(N):
    g = int(rng.integers(2, 9)); gic = int(rng.integers(1, 70)); goc = int(rng.integers(1, 70))
    cases.append(ConvCase(f"st_dense_{i}", (int(rng.integers(1, 12)), int(rng.integers(1, 12))), groups=g, gic=gic, goc=goc,
                          batch=int(rng.integers(1, 6)), **zp(rng), **clamp(rng)))
bad += run("grouped dense", "gemm_kernel", 31, cases)
# alignment-free GEMM ("gemm_kernel" 29), flat rows included (dense pixels, N <= 128, N % 16 != 0)
cases = []
for i in range(N):
    k = int(rng.integers(1, 300)); n = int(rng.integers(1, 260)); g = int(rng.choice([1, 1, 1, 2, 3]))
    kw = {}
    if rng.random() < 0.3: kw = dict(input_pixel_stride=g * k + int(rng.integers(0, 9)), output_pixel_stride=g * n + int(rng.integers(0, 9)))
    cases.append(ConvCase(f"st_u16_{i}", (int(rng.integers(1, 30)), int(rng.integers(1, 30))), groups=g, gic=k, goc=n,
                          batch=int(rng.integers(1, 5)), **kw, **zp(rng), **clamp(rng)))
bad += run("u16 gemm", "gemm_kernel", 29, cases, "q8_gemm_mfma_128x")
# depthwise sliding window on unaligned dwords ("dwconv_kernel" 8)
cases = []
for i in range(N):
    c = int(rng.integers(4, 140)); s = int(rng.choice([1, 2]))
    h, w = int(rng.integers(1, 40)), int(rng.integers(1, 40))
    pad = tuple(int(x) for x in rng.integers(0, 3, size=4))
    if h + pad[0] + pad[2] < 3 or w + pad[1] + pad[3] < 3: pad = (1, 1, 1, 1)
    kw = {}
    if rng.random() < 0.3: kw = dict(input_pixel_stride=c + int(rng.integers(0, 9)), output_pixel_stride=c + int(rng.integers(0, 9)))
    cases.append(ConvCase(f"st_dw_{i}", (h, w), (3, 3), pad, subsampling=(s, s), groups=c, gic=1, goc=1, batch=int(rng.integers(1, 5)),
                          **kw, **zp(rng), **clamp(rng)))
bad += run("dw row any", "dwconv_kernel", 8, cases, "q8_dwconv_row_3x3_any")
# LDS-staged first-layer kernels ("gemm_kernel" 30): 16-byte slots (<= 4 rows) and 32-byte slots (5 / 7 rows)
cases = []
for i in range(N):
    big = rng.random() < 0.5
    kh = int(rng.choice([5, 7])) if big else int(rng.integers(1, 5))
    kwid = int(rng.integers(1, 11 if big else 6))
    w = 16 * int(rng.integers(1, 6)); h = int(rng.integers(1, 50)); s = int(rng.choice([1, 2, 3]))
    pad = (int(rng.integers(0, kh)), int(rng.integers(0, min(kwid, 6))), int(rng.integers(0, kh)), int(rng.integers(0, min(kwid, 6))))
    if h + pad[0] + pad[2] < kh or w + pad[1] + pad[3] < kwid: continue
    cout = int(rng.choice([16, 32, 48, 64] if big else [8, 16, 24, 32, 40, 48, 56, 64]))
    cases.append(ConvCase(f"st_c3_{i}", (h, w), (kh, kwid), pad, subsampling=(s, s), gic=3, goc=cout, batch=int(rng.integers(1, 5)),
                          **zp(rng, not big), **clamp(rng)))
bad += run("c3 lds", "gemm_kernel", 30, cases, "q8_conv_c3rows")
sys.exit(1 if bad else 0)
