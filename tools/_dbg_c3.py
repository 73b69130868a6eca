import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch, qnnpack_amd
from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import conv_expected, conv_run
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
lib.set_option("gemm_kernel", 14)
for case in [ConvCase("w_7x7_nopad", (12, 12), (7, 7), gic=3, goc=32),
             ConvCase("w_7x7_kzp128", (12, 12), (7, 7), gic=3, goc=32, kzp=128),
             ConvCase("w_5x3", (12, 12), (5, 3), (2, 1, 2, 1), gic=3, goc=32, kzp=128)]:
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(lib, case, quant, out_hw, to_device=to_device, from_device=from_device)
    oh, ow = out_hw
    e = expected.reshape(-1, oh, ow, 32).astype(int); o = out.reshape(-1, oh, ow, 32).astype(int)
    bad = (e != o)
    print(case.name, kname, "bad", bad.sum(), "of", bad.size, "by channel", bad.sum(axis=(0,1,2))[:8], "by ox", bad.sum(axis=(0,1,3)), "by oy", bad.sum(axis=(0,2,3)))
    print(" sample exp", e[0,0,0,:8], "got", o[0,0,0,:8], " diff", (o-e)[0,0,:,0])
