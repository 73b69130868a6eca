import sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import qnnpack_amd
from _cases import ConvCase
from _gpu import from_device, to_device
from _runner import conv_expected, conv_run
import torch; torch.cuda.set_device(0); torch.zeros(1, device="cuda")
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream); lib.set_async(False); lib.set_option("gemm_kernel", 8)
for case in [ConvCase("w_3x3_zp0", (9, 9), (3, 3), (1, 1, 1, 1), gic=32, goc=32, izp=0, kzp=0),
             ConvCase("w_3x3_c32_n64", (8, 8), (3, 3), (1, 1, 1, 1), gic=32, goc=64)]:
    expected, quant, out_hw = conv_expected(case)
    out, kname = conv_run(lib, case, quant, out_hw, to_device=to_device, from_device=from_device)
    out = np.asarray(out).reshape(expected.shape); 
    bad = out != expected
    print(kname, case.name, expected.shape, "bad", bad.sum(), "of", bad.size)
    e2 = expected.reshape(-1, expected.shape[-1]); o2 = out.reshape(e2.shape); b2 = bad.reshape(e2.shape)
    print("bad per position:", b2.sum(axis=1)[:90])
    print("bad per channel:", b2.sum(axis=0))
    idx = np.argwhere(b2)[:8]
    for i, c in idx: print(i, c, "exp", e2[i, c], "got", o2[i, c])
