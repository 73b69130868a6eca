"""Measurement aid (round 6): ShuffleNet v2's odd-channel 3x3 depthwise layers at batch 128 on the generic four-channel kernel ("dwconv_kernel" 9),
the sliding-window kernel on unaligned dwords (8) and the automatic choice:  python tools/dw_any_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
from qnnpack_amd import QnnpackError
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
shapes = [(56, 56, 2, 58), (28, 28, 1, 58), (28, 28, 2, 116), (56, 56, 2, 122), (28, 28, 1, 122), (28, 28, 2, 244), (14, 14, 1, 244), (14, 14, 2, 488), (7, 7, 1, 488),
          (28, 28, 1, 120), (56, 56, 1, 27)]
for (H, W, S, C) in shapes:
    row = []
    for v in (9, 8, 0):
        lib.set_option("dwconv_kernel", v)
        try:
            layer = bench.ConvLayer(lib, torch, 128, H, W, 3, 3, S, 1, C, 1, 1, seed=5, min_bytes_between_reuse=512 << 20)
            ms = layer.time_ms(2, 8)
            row.append(f"{v}:{layer.kernel.replace('q8_dwconv_', '')} {ms*1e3:.1f}")
            layer.close()
        except QnnpackError:
            row.append(f"{v}:-")
    lib.set_option("dwconv_kernel", 0)
    print([H, W, S, C], " | ".join(row), flush=True)
