"""Measurement aid: pointwise / strided rows of the ResNet lists at batch 128 under several forced kernels ("gemm_kernel" option):
python tools/conv_variants_time.py   -- prints, per shape, us by variant (0 = auto)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
from qnnpack_amd import QnnpackError
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
shapes = [s for s in bench.RESNET50 + bench.RESNET18 if s[2] == 1]
seen = []
for s in shapes:
    if s not in seen: seen.append(s)
variants = [0, 2, 20, 9, 6, 1, 5]
for (H, W, KH, KW, S, D, G, GIC, GOC) in seen:
    row = []
    for v in variants:
        lib.set_option("gemm_kernel", v)
        try:
            layer = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, G, GIC, GOC, seed=5, min_bytes_between_reuse=512 << 20)
            ms = layer.time_ms(2, 8)
            row.append(f"{v}:{layer.kernel.replace('q8_', '')} {ms*1e3:.1f}")
            layer.close()
        except QnnpackError:
            row.append(f"{v}:-")
        except Exception as exc:  # noqa
            row.append(f"{v}:ERR")
    lib.set_option("gemm_kernel", 0)
    print([H, W, S, GIC, GOC], " | ".join(row), flush=True)
