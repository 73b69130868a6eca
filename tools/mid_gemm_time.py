"""Measurement aid (round 6): the pointwise rows of the MobileNetV2 sweep and of the ResNet lists at batch 128, automatic choice
against the 128x128-tile centred GEMM ("gemm_kernel" 24, q8gemm128x.hip) and the 256-wide one (23).
python tools/mid_gemm_time.py [--nets mobilenetv2,resnet50,resnet18]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
from qnnpack_amd import QnnpackError
ap = argparse.ArgumentParser(); ap.add_argument("--nets", default="mobilenetv2,resnet50,resnet18"); ap.add_argument("--variants", default="0,25,26,23")
args = ap.parse_args()
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
lists = {"mobilenetv2": bench.MOBILENETV2, "resnet50": bench.RESNET50, "resnet18": bench.RESNET18}
variants = [int(v) for v in args.variants.split(",")]
for net in args.nets.split(","):
    seen = []
    for s in lists[net]:
        if s[2] == 1 and s[3] == 1 and s[6] == 1 and s not in seen: seen.append(s)
    print("==", net, flush=True)
    total = {v: 0.0 for v in variants}
    for (H, W, KH, KW, S, D, G, GIC, GOC) in seen:
        row, best = [], None
        for v in variants:
            lib.set_option("gemm_kernel", v)
            try:
                layer = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, G, GIC, GOC, seed=5, min_bytes_between_reuse=512 << 20)
                ms = layer.time_ms(2, 10)
                row.append(f"{v}:{layer.kernel.replace('q8_', '')} {ms*1e3:.1f}")
                total[v] += ms * 1e3 if v == 0 else 0
                if v == 0: auto_ms = ms * 1e3
                best = ms * 1e3 if best is None else min(best, ms * 1e3)
                layer.close()
            except QnnpackError:
                row.append(f"{v}:-")
        lib.set_option("gemm_kernel", 0)
        print([H, W, S, GIC, GOC], " | ".join(row), f"| best {best:.1f} (auto {auto_ms:.1f})", flush=True)
