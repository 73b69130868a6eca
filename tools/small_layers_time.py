"""Measurement aid: the late MobileNetV2 pointwise layers (sweep layers 16-30) at batch 128, automatic kernel against forced
"gemm_kernel" variants, interleaved on one box:  python tools/small_layers_time.py [variant ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
variants = [int(v) for v in sys.argv[1:]] or [0, 6, 9]
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
shapes = [s for s in bench.MOBILENETV2[15:30] if s[2] == 1]
totals = {v: 0.0 for v in variants}
for rnd in range(2):
    for (H, W, KH, KW, S, D, G, GIC, GOC) in shapes:
        line = f"{H}x{W} {GIC:4d} -> {GOC:4d} "
        for v in variants:
            lib.set_option("gemm_kernel", v)
            try:
                layer = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, G, GIC, GOC, seed=3, min_bytes_between_reuse=512 << 20)
            except Exception as exc:          # a forced kernel refuses what it cannot take
                lib.set_option("gemm_kernel", 0)
                line += f"| {v}: refused "
                continue
            lib.set_option("gemm_kernel", 0)
            ms = layer.time_ms(2, 10)
            if rnd == 1: totals[v] += ms
            line += f"| {v}: {layer.kernel:26s} {ms*1e3:6.2f} us "
            layer.close()
        print(line)
print("sum over the layers each variant took (second round):", {v: round(t * 1e3, 2) for v, t in totals.items()})
