"""Measurement aid: dilated 3x3 depthwise shapes, alternating "dwconv_kernel" variants on one box, plus the ten
MobileNetV2 depthwise layers (regression check of kernel G's undilated flavours):
python tools/dw_dil_time.py [batch] [variant ...]   (2 = LDS-tiled, 0 = automatic: the dilated column walk where it applies)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = [int(v) for v in sys.argv[2:]] or [0]
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
shapes = [("dw3x3_dil2_28x28x192", (28, 28, 3, 3, 1, 2, 192, 1, 1)), ("dw3x3_dil2_56x56x144", (56, 56, 3, 3, 1, 2, 144, 1, 1)),
          ("dw3x3_dil4_28x28x192", (28, 28, 3, 3, 1, 4, 192, 1, 1)), ("dw3x3_dil6_33x33x960", (33, 33, 3, 3, 1, 6, 960, 1, 1)),
          ("dw3x3_dil1_28x28x192", (28, 28, 3, 3, 1, 1, 192, 1, 1))]
for rnd in range(2):
    for name, (H, W, KH, KW, S, D, G, GIC, GOC) in shapes:
        for v in variants:
            lib.set_option("dwconv_kernel", v)
            layer = bench.ConvLayer(lib, torch, batch if H < 33 or G < 500 else batch // 4, H, W, KH, KW, S, D, G, GIC, GOC, seed=700,
                                    min_bytes_between_reuse=512 << 20, out_scale=0.5)
            lib.set_option("dwconv_kernel", 0)
            ms = layer.time_ms(2, 10)
            b = layer.in_bytes + layer.out_bytes
            print(f"{name:22s} variant {v} {layer.kernel:32s} {ms*1e3:8.2f} us {b/ms/1e6:8.1f} GB/s  {b/ms/1e6/80:.1f} % of 8 TB/s")
            layer.close()
total_ms = total_b = 0
for (H, W, KH, KW, S, D, G, GIC, GOC) in [s for s in bench.MOBILENETV2 if s[6] > 1] if hasattr(bench, "MOBILENETV2") else []:
    layer = bench.ConvLayer(lib, torch, batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=3, min_bytes_between_reuse=512 << 20)
    ms = layer.time_ms(2, 10); b = layer.in_bytes + layer.out_bytes
    total_ms += ms; total_b += b
    print(f"mnv2 dw {H}x{W}x{G} s{S} {layer.kernel:26s} {ms*1e3:8.2f} us {b/ms/1e6:8.1f} GB/s")
    layer.close()
if total_ms:
    print(f"ten depthwise layers: {total_ms:.4f} ms, {total_b/total_ms/1e6:.1f} GB/s = {total_b/total_ms/1e6/80:.1f} % of 8 TB/s")
