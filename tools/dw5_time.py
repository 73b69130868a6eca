"""Measurement aid: bench.py's 5x5 / dilated depthwise shapes, alternating "dwconv_kernel" variants on one box:
python tools/dw5_time.py [batch] [variant ...]   (2 = LDS-tiled, 0 = automatic: the 5x5 column walk where it applies)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = [int(v) for v in sys.argv[2:]] or [0]
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
shapes = [("dw5x5_56x56x72_s2", (56, 56, 5, 5, 2, 1, 72, 1, 1)), ("dw5x5_28x28x240_s1", (28, 28, 5, 5, 1, 1, 240, 1, 1)),
          ("dw5x5_14x14x672_s1", (14, 14, 5, 5, 1, 1, 672, 1, 1)), ("dw5x5_112x112x32_s1", (112, 112, 5, 5, 1, 1, 32, 1, 1))]
only = os.environ.get("DW5_ONLY")          # one shape (PMC passes): a substring of its name
rounds = int(os.environ.get("DW5_ROUNDS", "2"))
for rnd in range(rounds):
    for name, (H, W, KH, KW, S, D, G, GIC, GOC) in shapes:
        if only and only not in name:
            continue
        for v in variants:
            lib.set_option("dwconv_kernel", v)
            layer = bench.ConvLayer(lib, torch, batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=700, min_bytes_between_reuse=512 << 20,
                                    out_scale=0.5)
            lib.set_option("dwconv_kernel", 0)
            ms = layer.time_ms(2, 10)
            b = layer.in_bytes + layer.out_bytes
            print(f"{name:22s} variant {v} {layer.kernel:26s} {ms*1e3:8.2f} us {b/ms/1e6:8.1f} GB/s")
            layer.close()
