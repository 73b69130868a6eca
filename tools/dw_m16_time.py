"""Measurement aid (round 6): the stride-1 depthwise layers of the MobileNetV2 sweep at batch 128, kernel G (automatic) against the
16x16x64 matrix-core walk ("dwconv_kernel" 7), interleaved: python tools/dw_m16_time.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
shapes = [(i + 1, s) for i, s in enumerate(bench.MOBILENETV2) if s[6] > 1 and s[4] == 1]
for idx, (H, W, KH, KW, S, D, G, GIC, GOC) in shapes:
    layers = {}
    for v in (0, 7):
        lib.set_option("dwconv_kernel", v)
        layers[v] = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, G, GIC, GOC, seed=100 + idx, min_bytes_between_reuse=512 << 20)
    lib.set_option("dwconv_kernel", 0)
    row = []
    for rnd in range(rounds):
        for v in ((0, 7) if rnd % 2 == 0 else (7, 0)):
            row.append((v, layers[v].time_ms(2, 10) * 1e3))
    b = layers[0].in_bytes + layers[0].out_bytes
    for v in (0, 7):
        ts = sorted(t for vv, t in row if vv == v)
        med = ts[len(ts) // 2]
        print(f"layer {idx:2d} {H}x{W}x{G}  {layers[v].kernel:26s} {med:7.2f} us  {b / med / 1e6:7.1f} GB/s  {[round(t, 1) for t in ts]}", flush=True)
    for v in layers: layers[v].close()
