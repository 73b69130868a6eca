"""Measurement aid: BASELINE configs[2] (3x3 convolution 56x56x64 -> 64, batch 128) with kernel zero points 127 / 128 (the
zero-point-centred flavour of the weight-stationary kernel) and 126 (pixel sums), interleaved on one box:
python tools/conv33_time.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, qnnpack_amd, bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
layers = {}
for kzp in (127, 128, 126):
    layers[kzp] = bench.ConvLayer(lib, torch, 128, 56, 56, 3, 3, 1, 1, 1, 64, 64, seed=3, min_bytes_between_reuse=544 << 20, kzp=kzp)
for rnd in range(rounds):
    for kzp, layer in layers.items():
        ms = layer.time_ms(2, 10)
        print(f"kzp {kzp} {layer.kernel:26s} {ms*1e3:8.2f} us  {2*128*56*56*64*576/ms/1e9:8.1f} TOP/s")
