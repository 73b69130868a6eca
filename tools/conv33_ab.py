"""Measurement aid (round 6): BASELINE configs[2] (3x3 convolution 56x56x64 -> 64, batch 128, kernel zero point 127) on the 16x16x64
flavour of the weight-stationary kernel (auto) against the 32x32x32 one ("gemm_kernel" 27), interleaved on one box:
python tools/conv33_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, qnnpack_amd, bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
layers = {}
for v in (0, 27):
    lib.set_option("gemm_kernel", v)
    layers[v] = bench.ConvLayer(lib, torch, 128, 56, 56, 3, 3, 1, 1, 1, 64, 64, seed=3, min_bytes_between_reuse=544 << 20)
lib.set_option("gemm_kernel", 0)
for rnd in range(rounds):
    for v, layer in (layers.items() if rnd % 2 == 0 else reversed(list(layers.items()))):
        ms = layer.time_ms(2, 10)
        print(f"gemm_kernel {v:2d} {layer.kernel:28s} {ms*1e3:8.2f} us  {2*128*56*56*64*576/ms/1e9:8.1f} TOP/s", flush=True)
