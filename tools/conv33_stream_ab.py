"""Measurement aid: BASELINE configs[2] with the streaming-store hint on (whole-line stores, four v_permlane16_swap) and off
(two direct 32-byte-run stores per position), interleaved on one box: python tools/conv33_stream_ab.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
layer = bench.ConvLayer(lib, torch, 128, 56, 56, 3, 3, 1, 1, 1, 64, 64, seed=3, min_bytes_between_reuse=544 << 20)
for rnd in range(rounds):
    for hint in (1, 0):
        lib.operator_set_streaming_stores(layer.op, hint)
        ms = layer.time_ms(2, 10)
        print(f"streaming_stores {hint} {layer.kernel:26s} {ms*1e3:8.2f} us")
