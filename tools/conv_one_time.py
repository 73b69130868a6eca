"""Measurement aid: one convolution shape of the reference bench's lists at batch 128 (bench.ConvLayer, rotating buffers):
python tools/conv_one_time.py H W K S G GIC GOC [rounds] [gemm_kernel]   -- prints kernel, us, TOP/s, fraction of bound"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
H, W, K, S, G, GIC, GOC = (int(x) for x in sys.argv[1:8])
rounds = int(sys.argv[8]) if len(sys.argv) > 8 else 3
variant = int(sys.argv[9]) if len(sys.argv) > 9 else 0
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
lib.set_option("gemm_kernel", variant)
layer = bench.ConvLayer(lib, torch, 128, H, W, K, K, S, 1, G, GIC, GOC, seed=5, min_bytes_between_reuse=512 << 20, kzp=int(os.environ.get("KZP", "127")))
bound = bench.layer_bound_ms(layer, G * GOC * K * K * GIC)
for _ in range(rounds):
    ms = layer.time_ms(2, 10)
    print(f"{[H, W, K, S, G, GIC, GOC]} {layer.kernel:28s} {ms*1e3:8.2f} us {layer.ops/ms/1e9:8.1f} TOP/s  frac of bound {bound/ms:.3f}")
