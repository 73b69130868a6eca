"""Measurement aid: bench.py's "next_rows" leg alone (deconvolution, add, pooling), optionally with a forced
"gemm_kernel" for A/B: python tools/next_rows_time.py [batch] [variant ...]."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
variants = [int(v) for v in sys.argv[2:]] or [0]
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
for rnd in range(2):
    for v in variants:
        lib.set_option("gemm_kernel", v)
        rows = bench.next_rows_bench(lib, torch, batch, 3, 20)
        lib.set_option("gemm_kernel", 0)
        for name, r in rows.items():
            if "deconv" in name:
                print(f"variant {v:2d} {name:32s} {r['kernel']:28s} {r['ms']*1e3:8.2f} us {r['gbs']:8.1f} GB/s")
