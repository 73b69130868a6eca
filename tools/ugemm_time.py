"""Measurement aid (round 6): the grouped / odd-channel 1x1 rows of the ShuffleNet lists at batch 128, automatic choice against the generic
tile kernel ("gemm_kernel" 1) and the register-staged 128 x 128 GEMM (29):  python tools/ugemm_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
from qnnpack_amd import QnnpackError
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
table = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_bench_shapes.json")))["lists"]
seen = []
for name in ("ShuffleNetV1G1", "ShuffleNetV1G2", "ShuffleNetV1G3", "ShuffleNetV1G4", "ShuffleNetV1G8", "ShuffleNetV2X05", "ShuffleNetV2X10", "ShuffleNetV2X20"):
    for s in table[name]:
        s = tuple(s)
        if s[2] == 1 and s[4] == 1 and s not in seen: seen.append(s)
tot = {0: 0.0, 1: 0.0, 29: 0.0}
for (H, W, KH, KW, S, D, G, GIC, GOC) in seen:
    row = []
    for v in (0, 1, 29):
        lib.set_option("gemm_kernel", v)
        try:
            layer = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, G, GIC, GOC, seed=5, min_bytes_between_reuse=512 << 20)
            ms = layer.time_ms(2, 8)
            tot[v] += ms * 1e3
            row.append(f"{v}:{layer.kernel.replace('q8_', '')} {ms*1e3:.1f}")
            layer.close()
        except QnnpackError:
            row.append(f"{v}:-")
    lib.set_option("gemm_kernel", 0)
    print([H, W, G, GIC, GOC], " | ".join(row), flush=True)
print("sums us", tot)
