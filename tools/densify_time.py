"""Measurement aid (round 6): every grouped 1x1 row of the ShuffleNet lists at batch 128 as it runs (one group per tile) against the DENSE
1x1 of the same tensor shape (G * GIC -> G * GOC, what a block-diagonal weight matrix would run as):  python tools/densify_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
table = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_bench_shapes.json")))["lists"]
seen = []
for name in ("ShuffleNetV1G2", "ShuffleNetV1G3", "ShuffleNetV1G4", "ShuffleNetV1G8"):
    for s in table[name]:
        s = tuple(s)
        if s[2] == 1 and s[4] == 1 and s[6] > 1 and s not in seen: seen.append(s)
tot = [0.0, 0.0]
for (H, W, KH, KW, S, D, G, GIC, GOC) in seen:
    row = []
    for i, (g, ic, oc) in enumerate(((G, GIC, GOC), (1, G * GIC, G * GOC))):
        layer = bench.ConvLayer(lib, torch, 128, H, W, KH, KW, S, D, g, ic, oc, seed=5, min_bytes_between_reuse=512 << 20)
        ms = layer.time_ms(2, 8)
        tot[i] += ms * 1e3
        row.append(f"{layer.kernel.replace('q8_', '')} {ms*1e3:.1f}")
        layer.close()
    print([H, W, G, GIC, GOC], " | dense: ".join(row), flush=True)
print("sums us: grouped %.1f, dense %.1f" % tuple(tot))
