"""Generator of tests/golden/reference_bench_shapes.json: the shape tables of the reference's convolution benchmark
(/root/reference/bench/convolution.cc:108-942 -- one `b->Args({N, H, W, KH, KW, S, D, G, GCin, GCout})` line per layer, grouped by the
function that registers them), read here where the reference tree exists and committed as DATA (numbers only): bench.py and the GPU
tier time / check these shapes on the GPU box, where /root/reference does not exist.
    python tests/golden/generate_bench_shapes.py
The lists bench.py already carries by hand (MobileNetV2, ResNet-18/50, ShuffleNet v1 g2) are cross-checked against the file."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/bench/convolution.cc"


def main():
    text = open(SRC).read()
    lists, current = {}, None
    for line in text.splitlines():
        m = re.match(r"static void (\w+)\(benchmark::internal::Benchmark\* b\)", line)
        if m:
            current = m.group(1)
            lists[current] = []
            continue
        m = re.search(r"b->Args\(\{([^}]*)\}\)", line)
        if m and current and not line.lstrip().startswith("//"):        # (repeated blocks are commented out in the file)
            vals = [int(v) for v in m.group(1).split(",")]
            assert len(vals) == 10 and vals[0] == 1, line
            lists[current].append(vals[1:])        # (H, W, KH, KW, S, D, G, GCin, GCout): bench.py's tuple order
    lists = {k: v for k, v in lists.items() if v}
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import bench
    for name, mine in (("MobileNetV2", bench.MOBILENETV2), ("ResNet18", bench.RESNET18), ("ResNet50", bench.RESNET50),
                       ("ShuffleNetV1G2", bench.SHUFFLENET_V1_G2)):
        # (bench.py keeps a repeated row once)
        assert all(list(t) in lists[name] for t in mine) and all(tuple(r) in mine for r in lists[name]), name
    out = os.path.join(HERE, "reference_bench_shapes.json")
    with open(out, "w") as f:
        json.dump({"source": "bench/convolution.cc:108-942 (b->Args lines; N = 1 dropped)", "order": "H W KH KW S D G GCin GCout",
                   "lists": lists}, f, indent=0, separators=(",", ":"))
        f.write("\n")
    print({k: len(v) for k, v in lists.items()})


if __name__ == "__main__":
    main()
