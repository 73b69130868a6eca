"""Measurement aid: the reference bench's ResNet-18 / ResNet-50 / ShuffleNet-v1-g2 shape lists (bench.py `conv_lists`) at
batch 128, per layer: kernel, us, TOP/s, fraction of max(MFMA, HBM) bound.  python tools/conv_lists_time.py [resnet18|resnet50|shufflenet|all|dense3x3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
which = sys.argv[1] if len(sys.argv) > 1 else "all"
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
lists = {"resnet18": bench.RESNET18, "resnet50": bench.RESNET50, "shufflenet": bench.SHUFFLENET_V1_G2}
if which == "dense3x3":
    seen, rows = set(), []
    for s in bench.RESNET18 + bench.RESNET50:
        if (s[2] == 3 and s[6] == 1 and s[7] >= 64 or s[2] == 7) and s not in seen:
            seen.add(s); rows.append(s)
    lists = {"dense3x3": rows}
elif which != "all":
    lists = {which: lists[which]}
for name, shapes in lists.items():
    r = bench.conv_list_bench(lib, torch, 128, shapes, 1800)
    print(f"{name}: {r['images_per_s_by_sum_of_layers']:.0f} images/s by sum of layers, {r['sum_of_layer_ms']*1e3:.1f} us, frac of bound {r['frac_of_bound']}, worst dense 3x3 {r['worst_dense_3x3_frac']}")
    for row in r["layers"]:
        print("   %-32s %-28s %8.2f us %7.1f TOP/s %7.1f GB/s  %s-bound %.3f" % (row["shape"], row["kernel"], row["us"], row["tops"], row["gbs"], row["bound"], row["frac_of_bound"]))
