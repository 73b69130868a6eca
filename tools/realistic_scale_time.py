"""Measurement aid: the MobileNetV2 sweep (bench.MOBILENETV2, batch 128, rotating buffers) with a requantization scale that
takes the shift >= 1 epilogue (output scale 20 -> 0.25 / 20 = 0.0125, shift 6) beside the reference bench's 0.5 (shift 0):
python tools/realistic_scale_time.py [out_scale ...]   -- per layer kernel and us, per sweep images/s by sum of layers"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd, bench
scales = [float(x) for x in sys.argv[1:]] or [20.0, 0.5]
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
res = {}
for osc in scales:
    res[osc] = bench.conv_list_bench(lib, torch, 128, bench.MOBILENETV2, 2100, out_scale=osc)
    r = res[osc]
    print(f"requant scale {0.25 / osc:.6g}: {r['images_per_s_by_sum_of_layers']:.0f} images/s by sum of layers, {r['sum_of_layer_ms'] * 1e3:.1f} us")
for i in range(len(bench.MOBILENETV2)):
    print("   %2d %-30s " % (i + 1, res[scales[0]]["layers"][i]["shape"]) +
          " | ".join("%-26s %7.2f us %.3f" % (res[o]["layers"][i]["kernel"], res[o]["layers"][i]["us"], res[o]["layers"][i]["frac_of_bound"]) for o in scales))
