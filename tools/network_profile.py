"""Measurement aid: per-operator kernel time of the whole-network example (examples/mobilenetv2.py), batch from argv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd
from examples import mobilenetv2 as mnv2
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
plan = mnv2.build_plan()
net = mnv2.DeviceNetwork(lib, torch, plan, batch)
net.buffers[0].copy_(torch.randint(0, 256, (net.buffers[0].numel(),), dtype=torch.uint8, device="cuda"))
net.run(); net.capture()
total = net.time_ms(3, 20)
rows = []
for op, h in zip(plan.ops, net.handles):
    ms = lib.time_operator(h, 2, 20)
    nbytes = sum(mnv2.tensor_bytes(plan, s, batch) for s in op.src) + mnv2.tensor_bytes(plan, op.dst, batch)
    rows.append((ms, op.name, net.kernels[op.name], plan.shapes[op.src[0]], plan.shapes[op.dst], nbytes))
print(f"graph replay {total*1e3:.1f} us; sum of individually timed operators {sum(r[0] for r in rows)*1e3:.1f} us")
for ms, name, kern, sin, sout, nbytes in rows:
    print(f"{ms*1e3:7.2f} us  {nbytes/ms/1e6:7.0f} GB/s  {name:22s} {kern:26s} {sin} -> {sout}")
net.close()
