"""Measurement aid: per-operator kernel time of the whole-network example (examples/mobilenetv2.py), batch from argv."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd
from examples import mobilenetv2 as mnv2
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
fuse = len(sys.argv) > 2 and sys.argv[2] == "fuse"
fold = len(sys.argv) > 2 and sys.argv[2] == "fold"      # residual adds in the project convolutions' epilogues
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
if os.environ.get("QNNP_FUSED_WEIGHTS"):      # A/B: 2 = the strip kernel fetches its expand / project fragments from L2
    lib.set_option("fused_weights", int(os.environ["QNNP_FUSED_WEIGHTS"]))
plan = mnv2.build_plan()
net = mnv2.DeviceNetwork(lib, torch, plan, batch, fuse=fuse, fold_adds=fold)
net.buffers[0].copy_(torch.randint(0, 256, (net.buffers[0].numel(),), dtype=torch.uint8, device="cuda"))
net.run(); net.capture()
total = net.time_ms(3, 20)
rows = []
by_name = {op.name: op for op in plan.ops}
for name, h in net.schedule:
    ms = lib.time_operator(h, 2, 20)
    if name in net.fused:
        first, last = net.fused[name]
        src, dst = plan.ops[first].src[0], plan.ops[last].dst
        hidden = plan.shapes[plan.ops[first + (1 if plan.ops[first].name.endswith("_expand") else 0)].dst] if True else None
        nbytes = mnv2.tensor_bytes(plan, src, batch) + mnv2.tensor_bytes(plan, dst, batch)
        rows.append((ms, name, net.kernels[name], plan.shapes[src], plan.shapes[dst], nbytes))
    else:
        op = by_name[name]
        nbytes = sum(mnv2.tensor_bytes(plan, s, batch) for s in op.src) + mnv2.tensor_bytes(plan, op.dst, batch)
        kern = net.kernels[name]
        if name in net.folded:
            nbytes += mnv2.tensor_bytes(plan, op.dst, batch)                   # the residual read
            kern += "+add" if lib.operator_residual_folded(h) == 1 else " then add"
        rows.append((ms, name, kern, plan.shapes[op.src[0]], plan.shapes[op.dst], nbytes))
print(f"graph replay {total*1e3:.1f} us; sum of individually timed operators {sum(r[0] for r in rows)*1e3:.1f} us")
for ms, name, kern, sin, sout, nbytes in rows:
    print(f"{ms*1e3:7.2f} us  {nbytes/ms/1e6:7.0f} GB/s  {name:22s} {kern:26s} {sin} -> {sout}")
net.close()
