"""Measurement aid (ABLATION=1 build via QNNP_GFX950_LIBRARY): cycle stamps of the fused strip kernel for chosen blocks of
the MobileNetV2 example at batch 128: python tools/trace_fused.py b7_fused b14_fused b1_fused"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QNNP_GFX950_TRACE"] = "1"
import torch, qnnpack_amd
from examples import mobilenetv2 as mnv2
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
if os.environ.get("QNNP_FUSED_WEIGHTS"):
    lib.set_option("fused_weights", int(os.environ["QNNP_FUSED_WEIGHTS"]))
plan = mnv2.build_plan()
net = mnv2.DeviceNetwork(lib, torch, plan, 128, fuse=True)
net.buffers[0].copy_(torch.randint(0, 256, (net.buffers[0].numel(),), dtype=torch.uint8, device="cuda"))
net.run()
lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int; lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
names = ["staging", "E0", "barrier", "D0", "barrier", "P0", "other chunks"]
for want in sys.argv[1:]:
    h = dict(net.schedule)[want]
    for _ in range(3): lib.run_operator(h)
    torch.cuda.synchronize()
    ev = lib.time_operator(h, 2, 20) * 1e3
    lib.run_operator(h); torch.cuda.synchronize()
    n = 4096 * 4 * 8
    buf = np.zeros(n, dtype=np.uint64)
    lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
    t = buf.reshape(4096, 32).astype(np.int64)
    ok = (t[:, 7] > t[:, 0]) & (t[:, 0] > 0)
    t = t[ok][:256]
    def seg(a, b): return int((t[:, b] - t[:, a]).mean())
    print(want, f"event {ev:.1f} us; blocks {len(t)}; cycles:",
          {"params issue+store": seg(0, 8), "hid fill": seg(8, 9), "input": seg(9, 10), "barrier": seg(10, 11), "pairs+barrier": seg(11, 1),
           "E0": seg(1, 2), "bar": seg(2, 3), "D0": seg(3, 4), "bar2": seg(4, 5), "P0": seg(5, 6),
           "E1": seg(16, 17), "bar1": seg(17, 18), "D1": seg(18, 19), "bar12": seg(19, 20), "P1": seg(20, 21),
           "rest": seg(21, 7), "epilogue": seg(7, 22), "total": seg(0, 22)})
net.close()
