"""Measurement aid (ABLATION=1 build): cycle stamps of the patch kernel (q8convpatch.hip), wave 0 of every workgroup.
python tools/trace_patch.py H W K S G GIC GOC"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QNNP_GFX950_TRACE"] = "1"
import torch, qnnpack_amd, bench
H, W, K, S, G, GIC, GOC = (int(x) for x in sys.argv[1:8])
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
layer = bench.ConvLayer(lib, torch, 128, H, W, K, K, S, 1, G, GIC, GOC, seed=5, min_bytes_between_reuse=512 << 20)
print("kernel", layer.kernel, "event us %.2f" % (layer.time_ms(2, 10) * 1e3))
for _ in range(3): lib.run_operator(layer.op)
n = 4096 * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int
lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
full = buf.reshape(4096, 4, 8).astype(np.int64)
t = full[:, 0, :6]
ok = t[:, 0] > 0
t = t[ok]
w = full[:, 1, :6][ok] * 10          # ns
d = np.diff(t, axis=1)
names = ["entry->requests issued", "->patch landed", "->patch re-centred", "->K loop done", "->stores issued"]
print(f"{ok.sum()} workgroups, wave 0, mean / median / max cycles:")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {d[:, i].max():9.0f}")
print(f"  {'entry->stores issued':26s} {(t[:, 5] - t[:, 0]).mean():9.0f}   span of all workgroups {t[:, 5].max() - t[:, 0].min()}")
base = w[:, 0].min()
print("wall clock (ns after the first workgroup's entry), mean over workgroups:", [int(x) for x in (w - base).mean(axis=0)],
      f"; entries 0..{int(w[:, 0].max() - base)}, last exit {int(w[:, 5].max() - base)}")
print("shader clock over wave 0's life: %.2f GHz" % ((t[:, 5] - t[:, 0]).mean() / (w[:, 5] - w[:, 0]).mean()))
