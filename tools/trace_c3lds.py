"""Measurement aid (ABLATION=1 build): cycle stamps of the LDS-staged entry-layer kernel (q8convc3.hip), every wave of every workgroup.
python tools/trace_c3lds.py [GOC]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QNNP_GFX950_TRACE"] = "1"
import torch, qnnpack_amd, bench
GOC = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
lib.set_option("gemm_kernel", 30)
layer = bench.ConvLayer(lib, torch, 128, 224, 224, 7, 7, 2, 1, 1, 3, GOC, seed=5, min_bytes_between_reuse=512 << 20)
print("kernel", layer.kernel, "event us %.2f" % (layer.time_ms(2, 10) * 1e3))
for _ in range(3): lib.run_operator(layer.op)
n = 4096 * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int
lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
full = buf.reshape(4096, 4, 8).astype(np.int64)
t = full[:, :, :5].reshape(-1, 5)
t = t[t[:, 0] > 0]
d = np.diff(t, axis=1)
names = ["entry->staged (loads + LDS writes)", "->barrier passed", "->first unit stored", "->all units stored"]
print(f"{len(t)} waves, mean / median / max cycles:")
for i, nm in enumerate(names):
    print(f"  {nm:36s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {d[:, i].max():9.0f}")
print(f"  {'entry->exit':36s} {(t[:, 4] - t[:, 0]).mean():9.0f}   span of all waves {t[:, 4].max() - t[:, 0].min()}")
