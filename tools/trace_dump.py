"""Measurement aid (ABLATION=1 builds): run one bench layer with QNNP_GFX950_TRACE=1 and print the average
cycle deltas between the in-kernel stamps of workgroup wave 0."""
import ctypes, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QNNP_GFX950_TRACE"] = "1"
import torch, qnnpack_amd, bench
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
layer_id = int(sys.argv[1]); batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
H, W, KH, KW, S, D, G, GIC, GOC = (56, 56, 3, 3, 1, 1, 1, 64, 64) if layer_id == 99 else bench.MOBILENETV2[layer_id - 1]
layer = bench.ConvLayer(lib, torch, batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=1, min_bytes_between_reuse=512 << 20)
for _ in range(3): lib.run_operator(layer.op)
n = 4096 * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int
lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
t = buf.reshape(4096, 4, 8).astype(np.int64)
print("kernel", layer.kernel, "rc", rc)
for item in range(4):
    rows = t[:, item, :]
    ok = rows[:, 0] > 0
    if ok.sum() == 0: continue
    d = np.diff(rows[ok][:, :6], axis=1)
    print(f"item {item}: blocks {ok.sum()}  mean cycle deltas between stamps 0..5:", np.round(d.mean(axis=0)).astype(int).tolist(),
          " total", int(np.round((rows[ok][:, 5] - rows[ok][:, 0]).mean())))
first = t[:, 0, 0]; last = t[:, :, 5].max(axis=1)
ok = first > 0
print("span first-start..last-end (cycles): min start", int(first[ok].min()), " max end", int(last[ok].max()), " span", int(last[ok].max() - first[ok].min()))
