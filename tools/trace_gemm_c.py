"""Measurement aid (ABLATION=1 build, QNNP_GFX950_LIBRARY pointing at it): in-kernel cycle stamps of the centred GEMM
(hip/q8gemm256c.hip, QNNP_C_STAMP) for one launch of the 4096^3 problem: per-segment shader cycles of wave 0 and wave 4
of every workgroup, the wall-clock lifetime of the wave and of the whole grid, and the clock that implies.
   GEMM_KERNEL=20 QNNP_GFX950_ABLATE=0 python tools/trace_gemm_c.py"""
import ctypes, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QNNP_GFX950_TRACE"] = "1"
import torch, qnnpack_amd
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
M = N = K = 4096
rng = np.random.default_rng(1)
w = rng.integers(0, 256, size=(N, K), dtype=np.uint8); bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
a = torch.randint(0, 256, (M * K,), dtype=torch.uint8, device="cuda"); c = torch.empty(M * N, dtype=torch.uint8, device="cuda")
lib.set_option("gemm_kernel", int(os.environ.get("GEMM_KERNEL", "20")))
op = lib.create_fully_connected_nc_q8(K, N, 127, 0.75, 127, 1.0, w, bias, 127, 1.0, 1, 254)
lib.setup_fully_connected_nc_q8(op, M, a, K, c, N)
for _ in range(40): lib.run_operator(op)          # sustained clock
ev = lib.time_operator(op, 2, 30) * 1e3
for _ in range(20): lib.run_operator(op)
torch.cuda.synchronize()
n = 4096 * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int; lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
t = buf.reshape(4096, 4, 8).astype(np.int64)[:256]
names = ["prologue(wait tile0)", "tile 0", "steady", "last fetch + final sync", "tail", "epilogue half 1", "epilogue half 2"]
out = {"kernel": lib.operator_kernel(op), "ablate": os.environ.get("QNNP_GFX950_ABLATE", "0"), "event_us": round(ev, 2)}
for item, nm in ((0, "wave0"), (1, "wave4")):
    d = np.diff(t[:, item, :8], axis=1)
    out[nm] = {k: int(v) for k, v in zip(names, np.round(d.mean(axis=0)))}
    out[nm]["total"] = int((t[:, item, 7] - t[:, item, 0]).mean())
for item, nm in ((2, "wave0"), (3, "wave4")):
    wall = (t[:, item, 1] - t[:, item, 0]) * 10.0          # ns: start .. stores acknowledged
    out[nm]["lifetime_ns(start..stores acked)"] = int(wall.mean())
    out[nm]["clock_GHz(stamp 0..7 / lifetime)"] = round(float((t[:, item - 2, 7] - t[:, item - 2, 0]).mean() / wall.mean()), 3)
w0 = t[:, 2, 0] * 10; w1 = np.maximum(t[:, 2, 1], t[:, 3, 1]) * 10
out["grid_ns"] = {"last start": int(w0.max() - w0.min()), "first end": int(w1.min() - w0.min()), "last end": int(w1.max() - w0.min())}
print(json.dumps(out))
