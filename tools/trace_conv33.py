"""Measurement aid (ABLATION=1 build, QNNP_GFX950_LIBRARY pointing at it): cycle stamps of the weight-stationary 3x3
kernel on BASELINE configs[2] -- the prologue of wave 0 of every workgroup (entry -> first patch requested -> weights
requested -> patch in LDS -> barrier passed -> weights in registers -> loop done), the unit loop's phases, and the 100 MHz
wall clock at entry / exit (dispatch skew, lifetime, shader clock).   python tools/trace_conv33.py [batch]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QNNP_GFX950_TRACE"] = "1"
import torch, qnnpack_amd, bench
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
layer = bench.ConvLayer(lib, torch, batch, 56, 56, 3, 3, 1, 1, 1, 64, 64, seed=1, min_bytes_between_reuse=512 << 20)
print("kernel", layer.kernel, "event us %.2f" % (layer.time_ms(2, 10) * 1e3))
for _ in range(3): lib.run_operator(layer.op)
n = 4096 * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int
lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
t = buf.reshape(4096, 4, 8).astype(np.int64)
pro = t[256:512, 0, :7]
ok = pro[:, 0] > 0
pro = pro[ok]
d = np.diff(pro, axis=1)
names = ["entry->weights requested", "->patch requested", "->patch in LDS", "->barrier passed", "->weights in registers", "->loop done"]
print(f"{ok.sum()} workgroups, wave 0, mean / median / max cycles:")
for i, nm in enumerate(names):
    print(f"  {nm:26s} {d[:, i].mean():9.0f} {np.median(d[:, i]):9.0f} {d[:, i].max():9.0f}")
print(f"  {'entry->loop done':26s} {(pro[:, 6] - pro[:, 0]).mean():9.0f}")
for item in range(4):
    rows = t[:256, item, :6][ok]
    good = rows[:, 0] > 0
    if good.sum():
        dd = np.diff(rows[good], axis=1)
        print(f"unit {item}: [acc init, K loop, epilogue, -, fix-up] mean cycles", np.round(dd.mean(axis=0)).astype(int).tolist(), "total", int((rows[good][:, 5] - rows[good][:, 0]).mean()))
wall = t[256:512, 1, :2][ok] * 10          # ns
base = wall[:, 0].min()
life = wall[:, 1] - wall[:, 0]
print(f"wall (ns): entries 0..{int(wall[:, 0].max() - base)} (median {int(np.median(wall[:, 0]) - base)}); exits {int(wall[:, 1].min() - base)}..{int(wall[:, 1].max() - base)}; "
      f"lifetime min/median/max {int(life.min())}/{int(np.median(life))}/{int(life.max())}")
print("shader clock over wave 0's life: %.2f GHz" % ((pro[:, 6] - pro[:, 0]).mean() / life.mean()))
