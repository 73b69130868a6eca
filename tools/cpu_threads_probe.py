"""Measurement aid: reference (compiled SSE2 QNNPACK) throughput on the host vs pthreadpool thread count."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from oracle import ref
lib = ref.lib()
N = K = 4096; M = 512
rng = np.random.default_rng(1)
w = rng.integers(0, 256, size=(N, K), dtype=np.uint8); bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
a = rng.integers(0, 256, size=M * K + 16, dtype=np.uint8); c = np.zeros(M * N, dtype=np.uint8)
op = lib.create_fully_connected_nc_q8(K, N, 127, 0.5, 127, 0.5, w, bias, 127, 0.5, 0, 255)
lib.setup_fully_connected_nc_q8(op, M, a[8:], K, c, N)
for t in [8, 16, 32, 64, 128, 256]:
    if t > (os.cpu_count() or 1): break
    pool = lib.threadpool(t)
    lib.run_operator(op, pool)
    it, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        lib.run_operator(op, pool); it += 1
    dt = time.perf_counter() - t0
    lib.destroy_threadpool(pool)
    print(f"gemm threads={t}: {2.0 * M * N * K * it / dt / 1e12:.4f} TOPS")
for t in [8, 16, 32, 64, 128, 256]:
    if t > (os.cpu_count() or 1): break
    r = bench.cpu_baseline_sweep(batch=16, seconds_budget=2.0, threads=t)
    print(f"sweep threads={t}: {r['images_per_s']} img/s")
