"""Measurement aid (ABLATION=1 builds): run one bench layer with QNNP_GFX950_TRACE=1 and print the average
cycle deltas between the in-kernel stamps of workgroup wave 0."""
import ctypes, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["QNNP_GFX950_TRACE"] = "1"
import torch, qnnpack_amd, bench
lib = qnnpack_amd.load(); lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
layer_id = int(sys.argv[1]); batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
if layer_id == 0:
    # headline GEMM 4096^3
    M = N = K = 4096
    rng = np.random.default_rng(1)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8); bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    a = torch.randint(0, 256, (M * K,), dtype=torch.uint8, device="cuda"); c = torch.empty(M * N, dtype=torch.uint8, device="cuda")
    lib.set_option("gemm_kernel", int(os.environ.get("GEMM_KERNEL", "0")))
    op = lib.create_fully_connected_nc_q8(K, N, 127, 0.75, 127, 1.0, w, bias, 127, 1.0, 1, 254)
    lib.setup_fully_connected_nc_q8(op, M, a, K, c, N)
    for _ in range(5): lib.run_operator(op)
    print("hipEvent avg per launch, 1 launch timed: %.2f us; 30 back-to-back: %.2f us" % (
        lib.time_operator(op, 2, 1) * 1e3, lib.time_operator(op, 2, 30) * 1e3))
    lib.run_operator(op)
    n = 4096 * 4 * 8
    buf = np.zeros(n, dtype=np.uint64)
    lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int; lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
    t = buf.reshape(4096, 4, 8).astype(np.int64)[:256]
    d0 = np.diff(t[:, 0, :5], axis=1)
    print("gemm256: mean cycles [prologue, main loop, rowsum+bias, epilogue]:", np.round(d0.mean(axis=0)).astype(int).tolist(), "total", int((t[:, 0, 4] - t[:, 0, 0]).mean()))
    wall = (t[:, 3, 1] - t[:, 3, 0]).mean() * 10.0   # ns
    print(f"gemm256: wave-0 lifetime {wall:.0f} ns wall -> shader clock {(t[:, 0, 4] - t[:, 0, 0]).mean() / wall:.3f} GHz during the kernel")
    tw = buf.reshape(4096, 4, 8).astype(np.int64)[1024:1280]
    if (tw[:, :, 0] > 0).all():
        base = tw[:, :, 0].min(axis=1)[:, None, None]
        rel = (tw - base).mean(axis=0)
        print("per-wave stamps rel. to the earliest wave's first stamp [p1 end, vmcnt done, barrier done, p2 end] x tiles 20,21:")
        for w_ in range(4): print("  wave", w_, np.round(rel[w_]).astype(int).tolist())
    w0 = t[:, 3, 0] * 10; w1 = t[:, 3, 1] * 10
    print(f"wall (ns): first start 0, last start {w0.max() - w0.min()}, first end {w1.min() - w0.min()}, last end {w1.max() - w0.min()}")
    order = np.argsort(w0)
    print("start offsets (ns) by block, sorted sample:", (w0[order] - w0.min())[::16].tolist())
    print("block ids in start order sample:", order[::16].tolist())
    print("end offsets (ns) sample:", np.sort(w1 - w0.min())[::16].tolist())
    print("start skew across blocks (cycles):", int(t[:, 0, 0].max() - t[:, 0, 0].min()), " end skew:", int(t[:, 0, 4].max() - t[:, 0, 4].min()))
    sys.exit(0)
lib.set_option("dwconv_kernel", int(os.environ.get("DW_KERNEL", "0")))
lib.set_option("gemm_kernel", int(os.environ.get("GEMM_KERNEL", "0")))
H, W, KH, KW, S, D, G, GIC, GOC = (56, 56, 3, 3, 1, 1, 1, 64, 64) if layer_id == 99 else bench.MOBILENETV2[layer_id - 1]
layer = bench.ConvLayer(lib, torch, batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=1, min_bytes_between_reuse=512 << 20)
for _ in range(3): lib.run_operator(layer.op)
n = 4096 * 4 * 8
buf = np.zeros(n, dtype=np.uint64)
lib.lib.qnnp_hip_trace_dump.restype = ctypes.c_int
lib.lib.qnnp_hip_trace_dump.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.lib.qnnp_hip_trace_dump(buf.ctypes.data, n)
t = buf.reshape(4096, 4, 8).astype(np.int64)
print("kernel", layer.kernel, "rc", rc, "event ms", lib.time_operator(layer.op, 2, 10))
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
w = t[:, 3, :2] * 10   # ns (100 MHz wall clock), kernel G: item 3 = [start, end] of wave 0 of each workgroup
okw = w[:, 0] > 0
if okw.sum():
    w = w[okw]; base = w[:, 0].min()
    life = w[:, 1] - w[:, 0]
    print(f"wall: {okw.sum()} workgroups; starts 0..{int(w[:,0].max()-base)} ns (median {int(np.median(w[:,0])-base)}); ends {int(w[:,1].min()-base)}..{int(w[:,1].max()-base)} ns; "
          f"lifetime min/median/max {int(life.min())}/{int(np.median(life))}/{int(life.max())} ns")
    cyc = (t[okw][:, 0, 5] - t[okw][:, 0, 0]); print("shader clock during the kernel: %.2f GHz" % (cyc.mean() / life.mean()))
    hw = t[okw][:, 2, 0]; xcc = t[okw][:, 2, 1] & 0xF
    if hw.any():
        import collections
        cu_key = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF)      # (XCC, SE, SH, CU)
        per_cu = collections.Counter(cu_key.tolist())
        hist = collections.Counter(per_cu.values())
        print("workgroups per CU -> number of CUs:", dict(sorted(hist.items())), " distinct CUs:", len(per_cu))
        n_on_cu = np.array([per_cu[k] for k in cu_key.tolist()])
        for k in sorted(hist):
            sel = n_on_cu == k
            print(f"  CUs holding {k} workgroups: lifetime min/median/max {int(life[sel].min())}/{int(np.median(life[sel]))}/{int(life[sel].max())} ns, last end {int((w[sel][:,1]-base).max())} ns")
        # inside one CU: lifetimes in start order (does the oldest wave finish first?)
        ranks = collections.defaultdict(list)
        for key in per_cu:
            idx = np.where(cu_key == key)[0]
            order = idx[np.argsort(w[idx, 0])]
            for r, i in enumerate(order): ranks[r].append(life[i])
        print("  mean lifetime by arrival order on the CU:", {r: int(np.mean(v)) for r, v in sorted(ranks.items())})
        print("  xcc of block b == b % 8 for", int((xcc == (np.where(okw)[0] % 8)).sum()), "of", int(okw.sum()), "blocks")
