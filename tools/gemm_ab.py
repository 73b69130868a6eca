#!/usr/bin/env python3
"""Same-process, interleaved A/B of the q8gemm 4096^3 kernel structures (guide rule 24: N variants x M rounds in ONE
process, report the distribution).   python tools/gemm_ab.py [--variants 0,10,11,4] [--rounds 5] [--only V]
Each variant = one qnnp_fully_connected_nc_q8 operator created under `gemm_kernel` = V on the same random operands;
a round times each variant as a 32-launch hipGraph replayed for >= 150 ms. --only V runs just that variant for a few
hundred launches (the form the rocprofv3 --pmc passes profile)."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NAMES = {0: "shipped 256x256 (8 waves, interleaved)", 2: "256x256 forced", 4: "256x256 4 waves 128x128/wave",
         10: "128x256 x 2 workgroups per CU", 11: "256x256 ping-pong + setprio",
         16: "256x256 4 waves, lean", 15: "256x256 lean (saddr DMA, ring unrolled)",
         20: "centred, 32x32x32 MFMA", 21: "centred, fragment reads in one burst",
         23: "centred, 16x16x64 MFMA (round 6)", 28: "standard image + row sums, 16x16x64 MFMA (round 6)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,10,11,4")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--only", type=int, default=-1)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--c16-opts", default="", help="measurement builds: time variant 23 once per QNNP_C16_OPT value of this comma list")
    ap.add_argument("--kzp", type=int, default=127, help="kernel zero point (126: no centred image -> lean / row-sum flavours)")
    args = ap.parse_args()
    import torch
    import qnnpack_amd
    lib = qnnpack_amd.load()
    torch.cuda.set_device(0); torch.zeros(1, device="cuda")
    lib.initialize(); lib.set_stream(torch.cuda.current_stream().cuda_stream)
    M = N = K = args.size
    rng = np.random.default_rng(0x51A0 + 2)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    gen = torch.Generator(device="cuda"); gen.manual_seed(0x51A0)
    a = torch.randint(0, 256, (M * K,), dtype=torch.uint8, device="cuda", generator=gen)
    variants = [args.only] if args.only >= 0 else [int(v) for v in args.variants.split(",")]
    ops, outs = {}, {}
    for v in variants:
        lib.set_option("gemm_kernel", v)
        op = lib.create_fully_connected_nc_q8(K, N, 127, 0.75, args.kzp, 1.0, w, bias, 127, 1.0, 1, 254)
        outs[v] = torch.empty(M * N, dtype=torch.uint8, device="cuda")
        lib.setup_fully_connected_nc_q8(op, M, a, K, outs[v], N)
        lib.run_operator(op)
        ops[v] = op
    lib.set_option("gemm_kernel", 0)
    torch.cuda.synchronize()
    ref = outs[variants[0]]
    for v in variants[1:]:
        assert torch.equal(outs[v], ref), f"variant {v} differs from variant {variants[0]}"
    lib.set_async(True)
    if args.only >= 0:
        for _ in range(300):
            lib.run_operator(ops[args.only])
        torch.cuda.synchronize()
        print(json.dumps({"only": args.only, "kernel": lib.operator_kernel(ops[args.only])}))
        return
    if args.c16_opts:
        # the same operator (code 23), one graph per value of QNNP_C16_OPT: the launcher reads it when the launches are captured
        base = ops[variants[0]]
        keys = [f"opt{o}" for o in args.c16_opts.split(",")]
        for key, o in zip(keys, args.c16_opts.split(",")):
            os.environ["QNNP_C16_OPT"] = o
            outs[key] = torch.empty(M * N, dtype=torch.uint8, device="cuda")
            lib.setup_fully_connected_nc_q8(base, M, a, K, outs[key], N)
            lib.run_operator(base); torch.cuda.synchronize()
            assert torch.equal(outs[key], ref), key
            ops[key] = base
        variants = keys
        NAMES.update({k: "c16 with QNNP_C16_OPT=" + k[3:] for k in keys})
    graphs = {}
    for v in variants:
        if args.c16_opts:
            os.environ["QNNP_C16_OPT"] = v[3:]
        lib.graph_begin()
        for _ in range(32):
            lib.run_operator(ops[v])
        graphs[v] = lib.graph_end()
    times = {v: [] for v in variants}
    for v in variants:                       # sustained-clock warm-up
        lib.graph_time(graphs[v], 2, 40)
    for r in range(args.rounds):
        for v in (variants if r % 2 == 0 else variants[::-1]):
            times[v].append(lib.graph_time(graphs[v], 1, 80) / 32.0)
    ops_count = 2.0 * M * N * K
    rows = []
    for v in variants:
        t = sorted(times[v]); med = t[len(t) // 2]
        rows.append({"gemm_kernel": v, "structure": NAMES.get(v, "?"), "kernel": lib.operator_kernel(ops[v]),
                     "us_median": round(med * 1e3, 2), "us_min": round(t[0] * 1e3, 2), "us_max": round(t[-1] * 1e3, 2),
                     "tops_median": round(ops_count / (med * 1e-3) / 1e12, 1),
                     "frac_of_5033": round(ops_count / (med * 1e-3) / 1e12 / 5033.0, 4)})
        print(json.dumps(rows[-1]), flush=True)
    for v in variants:
        lib.graph_destroy(graphs[v]); lib.delete_operator(ops[v])


if __name__ == "__main__":
    main()
