#!/usr/bin/env python3
"""bench.py -- throughput of the q8 conv/GEMM hot path on MI355X, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: one rank per GPU. Under torch.distributed.run the ranks come from RANK/LOCAL_RANK/WORLD_SIZE; started
     plainly, bench.py re-launches ITSELF under torch.distributed.run with N ranks on 127.0.0.1. Fewer than N
     visible GPUs, or a WORLD_SIZE that disagrees with --gpus, is an error -- never a silent 1-GPU run.)

Headline (BASELINE.json `metric`, configs[1]): int8 TOPS of q8gemm M=N=K=4096 (uint8 in, int32 MFMA
accumulate, fused Q31 requantize, uint8 out) through qnnp_*_fully_connected_nc_q8. A "step" is one
whole GEMM launch. ops = 2*M*N*K, the reference's accounting (bench/q8gemm.cc:108).
At N > 1 every rank runs its own replica of the GEMM (no batch dimension to shard: "replicas only",
DESIGN.md) -> weak scaling, value = sum over ranks.

Secondary numbers travel in the same JSON line under "extra" (not separate bench lines):
  * q8conv 3x3 56x56x64->64 batch 128 (configs[2]),
  * the MobileNetV2 depthwise layers (configs[3]) as HBM GB/s,
  * the 31-layer MobileNetV2 conv sweep (configs[4], bench/convolution.cc:453-536) as images/s with the
    batch sharded across ranks (no collective), each layer timed as its own operator like the reference bench,
  * "mobilenetv2_network": the WHOLE network (examples/mobilenetv2.py: 52 convolutions, 10 residual adds, global
    average pooling, classifier = 64 chained operators) as one hipGraph replay, images/s; "..._adds_folded": the
    same with the residual adds carried by the project convolutions (54 launches),
  * "next_rows": the operators SURVEY.md section 8f ranks after the hot path (deconvolution, add, pooling),
  * "q8gemm_4096_variants": the headline GEMM with a kernel zero point that has no centred image and with a shift >= 1
    requantization scale; "q8fc_m1_k1024_n1000": configs[0] on the device,
  * "conv_lists": the reference bench's ResNet-18 / ResNet-50 / ShuffleNet-v1-g2 lists (bench/convolution.cc:642-718,
    147-184) through whatever kernel auto picks -- the general implicit-GEMM convolution path.
Because the driver's record keeps `config`, `roofline` and `cpu_baseline` but drops `extra`, the figures of BASELINE's
other configs are repeated as flat scalars in `roofline.secondary` (secondary_block()).

The timed region is EXACTLY K steps between barriers (wall clock, max over ranks -> `value`), bracketed on the
launch stream by HIP events as well (-> "roofline", same launches); the K launches are captured into one hipGraph before the
region and replayed inside it (K kernels back to back; K separate launches if capture is unavailable). It directly follows >= 1 s of the same GEMM
replayed as a hipGraph (five batches of >= 200 ms, median reported as roofline.sustained_launch_ms), so the chip is in its
sustained clock / power state, not in a boost burst. "cpu_baseline" times the reference's own SSE2 path (oracle/_ref, built from
the reference sources) on this box's host cores on a bounded sample -- rank 0, N = 1 only.
Inputs are synthetic uniform-random uint8 already resident in HBM when the timed region starts.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# chip ceilings, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_I8_TOPS = 256 * 4 * 1024 * 2 * 2.4e9 / 1e12   # 5033: 256 CU x 4 SIMD x 1024 MAC/clk x 2.4 GHz (i8 = 2x the 2.5 PF bf16 dense rate)
PEAK_HBM_GBS = 8000.0                               # HBM3E spec; ~6300 achievable

# bench/convolution.cc:453-536 (the 31 active rows): N is replaced by the per-GPU batch
#                 H    W   KH KW S  D   G   GCin  GCout
MOBILENETV2 = [(224, 224, 3, 3, 2, 1,   1,    3,   32),
               (112, 112, 3, 3, 1, 1,  32,    1,    1), (112, 112, 1, 1, 1, 1, 1,  32,  16),
               (112, 112, 1, 1, 1, 1,   1,   16,   96), (112, 112, 3, 3, 2, 1, 96,  1,   1),
               (56, 56, 1, 1, 1, 1, 1,  96,  24), (56, 56, 1, 1, 1, 1, 1,  24, 144),
               (56, 56, 3, 3, 1, 1, 144, 1,   1), (56, 56, 1, 1, 1, 1, 1, 144,  24),
               (56, 56, 3, 3, 2, 1, 144, 1,   1), (28, 28, 1, 1, 1, 1, 1, 144,  32),
               (28, 28, 1, 1, 1, 1, 1,  32, 192), (28, 28, 3, 3, 1, 1, 192, 1,   1),
               (28, 28, 1, 1, 1, 1, 1, 192,  32), (28, 28, 3, 3, 2, 1, 192, 1,   1),
               (14, 14, 1, 1, 1, 1, 1, 192,  64), (14, 14, 1, 1, 1, 1, 1,  64, 384),
               (14, 14, 3, 3, 1, 1, 384, 1,   1), (14, 14, 1, 1, 1, 1, 1, 384,  64),
               (14, 14, 1, 1, 1, 1, 1, 384,  96), (14, 14, 1, 1, 1, 1, 1,  96, 576),
               (14, 14, 3, 3, 1, 1, 576, 1,   1), (14, 14, 1, 1, 1, 1, 1, 576,  96),
               (14, 14, 3, 3, 2, 1, 576, 1,   1), (7, 7, 1, 1, 1, 1, 1, 576, 160),
               (7, 7, 1, 1, 1, 1, 1, 160, 960), (7, 7, 3, 3, 1, 1, 960, 1,   1),
               (7, 7, 1, 1, 1, 1, 1, 960, 160), (7, 7, 1, 1, 1, 1, 1, 960, 320),
               (7, 7, 1, 1, 1, 1, 1, 320, 1280), (1, 1, 1, 1, 1, 1, 1, 1280, 1000)]


# bench/convolution.cc:642-718 -- the dense convolutions that reach the implicit-GEMM family (7x7 s2, 3x3 with 64..512
# channels, stride-2 3x3 and 1x1). Same columns; rows the reference comments out are left out here too.
RESNET18 = [(224, 224, 7, 7, 2, 1, 1, 3, 64), (56, 56, 3, 3, 1, 1, 1, 64, 64),
            (56, 56, 3, 3, 2, 1, 1, 64, 128), (28, 28, 3, 3, 1, 1, 1, 128, 128), (56, 56, 1, 1, 2, 1, 1, 64, 128),
            (28, 28, 3, 3, 2, 1, 1, 128, 256), (14, 14, 3, 3, 1, 1, 1, 256, 256), (28, 28, 1, 1, 2, 1, 1, 128, 256),
            (14, 14, 3, 3, 2, 1, 1, 256, 512), (7, 7, 3, 3, 1, 1, 1, 512, 512), (14, 14, 1, 1, 2, 1, 1, 256, 512)]
RESNET50 = [(224, 224, 7, 7, 2, 1, 1, 3, 64),
            (56, 56, 1, 1, 1, 1, 1, 64, 64), (56, 56, 3, 3, 1, 1, 1, 64, 64), (56, 56, 1, 1, 1, 1, 1, 64, 256),
            (56, 56, 1, 1, 1, 1, 1, 256, 64),
            (56, 56, 1, 1, 1, 1, 1, 256, 128), (56, 56, 3, 3, 2, 1, 1, 128, 128), (28, 28, 1, 1, 1, 1, 1, 128, 512),
            (56, 56, 1, 1, 2, 1, 1, 256, 512),
            (28, 28, 1, 1, 1, 1, 1, 512, 128), (28, 28, 3, 3, 1, 1, 1, 128, 128),
            (28, 28, 1, 1, 1, 1, 1, 512, 256), (28, 28, 3, 3, 2, 1, 1, 256, 256), (14, 14, 1, 1, 1, 1, 1, 256, 1024),
            (28, 28, 1, 1, 2, 1, 1, 512, 1024),
            (14, 14, 1, 1, 1, 1, 1, 1024, 256), (14, 14, 3, 3, 1, 1, 1, 256, 256),
            (14, 14, 1, 1, 1, 1, 1, 1024, 512), (14, 14, 3, 3, 2, 1, 1, 512, 512), (7, 7, 1, 1, 1, 1, 1, 512, 2048),
            (14, 14, 1, 1, 2, 1, 1, 1024, 2048),
            (7, 7, 1, 1, 1, 1, 1, 2048, 512), (7, 7, 3, 3, 1, 1, 1, 512, 512)]
# bench/convolution.cc:147-184 -- ShuffleNet v1 with 2 groups: grouped 1x1 (25 / 50 / 100 channels per group), depthwise s2
SHUFFLENET_V1_G2 = [(224, 224, 3, 3, 2, 1, 1, 3, 24),
                    (56, 56, 1, 1, 1, 1, 1, 24, 50), (56, 56, 3, 3, 2, 1, 50, 1, 1), (28, 28, 1, 1, 1, 1, 2, 25, 88),
                    (28, 28, 1, 1, 1, 1, 2, 100, 25), (28, 28, 3, 3, 2, 1, 50, 1, 1), (28, 28, 1, 1, 1, 1, 2, 25, 100),
                    (28, 28, 1, 1, 1, 1, 2, 100, 50), (28, 28, 3, 3, 2, 1, 100, 1, 1), (14, 14, 1, 1, 1, 1, 2, 50, 100),
                    (14, 14, 1, 1, 1, 1, 2, 200, 50), (14, 14, 3, 3, 2, 1, 100, 1, 1), (14, 14, 1, 1, 1, 1, 2, 50, 200),
                    (14, 14, 1, 1, 1, 1, 2, 200, 100), (14, 14, 3, 3, 2, 1, 200, 1, 1), (7, 7, 1, 1, 1, 1, 2, 100, 200),
                    (7, 7, 1, 1, 1, 1, 2, 400, 100), (7, 7, 3, 3, 2, 1, 200, 1, 1), (7, 7, 1, 1, 1, 1, 2, 100, 400)]


def conv_geometry(H, W, KH, KW, S, D):
    """Padding and output size exactly as bench/convolution.cc:37-47 computes them."""
    eh, ew = (KH - 1) * D + 1, (KW - 1) * D + 1
    pl, pt = ew // 2, eh // 2
    pr, pb = ew - 1 - pl, eh - 1 - pt
    oh = (pt + H + pb - eh) // S + 1
    ow = (pl + W + pr - ew) // S + 1
    return (pt, pr, pb, pl), oh, ow


class ConvLayer:
    """One qnnp convolution operator with device-resident synthetic tensors (rotating buffer sets)."""

    def __init__(self, lib, torch, batch, H, W, KH, KW, S, D, G, GIC, GOC, seed, min_bytes_between_reuse=0, out_scale=0.5, kzp=127):
        self.lib = lib
        (pt, pr, pb, pl), oh, ow = conv_geometry(H, W, KH, KW, S, D)
        rng = np.random.default_rng(seed)
        kernel = rng.integers(0, 256, size=(G, GOC, KH, KW, GIC), dtype=np.uint8)
        bias = rng.integers(-10000, 10001, size=G * GOC, dtype=np.int32)
        # quantization parameters of the reference bench (bench/convolution.cc:71-74)
        self.op = lib.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                                   127, 0.5, kzp, 0.5, kernel, bias, 127, out_scale, 0, 255, 0)
        self.batch, self.H, self.W = batch, H, W
        self.cin, self.cout = G * GIC, G * GOC
        self.in_bytes = batch * H * W * self.cin
        self.out_bytes = batch * oh * ow * self.cout
        self.ops = 2 * batch * oh * ow * G * GIC * GOC * KH * KW          # bench/convolution.cc:100-104
        nsets = 1
        if min_bytes_between_reuse:
            nsets = max(1, -(-min_bytes_between_reuse // (self.in_bytes + self.out_bytes)))
            nsets = min(nsets, 64)
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        self.inputs = [torch.randint(0, 256, (self.in_bytes,), dtype=torch.uint8, device="cuda", generator=gen)
                       for _ in range(nsets)]
        self.outputs = [torch.empty(self.out_bytes, dtype=torch.uint8, device="cuda") for _ in range(nsets)]
        lib.setup_convolution2d_nhwc_q8(self.op, batch, H, W, self.inputs[0], self.cin, self.outputs[0], self.cout)
        lib.run_operator(self.op)       # also resolves kernel_name
        self.kernel = lib.operator_kernel(self.op)

    def time_ms(self, warmup, iters):
        return self.lib.time_operator_rotating(self.op, self.inputs, self.outputs, warmup, iters)

    def close(self):
        self.lib.delete_operator(self.op)
        self.inputs = self.outputs = None


def network_bench(lib, torch, batch, total_batch, world, warmup, iters, fuse=False, fold_adds=False):
    """Every operator of a real quantized MobileNetV2 forward pass (examples/mobilenetv2.py: 52 convolutions, 10 residual
    adds, global average pooling, classifier), chained on device buffers with their true dependencies and replayed as one
    hipGraph. Unlike the 31-shape sweep each tensor is produced by the previous operator, so it may still sit in the
    256 MB Infinity Cache when it is consumed."""
    from examples import mobilenetv2 as mnv2
    from qnnpack_amd.shard import job_time_ms
    plan = mnv2.build_plan()
    net = mnv2.DeviceNetwork(lib, torch, plan, batch, fuse=fuse, fold_adds=fold_adds)
    try:
        gen = torch.Generator(device="cuda")
        gen.manual_seed(91)
        net.buffers[0].copy_(torch.randint(0, 256, (net.buffers[0].numel(),), dtype=torch.uint8, device="cuda", generator=gen))
        net.run()
        net.capture()
        ms = net.time_ms(max(warmup, 2), iters)
        job_ms = job_time_ms(ms, world)
        act = mnv2.algorithmic_bytes(plan, batch)
        return {"operators": len(plan.ops), "launches": len(net.schedule), "fused_blocks": len(net.fused),
                "adds_in_conv_epilogues": len(net.folded),
                "images_per_s": round(total_batch / (job_ms * 1e-3), 1),
                "batch_per_gpu": batch, "ms_per_batch": round(job_ms, 4), "timed_as": "one hipGraph replay of the chained operators",
                "activation_gbs": round(act / (ms * 1e-3) / 1e9, 1), "tops": round(mnv2.operations(plan, batch) / (ms * 1e-3) / 1e12, 2),
                "kernels": sorted(set(net.kernels.values()))}
    finally:
        net.close()


def layer_bound_ms(layer, weight_bytes=0):
    """The roofline of one layer: max(ops at the dense int8 MFMA peak, activation (+ weight) bytes at the HBM peak)."""
    return max(layer.ops / (PEAK_I8_TOPS * 1e12), (layer.in_bytes + layer.out_bytes + weight_bytes) / (PEAK_HBM_GBS * 1e9)) * 1e3


def conv_list_bench(lib, torch, batch, shapes, seed0, warmup=2, iters=8, out_scale=0.5):
    """One reference shape list (bench/convolution.cc) at `batch` images, every row its own operator on rotating buffers
    (> 512 MB between reuses), as the reference bench times them: per layer the kernel auto chose, time, TOP/s and the
    fraction of max(MFMA, HBM) roofline; per list the sum of layer times as images/s."""
    rows, total_ms, total_bound = [], 0.0, 0.0
    for i, (H, W, KH, KW, S, D, G, GIC, GOC) in enumerate(shapes):
        layer = ConvLayer(lib, torch, batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=seed0 + i, min_bytes_between_reuse=512 << 20,
                          out_scale=out_scale)
        ms = layer.time_ms(warmup, iters)
        wbytes = G * GOC * KH * KW * GIC
        bound = layer_bound_ms(layer, wbytes)
        total_ms += ms
        total_bound += bound
        rows.append({"shape": [H, W, KH, S, G, GIC, GOC], "kernel": layer.kernel, "us": round(ms * 1e3, 2),
                     "tops": round(layer.ops / (ms * 1e-3) / 1e12, 1),
                     "gbs": round((layer.in_bytes + layer.out_bytes) / (ms * 1e-3) / 1e9, 1),
                     "bound": "mfma" if layer.ops / (PEAK_I8_TOPS * 1e12) * 1e3 >= bound else "hbm",
                     "frac_of_bound": round(bound / ms, 3)})
        layer.close()
    dense3 = [r for r in rows if r["shape"][2] == 3 and r["shape"][4] == 1 and r["shape"][5] >= 64]
    return {"batch": batch, "images_per_s_by_sum_of_layers": round(batch / (total_ms * 1e-3), 1),
            "sum_of_layer_ms": round(total_ms, 4), "sum_of_bounds_ms": round(total_bound, 4),
            "frac_of_bound": round(total_bound / total_ms, 3),
            "worst_dense_3x3_frac": min((r["frac_of_bound"] for r in dense3), default=None),
            "layers": rows}


def reference_lists_bench(lib, torch, batch, skip=("MobileNetV2", "ResNet18", "ResNet50", "ShuffleNetV1G2"), seconds_budget=240.0):
    """Every OTHER shape list of the reference's convolution benchmark (bench/convolution.cc:108-942: ShuffleNet v1 g1/g3/g4/g8 and
    v2, MobileNet v1, SqueezeNet 1.0 / 1.1, VGG, the three depthwise lists; the table is the committed fixture
    tests/golden/reference_bench_shapes.json) at `batch` images: each DISTINCT shape timed once on rotating buffers as
    conv_list_bench does, each list summed over its rows as the reference bench runs them (a repeated row counts every time).
    Rows are compact: [H, W, KH, S, D, G, GCin, GCout, kernel, us, fraction of max(MFMA, HBM) bound]."""
    path = os.path.join(ROOT, "tests", "golden", "reference_bench_shapes.json")
    table = json.load(open(path))["lists"]
    cache, out, t0 = {}, {}, time.perf_counter()
    for name, shapes in table.items():
        if name in skip:
            continue
        rows, total_ms, total_bound, partial = [], 0.0, 0.0, False
        for shape in shapes:
            key = tuple(shape)
            if key not in cache:
                if time.perf_counter() - t0 > seconds_budget:
                    partial = True
                    continue
                H, W, KH, KW, S, D, G, GIC, GOC = key
                layer = ConvLayer(lib, torch, batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=9000 + len(cache),
                                  min_bytes_between_reuse=512 << 20)
                ms = layer.time_ms(2, 6)
                cache[key] = (layer.kernel, ms, layer_bound_ms(layer, G * GOC * KH * KW * GIC))
                layer.close()
            kernel, ms, bound = cache[key]
            H, W, KH, KW, S, D, G, GIC, GOC = key
            total_ms += ms
            total_bound += bound
            rows.append([H, W, KH, S, D, G, GIC, GOC, kernel.replace("q8_", ""), round(ms * 1e3, 2), round(bound / ms, 3)])
        out[name] = {"rows": len(shapes), "timed_rows": len(rows), "partial": partial,
                     "images_per_s_by_sum_of_layers": round(batch / (total_ms * 1e-3), 1) if total_ms and not partial else None,
                     "sum_of_layer_ms": round(total_ms, 4), "frac_of_bound": round(total_bound / total_ms, 3) if total_ms else None,
                     "worst_row": min(rows, key=lambda r: r[-1]) if rows else None, "layers": rows}
    return out


def gemm_variant_bench(lib, torch, kzp, in_scale, warmup, iters, seed):
    """The 4096^3 GEMM of the headline in a less favourable class: another kernel zero point (no centred image ->
    the lean kernel with its row term) or a requantization scale < 0.5 (shift >= 1 epilogue)."""
    M = N = K = 4096
    rng = np.random.default_rng(seed)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    a = torch.randint(0, 256, (M * K,), dtype=torch.uint8, device="cuda", generator=gen)
    c = torch.empty(M * N, dtype=torch.uint8, device="cuda")
    op = lib.create_fully_connected_nc_q8(K, N, 127, in_scale, kzp, 1.0, w, bias, 127, 1.0, 1, 254)
    lib.setup_fully_connected_nc_q8(op, M, a, K, c, N)
    lib.run_operator(op)
    # timed as the headline's `sustained_launch_ms`: a hipGraph of 64 launches, median of five batches of `iters` replays
    # (a short back-to-back loop right after another kernel family measures the clock ramp, not the kernel: 66 us for the
    #  headline configuration itself against 57 in the sustained state)
    lib.set_async(True)
    lib.graph_begin()
    for _ in range(64):
        lib.run_operator(op)
    graph = lib.graph_end()
    ms = lib.graph_time(graph, warmup, iters) / 64.0
    lib.graph_destroy(graph)
    lib.set_async(False)
    tops = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    out = {"kernel": lib.operator_kernel(op), "kernel_zero_point": kzp, "requant_scale": in_scale, "us": round(ms * 1e3, 2),
           "tops": round(tops, 1), "frac": round(tops / PEAK_I8_TOPS, 4),
           "timed_as": "median of 5 batches of replays of a 64-launch hipGraph"}
    lib.delete_operator(op)
    return out


def fc_m1_bench(lib, torch, warmup, iters):
    """BASELINE configs[0] on the device: qnnp_fully_connected_nc_q8 M=1, K=1024, N=1000 (a launch-bound GEMV)."""
    K, N = 1024, 1000
    rng = np.random.default_rng(5)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    a = torch.randint(0, 256, (K,), dtype=torch.uint8, device="cuda")
    c = torch.empty(N, dtype=torch.uint8, device="cuda")
    op = lib.create_fully_connected_nc_q8(K, N, 127, 0.5, 127, 0.5, w, bias, 127, 0.5, 0, 255)
    lib.setup_fully_connected_nc_q8(op, 1, a, K, c, N)
    lib.run_operator(op)
    ms = lib.time_operator(op, warmup, iters)
    out = {"kernel": lib.operator_kernel(op), "us": round(ms * 1e3, 2), "weight_gbs": round(K * N / (ms * 1e-3) / 1e9, 1)}
    lib.delete_operator(op)
    return out


def next_rows_bench(lib, torch, batch, warmup, iters):
    """The operators SURVEY.md section 8f ranks after the hot path, on MobileNetV2-sized tensors: kernel time
    (rotating buffers, > 512 MB between reuses) and algorithmic GB/s. All three are HBM-bound byte kernels except
    the deconvolution, which runs the offset-table MFMA kernel (stride 2: three taps in four are padding)."""
    out = {}
    gen = torch.Generator(device="cuda")
    gen.manual_seed(77)

    def rotating(nbytes_in, nbytes_out, min_between=512 << 20):
        nsets = min(64, max(1, -(-min_between // (nbytes_in + nbytes_out))))
        ins = [torch.randint(0, 256, (nbytes_in,), dtype=torch.uint8, device="cuda", generator=gen) for _ in range(nsets)]
        outs = [torch.empty(nbytes_out, dtype=torch.uint8, device="cuda") for _ in range(nsets)]
        return ins, outs

    # residual add of the 56x56x24 bottleneck output (reference bench/add.cc shape family)
    rows, ch = batch * 56 * 56, 24
    ins, outs = rotating(rows * ch, rows * ch)
    b_operand = torch.randint(0, 256, (rows * ch,), dtype=torch.uint8, device="cuda", generator=gen)
    op = lib.create_add_nc_q8(ch, 121, 0.75, 127, 1.25, 133, 0.96875, 0, 255, 0)
    lib.setup_add_nc_q8(op, rows, ins[0], ch, b_operand, ch, outs[0], ch)
    lib.run_operator(op)
    ms = lib.time_operator_rotating(op, ins, outs, warmup, iters)
    out["q8add_56x56x24"] = {"kernel": lib.operator_kernel(op), "ms": round(ms, 5), "bytes": 3 * rows * ch,
                             "gbs": round(3 * rows * ch / (ms * 1e-3) / 1e9, 1)}
    lib.delete_operator(op)

    # global average pooling in front of the classifier: 7x7x1280
    width, ch = 49, 1280
    ins, outs = rotating(batch * width * ch, batch * ch, min_between=64 << 20)
    op = lib.create_global_average_pooling_nwc_q8(ch, 121, 1.0, 133, 1.0, 0, 255, 0)
    lib.setup_global_average_pooling_nwc_q8(op, batch, width, ins[0], ch, outs[0], ch)
    lib.run_operator(op)
    ms = lib.time_operator_rotating(op, ins, outs, warmup, iters)
    nbytes = batch * width * ch + batch * ch
    out["q8gavgpool_7x7x1280"] = {"kernel": lib.operator_kernel(op), "ms": round(ms, 5), "bytes": nbytes,
                                  "gbs": round(nbytes / (ms * 1e-3) / 1e9, 1)}
    lib.delete_operator(op)

    # 2x upsampling deconvolution 28x28x64 -> 56x56x32 (2x2 kernel, stride 2)
    H = W = 28
    cin, cout = 64, 32
    rng = np.random.default_rng(78)
    kernel = rng.integers(0, 256, size=(1, cin, 2, 2, cout), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=cout, dtype=np.int32)
    ins, outs = rotating(batch * H * W * cin, batch * 4 * H * W * cout)
    op = lib.create_deconvolution2d_nhwc_q8(0, 0, 0, 0, 0, 0, 2, 2, 2, 2, 1, 1, 1, cin, cout,
                                            127, 0.5, 127, 0.5, kernel, bias, 127, 0.5, 0, 255, 0)
    lib.setup_deconvolution2d_nhwc_q8(op, batch, H, W, ins[0], cin, outs[0], cout)
    lib.run_operator(op)
    ms = lib.time_operator_rotating(op, ins, outs, warmup, iters)
    nbytes = batch * H * W * cin + batch * 4 * H * W * cout
    useful_ops = 2 * batch * H * W * 4 * cin * cout          # every input pixel meets each of the 2x2 taps once
    out["q8deconv_2x2s2_28x28x64_32"] = {"kernel": lib.operator_kernel(op), "ms": round(ms, 5), "bytes": nbytes,
                                         "gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                         "useful_tops": round(useful_ops / (ms * 1e-3) / 1e12, 2)}
    lib.delete_operator(op)

    # the same upsampling with a 3x3 kernel (stride 2, padding 1, output adjustment 1): four phase GEMMs of 1/2/2/4 taps
    kernel = rng.integers(0, 256, size=(1, cin, 3, 3, cout), dtype=np.uint8)
    op = lib.create_deconvolution2d_nhwc_q8(1, 1, 1, 1, 1, 1, 3, 3, 2, 2, 1, 1, 1, cin, cout,
                                            127, 0.5, 127, 0.5, kernel, bias, 127, 0.5, 0, 255, 0)
    lib.setup_deconvolution2d_nhwc_q8(op, batch, H, W, ins[0], cin, outs[0], cout)
    lib.run_operator(op)
    ms = lib.time_operator_rotating(op, ins, outs, warmup, iters)
    useful_ops = 2 * batch * H * W * 9 * cin * cout
    out["q8deconv_3x3s2_28x28x64_32"] = {"kernel": lib.operator_kernel(op), "ms": round(ms, 5), "bytes": nbytes,
                                         "gbs": round(nbytes / (ms * 1e-3) / 1e9, 1),
                                         "useful_tops": round(useful_ops / (ms * 1e-3) / 1e12, 2)}
    lib.delete_operator(op)
    return out


def cpu_baseline_gemm(seconds_budget=12.0):
    """Reference SSE2 q8gemm (qnnp_fully_connected_nc_q8 of the compiled reference) on the host cores:
    a 512-row slice of the same 4096^3 problem (same N, K, data distribution), all cores via the
    OpenMP pthreadpool shim, repeated for ~seconds_budget."""
    from oracle import o1, ref
    N = K = 4096
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0x51A0 + 2)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    if ref.available():
        lib = ref.lib()
        M = 512
        a = rng.integers(0, 256, size=M * K + 16, dtype=np.uint8)
        c = np.zeros(M * N, dtype=np.uint8)
        op = lib.create_fully_connected_nc_q8(K, N, 127, 0.5, 127, 0.5, w, bias, 127, 0.5, 0, 255)
        lib.setup_fully_connected_nc_q8(op, M, a[8:], K, c, N)
        def timed(threads, budget, max_iters):
            pool = lib.threadpool(threads)
            lib.run_operator(op, pool)      # warm-up
            iters, t0 = 0, time.perf_counter()
            while True:
                lib.run_operator(op, pool)
                iters += 1
                dt = time.perf_counter() - t0
                if dt >= budget or iters >= max_iters:
                    break
            lib.destroy_threadpool(pool)
            return iters, dt
        # the box reports more hardware threads than it can usefully run (256 threads measured 10x slower than
        # 16): give the reference its best thread count, found with a short calibration
        best_threads, best_rate = cores, 0.0
        for threads in sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores}):
            iters, dt = timed(threads, 0.7, 20)
            if iters / dt > best_rate:
                best_threads, best_rate = threads, iters / dt
        iters, dt = timed(best_threads, seconds_budget, 400)
        lib.delete_operator(op)
        tops = 2.0 * M * N * K * iters / dt / 1e12
        return {"value": round(tops, 5), "unit": "TOPS", "cores": best_threads, "kind": "reference",
                "sample": f"reference SSE2 q8gemm, M={M} rows of N=K=4096 x {iters} runs, {best_threads} of {cores} "
                          f"threads (best), {dt:.1f} s"}
    M = 32
    a = rng.integers(0, 256, size=(M, K), dtype=np.uint8)
    o1.set_threads(cores)
    t0 = time.perf_counter()
    acc = o1.gemm_acc(a, w, bias, 127, 127)
    o1.requantize_rows(acc, 0.5, 127, 0, 255)
    dt = time.perf_counter() - t0
    o1.set_threads(1)
    return {"value": round(2.0 * M * N * K / dt / 1e12, 6), "unit": "TOPS", "cores": cores, "kind": "port",
            "sample": f"scalar oracle port, M={M} rows of the N=K=4096 problem, {cores} OpenMP threads, {dt:.1f} s"}


def cpu_baseline_sweep(batch=16, seconds_budget=10.0, threads=None):
    """The MobileNetV2 conv-layer sweep on the compiled REFERENCE (its SSE2 q8conv / q8gemm / q8dwconv
    microkernels under qnnp_run_operator) with all host threads: `batch` images per layer, every layer its own
    operator as bench/convolution.cc, whole passes repeated for ~seconds_budget. None if the reference is not built."""
    from oracle import ref
    if not ref.available():
        return None
    lib = ref.lib()
    cores = threads or (os.cpu_count() or 1)
    pool = lib.threadpool(cores)
    ops, keep = [], []
    rng = np.random.default_rng(7)
    for i, (H, W, KH, KW, S, D, G, GIC, GOC) in enumerate(MOBILENETV2):
        (pt, pr, pb, pl), oh, ow = conv_geometry(H, W, KH, KW, S, D)
        kernel = rng.integers(0, 256, size=(G, GOC, KH, KW, GIC), dtype=np.uint8)
        bias = rng.integers(-10000, 10001, size=G * GOC, dtype=np.int32)
        op = lib.create_convolution2d_nhwc_q8(pt, pr, pb, pl, KH, KW, S, S, D, D, G, GIC, GOC,
                                              127, 0.5, 127, 0.5, kernel, bias, 127, 0.5, 0, 255, 0)
        inp = rng.integers(0, 256, size=batch * H * W * G * GIC + 16, dtype=np.uint8)
        out = np.zeros(batch * oh * ow * G * GOC, dtype=np.uint8)
        lib.setup_convolution2d_nhwc_q8(op, batch, H, W, inp[8:], G * GIC, out, G * GOC)
        ops.append(op)
        keep.append((inp, out, kernel, bias))
    lib.destroy_threadpool(pool)

    def timed(threads_, budget, max_passes):
        pool_ = lib.threadpool(threads_)
        for op in ops:
            lib.run_operator(op, pool_)      # warm-up pass
        passes, t0 = 0, time.perf_counter()
        while True:
            for op in ops:
                lib.run_operator(op, pool_)
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= budget or passes >= max_passes:
                break
        lib.destroy_threadpool(pool_)
        return passes, dt
    host = os.cpu_count() or 1
    best_threads = cores
    if threads is None:                      # best thread count for the reference (see cpu_baseline_gemm)
        best_rate = 0.0
        for t in sorted({t for t in (8, 16, 32, 64, 128, host) if t <= host}):
            passes, dt = timed(t, 0.7, 50)
            if passes / dt > best_rate:
                best_threads, best_rate = t, passes / dt
    passes, dt = timed(best_threads, seconds_budget, 2000)
    for op in ops:
        lib.delete_operator(op)
    return {"images_per_s": round(batch * passes / dt, 1), "cores": best_threads, "kind": "reference",
            "sample": f"reference SSE2 microkernels, all 31 layers, batch {batch} x {passes} passes, {best_threads} of {host} "
                      f"threads (best), {dt:.1f} s"}


def secondary_block(extra):
    """BASELINE's other configs as flat scalars INSIDE `roofline` (the driver's record keeps `config`, `roofline` and
    `cpu_baseline` and drops `extra`): c0 = configs[0] ... c4 = configs[4]; every fraction is of the chip peak named
    in the key (HBM 8 TB/s, int8 MFMA 5033 TOP/s) or of max(MFMA, HBM) time ("bound")."""
    def get(path, default=None):
        node = extra
        for key in path:
            if not isinstance(node, dict) or key not in node:
                return default
            node = node[key]
        return node
    out = {}
    c0 = get(["q8fc_m1_k1024_n1000"])
    if c0:
        out["c0_fc_m1_k1024_n1000_us"] = c0["us"]
        out["c0_kernel"] = c0["kernel"]
    c2 = get(["q8conv_3x3_56x56x64_b128"])
    if c2:
        out["c2_conv3x3_56x56x64_b128_ms"] = c2["ms"]
        out["c2_bound_ms"] = c2["roofline_ms"]
        out["c2_frac_of_bound"] = round(c2["roofline_ms"] / c2["ms"], 4)
        out["c2_tops"] = c2["tops"]
        out["c2_kernel"] = c2["kernel"]
    c3 = get(["q8dwconv_mobilenetv2_layers"])
    if c3:
        out["c3_dwconv_layers_hbm_gbs"] = c3["hbm_gbs"]
        out["c3_frac_of_hbm_peak"] = c3["frac_of_hbm_peak"]
    c4 = get(["mobilenetv2_sweep"])
    if c4:
        out["c4_sweep_images_per_s_graph"] = c4["images_per_s"]
        out["c4_sweep_images_per_s_sum_of_layers"] = c4["images_per_s_by_sum_of_layers"]
        out["c4_frac_of_hbm_peak"] = c4["frac_of_hbm_peak"]
        out["c4_batch_per_gpu"] = c4["batch_per_gpu"]
    for key, name in (("mobilenetv2_network", "network_images_per_s"), ("mobilenetv2_network_adds_folded", "network_adds_folded_images_per_s"),
                      ("mobilenetv2_network_fused", "network_fused_images_per_s"),
                      ("mobilenetv2_network_fused_expanding_blocks", "network_fused_expanding_images_per_s")):
        v = get([key, "images_per_s"])
        if v is not None:
            out[name] = v
    for key, name in (("kernel_zero_point_126", "gemm4096_kzp126"), ("requant_scale_0.3_shift1", "gemm4096_shift1"),
                      ("kernel_zero_point_126_scale_0.3", "gemm4096_kzp126_shift1")):
        v = get(["q8gemm_4096_variants", key])
        if v:
            out[name + "_frac"] = v["frac"]
            out[name + "_us"] = v["us"]
    v = get(["mobilenetv2_sweep_realistic_scale", "images_per_s_by_sum_of_layers"])
    if v is not None:
        out["c4_sweep_realistic_scale_images_per_s_sum_of_layers"] = v
    more = get(["q8dwconv_5x5_dilated_and_realistic_scale"], {})
    for key, name in (("dw5x5_56x56x72_s2", "dw5x5_s2"), ("dw5x5_28x28x240_s1", "dw5x5_28"), ("dw5x5_14x14x672_s1", "dw5x5_14"),
                      ("dw3x3_dil2_28x28x192", "dw3x3_dil2"), ("dw3x3_56x56x144_scale0.0125", "dw3x3_realistic_scale"),
                      ("pw_112x112x16_96_scale0.0125", "pw_layer4_realistic_scale")):
        if key in more:
            out[name + "_frac_of_hbm_peak"] = more[key]["frac_of_hbm_peak"]
    nxt = get(["next_rows"], {})
    for key, name in (("q8deconv_3x3s2_28x28x64_32", "deconv3x3s2"), ("q8deconv_2x2s2_28x28x64_32", "deconv2x2s2"),
                      ("q8add_56x56x24", "add"), ("q8gavgpool_7x7x1280", "gavgpool")):
        if key in nxt:
            out[name + "_frac_of_hbm_peak"] = round(nxt[key]["gbs"] / PEAK_HBM_GBS, 4)
    for net in ("resnet18", "resnet50", "shufflenet_v1_g2"):
        v = get(["conv_lists", net])
        if v:
            out[net + "_images_per_s"] = v["images_per_s_by_sum_of_layers"]
            out[net + "_frac_of_bound"] = v["frac_of_bound"]
            if v["worst_dense_3x3_frac"] is not None:
                out[net + "_worst_dense_3x3_frac_of_bound"] = v["worst_dense_3x3_frac"]
    for name, v in (get(["reference_bench_lists"], {}) or {}).items():
        if v.get("images_per_s_by_sum_of_layers") is not None:
            out["list_" + name + "_images_per_s"] = v["images_per_s_by_sum_of_layers"]
            out["list_" + name + "_frac_of_bound"] = v["frac_of_bound"]
    return out


def stub_mode():
    """QNNP_BENCH_STUB=1: harness self-test (tests/test_bench_harness.py). The launcher, the rank / world-size checks,
    the barrier-bracketed timed region, the max-over-ranks reduction and the JSON assembly run exactly as in a real run,
    over gloo on CPU, with a sleep standing in for the device step. The line it prints says so (`data`: "stub")."""
    return os.environ.get("QNNP_BENCH_STUB") == "1"


def assemble_line(*, world, steps, warmup, ms_per_step, ev_ms_per_rank, gemm_kernel, info, roofline, cpu, extra, data):
    """The one JSON line of the contract. value = whole-job int8 TOPS: every rank runs its own replica of the 4096^3
    GEMM (no batch to shard: "replicas only"), so the job processes `world` GEMMs per step in max-over-ranks time."""
    gemm_ops = 2.0 * 4096 ** 3
    value = world * gemm_ops / (ms_per_step * 1e-3) / 1e12
    if roofline is not None and world > 1:
        # N > 1: rank 0's kernel is the one in `roofline`; every rank's launch time travels beside it
        roofline = dict(roofline)
        roofline["per_rank_launch_ms"] = [round(v, 5) for v in ev_ms_per_rank]
        roofline["per_rank_frac"] = [round(gemm_ops / (v * 1e-3) / 1e12 / PEAK_I8_TOPS, 4) for v in ev_ms_per_rank]
    return {
        "metric": "q8gemm_int8_tops", "value": round(value, 2), "unit": "TOPS",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": data,
        "config": {"workload": "q8gemm M=N=K=4096 uint8 (qnnp_fully_connected_nc_q8, int8 MFMA, fused Q31 requantize)"
                               + (" -- one replica per GPU" if world > 1 else ""),
                   "kernel": gemm_kernel, "device": info["arch"], "compute_units": info["compute_units"],
                   "pct_of_i8_mfma_peak": round(100.0 * value / world / PEAK_I8_TOPS, 2)},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "extra": extra,
    }


CONTRACT_LINE_MAX = 6000      # bytes: the driver keeps the last 8 KB of stdout (round 5's 23 KB line was cut and never parsed)


def emit(line, full_out):
    """Print the contract line (LAST line of stdout) without `extra`; the whole object, `extra` included, goes to
    `full_out` (default gpurun_out/bench_full.json). Per-layer tables belong in the file, never on the line."""
    contract = {k: v for k, v in line.items() if k != "extra"}
    text = json.dumps(contract)
    if full_out:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_out)), exist_ok=True)
            with open(full_out, "w") as f:
                json.dump(line, f)
                f.write("\n")
            print(f"# full record (per-layer tables) written to {full_out}", file=sys.stderr)
        except OSError as exc:
            print(f"# could not write {full_out}: {exc}", file=sys.stderr)
    if len(text) > CONTRACT_LINE_MAX:
        # never let the line outgrow the capture again: drop the optional blocks, largest first, and say so
        sec = contract["roofline"].get("secondary") if isinstance(contract.get("roofline"), dict) else None
        if isinstance(sec, dict) and any(k.startswith("list_") for k in sec):     # the per-list scalars of the other bench lists first
            contract["roofline"] = dict(contract["roofline"], secondary={k: v for k, v in sec.items() if not k.startswith("list_")},
                                        dropped_for_length=["secondary.list_*"])
            text = json.dumps(contract)
        for key in ("secondary", "per_rank_frac"):
            if isinstance(contract.get("roofline"), dict) and key in contract["roofline"] and len(text) > CONTRACT_LINE_MAX:
                contract["roofline"] = {k: v for k, v in contract["roofline"].items() if k != key}
                contract["roofline"]["dropped_for_length"] = contract["roofline"].get("dropped_for_length", []) + [key]
                text = json.dumps(contract)
    sys.stderr.flush()
    print(text, flush=True)
    return text


def gather_floats(value, world):
    """[value of rank 0, ..., value of rank world-1] on every rank (all_gather over the harness process group)."""
    if world <= 1:
        return [float(value)]
    import torch
    import torch.distributed as dist
    device = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([float(value)], dtype=torch.float64, device=device)
    out = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(out, mine)
    return [float(t.item()) for t in out]


def run_stub(args, world, rank):
    """The harness with a sleep for a step (see stub_mode): same barriers, same reductions, same line."""
    import torch.distributed as dist
    from qnnpack_amd.shard import job_time_ms, shard_batch
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()

    step_s = 0.002 * (1.0 + 0.5 * rank)                    # the last rank is the slow one
    for _ in range(args.warmup):
        time.sleep(step_s)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(step_s)
    local_ms = (time.perf_counter() - t0) * 1e3 / args.steps
    barrier()
    ms_per_step = job_time_ms(local_ms, world)
    per_rank = gather_floats(local_ms, world)
    total_batch = args.sweep_batch * world
    start, my_batch = shard_batch(total_batch, world, rank)
    sweep_ms = job_time_ms(0.5 * (1.0 + 0.5 * rank), world)
    act_bytes = 1.0e7 * my_batch                           # (a stand-in for the sweep's algorithmic bytes per rank)
    extra = {"mobilenetv2_sweep": {"images_per_s": round(total_batch / (sweep_ms * 1e-3), 1), "batch_per_gpu": my_batch,
                                   "shard_start": start, "ms_per_batch": round(sweep_ms, 4), "timed_as": "stub",
                                   "images_per_s_by_sum_of_layers": round(total_batch / (sweep_ms * 1e-3), 1),
                                   "frac_of_hbm_peak": round(act_bytes / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                                   "aggregate_hbm_gbs": round(world * act_bytes / (sweep_ms * 1e-3) / 1e9, 1),
                                   "aggregate_frac_of_hbm_peak": round(act_bytes / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}}
    roofline = {"bound": "mfma", "kernel": "stub", "achieved": None, "peak": round(PEAK_I8_TOPS, 1), "unit": "TOP/s",
                "frac": None, "traffic": None, "secondary": secondary_block(extra)}
    cpu = {"value": None, "unit": "TOPS", "cores": 0, "kind": "stub", "sample": "none (harness self-test)"} if rank == 0 else None
    if rank == 0:
        emit(assemble_line(world=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
                           ev_ms_per_rank=per_rank, gemm_kernel="stub",
                           info={"arch": "stub", "compute_units": 0}, roofline=roofline, cpu=cpu,
                           extra=extra, data="stub (no device work: harness self-test)"), args.full_out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: re-run this file under torch.distributed.run, one rank per GPU.
    Fails loudly when the node has fewer than N GPUs (a 1-GPU number must never be reported as an N-GPU one)."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if stub_mode():
        have = n_gpus                                      # harness self-test: ranks are CPU processes
    if have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} requested but this node shows {have} GPU(s); refusing to run fewer ranks",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--sweep-batch", type=int, default=128, help="MobileNetV2 sweep images per GPU")
    ap.add_argument("--no-extra", action="store_true", help="headline GEMM only")
    ap.add_argument("--full-out", default=os.path.join(ROOT, "gpurun_out", "bench_full.json"),
                    help="file that receives the whole record (the contract line plus `extra`: per-layer tables); '' = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-conv-lists", action="store_true", help="skip the ResNet-18 / ResNet-50 / ShuffleNet shape lists")
    ap.add_argument("--gemm-kernel", type=int, default=0, help="0 auto, 1 generic MFMA kernel, 2 big-tile kernel")
    ap.add_argument("--dw-kernel", type=int, default=0,
                    help="measurement aid: 0 auto, 1 direct, 2 LDS-tiled, 3 register sliding window (depthwise layers)")
    ap.add_argument("--out-scale", type=float, default=0.5,
                    help="--layer mode only: output scale (requantization scale = 0.25 / this; 0.5 = the reference bench)")
    ap.add_argument("--layer", type=int, default=0,
                    help="measurement aid: time only MobileNetV2 sweep layer N (1-based) and print a short JSON line")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly with --gpus N: become the launcher of N ranks (one process per GPU, RCCL over xGMI for
        # the harness barrier only -- the data path has no collective)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a {world}-GPU run as {args.gpus}")
    if stub_mode():
        run_stub(args, world, rank)
        return

    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    if torch.cuda.device_count() < max(world, local_rank + 1):
        raise SystemExit(f"bench.py: {world} ranks need {world} GPUs, this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)              # before the process group: RCCL binds to the current device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)   # "nccl" is RCCL on ROCm
    torch.zeros(1, device="cuda")

    import qnnpack_amd
    from qnnpack_amd.shard import job_time_ms, shard_batch
    lib = qnnpack_amd.load()
    lib.set_device(local_rank)
    lib.initialize()
    lib.set_stream(torch.cuda.current_stream().cuda_stream)
    lib.set_option("gemm_kernel", args.gemm_kernel)
    lib.set_option("dwconv_kernel", args.dw_kernel)
    info = lib.device_info()

    def barrier():
        if world > 1:
            dist.barrier()

    if args.layer:
        # layer 99 = BASELINE configs[2]: 3x3 s1 conv 56x56x64 -> 64
        H, W, KH, KW, S, D, G, GIC, GOC = (56, 56, 3, 3, 1, 1, 1, 64, 64) if args.layer == 99 else MOBILENETV2[args.layer - 1]
        layer = ConvLayer(lib, torch, args.sweep_batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=100 + args.layer,
                          min_bytes_between_reuse=512 << 20, out_scale=args.out_scale)
        ms = layer.time_ms(args.warmup, args.steps)
        b = layer.in_bytes + layer.out_bytes
        print(json.dumps({"layer": args.layer, "shape": [H, W, KH, S, G, GIC, GOC], "kernel": layer.kernel,
                          "ms": round(ms, 5), "gbs": round(b / (ms * 1e-3) / 1e9, 1), "bytes": b,
                          "tops": round(layer.ops / (ms * 1e-3) / 1e12, 2)}), flush=True)
        layer.close()
        return

    # ------------------------------------------------------------------ headline: q8gemm 4096^3
    M = N = K = 4096
    rng = np.random.default_rng(0x51A0 + 2 + rank)
    w = rng.integers(0, 256, size=(N, K), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=N, dtype=np.int32)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x51A0 + rank)
    a = torch.randint(0, 256, (M * K,), dtype=torch.uint8, device="cuda", generator=gen)   # random, not zeros (DVFS)
    c = torch.empty(M * N, dtype=torch.uint8, device="cuda")
    # zero points 127, requantization scale 0.75, clamp [1, 254]: bench/q8gemm.cc:103
    op = lib.create_fully_connected_nc_q8(K, N, 127, 0.75, 127, 1.0, w, bias, 127, 1.0, 1, 254)
    lib.setup_fully_connected_nc_q8(op, M, a, K, c, N)
    lib.set_async(True)
    # the headline runs on a stream of its own: a graph captured on the legacy default stream would replay on a private one, out of
    # reach of the events below
    torch.cuda.synchronize()
    head_stream = torch.cuda.Stream()
    lib.set_stream(head_stream.cuda_stream)
    for _ in range(args.warmup):
        lib.run_operator(op)
    torch.cuda.synchronize()
    # Sustained state first: the same GEMM replayed as a hipGraph of 64 launches, five event-bracketed batches of
    # >= 200 ms each (median reported). The K timed steps follow at once, so they run at the sustained clock.
    lib.graph_begin()
    for _ in range(64):
        lib.run_operator(op)
    graph = lib.graph_end()
    sustained_ms = lib.graph_time(graph, 1, 48) / 64.0          # 5 batches x 48 replays x 64 launches
    lib.graph_destroy(graph)
    # The K timed steps are K `qnnp_run_operator` launches captured ONCE, before the timed region, into a hipGraph
    # (qnnp_gfx950_graph_*: how a caller that repeats a fixed sequence submits it) and replayed inside it: the K kernels then
    # run back to back, where K separate asynchronous calls from Python leave ~1 us between kernels (r05e: 58.66 us per
    # step by HIP events against 57.72 by rocprofv3 per kernel). One untimed replay first (the graph's upload).
    timed_graph = None
    capturing = False
    try:
        lib.graph_begin()
        capturing = True
        for _ in range(args.steps):
            lib.run_operator(op)
        timed_graph = lib.graph_end()
        capturing = False
        lib.graph_launch(timed_graph)
    except Exception as exc:                                  # capture unavailable: K separate launches
        print(f"# graph capture of the timed steps unavailable ({exc}); timing {args.steps} separate launches", file=sys.stderr)
        if capturing:                                         # never launch into a stream that is still capturing
            try:
                lib.graph_destroy(lib.graph_end())
            except Exception:  # noqa: BLE001
                pass
        elif timed_graph is not None:
            try:
                lib.graph_destroy(timed_graph)
            except Exception:  # noqa: BLE001
                pass
        timed_graph = None
        torch.cuda.synchronize()
    # the same K steps as K separate asynchronous launches (rounds 1-4's method; ~1 us of launch gap per step): reported beside
    # the graph figure so the two methods can be compared round to round
    sep0, sep1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    sep0.record(head_stream)
    for _ in range(args.steps):
        lib.run_operator(op)
    sep1.record(head_stream)
    torch.cuda.synchronize()
    separate_ms = sep0.elapsed_time(sep1) / args.steps
    stream = head_stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)   # (the launch stream: set_stream above)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record(stream)
    if timed_graph is not None:
        lib.graph_launch(timed_graph)                         # (replays on the stream it was captured on: head_stream)
    else:
        for _ in range(args.steps):
            lib.run_operator(op)
    ev1.record(stream)
    torch.cuda.synchronize()
    local_ms = (time.perf_counter() - t0) * 1e3 / args.steps
    barrier()
    if timed_graph is not None:
        lib.graph_destroy(timed_graph)
    lib.set_stream(torch.cuda.current_stream().cuda_stream)
    ms_per_step = job_time_ms(local_ms, world)
    gemm_kernel = lib.operator_kernel(op)
    # the same K launches by HIP events on the launch stream: kernel time without the host's share of the region
    ev_ms = ev0.elapsed_time(ev1) / args.steps
    ev_ms_per_rank = gather_floats(ev_ms, world)
    gemm_ops = 2.0 * M * N * K
    achieved = gemm_ops / (ev_ms * 1e-3) / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(gemm_kernel)
        except Exception:
            traffic = None
    roofline = {"bound": "mfma", "kernel": gemm_kernel, "achieved": round(achieved, 2), "peak": round(PEAK_I8_TOPS, 1),
                "unit": "TOP/s", "frac": round(achieved / PEAK_I8_TOPS, 4), "traffic": traffic,
                "launch_ms": round(ev_ms, 5), "algorithmic_bytes_per_launch": 3 * M * N + 4 * N,
                "timed_as": ("HIP events around the K timed steps (one replay of a hipGraph of K launches)"
                             if timed_graph is not None else "HIP events around the K timed steps (K separate launches)"),
                "separate_launches_ms": round(separate_ms, 5),
                "sustained_launch_ms": round(sustained_ms, 5),
                "sustained_tops": round(gemm_ops / (sustained_ms * 1e-3) / 1e12, 2),
                "sustained_as": "median of 5 x 48 replays of a 64-launch hipGraph, right before the timed steps"}
    copy_gbs = None
    try:
        # this chip's own streaming ceiling: a plain 16-byte-per-lane copy / read kernel over 1 GiB buffers
        dbg = qnnpack_amd.load_debug()       # measurement companion library, not the product
        # (round 4: the ceiling is the BEST streaming form -- nt loads and stores -- and the plain form is printed beside it:
        #  the product's layer 4 had beaten the plain copy kernel, which a ceiling must not allow)
        plain_gbs = round(dbg.copy_probe(False, 1024, 5), 1)
        copy_gbs = max(plain_gbs, round(dbg.copy_probe(False, 1024, 5, streaming=True), 1))
        roofline["hbm_copy_kernel_gbs"] = copy_gbs
        roofline["hbm_copy_kernel_plain_policy_gbs"] = plain_gbs
        roofline["hbm_read_kernel_gbs"] = round(dbg.copy_probe(True, 1024, 5), 1)
    except Exception as exc:  # noqa: BLE001
        print(f"# copy probe failed: {exc}", file=sys.stderr)
    if rank == 0:
        # the bare-MFMA rate of this very chip, measured in this process: with random operands the power
        # management holds a lower clock, so this -- not the nominal peak -- is what a kernel can reach at best
        try:
            dbg = qnnpack_amd.load_debug()   # measurement companion library, not the product
            roofline["mfma_only_random_operands"] = round(dbg.mfma_probe(True, 6400), 1)
            roofline["mfma_only_zero_operands"] = round(dbg.mfma_probe(False, 6400), 1)
            roofline["frac_of_mfma_only_random"] = round(achieved / roofline["mfma_only_random_operands"], 4)
        except Exception as exc:  # noqa: BLE001
            print(f"# mfma probe failed: {exc}", file=sys.stderr)
        # the vendor library on the same problem, same box, same minute (torch._int_mm -> hipBLASLt: int8 x int8 ->
        # int32, random operands, NO uint8 re-centring / row sums / requantization): a yardstick, not a baseline
        try:
            va = torch.randint(-128, 128, (M, K), dtype=torch.int8, device="cuda")
            vb = torch.randint(-128, 128, (N, K), dtype=torch.int8, device="cuda").t()
            for _ in range(5):
                torch._int_mm(va, vb)
            torch.cuda.synchronize()
            ve0, ve1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ve0.record()
            for _ in range(300):
                torch._int_mm(va, vb)
            ve1.record()
            torch.cuda.synchronize()
            roofline["vendor_int8_gemm_no_epilogue_tops"] = round(gemm_ops / (ve0.elapsed_time(ve1) * 1e-3 / 300) / 1e12, 1)
            del va, vb
        except Exception as exc:  # noqa: BLE001
            print(f"# vendor int8 GEMM probe failed: {exc}", file=sys.stderr)
    lib.delete_operator(op)
    del a, c

    extra = {}
    if not args.no_extra:
        lib.set_async(False)
        # ---------------------------------------------------------- configs[2]: 3x3 conv 56x56x64->64, batch 128
        layer = ConvLayer(lib, torch, 128, 56, 56, 3, 3, 1, 1, 1, 64, 64, seed=3, min_bytes_between_reuse=544 << 20)
        ms = layer.time_ms(3, 20)
        extra["q8conv_3x3_56x56x64_b128"] = {
            "kernel": layer.kernel, "ms": round(ms, 4), "tops": round(layer.ops / (ms * 1e-3) / 1e12, 2),
            "gbs": round((layer.in_bytes + layer.out_bytes) / (ms * 1e-3) / 1e9, 1),
            "frac_of_copy_kernel": round((layer.in_bytes + layer.out_bytes) / (ms * 1e-3) / 1e9 / copy_gbs, 4) if copy_gbs else None,
            "roofline_ms": round(max(layer.ops / (PEAK_I8_TOPS * 1e12), (layer.in_bytes + layer.out_bytes) / (PEAK_HBM_GBS * 1e9)) * 1e3, 4)}
        layer.close()

        # ---------------------------------------------------------- configs[3] + [4]: MobileNetV2 sweep, batch sharded
        total_batch = args.sweep_batch * world
        _, my_batch = shard_batch(total_batch, world, rank)
        sweep_ms, dw_ms, dw_bytes, act_bytes, total_ops, rows, layers = 0.0, 0.0, 0, 0, 0.0, [], []
        for i, (H, W, KH, KW, S, D, G, GIC, GOC) in enumerate(MOBILENETV2):
            layer = ConvLayer(lib, torch, my_batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=100 + i,
                              min_bytes_between_reuse=512 << 20)
            ms = layer.time_ms(2, 10)
            b = layer.in_bytes + layer.out_bytes
            sweep_ms += ms
            act_bytes += b
            total_ops += layer.ops
            if G > 1:
                dw_ms += ms
                dw_bytes += b
            rows.append({"layer": i + 1, "shape": [H, W, KH, S, G, GIC, GOC], "kernel": layer.kernel,
                         "ms": round(ms, 4), "gbs": round(b / (ms * 1e-3) / 1e9, 1),
                         "tops": round(layer.ops / (ms * 1e-3) / 1e12, 2)})
            layers.append(layer)
        # the whole sweep as ONE hipGraph (qnnp_gfx950_graph_*): 31 operator launches, one submission. Each layer
        # keeps its own tensors (as bench/convolution.cc); one pass touches 1.3 GB, so a replay finds nothing of
        # the previous one in the 256 MiB Infinity Cache.
        sum_of_layers_ms = sweep_ms
        graph_ms = None
        try:
            lib.set_async(True)
            lib.graph_begin()
            for layer in layers:
                lib.run_operator(layer.op)
            graph = lib.graph_end()
            graph_ms = lib.graph_time(graph, 2, 10)
            lib.graph_destroy(graph)
        except Exception as exc:  # noqa: BLE001 -- fall back to the per-layer sum, say so
            print(f"# sweep graph capture unavailable ({exc}); reporting the sum of per-layer times", file=sys.stderr)
        finally:
            lib.set_async(False)
        for layer in layers:
            layer.close()
        if graph_ms is not None:
            sweep_ms = graph_ms
        job_sweep_ms = job_time_ms(sweep_ms, world)
        launch_floor = None
        try:
            dbg_lib = qnnpack_amd.load_debug()
            launch_floor = {"blocks_256": round(dbg_lib.launch_floor_probe(31, 256, 20), 3),
                            "blocks_4096": round(dbg_lib.launch_floor_probe(31, 4096, 20), 3)}
        except Exception as exc:  # noqa: BLE001 -- a measurement aid, not part of the metric
            print(f"# launch-floor probe unavailable ({exc})", file=sys.stderr)
        extra["mobilenetv2_sweep"] = {
            "images_per_s": round(total_batch / (job_sweep_ms * 1e-3), 1), "batch_per_gpu": my_batch,
            "ms_per_batch": round(job_sweep_ms, 4), "timed_as": "one hipGraph replay of the 31 operators" if graph_ms is not None else "sum of per-layer times",
            "sum_of_layer_ms": round(sum_of_layers_ms, 4),
            # the reference bench's own accounting (each layer its own timed run, SURVEY.md 8d: N / sum of layer times);
            # `images_per_s` above is the more conservative whole-sequence replay, inter-kernel gaps included
            "images_per_s_by_sum_of_layers": round(my_batch / (sum_of_layers_ms * 1e-3) * world, 1),
            "hbm_gbs": round(act_bytes / (sweep_ms * 1e-3) / 1e9, 1),
            "frac_of_hbm_peak": round(act_bytes / (sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            # the whole job's %-of-roofline (north_star: images/s AND %-of-roofline at 1 / 2 / 4 / 8 GPUs): every rank moves
            # act_bytes in the slowest rank's time against `world` x the HBM peak -- at N = 1 the line above
            "aggregate_hbm_gbs": round(world * act_bytes / (job_sweep_ms * 1e-3) / 1e9, 1),
            "aggregate_frac_of_hbm_peak": round(act_bytes / (job_sweep_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "frac_of_copy_kernel": round(act_bytes / (sweep_ms * 1e-3) / 1e9 / copy_gbs, 4) if copy_gbs else None,
            "tops": round(total_ops / (sweep_ms * 1e-3) / 1e12, 2),
            "roofline_images_per_s_per_gpu": round(PEAK_HBM_GBS * 1e9 / (act_bytes / my_batch), 1),
            # what 31 dependent launches cost on this box with nothing in them (a hipGraph of empty kernels, one workgroup
            # per CU / sixteen per CU, hip/mfma_probe.hip): the floor under `ms_per_batch - sum_of_layer_ms`
            "empty_kernel_graph_us_per_launch": launch_floor,
            "layers": rows}
        extra["q8dwconv_mobilenetv2_layers"] = {
            "hbm_gbs": round(dw_bytes / (dw_ms * 1e-3) / 1e9, 1), "ms": round(dw_ms, 4),
            "frac_of_hbm_peak": round(dw_bytes / (dw_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "frac_of_copy_kernel": round(dw_bytes / (dw_ms * 1e-3) / 1e9 / copy_gbs, 4) if copy_gbs else None}

        # ---------------------------------------------------------- depthwise 5x5 / dilated 3x3 and a
        # realistic requantization scale (shift >= 1 epilogue; the reference bench's 0.5 takes the shift-0 one)
        more = {}
        for name, (H, W, KH, KW, S, D, G, GIC, GOC), oscale in [
                ("dw5x5_56x56x72_s2", (56, 56, 5, 5, 2, 1, 72, 1, 1), 0.5),      # MobileNetV3-large
                ("dw5x5_28x28x240_s1", (28, 28, 5, 5, 1, 1, 240, 1, 1), 0.5),
                ("dw5x5_14x14x672_s1", (14, 14, 5, 5, 1, 1, 672, 1, 1), 0.5),
                ("dw3x3_dil2_28x28x192", (28, 28, 3, 3, 1, 2, 192, 1, 1), 0.5),   # dilated (DeepLab-style)
                ("dw3x3_56x56x144_scale0.0125", (56, 56, 3, 3, 1, 1, 144, 1, 1), 20.0),
                ("pw_112x112x16_96_scale0.0125", (112, 112, 1, 1, 1, 1, 1, 16, 96), 20.0)]:
            layer = ConvLayer(lib, torch, my_batch, H, W, KH, KW, S, D, G, GIC, GOC, seed=700 + len(more),
                              min_bytes_between_reuse=512 << 20, out_scale=oscale)
            ms = layer.time_ms(2, 10)
            b = layer.in_bytes + layer.out_bytes
            more[name] = {"kernel": layer.kernel, "ms": round(ms, 4), "gbs": round(b / (ms * 1e-3) / 1e9, 1),
                          "frac_of_hbm_peak": round(b / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
                          "requant_scale": round(0.25 / oscale, 6)}
            layer.close()
        extra["q8dwconv_5x5_dilated_and_realistic_scale"] = more
        # the whole sweep at that scale (0.0125, shift 6: the packed shift >= 1 tail of requant.hip.h where the lane forms
        # apply), per layer on rotating buffers -- to be read beside `mobilenetv2_sweep.images_per_s_by_sum_of_layers`
        if not args.no_conv_lists:
            rs = conv_list_bench(lib, torch, my_batch, MOBILENETV2, 2100, out_scale=20.0)
            extra["mobilenetv2_sweep_realistic_scale"] = {
                "requant_scale": 0.0125, "batch_per_gpu": my_batch,
                "images_per_s_by_sum_of_layers": rs["images_per_s_by_sum_of_layers"], "sum_of_layer_ms": rs["sum_of_layer_ms"],
                "vs_scale_0.5_sum_of_layers": round(sum(r["ms"] for r in rows) / rs["sum_of_layer_ms"], 4),
                "layers": [{"layer": i + 1, "kernel": r["kernel"], "us": r["us"]} for i, r in enumerate(rs["layers"])]}

        # ---------------------------------------------------------- the whole network (64 chained operators, one hipGraph)
        extra["mobilenetv2_network"] = network_bench(lib, torch, my_batch, total_batch, world, args.warmup,
                                                     max(args.steps // 2, 5))
        # the same network with the ten residual adds riding in their project convolutions' epilogues
        # (qnnp_gfx950_attach_residual_add: 54 launches, the project outputs are never written)
        extra["mobilenetv2_network_adds_folded"] = network_bench(lib, torch, my_batch, total_batch, world, args.warmup,
                                                                 max(args.steps // 2, 5), fold_adds=True)
        # the same network with every inverted-residual block as ONE fused operator (qnnp_gfx950_create_fused_block)
        extra["mobilenetv2_network_fused"] = network_bench(lib, torch, my_batch, total_batch, world, args.warmup,
                                                           max(args.steps // 2, 5), fuse=True)
        # ... and with the one block that has no expand stage left to its two stand-alone kernels (faster there)
        extra["mobilenetv2_network_fused_expanding_blocks"] = network_bench(lib, torch, my_batch, total_batch, world, args.warmup,
                                                                            max(args.steps // 2, 5), fuse="expanding")

        # ---------------------------------------------------------- SURVEY 8f "next" rows: deconvolution, add, pooling
        extra["next_rows"] = next_rows_bench(lib, torch, my_batch, args.warmup, max(args.steps // 2, 5))

        # ---------------------------------------------------------- the headline GEMM outside its most favourable class, and
        # BASELINE configs[0] (M = 1 fully connected) on the device
        extra["q8gemm_4096_variants"] = {
            "kernel_zero_point_126": gemm_variant_bench(lib, torch, 126, 0.75, 1, 12, 11),     # no centred image: lean kernel
            "requant_scale_0.3_shift1": gemm_variant_bench(lib, torch, 127, 0.3, 1, 12, 12),   # shift >= 1 epilogue
            "kernel_zero_point_126_scale_0.3": gemm_variant_bench(lib, torch, 126, 0.3, 1, 12, 13)}
        extra["q8fc_m1_k1024_n1000"] = fc_m1_bench(lib, torch, 5, 50)

        # ---------------------------------------------------------- the reference bench's other convolution lists: the general
        # implicit-GEMM path (7x7 s2, 3x3 with 64..512 channels, stride-2 3x3 / 1x1, grouped 1x1), bench/convolution.cc:642-718, 147-184
        if not args.no_conv_lists:
            extra["conv_lists"] = {"resnet18": conv_list_bench(lib, torch, my_batch, RESNET18, 1800),
                                   "resnet50": conv_list_bench(lib, torch, my_batch, RESNET50, 5000),
                                   "shufflenet_v1_g2": conv_list_bench(lib, torch, my_batch, SHUFFLENET_V1_G2, 1200)}
            # ... and every other list of that file (kernel, us and fraction of bound per row: the full record only)
            extra["reference_bench_lists"] = reference_lists_bench(lib, torch, my_batch)
        roofline["secondary"] = secondary_block(extra)

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # the reference's SSE2 path on this box's host cores: the full sample at N = 1, a shorter one at N > 1 (the
        # other ranks wait at the closing barrier meanwhile) -- a SCALE line carries its baseline too
        cpu = cpu_baseline_gemm(12.0 if world == 1 else 5.0)
        if "mobilenetv2_sweep" in extra:
            sweep_cpu = cpu_baseline_sweep(seconds_budget=10.0 if world == 1 else 4.0)
            extra["mobilenetv2_sweep"]["cpu_baseline"] = sweep_cpu
            if sweep_cpu and isinstance(roofline.get("secondary"), dict):
                roofline["secondary"]["c4_cpu_reference_images_per_s"] = sweep_cpu["images_per_s"]
                roofline["secondary"]["c4_cpu_reference_threads"] = sweep_cpu["cores"]

    if rank == 0:
        emit(assemble_line(world=world, steps=args.steps, warmup=args.warmup, ms_per_step=ms_per_step,
                           ev_ms_per_rank=ev_ms_per_rank, gemm_kernel=gemm_kernel, info=info,
                           roofline=roofline, cpu=cpu, extra=extra, data="synthetic"), args.full_out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
