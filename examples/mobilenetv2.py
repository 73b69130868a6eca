"""A whole quantized MobileNetV2 (width 1.0) through the QNNPACK C ABI on gfx950.

The reference's benchmark (bench/convolution.cc:453-536) times the network's 31 DISTINCT convolution shapes one
by one. This module chains the real thing -- 52 convolutions (first 3x3, 17 inverted-residual blocks, last 1x1), the
10 residual adds, global average pooling and the classifier -- on device buffers, every operator created through
`include/qnnpack.h`, and replays the 65 launches as ONE hipGraph (`qnnp_gfx950_graph_*`). It is a caller of the
library, not part of it: `bench.py` reports its images/s next to the per-layer sweep, and
`tests/test_gpu_network.py` checks every intermediate tensor against the scalar oracle, layer by layer.

Quantization: every weighted operator is created with input / kernel scale 1 and an output scale + zero point handed
in per operator (`Quant`): the bench uses the reference bench's constants, the parity test derives them from the
oracle's accumulators exactly as the reference's operator testers do.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# torchvision / the paper's table 2: (expansion t, output channels c, repeats n, first stride s)
BLOCKS = [(1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1)]


@dataclass
class Op:
    kind: str                      # "conv" | "add" | "gap" | "fc"
    name: str
    src: Tuple[int, ...]           # tensor ids read
    dst: int                       # tensor id written
    # convolution geometry (kind == "conv")
    hw: Tuple[int, int] = (0, 0)
    k: int = 1
    stride: int = 1
    groups: int = 1
    gic: int = 0
    goc: int = 0
    channels: int = 0              # add / gap / fc output width
    width: int = 0                 # gap: pixels per image


@dataclass
class Quant:
    """Per-operator quantization handed to create: zero points of the operands, output scale / zero point / clamp."""
    in_zp: int = 127
    in2_zp: int = 127
    kernel_zp: int = 127
    out_scale: float = 2.0         # input and kernel scales are 1: requantization scale = 1 / out_scale (0.5, the
                                   # reference bench's 0.5 * 0.5 / 0.5, bench/convolution.cc:71-74)
    out_zp: int = 127
    qmin: int = 0
    qmax: int = 255


@dataclass
class Plan:
    ops: List[Op] = field(default_factory=list)
    shapes: Dict[int, Tuple[int, int, int]] = field(default_factory=dict)   # tensor id -> (H, W, C); H = W = 1 after pooling
    weights: Dict[str, Tuple[np.ndarray, np.ndarray]] = field(default_factory=dict)


def _out_hw(h, k, s):
    pad = k // 2
    return (h + 2 * pad - k) // s + 1


def build_plan(input_hw: int = 224, classes: int = 1000, seed: int = 0x51A0) -> Plan:
    rng = np.random.default_rng(seed)
    plan = Plan()
    tid = 0
    plan.shapes[0] = (input_hw, input_hw, 3)

    def conv(name, src, k, stride, groups, gic, goc):
        nonlocal tid
        h, w, c = plan.shapes[src]
        assert c == groups * gic, (name, c, groups, gic)
        tid += 1
        oh, ow = _out_hw(h, k, stride), _out_hw(w, k, stride)
        plan.shapes[tid] = (oh, ow, groups * goc)
        plan.ops.append(Op("conv", name, (src,), tid, hw=(h, w), k=k, stride=stride, groups=groups, gic=gic, goc=goc))
        plan.weights[name] = (rng.integers(0, 256, size=(groups, goc, k, k, gic), dtype=np.uint8),
                              rng.integers(-10000, 10001, size=groups * goc, dtype=np.int32))
        return tid

    x = conv("conv0_3x3s2", 0, 3, 2, 1, 3, 32)
    cin = 32
    b = 0
    for t, c, n, s in BLOCKS:
        for i in range(n):
            stride = s if i == 0 else 1
            inp = x
            hidden = cin * t
            if t != 1:
                x = conv(f"b{b}_expand", x, 1, 1, 1, cin, hidden)
            x = conv(f"b{b}_dw", x, 3, stride, hidden, 1, 1)
            x = conv(f"b{b}_project", x, 1, 1, 1, hidden, c)
            if stride == 1 and cin == c:
                tid += 1
                plan.shapes[tid] = plan.shapes[x]
                plan.ops.append(Op("add", f"b{b}_add", (inp, x), tid, channels=c))
                x = tid
            cin = c
            b += 1
    x = conv("conv_last_1x1", x, 1, 1, 1, cin, 1280)
    h, w, c = plan.shapes[x]
    tid += 1
    plan.shapes[tid] = (1, 1, c)
    plan.ops.append(Op("gap", "global_average_pooling", (x,), tid, channels=c, width=h * w))
    x = tid
    tid += 1
    plan.shapes[tid] = (1, 1, classes)
    plan.ops.append(Op("fc", "classifier", (x,), tid, gic=c, channels=classes))
    plan.weights["classifier"] = (rng.integers(0, 256, size=(classes, c), dtype=np.uint8),
                                  rng.integers(-10000, 10001, size=classes, dtype=np.int32))
    return plan


def tensor_bytes(plan: Plan, tid: int, batch: int) -> int:
    h, w, c = plan.shapes[tid]
    return batch * h * w * c


def algorithmic_bytes(plan: Plan, batch: int) -> int:
    """Activation bytes every operator reads and writes (weights excluded), the figure a per-operator HBM roofline uses."""
    total = 0
    for op in plan.ops:
        total += sum(tensor_bytes(plan, s, batch) for s in op.src) + tensor_bytes(plan, op.dst, batch)
    return total


def operations(plan: Plan, batch: int) -> int:
    """2 * MACs of the convolutions and the classifier, the reference's op count (bench/convolution.cc:100-104)."""
    ops = 0
    for op in plan.ops:
        if op.kind == "conv":
            oh, ow, _ = plan.shapes[op.dst]
            ops += 2 * batch * oh * ow * op.groups * op.gic * op.goc * op.k * op.k
        elif op.kind == "fc":
            ops += 2 * batch * op.gic * op.channels
    return ops


def blocks_of(plan: Plan):
    """The inverted-residual blocks of the plan as index tuples (expand | None, depthwise, project, add | None)."""
    out = []
    ops = plan.ops
    i = 0
    while i < len(ops):
        op = ops[i]
        if op.kind == "conv" and op.name.endswith("_dw"):
            ex = i - 1 if i > 0 and ops[i - 1].name.endswith("_expand") else None
            pr = i + 1
            ad = i + 2 if i + 2 < len(ops) and ops[i + 2].kind == "add" else None
            out.append((ex, i, pr, ad))
            i = (ad if ad is not None else pr) + 1
        else:
            i += 1
    return out


class DeviceNetwork:
    """The plan's operators created through the C ABI and bound to device buffers (one per tensor).

    fuse=True additionally builds one fused operator per inverted-residual block from the stand-alone operators
    (qnnp_gfx950_create_fused_block) and runs it in their place wherever the fused kernel takes the block; the
    expanded tensors of those blocks are then never written. fuse="expanding": the same for the blocks that have an
    expand stage only."""

    def __init__(self, lib, torch, plan: Plan, batch: int, quant: Optional[Dict[str, Quant]] = None, fuse: bool = False,
                 fold_adds: bool = False):
        self.lib, self.plan, self.batch = lib, plan, batch
        self.buffers = {t: torch.empty(tensor_bytes(plan, t, batch), dtype=torch.uint8, device="cuda")
                        for t in plan.shapes}
        self.handles = []
        self.kernels = {}
        self._plain_stores = set()
        quant = quant or {}
        for op in plan.ops:
            q = quant.get(op.name, Quant())
            src = [self.buffers[s] for s in op.src]
            dst = self.buffers[op.dst]
            if op.kind == "conv":
                kernel, bias = plan.weights[op.name]
                pad = op.k // 2
                h = lib.create_convolution2d_nhwc_q8(pad, pad, pad, pad, op.k, op.k, op.stride, op.stride, 1, 1,
                                                     op.groups, op.gic, op.goc, q.in_zp, 1.0, q.kernel_zp, 1.0,
                                                     kernel, bias, q.out_zp, float(q.out_scale), q.qmin, q.qmax, 0)
                cin, cout = op.groups * op.gic, op.groups * op.goc
                lib.setup_convolution2d_nhwc_q8(h, batch, op.hw[0], op.hw[1], src[0], cin, dst, cout)
            elif op.kind == "add":
                h = lib.create_add_nc_q8(op.channels, q.in_zp, 1.0, q.in2_zp, 1.0, q.out_zp, float(q.out_scale),
                                         q.qmin, q.qmax, 0)
                hh, ww, c = plan.shapes[op.dst]
                lib.setup_add_nc_q8(h, batch * hh * ww, src[0], c, src[1], c, dst, c)
            elif op.kind == "gap":
                h = lib.create_global_average_pooling_nwc_q8(op.channels, q.in_zp, 1.0, q.out_zp, float(q.out_scale),
                                                             q.qmin, q.qmax, 0)
                lib.setup_global_average_pooling_nwc_q8(h, batch, op.width, src[0], op.channels, dst, op.channels)
            else:
                kernel, bias = plan.weights[op.name]
                h = lib.create_fully_connected_nc_q8(op.gic, op.channels, q.in_zp, 1.0, q.kernel_zp, 1.0, kernel, bias,
                                                     q.out_zp, float(q.out_scale), q.qmin, q.qmax, 0)
                lib.setup_fully_connected_nc_q8(h, batch, src[0], op.gic, dst, op.channels)
            self.handles.append(h)
        self.graph = None
        # execution schedule: (name, handle) in order; fused blocks replace their members
        self.fused = {}
        self.fused_handles = []
        self.schedule = [(op.name, h) for op, h in zip(plan.ops, self.handles)]
        # fold_adds: every residual add rides in its project convolution (qnnp_gfx950_attach_residual_add): the
        # convolution is re-bound to write the add's output tensor, the project output tensor is never written
        self.folded = {}
        if fold_adds and not fuse:
            skip = set()
            for i, op in enumerate(plan.ops):
                if op.kind != "add" or i == 0 or plan.ops[i - 1].dst != op.src[1] or plan.ops[i - 1].kind != "conv":
                    continue
                conv = plan.ops[i - 1]
                cin, cout = conv.groups * conv.gic, conv.groups * conv.goc
                lib.setup_convolution2d_nhwc_q8(self.handles[i - 1], batch, conv.hw[0], conv.hw[1],
                                                self.buffers[conv.src[0]], cin, self.buffers[op.dst], cout)
                lib.attach_residual_add(self.handles[i - 1], self.handles[i], self.buffers[op.src[0]], cout)
                self.folded[conv.name] = (i - 1, i)
                skip.add(i)
            self.schedule = [(op.name, h) for i, (op, h) in enumerate(zip(plan.ops, self.handles)) if i not in skip]
        if fuse:
            from qnnpack_amd import QnnpackError
            replaced = {}
            for ex, dw, pr, ad in blocks_of(plan):
                first = ex if ex is not None else dw
                last = ad if ad is not None else pr
                if fuse == "expanding" and ex is None:
                    # the block without an expand stage (MobileNetV2's first): its "hidden" tensor IS its input, fusing saves
                    # one write + read of the depthwise output only, and its two stand-alone kernels are
                    # faster (38 against 53 us at batch 128): a builder that wants the last 3 % leaves it alone
                    continue
                try:
                    fh = lib.create_fused_block(self.handles[ex] if ex is not None else None, self.handles[dw],
                                                self.handles[pr], self.handles[ad] if ad is not None else None)
                except QnnpackError:
                    continue
                src = plan.ops[first].src[0]
                dst = plan.ops[last].dst
                hh, ww, cin = plan.shapes[src]
                cout = plan.shapes[dst][2]
                st = lib.setup_fused_block_status(fh, batch, hh, ww, self.buffers[src], cin, self.buffers[dst], cout)
                if st != 0:                       # outside the fused kernel's range: the stand-alone operators run it
                    lib.delete_operator(fh)
                    continue
                self.fused_handles.append(fh)
                name = plan.ops[dw].name.replace("_dw", "_fused")
                self.fused[name] = (first, last)
                replaced[first] = (name, fh, last)
            schedule, i = [], 0
            while i < len(plan.ops):
                if i in replaced:
                    name, fh, last = replaced[i]
                    schedule.append((name, fh))
                    i = last + 1
                else:
                    schedule.append((plan.ops[i].name, self.handles[i]))
                    i += 1
            self.schedule = schedule

    # Every operator's output is the next operator's input: keep the outputs cacheable. The library's default marks
    # whole-line output stores as streaming ("streaming_stores" = 1), which is right for an operator run on its own --
    # the per-layer sweep gains 4-5 % -- and costs a chained network about 1 %: the consumer finds nothing of a streamed
    # tensor in the last-level cache. The hint is a property of each of THIS network's operators
    # (qnnp_gfx950_operator_set_streaming_stores): the process-wide option, and whatever other threads launch, stay as they are.
    def _launch_all(self, record_names: bool):
        for name, h in self.schedule:
            if h not in self._plain_stores:
                self.lib.operator_set_streaming_stores(h, 0)
                self._plain_stores.add(h)
            self.lib.run_operator(h)
            if record_names:
                self.kernels[name] = self.lib.operator_kernel(h)

    def run(self):
        """One forward pass, operator by operator."""
        self._launch_all(True)

    def capture(self):
        """Record the whole forward pass into one hipGraph (device pointers only; nothing runs yet)."""
        self.lib.graph_begin()
        try:
            self._launch_all(False)
        finally:
            self.graph = self.lib.graph_end()
        return self.graph

    def replay(self):
        self.lib.graph_launch(self.graph)
        self.lib.graph_synchronize(self.graph)

    def time_ms(self, warmup: int, iters: int) -> float:
        return self.lib.graph_time(self.graph, warmup, iters)

    def close(self):
        if self.graph is not None:
            self.lib.graph_destroy(self.graph)
            self.graph = None
        for h in self.fused_handles:
            self.lib.delete_operator(h)
        for h in self.handles:
            self.lib.delete_operator(h)
        self.handles = []
        self.fused_handles = []
        self.buffers = {}
