"""GPU tier: a whole quantized MobileNetV2 (examples/mobilenetv2.py: 52 convolutions, 10 residual adds, global
average pooling, classifier = 64 operators through the C ABI) against the scalar oracle, EVERY intermediate tensor
bit for bit, run operator by operator and replayed as one hipGraph. Quantization parameters are derived from the
oracle's accumulators layer by layer, as the reference's operator testers derive theirs
(test/convolution-operator-tester.h:407-413), so every tensor spans its 0..255 range instead of saturating."""
import numpy as np
import pytest
import torch

from _cases import output_quantization
from _gpu import from_device
from examples import mobilenetv2 as mnv2
from oracle import o1

pytestmark = pytest.mark.gpu


def oracle_forward(plan, image, batch):
    """Returns ({tensor id: uint8 array}, {op name: Quant}) -- the oracle's tensors and the parameters it chose."""
    tensors = {0: image}
    zps = {0: 127}
    quant = {}
    for op in plan.ops:
        if op.kind == "conv":
            kernel, bias = plan.weights[op.name]
            pad = op.k // 2
            shape = o1.conv_shape(batch, op.hw[0], op.hw[1], (pad,) * 4, (op.k, op.k), (op.stride, op.stride), (1, 1),
                                  op.groups, op.gic, op.goc)
            acc = o1.conv2d_acc(shape, tensors[op.src[0]], kernel, bias, zps[op.src[0]], 127)
            oscale, ozp = output_quantization(acc)
            cout = op.groups * op.goc
            out = o1.requantize_rows(acc.reshape(-1, cout), np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
            quant[op.name] = mnv2.Quant(in_zp=zps[op.src[0]], kernel_zp=127, out_scale=float(oscale), out_zp=ozp)
        elif op.kind == "add":
            a, b = tensors[op.src[0]], tensors[op.src[1]]
            out = np.empty_like(a)
            rows = a.size // op.channels
            o1.add_q8(rows, op.channels, zps[op.src[0]], 1.0, zps[op.src[1]], 1.0, 128, 2.0, 0, 255,
                      a, op.channels, b, op.channels, out, op.channels)
            ozp = 128
            quant[op.name] = mnv2.Quant(in_zp=zps[op.src[0]], in2_zp=zps[op.src[1]], out_scale=2.0, out_zp=128)
        elif op.kind == "gap":
            x = tensors[op.src[0]]
            out = np.empty(batch * op.channels, np.uint8)
            ozp = zps[op.src[0]]
            o1.global_average_pooling_q8(batch, op.width, op.channels, ozp, 1.0, ozp, 1.0, 0, 255,
                                         x, op.channels, out, op.channels)
            quant[op.name] = mnv2.Quant(in_zp=ozp, out_scale=1.0, out_zp=ozp)
        else:
            kernel, bias = plan.weights[op.name]
            a = tensors[op.src[0]].reshape(batch, op.gic)
            acc = o1.gemm_acc(a, kernel, bias, zps[op.src[0]], 127)
            oscale, ozp = output_quantization(acc)
            out = o1.requantize_rows(acc, np.float32(1.0) / oscale, ozp, 0, 255).reshape(-1)
            quant[op.name] = mnv2.Quant(in_zp=zps[op.src[0]], kernel_zp=127, out_scale=float(oscale), out_zp=ozp)
        tensors[op.dst] = np.ascontiguousarray(out)
        zps[op.dst] = ozp
    return tensors, quant


@pytest.mark.parametrize("input_hw,batch", [(96, 2), (224, 1)])
def test_whole_network_matches_oracle_tensor_by_tensor(qnnp, input_hw, batch):
    plan = mnv2.build_plan(input_hw=input_hw, classes=1000, seed=0x51A0 + input_hw)
    rng = np.random.default_rng(input_hw)
    image = rng.integers(0, 256, size=batch * input_hw * input_hw * 3, dtype=np.uint8)
    o1.set_threads(16)
    try:
        expected, quant = oracle_forward(plan, image, batch)
    finally:
        o1.set_threads(1)
    assert len(plan.ops) == 64 and sum(op.kind == "conv" for op in plan.ops) == 52

    net = mnv2.DeviceNetwork(qnnp, torch, plan, batch, quant)
    try:
        net.buffers[0].copy_(torch.from_numpy(image))
        net.run()                                                   # operator by operator
        for op in plan.ops:
            got = from_device(net.buffers[op.dst])
            bad = np.flatnonzero(got != expected[op.dst])
            assert bad.size == 0, f"{op.name} ({net.kernels[op.name]}): {bad.size} of {got.size} bytes differ"
        # the same forward pass as ONE hipGraph: wipe every tensor, replay, compare again
        net.capture()
        for t in net.buffers:
            if t != 0:
                net.buffers[t].zero_()
        torch.cuda.synchronize()
        net.replay()
        last = plan.ops[-1].dst
        assert np.array_equal(from_device(net.buffers[last]), expected[last]), "graph replay: classifier output"
        for op in plan.ops:
            assert np.array_equal(from_device(net.buffers[op.dst]), expected[op.dst]), f"graph replay: {op.name}"
        # every kernel family of the hot path took part
        used = set(net.kernels.values())
        assert any(k.startswith("q8_dwconv") for k in used) and any(k.startswith("q8_pw_stream") for k in used)
        assert "q8_vadd_flat" in used and any(k.startswith("q8_gavgpool") for k in used)
    finally:
        net.close()
