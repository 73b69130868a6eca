"""GPU tier: the fused inverted-residual block operator (qnnp_gfx950_create/setup_fused_block; hip/q8fusedstrip.hip,
round 4's strip kernel, and hip/q8fused.hip, the tile kernel it replaces where it applies) against the scalar oracle
and against the stand-alone operators it is built from: EVERY block of a MobileNetV2 (expand / no expand, stride 1 / 2,
with / without residual, 16..960 hidden channels, 112^2 .. 7^2 pixels, strip and tile edges at odd sizes), bit for
bit; the whole network with the blocks fused, eagerly and as one hipGraph; forced strip heights; the bench's batch; and
the create / setup status codes."""
import numpy as np
import pytest
import torch

from _gpu import from_device, to_device
from examples import mobilenetv2 as mnv2
from oracle import o1
from test_gpu_network import oracle_forward

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_kernel,rows,weights", [(0, 0, 0), (0, 1, 0), (0, 3, 0), (1, 0, 0), (0, 2, 0)],
                         ids=["auto", "strips_of_1", "strips_of_3", "tile_kernel", "strips_of_2"])
@pytest.mark.parametrize("input_hw,batch", [(96, 2), (224, 1), (72, 3)])
def test_fused_network_matches_oracle(qnnp, input_hw, batch, fused_kernel, rows, weights):
    """("fused_weights" = 1, a chunk's expand / project fragments staged in LDS by LDS-DMA, exists in measurement builds only:
    it measured level -- DESIGN 4.7b; the product library ignores the option)"""
    qnnp.set_option("fused_kernel", fused_kernel)
    qnnp.set_option("fused_rows", rows)
    qnnp.set_option("fused_weights", weights)
    try:
        _fused_network_matches_oracle(qnnp, input_hw, batch, fused_kernel)
    finally:
        qnnp.set_option("fused_kernel", 0)
        qnnp.set_option("fused_rows", 0)
        qnnp.set_option("fused_weights", 0)


def _fused_network_matches_oracle(qnnp, input_hw, batch, fused_kernel):
    plan = mnv2.build_plan(input_hw=input_hw, classes=1000, seed=0x51A0 + input_hw)
    rng = np.random.default_rng(1000 + input_hw)
    image = rng.integers(0, 256, size=batch * input_hw * input_hw * 3, dtype=np.uint8)
    o1.set_threads(16)
    try:
        expected, quant = oracle_forward(plan, image, batch)
    finally:
        o1.set_threads(1)
    net = mnv2.DeviceNetwork(qnnp, torch, plan, batch, quant, fuse=True)
    try:
        if fused_kernel == 1:
            # the tile kernel: the blocks whose weights fit LDS beside the tiles (b0..b9 of 17), the rest stay stand-alone
            assert len(net.fused) >= 8, (len(net.fused), sorted(net.fused))
        else:
            assert len(net.fused) == 17, (len(net.fused), sorted(net.fused))      # the strip kernel takes every block
        net.buffers[0].copy_(torch.from_numpy(image))
        net.run()
        hidden = set()
        for first, last in net.fused.values():
            hidden.update(plan.ops[i].dst for i in range(first, last))       # tensors the fused blocks never write
        for op in plan.ops:
            if op.dst in hidden:
                continue
            got = from_device(net.buffers[op.dst])
            bad = np.flatnonzero(got != expected[op.dst])
            assert bad.size == 0, f"{op.name}: {bad.size} of {got.size} bytes differ (first at {bad[:4].tolist()})"
        fused_names = {net.kernels[name] for name in net.fused}
        assert fused_names == ({"q8_fused_block"} if fused_kernel == 1 else {"q8_fused_strip"}), fused_names
        # one hipGraph replay of the fused schedule
        net.capture()
        for t in net.buffers:
            if t != 0:
                net.buffers[t].zero_()
        torch.cuda.synchronize()
        net.replay()
        last = plan.ops[-1].dst
        assert np.array_equal(from_device(net.buffers[last]), expected[last])
    finally:
        net.close()


@pytest.mark.parametrize("rows,weights", [(0, 0), (2, 0), (4, 0)], ids=["auto", "strips_of_2", "strips_of_4"])
def test_fused_network_equals_the_unfused_one_at_the_bench_batch(qnnp, rows, weights):
    """Batch 128, bench.py's network and quantization (extra.mobilenetv2_network_fused): strip heights, chunk widths and
    the workgroup count depend on the batch, so the fused chain is compared here, tensor by tensor, with the stand-alone
    operators' chain on the same random images (those operators are held to the oracle and, at this batch, to the
    compiled reference by tests/test_gpu_sweep_bench_batch.py)."""
    batch = 128
    plan = mnv2.build_plan()
    gen = torch.Generator(device="cuda")
    gen.manual_seed(91)
    plain = mnv2.DeviceNetwork(qnnp, torch, plan, batch)
    try:
        image = torch.randint(0, 256, (plain.buffers[0].numel(),), dtype=torch.uint8, device="cuda", generator=gen)
        plain.buffers[0].copy_(image)
        plain.run()
        torch.cuda.synchronize()
        want = {op.dst: plain.buffers[op.dst].clone() for op in plan.ops}
    finally:
        plain.close()
    qnnp.set_option("fused_rows", rows)
    qnnp.set_option("fused_weights", weights)
    try:
        net = mnv2.DeviceNetwork(qnnp, torch, plan, batch, fuse=True)
    finally:
        qnnp.set_option("fused_rows", 0)
        qnnp.set_option("fused_weights", 0)
    try:
        assert len(net.fused) == 17, sorted(net.fused)
        net.buffers[0].copy_(image)
        net.run()
        torch.cuda.synchronize()
        hidden = set()
        for first, last in net.fused.values():
            hidden.update(plan.ops[i].dst for i in range(first, last))
        for op in plan.ops:
            if op.dst in hidden:
                continue
            assert torch.equal(net.buffers[op.dst], want[op.dst]), f"{op.name}: fused chain differs from the stand-alone chain"
        assert {net.kernels[name] for name in net.fused} == {"q8_fused_strip"}
    finally:
        net.close()


def test_status_codes(qnnp):
    from qnnpack_amd import Status
    k1 = np.zeros((1, 32, 1, 1, 16), np.uint8)
    kd = np.zeros((32, 1, 3, 3, 1), np.uint8)
    k3 = np.zeros((1, 16, 1, 1, 32), np.uint8)
    b16, b32 = np.zeros(16, np.int32), np.zeros(32, np.int32)
    ex = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 16, 32, 1, 1.0, 2, 1.0, k1, b32, 3, 2.0, 0, 255, 0)
    dw = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 32, 1, 1, 3, 1.0, 2, 1.0, kd, b32, 4, 2.0, 0, 255, 0)
    pr = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 32, 16, 4, 1.0, 2, 1.0, k3, b16, 5, 2.0, 0, 255, 0)
    dw5 = qnnp.create_convolution2d_nhwc_q8(2, 2, 2, 2, 5, 5, 1, 1, 1, 1, 32, 1, 1, 3, 1.0, 2, 1.0,
                                            np.zeros((32, 1, 5, 5, 1), np.uint8), b32, 4, 2.0, 0, 255, 0)
    add = qnnp.create_add_nc_q8(16, 1, 1.0, 5, 1.0, 6, 2.0, 0, 255, 0)
    add8 = qnnp.create_add_nc_q8(8, 1, 1.0, 5, 1.0, 6, 2.0, 0, 255, 0)
    try:
        st, h = qnnp.create_fused_block_status(ex, dw, pr, add)
        assert st == Status.success and h
        buf = to_device(np.zeros(1 << 16, np.uint8))
        assert qnnp.setup_fused_block_status(h, 0, 0, 0, None, 16, None, 16) == Status.success
        assert qnnp.run_operator_status(h) == Status.success                       # empty batch: no-op
        assert qnnp.setup_fused_block_status(h, 1, 0, 8, buf, 16, buf, 16) == Status.invalid_parameter
        assert qnnp.setup_fused_block_status(h, 1, 8, 8, buf, 15, buf, 16) == Status.invalid_parameter
        assert qnnp.setup_fused_block_status(h, 1, 8, 8, buf, 16, buf, 16) == Status.success
        assert qnnp.setup_convolution2d_nhwc_q8_status(h, 1, 8, 8, buf, 16, buf, 16) == Status.invalid_parameter
        qnnp.delete_operator(h)
        assert qnnp.create_fused_block_status(ex, dw5, pr)[0] == Status.unsupported_parameter     # not a 3x3
        assert qnnp.create_fused_block_status(ex, pr, dw)[0] == Status.unsupported_parameter      # wrong order
        assert qnnp.create_fused_block_status(ex, dw, pr, add8)[0] == Status.unsupported_parameter
        assert qnnp.create_fused_block_status(None, None, pr)[0] == Status.invalid_parameter
    finally:
        for h in (ex, dw, pr, dw5, add, add8):
            qnnp.delete_operator(h)
