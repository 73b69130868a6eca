"""GPU tier: the multi-GPU contract of the hot path, on however many GPUs the box has.

north_star shards batches across the GPUs of a node with no collective: every output pixel depends on one image
(reference src/operator-run.c:675-679, 797-802, 837-842 -- batch / pixels are independent grid dimensions). So a
batch run as per-rank shards must be BYTE-IDENTICAL to the unsharded run. The reference also lets distinct
operators run from different threads at once (stack-local run contexts, src/operator-run.c:783-795): checked here
with two threads on one device and, when the box has two GPUs, one thread per device in ONE process.
"""
import threading

import numpy as np
import pytest

from _cases import ConvCase, FcCase, conv_tensors, fc_tensors
from _gpu import from_device, to_device
from _runner import FILL, assert_bytes_equal, conv_expected, fc_expected
from qnnpack_amd import Status
from qnnpack_amd.shard import shard_batch

pytestmark = pytest.mark.gpu


def _create_conv(lib, case, kernel, bias, quant):
    oscale, ozp = quant
    return lib.create_convolution2d_nhwc_q8(
        case.padding[0], case.padding[1], case.padding[2], case.padding[3],
        case.kernel_size[0], case.kernel_size[1], case.subsampling[0], case.subsampling[1],
        case.dilation[0], case.dilation[1], case.groups, case.gic, case.goc,
        case.izp, 1.0, case.kzp, 1.0, kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)


SHARD_CASES = [
    ConvCase("shard_3x3_c64", (28, 28), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=6),        # LDS-tiled convolution
    ConvCase("shard_dw3x3_c144", (28, 28), (3, 3), (1, 1, 1, 1), groups=144, batch=7),         # depthwise, odd split
    ConvCase("shard_1x1_c32_96", (56, 56), gic=32, goc=96, batch=5),                           # streaming pointwise
    ConvCase("shard_3x3s2_c3", (64, 64), (3, 3), (1, 1, 1, 1), subsampling=(2, 2), gic=3, goc=32, batch=4),
]


@pytest.mark.parametrize("case", SHARD_CASES, ids=lambda c: c.name)
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_batch_is_byte_identical_to_unsharded(qnnp, case, world):
    """Rank r runs images [start, start+count) with its OWN operator (weights replicated), exactly as bench.py's ranks
    do; the shards' outputs concatenated equal the unsharded run and the oracle."""
    inp, kernel, bias = conv_tensors(case)
    expected, quant, (oh, ow) = conv_expected(case, inp, kernel, bias)
    H, W = case.input_size
    cin, cout = case.groups * case.gic, case.groups * case.goc
    in_img, out_img = H * W * cin, oh * ow * cout
    d_in = to_device(inp)

    # unsharded
    d_out = to_device(np.full(case.batch * out_img, FILL, np.uint8))
    op = _create_conv(qnnp, case, kernel, bias, quant)
    qnnp.setup_convolution2d_nhwc_q8(op, case.batch, H, W, d_in, cin, d_out, cout)
    qnnp.run_operator(op)
    whole = from_device(d_out)
    qnnp.delete_operator(op)
    assert_bytes_equal(whole, expected, f"{case.name} unsharded")

    # sharded: one operator per rank over its slice of the same device tensors (no collective, no copies)
    d_out2 = to_device(np.full(case.batch * out_img, FILL, np.uint8))
    ops = []
    for rank in range(world):
        start, count = shard_batch(case.batch, world, rank)
        rop = _create_conv(qnnp, case, kernel, bias, quant)
        ops.append(rop)
        if count:
            qnnp.setup_convolution2d_nhwc_q8(rop, count, H, W, d_in[start * in_img:], cin, d_out2[start * out_img:], cout)
        else:
            qnnp.setup_convolution2d_nhwc_q8(rop, 0, H, W, d_in, cin, d_out2, cout)
    for rop in ops:
        qnnp.run_operator(rop)
    sharded = from_device(d_out2)
    for rop in ops:
        qnnp.delete_operator(rop)
    assert_bytes_equal(sharded, whole, f"{case.name} in {world} shards vs unsharded")


def test_set_device_after_initialize_is_per_thread(qnnp):
    n = qnnp.device_count()
    assert n >= 1
    assert qnnp.get_device() == 0
    qnnp.set_device(0)                                   # re-selecting the bound device is fine
    assert qnnp.get_device() == 0
    assert qnnp.lib.qnnp_gfx950_set_device(n) == Status.invalid_parameter      # one past the last GPU
    assert qnnp.lib.qnnp_gfx950_set_device(-1) == Status.invalid_parameter
    assert qnnp.get_device() == 0
    seen = {}

    def other_thread():
        seen["before"] = qnnp.get_device()               # a fresh thread starts on the primary device
        qnnp.set_device(n - 1)
        seen["after"] = qnnp.get_device()
    t = threading.Thread(target=other_thread)
    t.start()
    t.join()
    assert seen == {"before": 0, "after": n - 1}
    assert qnnp.get_device() == 0                        # the selection was that thread's own


def test_failed_setup_leaves_the_operator_unrunnable(qnnp):
    case = ConvCase("setup_fail", (12, 12), (3, 3), (1, 1, 1, 1), gic=16, goc=16, batch=2)
    inp, kernel, bias = conv_tensors(case)
    expected, quant, (oh, ow) = conv_expected(case, inp, kernel, bias)
    d_in, d_out = to_device(inp), to_device(np.full(expected.size, FILL, np.uint8))
    op = _create_conv(qnnp, case, kernel, bias, quant)
    assert qnnp.run_operator_status(op) == Status.invalid_parameter           # never set up
    qnnp.setup_convolution2d_nhwc_q8(op, case.batch, 12, 12, d_in, 16, d_out, 16)
    qnnp.run_operator(op)
    assert_bytes_equal(from_device(d_out), expected, "before the failed setup")
    # validation failure: nothing of the operator was touched, the previous setup stays runnable (as the reference)
    assert qnnp.setup_convolution2d_nhwc_q8_status(op, case.batch, 0, 12, d_in, 16, d_out, 16) == Status.invalid_parameter
    qnnp.run_operator(op)
    # failure AFTER the geometry was rewritten (index range): the operator must refuse to run, not launch stale tables
    st = qnnp.setup_convolution2d_nhwc_q8_status(op, 1 << 22, 1 << 10, 1 << 10, d_in, 16, d_out, 16)
    assert st == Status.unsupported_parameter
    assert qnnp.run_operator_status(op) == Status.invalid_parameter
    # and a good setup revives it
    qnnp.setup_convolution2d_nhwc_q8(op, case.batch, 12, 12, d_in, 16, d_out, 16)
    qnnp.run_operator(op)
    assert_bytes_equal(from_device(d_out), expected, "after re-setup")
    qnnp.delete_operator(op)


def _thread_job(lib, case, rounds, errors, tag, device=None, capture=False):
    """create -> (setup -> run -> compare) x rounds -> delete, all from this thread."""
    try:
        import torch
        if device is not None:
            lib.set_device(device)
            torch.cuda.set_device(device)
        if isinstance(case, FcCase):
            inp, kernel, bias = fc_tensors(case)
            expected, (oscale, ozp) = fc_expected(case, inp, kernel, bias)
            op = lib.create_fully_connected_nc_q8(case.input_channels, case.output_channels, case.izp, 1.0, case.kzp, 1.0,
                                                  kernel, bias, ozp, float(oscale), case.qmin, case.qmax, 0)
        else:
            inp, kernel, bias = conv_tensors(case)
            expected, quant, _ = conv_expected(case, inp, kernel, bias)
            op = _create_conv(lib, case, kernel, bias, quant)
        dev = f"cuda:{device}" if device is not None else "cuda"
        d_in = torch.from_numpy(inp.copy()).to(dev)
        for r in range(rounds):
            d_out = torch.full((expected.size,), FILL, dtype=torch.uint8, device=dev)
            if isinstance(case, FcCase):
                lib.setup_fully_connected_nc_q8(op, case.batch, d_in, case.in_stride, d_out, case.out_stride)
            else:
                lib.setup_convolution2d_nhwc_q8(op, case.batch, case.input_size[0], case.input_size[1],
                                                d_in, case.in_stride, d_out, case.out_stride)
            if capture and r % 2 == 1:
                # this thread records its launch into a hipGraph while the other thread keeps launching normally
                lib.graph_begin()
                lib.run_operator(op)
                graph = lib.graph_end()
                lib.graph_launch(graph)
                lib.graph_destroy(graph)
            else:
                lib.run_operator(op)
            got = d_out.cpu().numpy()
            if not np.array_equal(got, expected):
                errors.append(f"{tag}: round {r}: {int((got != expected).sum())} bytes differ")
                break
        lib.delete_operator(op)
    except Exception as exc:  # noqa: BLE001 -- reported by the main thread
        errors.append(f"{tag}: {type(exc).__name__}: {exc}")


def test_two_threads_run_distinct_operators_concurrently(qnnp):
    """reference src/operator-run.c:783-795: run contexts are per call, so distinct operators may run from different
    threads. Here: a 3x3 convolution and a fully connected operator, 24 setup/run rounds each, one of the threads
    capturing hipGraphs half of the time (capture state is thread-local)."""
    conv = ConvCase("mt_conv3x3", (20, 20), (3, 3), (1, 1, 1, 1), gic=32, goc=48, batch=3)
    fc = FcCase("mt_fc", 192, 320, 260)
    dw = ConvCase("mt_dw", (24, 24), (3, 3), (1, 1, 1, 1), groups=96, batch=2)
    errors = []
    qnnp.set_stream(None)      # the threads share the device's default stream
    try:
        threads = [threading.Thread(target=_thread_job, args=(qnnp, conv, 24, errors, "conv", None, True)),
                   threading.Thread(target=_thread_job, args=(qnnp, fc, 24, errors, "fc")),
                   threading.Thread(target=_thread_job, args=(qnnp, dw, 24, errors, "dw"))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
            assert not t.is_alive(), "worker thread hung"
    finally:
        import torch
        qnnp.set_stream(torch.cuda.current_stream().cuda_stream)
    assert not errors, errors


def test_one_process_drives_two_gpus(qnnp):
    """one host thread + one stream per device. Needs >= 2 GPUs; on the 1-GPU box this is a skip (the
    per-device machinery is still exercised by every other test through device 0)."""
    if qnnp.device_count() < 2:
        pytest.skip("box has one GPU")
    case = ConvCase("two_gpu_3x3", (28, 28), (3, 3), (1, 1, 1, 1), gic=64, goc=64, batch=4)
    errors = []
    threads = [threading.Thread(target=_thread_job, args=(qnnp, case, 6, errors, f"gpu{d}", d)) for d in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
        assert not t.is_alive()
    assert not errors, errors
    # memory of GPU 1 handed to an operator of GPU 0 is refused, not dereferenced
    import torch
    inp, kernel, bias = conv_tensors(case)
    expected, quant, _ = conv_expected(case, inp, kernel, bias)
    op = _create_conv(qnnp, case, kernel, bias, quant)
    foreign = torch.zeros(inp.size, dtype=torch.uint8, device="cuda:1")
    out = torch.zeros(expected.size, dtype=torch.uint8, device="cuda:0")
    assert qnnp.setup_convolution2d_nhwc_q8_status(op, case.batch, 28, 28, foreign, 64, out, 64) == Status.invalid_parameter
    qnnp.delete_operator(op)


def test_fused_block_lives_on_its_members_device_not_the_callers(qnnp):
    """A fused block borrows its members' device images and uploads its own (strip images) at create: those must land on
    the MEMBERS' device whatever device the creating thread has selected. On one GPU the same sequence runs with device 0
    on both sides (create from a worker thread that never selected a device); with two GPUs the members live on GPU 1
    while the creator has GPU 0 selected, and the block is then set up and run on GPU 1 and compared with its parts."""
    import torch
    two = qnnp.device_count() >= 2
    member_dev = 1 if two else 0
    rng = np.random.default_rng(77)
    k1 = rng.integers(0, 256, (1, 96, 1, 1, 16), dtype=np.uint8)
    kd = rng.integers(0, 256, (96, 1, 3, 3, 1), dtype=np.uint8)
    k3 = rng.integers(0, 256, (1, 24, 1, 1, 96), dtype=np.uint8)
    b96, b24 = rng.integers(-5000, 5000, 96).astype(np.int32), rng.integers(-5000, 5000, 24).astype(np.int32)
    made = {}

    def make_members():
        qnnp.set_device(member_dev)
        made["ex"] = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 16, 96, 120, 0.02, 127, 0.01, k1, b96, 110, 0.05, 0, 255, 0)
        made["dw"] = qnnp.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 96, 1, 1, 110, 0.05, 127, 0.01, kd, b96, 100, 0.06, 0, 255, 0)
        made["pr"] = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 96, 24, 100, 0.06, 127, 0.01, k3, b24, 128, 0.07, 0, 255, 0)
    t = threading.Thread(target=make_members)
    t.start(); t.join(timeout=120)
    assert set(made) == {"ex", "dw", "pr"}
    assert qnnp.get_device() == 0                          # this thread still has device 0 selected
    fused = qnnp.create_fused_block(made["ex"], made["dw"], made["pr"])
    dev = f"cuda:{member_dev}"
    batch, hw = 2, 20
    x = torch.randint(0, 256, (batch * hw * hw * 16,), dtype=torch.uint8, device=dev)
    h1 = torch.empty(batch * hw * hw * 96, dtype=torch.uint8, device=dev)
    h2 = torch.empty_like(h1)
    want = torch.empty(batch * hw * hw * 24, dtype=torch.uint8, device=dev)
    got = torch.full_like(want, FILL)
    qnnp.set_stream(None)
    try:
        qnnp.setup_convolution2d_nhwc_q8(made["ex"], batch, hw, hw, x, 16, h1, 96)
        qnnp.setup_convolution2d_nhwc_q8(made["dw"], batch, hw, hw, h1, 96, h2, 96)
        qnnp.setup_convolution2d_nhwc_q8(made["pr"], batch, hw, hw, h2, 96, want, 24)
        for name in ("ex", "dw", "pr"):
            qnnp.run_operator(made[name])
        qnnp.setup_fused_block(fused, batch, hw, hw, x, 16, got, 24)
        qnnp.run_operator(fused)
        torch.cuda.synchronize(member_dev)
        assert torch.equal(got, want), "fused block differs from its stand-alone chain"
        if two:
            # members of two devices in one block are refused
            other = qnnp.create_convolution2d_nhwc_q8(0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 96, 24, 100, 0.06, 127, 0.01, k3, b24, 128, 0.07, 0, 255, 0)
            assert qnnp.create_fused_block_status(made["ex"], made["dw"], other)[0] == Status.invalid_parameter
            qnnp.delete_operator(other)
    finally:
        qnnp.set_stream(torch.cuda.current_stream().cuda_stream)
        qnnp.delete_operator(fused)
        for h in made.values():
            qnnp.delete_operator(h)


def test_thread_per_gpu_example_runs(qnnp):
    """examples/multi_gpu_threads.py: the one-process / one-thread-per-GPU form on however many GPUs the box has (a second
    caller of the path the two-GPU test above covers; with one GPU it still selects and binds the device from a worker
    thread and checks the shard against the unsharded run)."""
    import torch
    from examples import multi_gpu_threads
    qnnp.set_stream(None)
    try:
        assert multi_gpu_threads.main(["--batch", "6", "--rounds", "2"]) == 0
    finally:
        qnnp.set_stream(torch.cuda.current_stream().cuda_stream)
        qnnp.set_device(0)
        qnnp.set_async(False)
