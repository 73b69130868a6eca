"""CPU tier, world_size 2 over gloo: the multi-GPU path is batch sharding with no data-path
collective. Checks that the per-rank slices partition the batch, that the
whole-job time is the max over ranks exactly as bench.py uses them, and -- with the scalar oracle
standing in for the device operator, which needs a GPU -- that a convolution run as two rank-local
shards reproduces the unsharded output byte for byte (the GPU tier repeats this with the HIP
operators: tests/test_gpu_multidevice.py)."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_batch_partitions():
    from qnnpack_amd.shard import shard_batch
    for total in [0, 1, 7, 128, 1024, 1025]:
        for world in [1, 2, 3, 4, 8]:
            slices = [shard_batch(total, world, r) for r in range(world)]
            assert slices[0][0] == 0
            for (s0, c0), (s1, _) in zip(slices, slices[1:]):
                assert s0 + c0 == s1
            assert slices[-1][0] + slices[-1][1] == total
            counts = [c for _, c in slices]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        shard_batch(8, 2, 2)


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from qnnpack_amd.shard import job_time_ms, shard_batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = shard_batch(total, world, rank)
    dist.barrier()
    local_ms = 10.0 + 5.0 * rank          # rank 1 is the slow one
    job_ms = job_time_ms(local_ms, world)
    q.put((rank, start, count, job_ms))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_job_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, total = 2, 129
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, c0, t0), (r1, s1, c1, t1) = results
    assert (s0, c0, s1, c1) == (0, 65, 65, 64)
    assert t0 == t1 == 15.0                    # max over ranks, seen by every rank


def _shard_conv_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import numpy as np
    import torch
    import torch.distributed as dist
    from _cases import ConvCase, conv_tensors
    from _runner import conv_expected
    from dataclasses import replace
    from qnnpack_amd.shard import shard_batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = ConvCase("gloo_shard_3x3", (9, 8), (3, 3), (1, 1, 1, 1), gic=8, goc=12, batch=5)
    inp, kernel, bias = conv_tensors(case)
    whole, quant, (oh, ow) = conv_expected(case, inp, kernel, bias)
    start, count = shard_batch(case.batch, world, rank)
    in_img = case.input_size[0] * case.input_size[1] * case.gic
    out_img = oh * ow * case.goc
    # the rank's operator: same weights, same quantization, its own images only
    from oracle import o1
    shape = o1.conv_shape(count, case.input_size[0], case.input_size[1], case.padding, case.kernel_size,
                          case.subsampling, case.dilation, case.groups, case.gic, case.goc, case.in_stride)
    acc = o1.conv2d_acc(shape, inp[start * in_img:(start + count) * in_img], kernel, bias, case.izp, case.kzp)
    oscale, ozp = quant
    mine = np.zeros(count * out_img, np.uint8)
    o1.requantize_rows(acc.reshape(count * oh * ow, case.goc), np.float32(1.0) / oscale, ozp, case.qmin, case.qmax,
                       mine, case.goc)
    # gather for the CHECK only (the data path itself has no collective): pad to the largest shard
    longest = (case.batch + world - 1) // world * out_img
    padded = torch.zeros(longest, dtype=torch.uint8)
    padded[:mine.size] = torch.from_numpy(mine)
    parts = [torch.zeros(longest, dtype=torch.uint8) for _ in range(world)]
    dist.all_gather(parts, padded)
    joined = np.concatenate([parts[r].numpy()[:shard_batch(case.batch, world, r)[1] * out_img] for r in range(world)])
    q.put((rank, bool(np.array_equal(joined, whole)), int(joined.size)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_convolution_matches_unsharded():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world = 2
    procs = [ctx.Process(target=_shard_conv_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results), results
