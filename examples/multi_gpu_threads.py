#!/usr/bin/env python3
"""One process, one host thread per GPU: the second deployment form of DESIGN.md section 6 (the first is one process
per GPU, as bench.py --gpus N runs it).

    python examples/multi_gpu_threads.py [--devices D] [--batch B] [--rounds R]

A batch of a 3x3 convolution (56x56x64 -> 64, BASELINE configs[2]'s shape) is split contiguously over the devices with
qnnpack_amd.shard.shard_batch -- the reference's own batch axis (src/operator-run.c:675-679, 797-802, 837-842) -- and
every device gets a host thread that selects it (qnnp_gfx950_set_device), creates ITS operator through the reference C
API (weights are replicated: 37 KB), sets it up on its slice of the tensors and runs it. No collective: the only thing the
threads share is the host-side weights. Each shard is then compared byte for byte with the same images run on device 0
alone, and the aggregate images/s is printed. With one visible GPU it runs as a one-device job (same code path: the
thread still selects its device and binds its context)."""
from __future__ import annotations

import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H = W = 56
CIN = COUT = 64


def device_job(lib, torch, device, images, kernel, bias, rounds, results, errors):
    """create -> setup -> run x rounds -> delete on `device`, all from this thread"""
    try:
        lib.set_device(device)                      # this THREAD now drives `device` (bound on first use)
        torch.cuda.set_device(device)
        op = lib.create_convolution2d_nhwc_q8(1, 1, 1, 1, 3, 3, 1, 1, 1, 1, 1, CIN, COUT,
                                              127, 0.5, 127, 0.5, kernel, bias, 127, 64.0, 0, 255, 0)
        n = images.shape[0]
        d_in = torch.from_numpy(images.reshape(-1).copy()).to(f"cuda:{device}")
        d_out = torch.empty(n * H * W * COUT, dtype=torch.uint8, device=f"cuda:{device}")
        lib.setup_convolution2d_nhwc_q8(op, n, H, W, d_in, CIN, d_out, COUT)
        lib.run_operator(op)                        # synchronous: outputs complete on return
        t0 = time.perf_counter()
        for _ in range(rounds):
            lib.run_operator(op)
        dt = time.perf_counter() - t0
        results[device] = (d_out.cpu().numpy(), dt, lib.operator_kernel(op))
        lib.delete_operator(op)
    except Exception as exc:  # noqa: BLE001 -- reported by the main thread
        errors.append(f"device {device}: {type(exc).__name__}: {exc}")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", type=int, default=0, help="GPUs to use (0 = all visible)")
    ap.add_argument("--batch", type=int, default=32, help="images in the whole job")
    ap.add_argument("--rounds", type=int, default=20)
    args = ap.parse_args(argv)

    import torch
    import qnnpack_amd
    from qnnpack_amd.shard import shard_batch
    assert torch.cuda.is_available(), "needs at least one MI355X; there is no CPU fallback"
    lib = qnnpack_amd.load()
    lib.initialize()
    ndev = min(args.devices or lib.device_count(), lib.device_count())
    rng = np.random.default_rng(11)
    kernel = rng.integers(0, 256, size=(1, COUT, 3, 3, CIN), dtype=np.uint8)
    bias = rng.integers(-10000, 10001, size=COUT, dtype=np.int32)
    images = rng.integers(0, 256, size=(args.batch, H * W * CIN), dtype=np.uint8)

    # the whole batch on device 0 from the main thread: what the shards must reproduce
    whole, errors = {}, []
    device_job(lib, torch, 0, images, kernel, bias, 1, whole, errors)
    assert not errors, errors
    reference = whole[0][0].reshape(args.batch, -1)

    results = {}
    threads = []
    for d in range(ndev):
        start, count = shard_batch(args.batch, ndev, d)
        threads.append(threading.Thread(target=device_job,
                                        args=(lib, torch, d, images[start:start + count], kernel, bias, args.rounds, results, errors)))
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    wall = time.perf_counter() - t0
    assert not errors, errors
    for d in range(ndev):
        start, count = shard_batch(args.batch, ndev, d)
        got = results[d][0].reshape(count, -1)
        assert np.array_equal(got, reference[start:start + count]), f"shard of device {d} differs from the unsharded run"
    slowest = max(results[d][1] for d in range(ndev))
    per_device = {d: round(shard_batch(args.batch, ndev, d)[1] * args.rounds / results[d][1], 1) for d in range(ndev)}
    print(f"{ndev} device(s), batch {args.batch} in shards of {[shard_batch(args.batch, ndev, d)[1] for d in range(ndev)]}: "
          f"byte-identical to the unsharded run; {args.batch * args.rounds / slowest:.0f} images/s over the timed rounds "
          f"(kernel {results[0][2]}, slowest thread {slowest * 1e3:.1f} ms, wall {wall * 1e3:.0f} ms incl. create/setup)")
    print(f"per-device images/s (each thread's own shard over its own timed rounds): {per_device}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
