"""Measurement aid: does the network gain from running sub-batches CONCURRENTLY? N graphs, one per sub-batch, each
captured and replayed on its own stream (hipGraph replays the chains of ONE graph on one queue, tools/network_chains.py),
against one graph over the whole batch. Wall time of `iters` rounds between two device synchronisations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, qnnpack_amd
from examples import mobilenetv2 as mnv2
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
counts = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
lib = qnnpack_amd.load(); lib.initialize()
plan = mnv2.build_plan()
image = torch.randint(0, 256, (batch * 224 * 224 * 3,), dtype=torch.uint8, device="cuda")
configs = {}
for n in counts:
    streams = [torch.cuda.Stream() for _ in range(n)]
    nets = []
    per = image.numel() // n
    for i, s in enumerate(streams):
        lib.set_stream(s.cuda_stream)
        net = mnv2.DeviceNetwork(lib, torch, plan, batch // n, fold_adds=True)
        net.buffers[0].copy_(image[i * per:(i + 1) * per])
        torch.cuda.synchronize()
        net.run(); torch.cuda.synchronize()
        net.capture()
        nets.append(net)
    configs[n] = (streams, nets)
lib.set_async(True)
for rnd in range(3):
    for n in counts:
        streams, nets = configs[n]
        for _ in range(3):
            for net in nets: lib.graph_launch(net.graph)
        torch.cuda.synchronize()
        iters = 50
        t0 = time.perf_counter()
        for _ in range(iters):
            for net in nets: lib.graph_launch(net.graph)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / iters * 1e6
        print(f"round {rnd} graphs {n}: {us:8.1f} us per batch of {batch}  {batch/us*1e6:9.0f} img/s")
lib.set_async(False)
last = plan.ops[-1].dst
ref = torch.cat([net.buffers[last] for net in configs[counts[0]][1]]).cpu()
for n in counts[1:]:
    out = torch.cat([net.buffers[last] for net in configs[n][1]]).cpu()
    print(f"graphs {n}: output {'identical' if torch.equal(out, ref) else 'DIFFERS'}")
