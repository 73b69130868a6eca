"""Measurement aid (not part of the product, not used by bench.py): the vendor int8 GEMM (torch._int_mm ->
hipBLASLt/rocBLAS, int8 x int8 -> int32, no requantization) on the same 4096^3 problem with random operands,
as a yardstick for what a tuned library sustains on this board under its power cap."""
import torch, time
M = N = K = 4096
for label, gen in (("random", lambda s: torch.randint(-128, 128, s, dtype=torch.int8, device="cuda")),
                   ("zeros", lambda s: torch.zeros(s, dtype=torch.int8, device="cuda"))):
    a = gen((M, K)); b = gen((K, N))
    for layout, bb in (("b row-major", b), ("b = (N,K).t()", gen((N, K)).t())):
        try:
            for _ in range(5): torch._int_mm(a, bb)
            torch.cuda.synchronize()
            for iters in (1, 30, 300):
                e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(iters): torch._int_mm(a, bb)
                e1.record(); torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / iters
                print(f"{label:7s} {layout:14s} iters={iters:4d}  {us:8.2f} us/launch  {2.0 * M * N * K / us / 1e6:8.1f} TOPS")
        except Exception as exc:  # noqa: BLE001
            print(label, layout, "failed:", exc)
