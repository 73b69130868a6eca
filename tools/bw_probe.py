"""Measurement aid: raw HBM copy / fill rates through torch on this box (context for the HBM rooflines)."""
import torch, time
torch.cuda.set_device(0)
def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for mb in [25, 180, 600]:
    n = mb << 20
    srcs = [torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda") for _ in range(4)]
    dsts = [torch.empty(n, dtype=torch.uint8, device="cuda") for _ in range(4)]
    i = [0]
    def cp():
        dsts[i[0] % 4].copy_(srcs[i[0] % 4]); i[0] += 1
    def fill():
        dsts[i[0] % 4].fill_(7); i[0] += 1
    def rd():
        srcs[i[0] % 4].view(torch.int32).sum(); i[0] += 1
    t = timeit(cp); print(f"copy  {mb} MB: {t*1e3:.1f} us  {(2*n)/t/1e6:.0f} GB/s (read+write)")
    t = timeit(fill); print(f"fill  {mb} MB: {t*1e3:.1f} us  {n/t/1e6:.0f} GB/s (write)")
    t = timeit(rd); print(f"sum   {mb} MB: {t*1e3:.1f} us  {n/t/1e6:.0f} GB/s (read)")
