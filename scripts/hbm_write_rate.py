"""What write rate does this GPU reach on a plain fill?  The reference point for the pillar forward launch, which writes the whole
BEV canvas (134 MB at B=4, 512^2, 64 bf16 channels) exactly once.  Prints torch's fill and copy rates at that size."""
import torch

dev = torch.device("cuda")
n = 4 * 512 * 512 * 64
x = torch.empty(n, dtype=torch.bfloat16, device=dev)
y = torch.empty(n, dtype=torch.bfloat16, device=dev)
for name, fn, bytes_ in (("fill 134 MB", lambda: x.zero_(), 2 * n), ("copy 134 MB (read + write)", lambda: y.copy_(x), 4 * n)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name}: {us:.1f} us per call = {bytes_ / us / 1e6:.2f} TB/s")
