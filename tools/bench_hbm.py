"""Streaming ceilings of this GPU with library kernels (ATen fill / copy / read-reduce) at the size of a level-3 tensor (164 MB)
and at 1 GB: what an HBM-bound kernel of this path can hope for."""
import torch

dev = "cuda:0"


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for numel in (12 * 64 * 128 * 416, 256 * 1024 * 1024):
    a = torch.empty(numel, device=dev)
    b = torch.empty(numel, device=dev)
    mb = numel * 4 / 1e6
    us = t(lambda: a.fill_(1.0))
    print(f"{mb:7.0f} MB fill : {us:7.1f} us  {mb / us:6.2f} TB/s written")
    us = t(lambda: b.copy_(a))
    print(f"{mb:7.0f} MB copy : {us:7.1f} us  {2 * mb / us:6.2f} TB/s read+written")
    us = t(lambda: a.sum())
    print(f"{mb:7.0f} MB sum  : {us:7.1f} us  {mb / us:6.2f} TB/s read")
