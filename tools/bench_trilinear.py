"""Time the head's x2 trilinear upsample alone: [1,12,32,64,208] -> [1,12,64,128,416], align_corners=True.
    python tools/bench_trilinear.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd as ra  # noqa: E402

dev = "cuda:0"
x = torch.randn((1, 12, 32, 64, 208), device=dev)
outs = [torch.empty((1, 12, 64, 128, 416), device=dev) for _ in range(8)]


def run(i):
    ra.ops.trilinear3d_act(x, (64, 128, 416), True, False, outs[i % len(outs)], 0)


for n_out in (1, 8):
    for i in range(3):
        run(i % n_out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(24):
        run(i % n_out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 24
    mb = (outs[0].numel() + x.numel()) * 4 / 1e6
    print(f"trilinear x2 upsample, {n_out} output buffer(s) in rotation: {us:.1f} us ({mb / us:.2f} TB/s)")
