"""Time the head's last_3_3d (3x3x3, 12 -> 1 channel, no BN / ReLU) alone at the headline shape [1,12,64,128,416], with input
buffers in rotation (2 x 164 MB: past the 256 MiB memory-side cache).
    python tools/bench_head.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd as ra  # noqa: E402

dev = "cuda:0"
shape = (1, 12, 64, 128, 416)
xs = [torch.randn(shape, device=dev) for _ in range(2)]
w = torch.randn((1, 12, 3, 3, 3), device=dev) * 0.05
out = torch.empty((1, 1) + shape[2:], device=dev)


def run(i):
    ra.ops.conv3d_k3_small(xs[i % len(xs)], w, None, None, False, out)


for i in range(4):
    run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40):
    run(i)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 40
mb = (xs[0].numel() + out.numel()) * 4 / 1e6
print(f"last_3_3d 12->1 at {shape}: {us:.1f} us ({mb / us:.2f} TB/s of {mb:.0f} MB algorithmic; {2 * 27 * 12 * out.numel() / us / 1e6:.1f} TFLOP/s)")
