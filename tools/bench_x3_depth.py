"""Dual level-3 launch (4 + 4 -> 12 channels, 128x416) at several depths: time per plane against the work items per resident
workgroup (208 columns x ceil(D/8) segments over 512 slots) — how much of the launch is the last, partly filled round."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
g = torch.Generator().manual_seed(1)
wa = (torch.randn((12, 4, 3, 3, 3), generator=g) * 0.1).to(dev)
wb = (torch.randn((12, 4, 3, 3, 3), generator=g) * 0.1).to(dev)
pa, pb = ops.conv3d_k3_pack(wa), ops.conv3d_k3_pack(wb)
for D in (48, 56, 64, 72, 80, 96):
    x = torch.randn((1, 8, D, 128, 416), generator=g).to(dev)
    y = torch.empty((1, 12, D, 128, 416), device=dev)
    for _ in range(3):
        ops.conv3d_k3_dual(x, 4, pa, None, None, pb, None, None, 12, True, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv3d_k3_dual(x, 4, pa, None, None, pb, None, None, 12, True, y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    nseg = -(-D // 8)
    items = 208 * nseg
    print(f"D={D}: {us:.1f} us, {us / D:.3f} us/plane; {items} items over 512 slots = {items / 512:.2f} per slot")
