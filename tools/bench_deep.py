"""Time the dual 3x3x3 ConvBR launch of the level-6 / level-12 cells (headline shapes) under both precisions.
    python tools/bench_deep.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
for cs, cout, shape in [(16, 48, (1, 16, 32, 104)), (8, 24, (1, 32, 64, 208)), (16, 16, (1, 16, 32, 104)), (8, 8, (1, 32, 64, 208))]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((shape[0], 2 * cs) + shape[1:], generator=g).to(dev)
    wa = (torch.randn((cout, cs, 3, 3, 3), generator=g) * 0.1).to(dev)
    wb = (torch.randn((cout, cs, 3, 3, 3), generator=g) * 0.1).to(dev)
    pa, pb = ops.conv3d_k3_pack(wa), ops.conv3d_k3_pack(wb)
    y = torch.empty((shape[0], cout) + shape[1:], device=dev)
    for prec in ("fp32", "f16x3"):
        with ops.conv_precision(prec):
            used = ops.conv3d_k3_uses_x3(2 * cs, cout, *shape, nset=2)
            for _ in range(3):
                ops.conv3d_k3_dual(x, cs, pa, None, None, pb, None, None, cout, True, y)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                ops.conv3d_k3_dual(x, cs, pa, None, None, pb, None, None, cout, True, y)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        fl = 2.0 * shape[0] * shape[1] * shape[2] * shape[3] * 2 * cs * cout * 27
        print(f"dual {cs}+{cs} -> {cout} {shape} [{prec}] x3={used}: {us:.1f} us ({fl / us * 1e-6:.0f} TFLOP/s fp32-equivalent)", flush=True)
