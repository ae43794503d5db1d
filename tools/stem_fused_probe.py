"""Timing probe: the fused stem launch (ops.costvol_stem_conv3d) at the headline shape with and without stem3d0's in-staging tail.
   rocprofv3 --kernel-trace --stats -- python3 tools/stem_fused_probe.py   (or plain: prints event timings)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rag_amd as ra  # noqa: E402

ops = ra.ops
dev = "cuda:0"
g1 = torch.Generator().manual_seed(5)
B, h, w, maxdisp = 1, 128, 416, 192
C = 12
L, R = torch.randn((B, C, h, w), generator=g1).to(dev), torch.randn((B, C, h, w), generator=g1).to(dev)
w0 = (torch.randn((12, 2 * C, 3, 3, 3), generator=g1) * 0.05).to(dev)
w1 = (torch.randn((12, 12, 3, 3, 3), generator=g1) * 0.1).to(dev)
tw0 = (torch.randn((4, 12), generator=g1) * 0.3).to(dev)
tw1 = [(torch.randn((4, 12), generator=g1) * 0.3).to(dev) for _ in range(2)]
d = maxdisp // 3
with ops.conv_precision("f16x3"):
    var = ops.costvol_stem_prepare(w0)
    pk = ops.conv3d_k3_pack(w1)
    pre0 = torch.empty((B, 8, d, h, w), device=dev)
    pre1 = torch.empty((B, 8, d, h, w), device=dev)
    for with_tail in (True, False, True, False):
        t0 = [ops.Tail(tw0, None, None, True, pre0, 0, g4=True)] if with_tail else None
        t1 = [ops.Tail(tw1[0], None, None, True, pre0, 4, g4=True), ops.Tail(tw1[1], None, None, False, pre1, 0, g4=True)]
        for _ in range(5):
            ops.costvol_stem_conv3d(L, R, maxdisp, var, 12, None, None, True, t0, pk, 12, None, None, True, None, None, tails=t1, store_main=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            ops.costvol_stem_conv3d(L, R, maxdisp, var, 12, None, None, True, t0, pk, 12, None, None, True, None, None, tails=t1, store_main=False)
        e1.record()
        torch.cuda.synchronize()
        print(f"stem3d0's tail in the staging: {with_tail}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (planes + stem3d1)")
