"""Single 3x3x3 convolutions on the SMALL (level-6 / level-12) volumes under the default precision (the deep-level f16x3 form
of DESIGN.md 4.7 where the channel counts are 8 / 16 per set, the fp32-MFMA kernel otherwise); tools/bench_deep.py times the dual
launches of the headline forward under both precisions."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
for cin, cout, shape in [(16, 48, (1, 16, 32, 104)), (8, 24, (1, 32, 64, 208)), (16, 16, (1, 16, 32, 104)), (24, 24, (1, 16, 32, 104))]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((shape[0], cin) + shape[1:], generator=g).to(dev)
    w = (torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.1).to(dev)
    pk = ops.conv3d_k3_pack(w)
    y = torch.empty((shape[0], cout) + shape[1:], device=dev)
    used = ops.conv3d_k3_uses_x3(cin, cout, *((shape[0],) + shape[1:]))
    for _ in range(3):
        ops.conv3d_k3(x, pk, cout, None, None, False, y)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            for _ in range(20):
                ops.conv3d_k3(x, pk, cout, None, None, False, y)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = 2.0 * shape[0] * shape[1] * shape[2] * shape[3] * cin * cout * 27
    print(f"Cin={cin} Cout={cout} {shape}: x3={used} {us:.1f} us ({fl / us * 1e-6:.0f} TFLOP/s fp32-equivalent)", flush=True)
