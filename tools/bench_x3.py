"""f16x3 convolution (conv3d_x3.hip) vs the fp32-MFMA kernel through the same entry point (profiling aid).
Run twice:  python tools/bench_x3.py   and   RAGMI_X3=0 python tools/bench_x3.py  (the variable only picks the DEFAULT precision
of rag_amd.ops at import: f16x3 / fp32; ops.conv_precision(...) is the API)"""
import os
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
print("RAGMI_X3 =", os.environ.get("RAGMI_X3", "1 (default)"))
for cin, cout, shape in [(12, 12, (1, 16, 40, 70)), (4, 12, (2, 9, 33, 65)), (12, 12, (1, 64, 128, 416)), (4, 12, (1, 64, 128, 416)),
                         (24, 12, (1, 64, 128, 416)), (12, 12, (4, 64, 64, 128))]:
    g = torch.Generator().manual_seed(1)
    x = torch.randn((shape[0], cin) + shape[1:], generator=g)
    w = torch.randn((cout, cin, 3, 3, 3), generator=g) * 0.1
    xg, pk = x.to(dev), ops.conv3d_k3_pack(w.to(dev))
    y = torch.empty((shape[0], cout) + shape[1:], device=dev)
    used = ops.conv3d_k3_uses_x3(cin, cout, *((shape[0],) + shape[1:]))
    ops.conv3d_k3(xg, pk, cout, None, None, False, y)
    if x.numel() < 3e6:
        ref = F.conv3d(x.double(), w.double(), padding=1)
        err = float((y.cpu().double() - ref).abs().max() / ref.abs().max())
        print(f"Cin={cin} Cout={cout} {shape}: x3={used} max rel err {err:.2e}", flush=True)
        continue
    for _ in range(3):
        ops.conv3d_k3(xg, pk, cout, None, None, False, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3d_k3(xg, pk, cout, None, None, False, y)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    fl = 2.0 * x.shape[0] * shape[1] * shape[2] * shape[3] * cin * cout * 27
    print(f"Cin={cin} Cout={cout} {shape}: x3={used} {us:.1f} us ({fl / us * 1e-6:.0f} TFLOP/s fp32-equivalent)", flush=True)
