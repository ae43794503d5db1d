"""Time ragmi_disp_softargmin at the headline shape (cost [B,1,64,128,416] -> disparity [B,384,1248], maxdisp 192).
    python tools/bench_disp.py [B]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd as ra  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x = torch.randn((B, 1, 64, 128, 416), device="cuda") * 1000.0
for _ in range(5):
    out = ra.ops.disp_softargmin(x, 192)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    out = ra.ops.disp_softargmin(x, 192)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
byt = x.numel() * 4 + out.numel() * 4
print(f"disp_softargmin B={B}: {us:.1f} us/launch; algorithmic {byt / 1e6:.1f} MB -> {byt / us / 1e6:.3f} TB/s")
