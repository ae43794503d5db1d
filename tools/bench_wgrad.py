"""Micro-benchmark of the 3x3x3 weight-gradient kernel (profiling aid; not part of the product path).
usage: python tools/bench_wgrad.py  [RAGMI_WGRAD_DIAG=bits]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
cases = [(4, 4, (4, 64, 64, 128)), (12, 12, (4, 64, 64, 128)), (24, 12, (4, 64, 64, 128)), (4, 12, (4, 64, 64, 128)),
         (8, 8, (4, 32, 32, 64)), (16, 16, (4, 16, 16, 32)), (12, 1, (4, 64, 64, 128))]
for cin, cout, (B, D, H, W) in cases:
    x = torch.randn((B, cin, D, H, W), device=dev)
    g = torch.randn((B, cout, D, H, W), device=dev)
    for _ in range(3):
        rag_amd.ops.conv3d_k3_wgrad(x, g, cout)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        rag_amd.ops.conv3d_k3_wgrad(x, g, cout)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    flops = 2.0 * B * D * H * W * cin * cout * 27
    nbytes = 4.0 * B * D * H * W * (cin + cout)
    print(f"Cin={cin:3d} Cout={cout:3d} {B}x{D}x{H}x{W}: {us:8.1f} us  {flops / us * 1e-6:6.1f} TFLOP/s  {nbytes / us * 1e-3:7.0f} GB/s (x+g once)", flush=True)
