"""Time the head's last steps at the headline shape: y6 [1,12,32,64,208] -> upsample x2 -> last_3_3d (12 -> 1) -> mat [1,1,64,128,416].
   (a) the fused kernel ragmi_upconv3d_c1_fwd; (b) standalone upsample + convolution (conv3d_c1 / generic small-Cout kernel).
    python tools/bench_head.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd as ra  # noqa: E402

dev = "cuda:0"
y6 = torch.randn((1, 12, 32, 64, 208), device=dev)
shape = (1, 12, 64, 128, 416)
ups = [torch.empty(shape, device=dev) for _ in range(2)]       # 2 x 164 MB in rotation: past the 256 MiB memory-side cache
w = torch.randn((1, 12, 3, 3, 3), device=dev) * 0.05
out = torch.empty((1, 1) + shape[2:], device=dev)


def timed(fn, n=40):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def two(i):
    ra.ops.trilinear3d_act(y6, shape[2:], True, False, ups[i % 2], 0)
    ra.ops.conv3d_k3_small(ups[i % 2], w, None, None, False, out)


t_conv = timed(lambda i: ra.ops.conv3d_k3_small(ups[i % 2], w, None, None, False, out))
t_two = timed(two)
t_fused = timed(lambda i: ra.ops.upconv3d_c1(y6, w, None, None, False, out))
print(f"last_3_3d alone {t_conv:.1f} us; upsample + last_3_3d {t_two:.1f} us; fused upconv3d_c1 {t_fused:.1f} us "
      f"({2 * 27 * 12 * out.numel() / t_fused / 1e6:.1f} TFLOP/s)")
