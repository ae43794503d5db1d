"""Time ragmi_costvol_stem_fwd (planes + combine launches) at the headline shape: features [B,12,128,416], maxdisp 192.
    python tools/bench_stem.py [B]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd as ra  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda:0"
g = torch.Generator().manual_seed(0)
lf = torch.randn((B, 12, 128, 416), generator=g).to(dev)
rf = torch.randn((B, 12, 128, 416), generator=g).to(dev)
w = (torch.randn((12, 24, 3, 3, 3), generator=g) * 0.1).to(dev)
var = ra.ops.costvol_stem_prepare(w)
sc, sh = torch.rand(12, generator=g).to(dev) + 0.5, torch.randn(12, generator=g).to(dev)
out = torch.empty((B, 12, 64, 128, 416), device=dev)
for _ in range(3):
    ra.ops.costvol_stem(lf, rf, 192, var, 12, sc, sh, True, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
e0.record()
for _ in range(n):
    ra.ops.costvol_stem(lf, rf, 192, var, 12, sc, sh, True, out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
byt = out.numel() * 4 + 2 * lf.numel() * 4
print(f"costvol_stem B={B} (no tails): {us:.1f} us per call (planes + combine); output {out.numel() * 4 / 1e6:.0f} MB -> {byt / us / 1e6:.2f} TB/s algorithmic")
