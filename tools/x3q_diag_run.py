"""One level-3 dual-cell launch shape (4 + 4 -> 12, two tails) on the quad-ring kernel for tools/x3q_diag.sh: plane input or G4 input + G4 tails
(argv[1] = g4 | planes), no main store when argv[2] = nomain."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
g4 = len(sys.argv) > 1 and sys.argv[1] == "g4"
nomain = len(sys.argv) > 2 and sys.argv[2] == "nomain"
g = torch.Generator().manual_seed(1)
shape = (1, 64, 128, 416)
r = lambda *s: (torch.randn(s, generator=g) * 0.1).to(dev)  # noqa: E731
tails_out = torch.empty((1, 8) + shape[1:], device=dev)
with ops.conv_precision("f16x3"):
    x8 = torch.randn((1, 8) + shape[1:], generator=g).to(dev)
    pa, pb = ops.conv3d_k3_pack(r(12, 4, 3, 3, 3)), ops.conv3d_k3_pack(r(12, 4, 3, 3, 3))
    y = torch.empty((1, 12) + shape[1:], device=dev)
    tl = [ops.Tail(r(4, 12), r(4).abs() + 0.5, r(4), True, tails_out, 4 * k, g4=g4) for k in range(2)]
    if len(sys.argv) > 3 and sys.argv[3] == "down":      # cell 1's launch: one full-resolution tail + the two down-sampling tails of cell 3
        half = torch.empty((1, 8, 32, 64, 208), device=dev)
        tl = tl[:1] + [ops.Tail(r(4, 12), r(4).abs() + 0.5, r(4), True, half, 4 * k, down=True) for k in range(2)]
    sa, ha, sb, hb = r(12).abs() + 0.5, r(12), r(12).abs() + 0.5, r(12)
    for _ in range(300):
        ops.conv3d_k3_dual(x8, 4, pa, sa, ha, pb, sb, hb, 12, True, y, tails=tl, store_main=not nomain, x_g4=g4)
torch.cuda.synchronize()
