"""The two level-3 launch shapes of the headline forward on the z-marching split-operand kernel (for tools/x3_diag.sh):
stem3d1 (12 -> 12, two tails, no main store) and a dual cell (4 + 4 -> 12, two tails)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
g = torch.Generator().manual_seed(1)
shape = (1, 64, 128, 416)
r = lambda *s: (torch.randn(s, generator=g) * 0.1).to(dev)  # noqa: E731
tails_out = torch.empty((1, 8) + shape[1:], device=dev)
def tails():
    return [ops.Tail(r(4, 12), r(4).abs() + 0.5, r(4), True, tails_out, 4 * k) for k in range(2)]
with ops.conv_precision("f16x3"):
    x = torch.randn((1, 12) + shape[1:], generator=g).to(dev)
    pk = ops.conv3d_k3_pack(r(12, 12, 3, 3, 3))
    y = torch.empty((1, 12) + shape[1:], device=dev)
    for _ in range(5):
        ops.conv3d_k3(x, pk, 12, r(12).abs() + 0.5, r(12), True, y, None, tails=tails(), store_main=False)
    x8 = x[:, :8].contiguous()
    pa, pb = ops.conv3d_k3_pack(r(12, 4, 3, 3, 3)), ops.conv3d_k3_pack(r(12, 4, 3, 3, 3))
    for _ in range(5):
        ops.conv3d_k3_dual(x8, 4, pa, r(12).abs() + 0.5, r(12), pb, r(12).abs() + 0.5, r(12), 12, True, y, tails=tails())
torch.cuda.synchronize()
