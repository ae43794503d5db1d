"""One level-3 dual-cell-shaped f16x3 launch set for counter passes (rocprofv3 --pmc ... -- python3 tools/x3_one.py)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
g = torch.Generator().manual_seed(1)
x = torch.randn((1, 12, 64, 128, 416), generator=g).to(dev)
w = (torch.randn((12, 12, 3, 3, 3), generator=g) * 0.1).to(dev)
pk = ops.conv3d_k3_pack(w)
y = torch.empty((1, 12, 64, 128, 416), device=dev)
for _ in range(5):
    ops.conv3d_k3(x, pk, 12, None, None, False, y)
torch.cuda.synchronize()
