"""A few level-3 dual-cell launches (4 + 4 -> 12 channels at 64x128x416, f16x3) for counter passes:
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d out -- python3 tools/x3_dual_one.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
g = torch.Generator().manual_seed(1)
x = torch.randn((1, 8, 64, 128, 416), generator=g).to(dev)
wa = (torch.randn((12, 4, 3, 3, 3), generator=g) * 0.1).to(dev)
wb = (torch.randn((12, 4, 3, 3, 3), generator=g) * 0.1).to(dev)
pa, pb = ops.conv3d_k3_pack(wa), ops.conv3d_k3_pack(wb)
y = torch.empty((1, 12, 64, 128, 416), device=dev)
assert ops.conv3d_k3_uses_x3(8, 12, 1, 64, 128, 416, nset=2)
for _ in range(5):
    ops.conv3d_k3_dual(x, 4, pa, None, None, pb, None, None, 12, True, y)
torch.cuda.synchronize()
