"""GPU-side time of the Feature Net launches (rocprofv3 kernel trace of a few end-to-end forwards; use under tools/ab style runs):
    rocprofv3 --kernel-trace --stats --output-format csv -d out -- python3 tools/bench_cell2d.py"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = rag_amd.Network(rag_amd.ALL_CONV_GENOTYPE, dev, maxdisp=192).to(dev).eval()
g = torch.Generator().manual_seed(1)
x = torch.randn((2, 3, 384, 1248), generator=g).to(dev)
with torch.no_grad():
    for _ in range(20):
        net.feature(x, net.arch_init, None)
torch.cuda.synchronize()
