"""Run-to-run determinism of the inference forward at the headline shape: N forwards per configuration, every output compared bit for bit
with the first (round 5 met a kernel whose results changed from run to run — NOTES — so this is checked explicitly).
   python tools/determinism.py [N]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rag_amd as ra  # noqa: E402
from oracle import matching_oracle as O  # noqa: E402

DEV = "cuda:0"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rows = O.ALL_CONV
sd = O.random_matching_state_dict(rows, seed=0)
net = ra.MatchingNet(ra.ALL_CONV_GENOTYPE, maxdisp=192)
net.load_state_dict(sd)
net = net.to(DEV).eval()
g = torch.Generator().manual_seed(1234)
bad = 0
for name, B, dt, prec in (("fp32 f16x3 B=1", 1, torch.float32, "f16x3"), ("fp32 strict B=1", 1, torch.float32, "fp32"), ("bf16 B=2", 2, torch.bfloat16, "f16x3")):
    lf, rf = (torch.randn((B, 12, 128, 416), generator=g).to(DEV).to(dt) for _ in range(2))
    with torch.no_grad(), ra.ops.conv_precision(prec):
        first = net(lf, rf)
        diff = 0
        for _ in range(N - 1):
            out = net(lf, rf)
            diff += int((out != first).sum())
    print(f"{name}: {N} forwards, {diff} differing output values", flush=True)
    bad += diff != 0
net2 = ra.Network(ra.ALL_CONV_GENOTYPE, DEV, maxdisp=192).to(DEV).eval()
left, right = (torch.randn((1, 3, 384, 1248), generator=g).to(DEV) for _ in range(2))
with torch.no_grad():
    first = net2(left, right, 0, net2.arch_init)
    diff = sum(int((net2(left, right, 0, net2.arch_init) != first).sum()) for _ in range(N // 2))
print(f"end to end (images -> disparity): {N // 2 + 1} forwards, {diff} differing output values")
sys.exit(1 if (bad or diff) else 0)
