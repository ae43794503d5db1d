"""Which ATen calls of one eager training step (or, with `infer`, one eval forward from images) end in a device-to-device MEMCPY (they would become memcpy nodes of a captured
step, which this runtime does not replay safely — DESIGN.md §4.4)?  Prints the rag_amd source lines behind them."""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, ".")
import rag_amd
from rag_amd.train import GradBucket, exchange_and_update, forward_backward, make_optimizer

dev = torch.device("cuda:0")
torch.manual_seed(0)
net = rag_amd.Network(rag_amd.ALL_CONV_GENOTYPE, dev, maxdisp=192).to(dev).train()
bucket = GradBucket(net.parameters())
opt = make_optimizer(net.parameters(), bucket=bucket)
g = torch.Generator().manual_seed(1)
left = torch.randn((2, 3, 96, 192), generator=g).to(dev)
right = torch.randn((2, 3, 96, 192), generator=g).to(dev)
gt = (torch.rand((2, 96, 192), generator=g) * 200).to(dev)
for _ in range(2):
    forward_backward(net, bucket, left, right, gt)
    exchange_and_update(opt, bucket, clip=5.0)
torch.cuda.synchronize()
infer = len(sys.argv) > 1 and sys.argv[1] == "infer"
if infer:
    net.eval()
    with torch.no_grad():
        net(left, right, 0, net.arch_init)          # builds the cached folded weights (not part of a captured pass)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    if infer:
        with torch.no_grad():
            net(left, right, 0, net.arch_init)
    else:
        forward_backward(net, bucket, left, right, gt)
    torch.cuda.synchronize()
ev = prof.events()
mem = [e for e in ev if "memcpy" in e.name.lower() or "memset" in e.name.lower() or "copyBuffer" in e.name]
print("memcpy/memset-like events:", collections.Counter(e.name for e in mem))
cnt = collections.Counter()
for e in ev:
    if e.name != "hipMemcpyAsync":
        continue
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 6:
        chain.append(q.name + (str([list(sh) for sh in q.input_shapes][:2]) if getattr(q, "input_shapes", None) else ""))
        q = q.cpu_parent
    cnt[" <- ".join(chain)] += 1
for k, v in cnt.most_common(25):
    print(v, k)
