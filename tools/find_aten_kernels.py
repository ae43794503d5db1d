"""Which ATen calls of one eager training step launch generic ATen kernels (copies, fills, adds) and from where?  Prints parent-op
chains with counts — candidates for folding into the HIP kernels next to them."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    forward_backward(net, bucket, left, right, gt)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name != "hipLaunchKernel" and e.name != "hipExtModuleLaunchKernel":
        continue
    chain, q = [], e.cpu_parent
    while q is not None and len(chain) < 5:
        chain.append(q.name + (str([list(sh) for sh in q.input_shapes][:2]) if getattr(q, "input_shapes", None) else ""))
        q = q.cpu_parent
    if chain and chain[0].startswith("aten::"):
        cnt[" <- ".join(chain[:4])] += 1
for k, v in cnt.most_common(30):
    print(v, k[:260])
