"""Host side of the training step (BASELINE config 5; reference: approaches/rag.py:155-219, Appr.train_epoch).

One step = forward (HIP) -> masked smooth-L1 -> backward (HIP, rag_amd.autograd) -> gradient all-reduce ->
clip_grad_norm_ -> SGD.  Data-parallel replicas (one process per GPU) exchange gradients as ONE flat fp32
bucket: the `.grad` of every trainable parameter is a view into `GradBucket.flat`, so autograd accumulates
straight into the buffer RCCL reduces — no pack/unpack copies and a single collective per step (the message is
< 1 MB, i.e. latency-bound on xGMI: one call is what matters).  Train-mode BatchNorm keeps
per-replica statistics, like the single-GPU reference per replica.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.nn.functional as F


class GradBucket:
    """All trainable parameters' gradients as views of one flat buffer (`requires_grad` is read at construction:
    build it after `freeze_model` / `modify_param`, rag.py:101-102)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradBucket: no trainable parameters")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            p._ragmi_direct = True          # rag_amd.autograd accumulates into these views in place (no AccumulateGrad adds)
            off += p.numel()

    def zero(self) -> None:
        """optimizer.zero_grad() that keeps the views (rag.py:213)."""
        self.flat.zero_()
        for p, g in zip(self.params, self._views()):
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def _views(self):
        off = 0
        for p in self.params:
            yield self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def all_reduce_mean(self, dist=None) -> None:
        """Sum over replicas in one collective, then divide by the world size.  A one-rank group still issues the collective
        (the identity): a single-GPU run of a distributed job exercises the same RCCL path as the 8-GPU one."""
        if dist is not None and dist.is_initialized():
            dist.all_reduce(self.flat)
            if dist.get_world_size() > 1:
                self.flat.div_(dist.get_world_size())

    def clip_(self, max_norm: float) -> torch.Tensor:
        """torch.nn.utils.clip_grad_norm_ over the bucket (rag.py:215); returns the total norm."""
        total = torch.linalg.vector_norm(self.flat)
        self.flat.mul_(torch.clamp(max_norm / (total + 1e-6), max=1.0))
        return total


def masked_smooth_l1(disp: torch.Tensor, gt: torch.Tensor, maxdisp: int) -> torch.Tensor:
    """F.smooth_l1_loss(disp[mask], gt[mask]) with mask = 0 < gt < maxdisp (rag.py:210-211) without the boolean gather
    (no stream synchronisation): the fused HIP loss of rag_amd.metrics on the GPU."""
    if disp.is_cuda:
        from .metrics import masked_smooth_l1 as fused
        return fused(disp, gt, maxdisp)
    mask = ((gt < maxdisp) & (gt > 0)).to(disp.dtype)          # host-side twin (CPU tests of the step logic)
    per = F.smooth_l1_loss(disp, gt, reduction="none")
    return (per * mask).sum() / mask.sum()


TRAIN_PRECISION = "fp32"     # the reference's training step is fp32 (rag.py:204-216); "f16x3" is opt-in (~6 % of the step)


def forward_backward(net, bucket: GradBucket, left, right, gt, *, task_arch=None, features: bool = False,
                     precision: Optional[str] = None):
    """forward -> masked smooth-L1 -> zero the bucket -> backward (rag.py:208-214).  `features=True`: `net` is a
    MatchingNet and left/right are Feature-Net outputs.  `precision`: arithmetic of the 3x3x3 convolutions of the step (forward and
    data gradient): "fp32" (default, TRAIN_PRECISION: every contraction on the fp32-input MFMA forms, the reference's arithmetic
    class) or "f16x3" (opt-in; bound in include/rag_amd.h).  Returns the (detached) loss."""
    from . import ops
    with ops.conv_precision(precision or TRAIN_PRECISION):
        disp = net(left, right, task_arch) if features else net(left, right, 0, task_arch if task_arch is not None else net.arch_init)
        loss = masked_smooth_l1(disp, gt, net.maxdisp)
        bucket.zero()
        loss.backward()
    return loss.detach()


class FlatSGD:
    """torch.optim.SGD(lr, momentum, weight_decay) of rag.py:64-70 over a GradBucket, with the parameters themselves moved
    into one flat buffer next to the flat gradient: clip_grad_norm_ + step is then ONE pair of HIP launches
    (ragmi_sgd_clip_step) instead of torch's multi-tensor launches over ~500 small tensors.  GPU only (no CPU fallback: on
    the CPU use torch.optim.SGD via make_optimizer).  `param_groups[0]["lr"]` may be changed between steps like torch's."""

    def __init__(self, bucket: GradBucket, lr: float = 1e-3, momentum: float = 0.9, weight_decay: float = 3e-3):
        if not bucket.flat.is_cuda:
            raise RuntimeError("FlatSGD runs on the MI355X only; use make_optimizer (torch.optim.SGD) for CPU tensors")
        self.bucket = bucket
        self.param_groups = [dict(params=bucket.params, lr=lr, momentum=momentum, weight_decay=weight_decay)]
        self.flat = torch.empty_like(bucket.flat)
        off = 0
        for p in bucket.params:                       # parameters become views of one buffer (values kept)
            v = self.flat[off:off + p.numel()].view_as(p)
            v.copy_(p.detach())
            p.data = v
            off += p.numel()
        self.momentum_buffer = torch.zeros_like(self.flat)
        self.steps = 0
        from . import ops
        self._ops = ops
        self._ws = torch.empty((ops.load_library().ragmi_sgd_workspace_bytes() // 4,), device=self.flat.device, dtype=torch.float32)
        self.total_norm = torch.zeros((1,), device=self.flat.device, dtype=torch.float32)

    def zero_grad(self, set_to_none: bool = False) -> None:
        self.bucket.zero()

    def step(self, clip: float = 0.0) -> torch.Tensor:
        """clip_grad_norm_(clip) (clip <= 0: none) + SGD step; returns the pre-clip total gradient norm (device tensor)."""
        g = self.param_groups[0]
        self._ops.sgd_clip_step(self.flat, self.bucket.flat, self.momentum_buffer, g["lr"], g["momentum"], g["weight_decay"], clip,
                                self.steps == 0 or g["momentum"] == 0.0, self._ws, self.total_norm)
        self.steps += 1
        torch.autograd.graph.increment_version(self.bucket.params)       # packed-weight caches key on ._version
        return self.total_norm

    def state_dict(self):
        return {"momentum_buffer": self.momentum_buffer.clone(), "steps": self.steps,
                "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}

    def load_state_dict(self, sd) -> None:
        self.momentum_buffer.copy_(sd["momentum_buffer"])
        self.steps = int(sd["steps"])
        self.param_groups[0].update(sd["param_groups"][0])


def exchange_and_update(optimizer, bucket: GradBucket, *, clip: float = 5.0, dist=None) -> None:
    """gradient all-reduce (mean over replicas) -> clip_grad_norm_ -> optimizer step (rag.py:215-216)."""
    bucket.all_reduce_mean(dist)
    if isinstance(optimizer, FlatSGD):
        optimizer.step(clip)
        return
    bucket.clip_(clip)
    optimizer.step()


def train_step(net, optimizer, bucket: GradBucket, left, right, gt, *, task_arch=None, clip: float = 5.0, dist=None,
               features: bool = False, precision: Optional[str] = None):
    """One optimisation step as in Appr.train_epoch (rag.py:204-216); `precision` as in forward_backward (default fp32).
    Returns the (detached) loss."""
    loss = forward_backward(net, bucket, left, right, gt, task_arch=task_arch, features=features, precision=precision)
    exchange_and_update(optimizer, bucket, clip=clip, dist=dist)
    return loss


def graph_census(graph: "torch.cuda.CUDAGraph") -> dict:
    """Node counts {kernel, memcpy, memset, other} of a captured hipGraph (ragmi_graph_node_census: hipGraphGetNodes /
    hipGraphNodeGetType).  `graph` must have been created with keep_graph=True and not yet released.  A graph that holds memcpy
    or memset nodes is not replay-safe on this runtime when null-stream copies run between replays."""
    import ctypes
    from ._lib import check, load_library
    n = [ctypes.c_int32() for _ in range(4)]
    check(load_library().ragmi_graph_node_census(graph.raw_cuda_graph(), *[ctypes.byref(v) for v in n]), "graph_node_census")
    return dict(zip(("kernel", "memcpy", "memset", "other"), (v.value for v in n)))


class GraphedTrainStep:
    """The same step with forward + loss + backward replayed as ONE captured hipGraph (the ~2500 kernel launches of a
    step cost more host time than GPU time when issued one by one); the gradient exchange, clipping and the optimizer
    step stay eager (a handful of launches, and the collective stays outside the graph).  Inputs are copied into static
    buffers; shapes, the architecture and which parameters train must not change after capture.

    Nothing on the captured path may be a memset or memcpy NODE (hipMemsetAsync, a contiguous same-dtype copy_/clone):
    on this runtime (ROCm 7.2) a memcpy issued on the null stream between two replays (a `.item()`, a `.clone()`) left the next
    replay computing garbage whenever the instantiated graph held such a node — measured: 6-7 of 8 runs with one 160-byte memset
    and one 4-byte clone in the graph, 0 of 80 once both were kernels, with the same null-stream copies between replays
    (DESIGN.md 4.4 lists what was ruled out; tests/test_hip_train.py drives exactly that pattern).  The cause inside the runtime
    is not established, so the invariant is ENFORCED rather than assumed: the captured graph's nodes are counted at capture time
    (ragmi_graph_node_census) and a capture holding any memcpy / memset node is refused — an ATen op that starts lowering to
    copy_ or memset after a torch upgrade fails loudly here instead of corrupting a replay.  The graph replays on the caller's
    current stream; no private stream is involved."""

    def __init__(self, net, optimizer, bucket: GradBucket, left, right, gt, *, task_arch=None, clip: float = 5.0, dist=None,
                 features: bool = False, warmup: int = 2, precision: Optional[str] = None):
        self.net, self.opt, self.bucket, self.clip, self.dist = net, optimizer, bucket, clip, dist
        self.left, self.right, self.gt = left.clone(), right.clone(), gt.clone()
        self.precision = precision or TRAIN_PRECISION
        kw = dict(task_arch=task_arch, features=features, precision=self.precision)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # warm-up on a side stream: lazy state (allocator pools, occupancy
            for _ in range(max(warmup, 1)):            # queries, momentum buffers) exists before capture
                forward_backward(net, bucket, self.left, self.right, self.gt, **kw)
                exchange_and_update(optimizer, bucket, clip=clip, dist=dist)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph(keep_graph=True)     # keep the hipGraph_t: its nodes are inspected below
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # RCCL's watchdog thread must not trip it
            self.loss = forward_backward(net, bucket, self.left, self.right, self.gt, **kw)
        self.node_census = self._census()
        if self.node_census["memcpy"] or self.node_census["memset"]:
            raise RuntimeError(
                f"GraphedTrainStep: the captured step holds {self.node_census['memcpy']} memcpy and {self.node_census['memset']} "
                "memset node(s); such nodes are not replay-safe on this runtime (DESIGN.md 4.4).  tools/find_memcpy_ops.py lists "
                "the ATen calls behind them; run the step eagerly (rag_amd.train.train_step) until they are kernels.")
        self.graph.instantiate()
        # train-mode BatchNorm buffers change at every replay through raw pointers: their version counters are bumped by hand
        # (the eval-mode caches of rag_amd.modules key on them)
        self._bn_buffers = [t for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm) and m.training
                            and m.track_running_stats for t in (m.running_mean, m.running_var, m.num_batches_tracked) if t is not None]

    def _census(self) -> dict:
        return graph_census(self.graph)

    def __call__(self, left=None, right=None, gt=None):
        for dst, src in ((self.left, left), (self.right, right), (self.gt, gt)):
            if src is not None and src.data_ptr() != dst.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        if self._bn_buffers:
            torch.autograd.graph.increment_version(self._bn_buffers)
        exchange_and_update(self.opt, self.bucket, clip=self.clip, dist=self.dist)
        return self.loss


def make_optimizer(params, lr: float = 1e-3, momentum: float = 0.9, weight_decay: float = 3e-3, bucket: Optional[GradBucket] = None):
    """rag.py:64-70: SGD over the parameters that require grad.  With `bucket` (on the GPU): the fused flat-buffer FlatSGD."""
    if bucket is not None and bucket.flat.is_cuda:
        return FlatSGD(bucket, lr=lr, momentum=momentum, weight_decay=weight_decay)
    return torch.optim.SGD([p for p in params if p.requires_grad], lr=lr, momentum=momentum, weight_decay=weight_decay)
