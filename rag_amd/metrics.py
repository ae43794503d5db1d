"""Loss and evaluation metrics of the stereo loops, fused on the device.

The reference computes, per batch, a masked smooth-L1 loss and five metrics with six boolean gathers and six
`.item()` synchronisations (approaches/rag.py:418-430, utilstool/metrics.py:21-65).  Here one kernel pass accumulates
everything per image, a second tiny kernel applies the reference's per-image rules, and the eight results stay in
one device tensor: a caller that wants Python floats pays ONE device-to-host copy (`StereoMetrics.floats()`), and the
training loss never leaves the device.
"""
from __future__ import annotations

from typing import Dict

import torch

from . import ops
from ._lib import check, load_library

NAMES = ("loss", "EPE", "D1", "Thres1", "Thres2", "Thres3")


def _raw(disp_est: torch.Tensor, disp_gt: torch.Tensor, maxdisp: float) -> torch.Tensor:
    ops._need_gpu(disp_est, disp_gt)
    if disp_est.shape != disp_gt.shape or disp_est.dim() != 3:
        raise ValueError("stereo metrics: disp_est and disp_gt must both be [B, H, W]")   # metrics.py:14-18
    est, gt = disp_est.contiguous(), disp_gt.contiguous()
    B, H, W = est.shape
    buf = torch.empty((B + 1, 8), device=est.device, dtype=torch.float32)
    check(load_library().ragmi_stereo_metrics_fwd(est.data_ptr(), gt.data_ptr(), B, H, W, float(maxdisp), buf.data_ptr(),
                                                  buf[B].data_ptr(), ops._stream()), "stereo_metrics")
    return buf[B]


class StereoMetrics:
    """Result of `stereo_metrics`: `.tensor` is the 8-float device vector (loss, EPE, D1, Thres1, Thres2, Thres3,
    masked pixel count, images kept); `[name]` gives a 0-d device tensor; `floats()` copies once to the host."""

    def __init__(self, tensor: torch.Tensor):
        self.tensor = tensor

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.tensor[NAMES.index(name)]

    def floats(self) -> Dict[str, float]:
        vals = self.tensor.tolist()        # the only synchronisation
        return dict(zip(NAMES, vals[:6]))


def stereo_metrics(disp_est: torch.Tensor, disp_gt: torch.Tensor, maxdisp: float = 192) -> StereoMetrics:
    """loss, EPE, D1, Thres1/2/3 of Appr.eval (rag.py:418-430) for disp_est, disp_gt [B,H,W]; no gradient."""
    with torch.no_grad():
        return StereoMetrics(_raw(disp_est.detach(), disp_gt, maxdisp))


class MaskedSmoothL1Fn(torch.autograd.Function):
    """F.smooth_l1_loss(disp_est[mask], disp_gt[mask]) with mask = 0 < gt < maxdisp (rag.py:210-211), forward and
    backward in one kernel each, without the boolean gather (no stream synchronisation)."""

    @staticmethod
    def forward(ctx, disp_est, disp_gt, maxdisp):
        est = disp_est.contiguous()
        out = _raw(est, disp_gt, maxdisp)
        ctx.save_for_backward(est, disp_gt.contiguous(), out)
        ctx.maxdisp = float(maxdisp)
        return out[0] * 1.0          # a kernel, not a memcpy (see ragmi_stereo_metrics_fwd on hipGraph memcpy nodes)

    @staticmethod
    def backward(ctx, gout):
        est, gt, out = ctx.saved_tensors
        B, H, W = est.shape
        gout = gout.reshape(1).contiguous().float()
        d = torch.empty_like(est)
        check(load_library().ragmi_masked_smooth_l1_bwd(est.data_ptr(), gt.data_ptr(), out.data_ptr(), gout.data_ptr(), d.data_ptr(),
                                                        B, H, W, ctx.maxdisp, ops._stream()), "masked_smooth_l1_bwd")
        return d, None, None


def masked_smooth_l1(disp_est: torch.Tensor, disp_gt: torch.Tensor, maxdisp: float = 192) -> torch.Tensor:
    return MaskedSmoothL1Fn.apply(disp_est, disp_gt, maxdisp)
