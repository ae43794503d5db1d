"""Training step of the Matching-Net path on the HIP kernels (BASELINE config 5).

The reference trains by running PyTorch autograd through the very modules it infers with
(approaches/rag.py:155-219: `model.train()`, reused units `.eval()`, smooth-L1, `loss.backward()`).
Here every node of that graph is a `torch.autograd.Function` whose forward AND backward enqueue the
hand-written kernels of librag_amd.so; autograd only orders the calls.  No ATen convolution,
batch-norm, interpolate or softmax runs: if the library is missing these raise, there is no fallback.

    ConvBRGroupFn sibling ConvBRs reading one tensor, as one stacked convolution (forward, data and weight gradient)
    ConvBRFn      conv (3x3x3 MFMA kernel / 1x1x1) -> BatchNorm (batch or running statistics) -> ReLU
                  backward: ReLU+BN adjoint (reduce + apply), data gradient = the forward conv kernel on the
                  output gradient with the weight transposed and its taps flipped, weight gradient kernels
    StridedStemFn the Feature Net's stride-3 2-D stem, same structure
    TrilinearFn   F.interpolate(mode='trilinear') and its scatter adjoint
    CostVolFn     the concat-and-shift cost volume and its gather adjoint
    DispFn        fused upsample x3 -> softmin -> expectation and its adjoint
    AddFn         sum of two branch outputs

Per-channel BatchNorm bookkeeping (mean/var -> scale/shift, the three backward coefficients, running-stat
updates) is host-side arithmetic on [C]-sized vectors, like the eval-mode folding in modules._ConvBR.prepared.
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import ops


def _direct(p) -> "torch.Tensor | None":
    """A parameter whose .grad is a persistent view of a flat gradient bucket (rag_amd.train.GradBucket): backward kernels
    accumulate into it in place and autograd gets None for that input — no temporary and no AccumulateGrad add launch."""
    if getattr(p, "_ragmi_direct", False) and p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32:
        return p.grad
    return None


def _dense(t: torch.Tensor) -> torch.Tensor:
    """Channel planes dense (a channel-slice view of a contiguous buffer qualifies); copy otherwise."""
    inner = 1
    for i in range(t.dim() - 1, 0, -1):
        if t.shape[i] != 1 and t.stride(i) != inner:
            return t.contiguous()
        inner *= t.shape[i]
    return t


def _vol(t: torch.Tensor) -> int:
    n = 1
    for s in t.shape[2:]:
        n *= s
    return n


class ConvBRFn(torch.autograd.Function):
    """y = act(bn(conv(x))) for the stride-1 'same' 1x1x1 / 3x3x3 ConvBR (operations_3d.py:40-47) on 5-D volumes
    (the 2-D flavour runs on depth-1 volumes with its 3x3 weight embedded in the middle z-slice)."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mod):
        k = mod._geometry()
        x = _dense(x)
        B, cout = x.shape[0], weight.shape[0]
        w = weight.detach()
        raw = torch.empty((B, cout) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
        if k == 3:
            ops.conv3d_k3(x, ops.conv3d_k3_pack(w, for_current_precision=True), cout, None, None, False, raw)
        else:
            ops.conv3d_k1(x, w.reshape(cout, -1), None, None, False, raw)
        n = B * _vol(x)
        y, scale, shift, mean, invstd, training = _bn_forward_act(raw, n, gamma, beta, mod)
        ctx.mod, ctx.k, ctx.n, ctx.training = mod, k, n, training
        ctx.prec = ops.get_conv_precision()        # the data gradient runs under the forward's arithmetic contract
        ctx.save_for_backward(x, weight, raw, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, raw, scale, shift, mean, invstd = ctx.saved_tensors
        mod, k, n = ctx.mod, ctx.k, ctx.n
        need_x, need_w, need_g, need_b = ctx.needs_input_grad[:4]
        draw, dgamma, dbeta = _bn_backward(_dense(dy), raw, scale, shift, mean, invstd, mod, n, ctx.training, need_g, need_b)
        dx = dw = None
        w = weight.detach()
        cout, cin = w.shape[:2]
        if need_x:
            dx = torch.empty_like(x)
            if k == 3:
                with ops.conv_precision(ctx.prec):
                    ops.conv3d_k3(draw, ops.conv3d_k3_pack(w, transpose=True, for_current_precision=True), cin, None, None, False, dx)
            else:
                ops.conv3d_k1(draw, w.reshape(cout, cin), None, None, False, dx, transposed=True)   # W^T read in place
        if need_w:
            tw = _direct(mod.conv.weight)
            if k == 3:
                dw = ops.conv3d_k3_wgrad(x, draw, cout, into=[tw] if tw is not None else None, planar2d=mod.NDIM == 2)
            else:
                dw = ops.conv3d_k1_wgrad(x, draw, cout, into=tw)
                dw = None if tw is not None else dw.reshape(weight.shape)
        return dx, dw, dgamma, dbeta, None


def _bump_running_stats(bn) -> None:
    """The train-mode BatchNorm kernels update running_mean / running_var / num_batches_tracked through raw pointers, which
    autograd's version counters do not see.  The eval-mode caches (modules._ConvBR.stamp, _Cell._fused) key on those versions: bump
    them by hand, or an eval forward after a train forward WITHOUT a weight update (BN re-estimation, a train pass under no_grad,
    lr = 0) would reuse the stale folded scale / shift."""
    ts = [t for t in (bn.running_mean, bn.running_var, bn.num_batches_tracked) if t is not None]
    torch.autograd.graph.increment_version(ts)


def _bn_forward(raw, n, gamma, beta, mod):
    """BatchNorm bookkeeping of the ConvBR forwards: (scale, shift, mean, invstd, training).  Train mode: one call computes the
    batch statistics, folds them and updates the running statistics (ragmi_bn_train_stats_fwd)."""
    bn = mod.bn
    cout = raw.shape[1]
    if not mod.use_bn:
        return torch.ones(cout, device=raw.device), torch.zeros(cout, device=raw.device), None, None, False
    g, b = gamma.detach(), beta.detach()
    if bn.training:
        if bn.momentum is None:
            raise NotImplementedError("rag_amd: BatchNorm momentum=None (cumulative average) is not built; the reference uses 0.1")
        track = bn.track_running_stats and bn.running_mean is not None
        st = ops.bn_train_stats(raw, g, b, bn.running_mean if track else None, bn.running_var if track else None,
                                bn.num_batches_tracked if track else None, bn.momentum, bn.eps)
        if track:
            _bump_running_stats(bn)
        return st[2], st[3], st[0], st[1], True
    mean, var = bn.running_mean.detach(), bn.running_var.detach()
    invstd = torch.rsqrt(var + bn.eps)
    scale = (g * invstd).contiguous()
    shift = (b - mean * scale).contiguous()
    return scale, shift, mean, invstd, False


def _bn_forward_act(raw, n, gamma, beta, mod, res=None):
    """BatchNorm + ReLU of a ConvBR forward: (y, scale, shift, mean, invstd, training).  Train mode is two launches
    (ragmi_bn_train_act_fwd: statistics, then finalize + running-stat update + affine + ReLU in one kernel).  `res` is added after
    the activation in the same pass (a cell's sum of branches)."""
    bn = mod.bn
    if mod.use_bn and bn.training:
        if bn.momentum is None:
            raise NotImplementedError("rag_amd: BatchNorm momentum=None (cumulative average) is not built; the reference uses 0.1")
        track = bn.track_running_stats and bn.running_mean is not None
        y, st = ops.bn_train_act(raw, gamma.detach(), beta.detach(), bn.running_mean if track else None,
                                 bn.running_var if track else None, bn.num_batches_tracked if track else None, bn.momentum, bn.eps,
                                 mod.relu, res=res)
        if track:
            _bump_running_stats(bn)
        return y, st[2], st[3], st[0], st[1], True
    scale, shift, mean, invstd, training = _bn_forward(raw, n, gamma, beta, mod)
    if mod.use_bn or mod.relu or res is not None:
        y = ops.bn_act(raw, scale, shift, mod.relu, res=res)
    else:
        y = raw
    return y, scale, shift, mean, invstd, training


def _bn_backward(dy, raw, scale, shift, mean, invstd, mod, n, training, need_g, need_b, out=None):
    """ReLU + BatchNorm adjoint: (gradient w.r.t. the raw conv output, dgamma, dbeta); `out`: destination channel slice.
    dgamma / dbeta come back as None when they were accumulated straight into bucket-backed .grad tensors."""
    if not (mod.use_bn or mod.relu):
        if out is not None:
            torch.mul(dy, 1.0, out=out)       # a kernel: contiguous copy_ would be a memcpy node in a captured step
            return out, None, None
        return dy, None, None
    dgamma = dbeta = None
    if mod.use_bn and (training or need_g or need_b):
        tg, tb = (_direct(mod.bn.weight), _direct(mod.bn.bias)) if (need_g and need_b) else (None, None)
        direct = tg is not None and tb is not None
        dx, dg, db = ops.bn_act_bwd(dy, raw, scale, shift, mod.relu, mean, invstd, training, out=out,
                                    dgamma_into=tg if direct else None, dbeta_into=tb if direct else None)
        if not direct:
            dgamma, dbeta = (dg if need_g else None), (db if need_b else None)
        return dx, dgamma, dbeta
    else:
        c1 = scale
        c2 = c3 = torch.zeros_like(scale)
    return ops.bn_act_bwd_apply(dy, 0, raw, scale, shift, mod.relu, c1, c2, c3, out=out), dgamma, dbeta


class ConvBRGroupFn(torch.autograd.Function):
    """Sibling 3x3(x3) ConvBRs that read the SAME tensor (a cell state feeding several new states, rag_model.py:163-172) as
    one convolution with the weights stacked along Cout: one forward conv, one data-gradient conv (which also sums the
    siblings' contributions to dx) and one weight-gradient launch instead of one of each per sibling; BatchNorm stays per
    unit (each keeps its own mode, statistics and parameters).  apply(x, mods, w0, gamma0, beta0, w1, ..., res0, res1, ...) ->
    (y0, y1, ...): the optional trailing `res_i` (one per unit, or none at all) is added to unit i's output after its activation —
    the other branch of the state it feeds (rag_model.py:170-172), so a cell's sum of two branches costs no launch of its own."""

    @staticmethod
    def forward(ctx, x, mods, *params):
        x = _dense(x)
        n, C = len(mods), mods[0].conv.out_channels
        res = [_dense(t) for t in params[3 * n:]]
        if res and len(res) != n:
            raise ValueError("ConvBRGroupFn: residual inputs come one per unit or not at all")
        ctx.has_res = bool(res)
        B = x.shape[0]
        # concatenated as 2-D rows: ATen's cat of 5-D tensors falls back to one contiguous copy_ (a MEMCPY, i.e. a memcpy node of a
        # captured step — DESIGN.md 4.4) per input; up to 4-D it is one batched kernel
        w0 = params[0]
        wcat = torch.cat([params[3 * i].detach().reshape(C, -1) for i in range(n)]).view((n * C,) + tuple(w0.shape[1:]))
        raw = torch.empty((B, n * C) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
        ops.conv3d_k3(x, ops.conv3d_k3_pack(wcat, for_current_precision=True), n * C, None, None, False, raw)
        nvox = B * _vol(x)
        outs, saved = [], [x, wcat, raw]
        ctx.training = []
        for i, m in enumerate(mods):
            r = raw[:, i * C:(i + 1) * C]
            y, scale, shift, mean, invstd, training = _bn_forward_act(r, nvox, params[3 * i + 1], params[3 * i + 2], m,
                                                                      res=res[i] if res else None)
            outs.append(y)
            saved += [scale, shift, mean, invstd]
            ctx.training.append(training)
        ctx.mods, ctx.n = mods, nvox
        ctx.prec = ops.get_conv_precision()
        ctx.save_for_backward(*saved)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        x, wcat, raw = ctx.saved_tensors[:3]
        mods, n = ctx.mods, len(ctx.mods)
        C = mods[0].conv.out_channels
        draw = torch.empty_like(raw)
        grads = [None] * (3 * n)
        for i, m in enumerate(mods):
            scale, shift, mean, invstd = ctx.saved_tensors[3 + 4 * i: 7 + 4 * i]
            need_g, need_b = ctx.needs_input_grad[2 + 3 * i + 1], ctx.needs_input_grad[2 + 3 * i + 2]
            r = raw[:, i * C:(i + 1) * C]
            d, grads[3 * i + 1], grads[3 * i + 2] = _bn_backward(_dense(dys[i]), r, scale, shift, mean, invstd, m, ctx.n, ctx.training[i],
                                                                 need_g, need_b, out=draw[:, i * C:(i + 1) * C])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            with ops.conv_precision(ctx.prec):
                ops.conv3d_k3(draw, ops.conv3d_k3_pack(wcat, transpose=True, for_current_precision=True), x.shape[1], None, None, False, dx)
        need_w = [ctx.needs_input_grad[2 + 3 * i] for i in range(n)]
        if any(need_w):
            planar = mods[0].NDIM == 2
            targets = [_direct(m.conv.weight) for m in mods]
            if all(need_w) and all(t is not None for t in targets) and n <= 8:
                ops.conv3d_k3_wgrad(x, draw, n * C, into=targets, planar2d=planar)      # every unit's .grad in place
            else:
                dw = ops.conv3d_k3_wgrad(x, draw, n * C, planar2d=planar)
                for i in range(n):
                    if need_w[i]:
                        grads[3 * i] = dw[i * C:(i + 1) * C]
        # the residual enters after the activation: its gradient is the output gradient itself
        gres = [dys[i] if ctx.needs_input_grad[2 + 3 * n + i] else None for i in range(n)] if ctx.has_res else []
        return (dx, None, *grads, *gres)


class StridedStemFn(torch.autograd.Function):
    """The Feature Net's strided 2-D 3x3 ConvBR (rag_model.py:200, stride 3) on 4-D tensors."""

    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mod):
        x = x.contiguous()
        stride = mod.conv.stride[0]
        raw = ops.conv2d_k3_strided(x, weight.detach(), None, None, False, stride)
        n = raw.shape[0] * _vol(raw)
        y, scale, shift, mean, invstd, training = _bn_forward_act(raw, n, gamma, beta, mod)
        ctx.mod, ctx.n, ctx.training, ctx.stride = mod, n, training, stride
        ctx.save_for_backward(x, weight, raw, scale, shift, mean, invstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, raw, scale, shift, mean, invstd = ctx.saved_tensors
        need_x, need_w, need_g, need_b = ctx.needs_input_grad[:4]
        draw, dgamma, dbeta = _bn_backward(_dense(dy), raw, scale, shift, mean, invstd, ctx.mod, ctx.n, ctx.training, need_g, need_b)
        dx = ops.conv2d_k3_strided_dgrad(draw, weight, x.shape[2:], ctx.stride) if need_x else None
        dw = None
        if need_w:
            tw = _direct(ctx.mod.conv.weight)
            dw = ops.conv2d_k3_strided_wgrad(x, draw, ctx.stride, into=tw)
            dw = None if tw is not None else dw
        return dx, dw, dgamma, dbeta, None


class DispRegFn(torch.autograd.Function):
    """DisparityRegression.forward (rag_model.py:23-29)."""

    @staticmethod
    def forward(ctx, prob, maxdisp):
        ctx.maxdisp = maxdisp
        return ops.disparity_regression(prob, maxdisp)

    @staticmethod
    def backward(ctx, dout):
        return ops.disparity_regression_bwd(dout, ctx.maxdisp), None


class TrilinearFn(torch.autograd.Function):
    """F.interpolate(x, size, mode='trilinear', align_corners=...) (rag_model.py:146-153, 355-362)."""

    @staticmethod
    def forward(ctx, x, size, align_corners):
        ctx.in_size, ctx.align = tuple(x.shape[2:]), bool(align_corners)
        return ops.trilinear3d(x, size, align_corners)

    @staticmethod
    def backward(ctx, dy):
        return ops.trilinear3d_bwd(dy, ctx.in_size, ctx.align), None, None


class CostVolFn(torch.autograd.Function):
    """The cost-volume loop of rag_model.py:375-383 (backward: the 128 CopySlices nodes as one gather kernel)."""

    @staticmethod
    def forward(ctx, left_fea, right_fea, maxdisp):
        return ops.costvol(left_fea, right_fea, maxdisp)

    @staticmethod
    def backward(ctx, dcost):
        dl, dr = ops.costvol_bwd(dcost)
        return dl, dr, None


class DispFn(torch.autograd.Function):
    """Disp.forward (rag_model.py:39-44) fused; backward recomputes the softmin statistics."""

    @staticmethod
    def forward(ctx, cost, maxdisp):
        ctx.save_for_backward(cost)
        ctx.maxdisp = maxdisp
        return ops.disp_softargmin(cost, maxdisp)

    @staticmethod
    def backward(ctx, dout):
        (cost,) = ctx.saved_tensors
        return ops.disp_softargmin_bwd(cost, dout, ctx.maxdisp), None


class AddFn(torch.autograd.Function):
    """a + b (the `sum(new_states)` of Cell_3d, rag_model.py:172)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _dense(a), _dense(b)
        out = torch.empty(a.shape, device=a.device, dtype=a.dtype)
        return ops.add(a, 0, b, 0, out, 0, a.shape[1])

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def needs_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def resample(x: torch.Tensor, size: Sequence[int], align_corners: bool = True) -> torch.Tensor:
    size = tuple(int(v) for v in size)
    if tuple(x.shape[2:]) == size:
        return x
    return TrilinearFn.apply(x, size, align_corners)
