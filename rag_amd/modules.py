"""Host-side mirror of the reference's Matching-Net modules, running on librag_amd.so.

Names, constructor signatures, attribute names and state_dict key layout follow
chzhang18/RAG (src/models/rag_model.py, src/automl/operations_3d.py,
src/automl/genotypes_{2d,3d}.py) so a reference checkpoint loads unchanged and the
approaches/automl growth loop can poke the same attributes.  The arithmetic is NOT
PyTorch: every forward below enqueues hand-written HIP kernels through ``rag_amd.ops``.

New seam (named by BASELINE.json north_star; the reference inlines it in Network.forward,
rag_model.py:375-386): ``MatchingNet.forward(left_fea, right_fea) -> disp[B, 3h, 3w]``.

Two execution modes, both on the HIP kernels, neither with a PyTorch fallback:
  * inference (``torch.no_grad()`` + eval-mode BatchNorm): the fused executor below (folded BN, concat-free cells,
    dual-input launches, cross-module tails);
  * training (autograd enabled and something requires grad, or a BatchNorm in train mode): the same modules compose
    the ``rag_amd.autograd`` Functions node by node like the reference's graph (approaches/rag.py:155-219).
"""
from __future__ import annotations

from collections import namedtuple
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

import torch
import torch.nn as nn

from . import autograd as ag
from . import ops

# src/automl/genotypes_2d.py:4-8 — the genotype handed to Network / Cell_3d; 3-D cells read `.reduce`
Genotype = namedtuple("Genotype_2D", "normal normal_concat reduce reduce_concat")
# src/automl/genotypes_3d.py:6-9
PRIMITIVES_3D = ["skip_connect_3d", "3d_conv_3x3"]

ALL_CONV_ROWS = np.array([[0, 1], [1, 1], [2, 1], [3, 1], [5, 1], [6, 1]])
ALL_SKIP_ROWS = np.array([[0, 0], [1, 0], [2, 0], [3, 0], [5, 0], [6, 0]])
ALL_CONV_GENOTYPE = Genotype(normal=ALL_CONV_ROWS, normal_concat=None, reduce=ALL_CONV_ROWS, reduce_concat=None)


def _volume(size) -> int:
    n = 1
    for v in size:
        n *= int(v)
    return n


ALL_SKIP_GENOTYPE = Genotype(normal=ALL_SKIP_ROWS, normal_concat=None, reduce=ALL_SKIP_ROWS, reduce_concat=None)


class Identity_3d(nn.Module):
    """src/automl/operations_3d.py:84-90."""

    def forward(self, x):
        return x


class _ConvBR(nn.Module):
    """Conv(bias=False) -> BatchNorm -> ReLU as ONE fused HIP kernel; shared by the 3-D (Matching Net) and 2-D (Feature
    Net, a depth-1 volume on the same kernels) flavours.  Sub-module names `conv` / `bn`, the always-constructed `bn`
    and the init follow the reference (operations_3d.py:31-55, operations_2d.py:31-55)."""

    NDIM = 3

    def __init__(self, C_in, C_out, kernel_size, stride, padding, bn=True, relu=True):
        super().__init__()
        self.relu = relu
        self.use_bn = bn
        if self.NDIM == 3:
            self.conv = nn.Conv3d(C_in, C_out, kernel_size, stride=stride, padding=padding, bias=False)
            self.bn = nn.BatchNorm3d(C_out)
        else:
            self.conv = nn.Conv2d(C_in, C_out, kernel_size, stride=stride, padding=padding, bias=False)
            self.bn = nn.BatchNorm2d(C_out)
        self._initialize_weights()
        self._cache = None

    def _initialize_weights(self):
        nn.init.kaiming_normal_(self.conv.weight, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(self.bn.weight, 1)
        nn.init.constant_(self.bn.bias, 0)

    # -- HIP-side parameters: packed weights + folded eval-mode BN, cached on tensor versions
    def _geometry(self) -> int:
        """kernel size 1 or 3 (stride 1, 'same' padding), or -3 for the strided 2-D 3x3 stem (Feature Net)."""
        k, n = self.conv.kernel_size, self.NDIM
        same = self.conv.padding == tuple((ki - 1) // 2 for ki in k) and self.conv.dilation == (1,) * n and self.conv.groups == 1
        if same and k in ((1,) * n, (3,) * n) and self.conv.stride == (1,) * n:
            return k[0]
        if same and n == 2 and k == (3, 3) and self.conv.stride[0] == self.conv.stride[1] > 1:
            return -3
        raise NotImplementedError("rag_amd ConvBR: only 1x1(x1)/pad0 and 3x3(x3)/pad1 at stride 1 (plus the strided 2-D 3x3 "
                                  "stem) are built (everything the reference instantiates)")

    def _small(self) -> bool:
        """Cout <= 2 (last_3_3d): VALU form of the 3x3x3 kernel instead of the 4-row MFMA."""
        return (self.NDIM == 3 and self.conv.out_channels <= 2 and self.conv.in_channels % 4 == 0
                and self.conv.in_channels <= 128)

    def stamp(self) -> tuple:
        w, bn = self.conv.weight, self.bn
        return (w.data_ptr(), w._version, bn.weight._version, bn.bias._version, bn.running_mean._version,
                bn.running_var._version, bn.weight.data_ptr(), str(w.device), self.use_bn)

    def prepared(self) -> Tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor]]:
        """(weights in kernel layout, scale, shift); scale/shift None when bn=False."""
        if self.use_bn and self.bn.training:
            raise RuntimeError("rag_amd ConvBR: folded BatchNorm parameters requested in train mode (internal error: "
                               "train-mode units run through rag_amd.autograd)")
        stamp = self.stamp()
        if self._cache is None or self._cache[0] != stamp:
            k = self._geometry()
            w = self.conv.weight.detach()
            with torch.no_grad():
                if k == -3 or (k == 3 and self._small()):
                    wk = w.contiguous()                      # strided 2-D stem / VALU form read the raw weight
                elif k == 3:
                    wk = ops.conv3d_k3_pack(w)               # a 2-D 3x3 packs as the dz = 1 plane of a 3x3x3 (depth-1 volumes)
                else:
                    wk = w.reshape(w.shape[0], w.shape[1]).contiguous()
                if self.use_bn:
                    bn = self.bn
                    # same folding ATen's eval batch_norm uses: alpha = gamma * rsqrt(var + eps); beta = b - mean * alpha
                    scale = (bn.weight.detach() * torch.rsqrt(bn.running_var.detach() + bn.eps)).float().contiguous()
                    shift = (bn.bias.detach() - bn.running_mean.detach() * scale).float().contiguous()
                else:
                    scale = shift = None
            self._cache = (stamp, wk, scale, shift)
        return self._cache[1], self._cache[2], self._cache[3]

    def autograd_mode(self, *inputs) -> bool:
        """Training composition needed: a gradient is wanted, or this unit's BatchNorm uses batch statistics."""
        return (ag.needs_grad(*inputs, self.conv.weight, self.bn.weight, self.bn.bias)
                or (self.use_bn and self.bn.training))

    def _forward_autograd(self, x: torch.Tensor, resample_to: Optional[Sequence[int]]) -> torch.Tensor:
        if x.dtype != torch.float32:
            raise NotImplementedError("rag_amd: the training path is fp32 (bf16 storage is inference only)")
        if self._geometry() == -3:
            return ag.StridedStemFn.apply(x if x.dim() == 4 else x.squeeze(2), self.conv.weight, self.bn.weight, self.bn.bias, self)
        squeeze = x.dim() == 4
        if squeeze:
            x = x.unsqueeze(2)
        if resample_to is not None:
            x = ag.resample(x, resample_to, True)
        y = ag.ConvBRFn.apply(x, self.conv.weight, self.bn.weight, self.bn.bias, self)
        return y.squeeze(2) if squeeze else y      # (a select's backward is zeros + copy_: a memcpy node when captured)

    def costvol_fusable(self, C_fea: int) -> bool:
        """This unit can consume (left_fea, right_fea) directly instead of the cost volume (ragmi_costvol_stem_fwd)."""
        return (self.NDIM == 3 and self._geometry() == 3 and self.conv.in_channels == 2 * C_fea and C_fea <= 16
                and self.conv.out_channels <= 16)

    def costvol_variants(self) -> torch.Tensor:
        """pre-summed weight variants of the fused cost-volume form, cached on the weight version."""
        w = self.conv.weight
        key = (w.data_ptr(), w._version, str(w.device))
        hit = getattr(self, "_cv_cache", None)
        if hit is None or hit[0] != key:
            with torch.no_grad():
                hit = (key, ops.costvol_stem_prepare(w.detach()))
            self._cv_cache = hit
        return hit[1]

    def forward_costvol(self, left_fea, right_fea, maxdisp, tails=None, out_g4: bool = False) -> torch.Tensor:
        """act(bn(conv(cost_volume(left_fea, right_fea)))) without materialising the cost volume (inference)."""
        _wk, scale, shift = self.prepared()
        return ops.costvol_stem(left_fea, right_fea, maxdisp, self.costvol_variants(), self.conv.out_channels, scale, shift,
                                self.relu, tails=tails, out_g4=out_g4)

    def as_tail(self, out: torch.Tensor, out_ch0: int, g4: bool = False) -> "ops.Tail":
        """This 1x1(x1) ConvBR (<= 4 output channels) as a tail of the kernel that produces its input (g4: `out` is a
        channel-group-interleaved buffer, ops.Tail)."""
        if self._geometry() != 1 or self.conv.out_channels > 4:
            raise ValueError("only 1x1x1 ConvBR with <= 4 output channels can be fused as a tail")
        wk, scale, shift = self.prepared()
        return ops.Tail(wk, scale, shift, self.relu, out, out_ch0, g4=g4)

    def as_down_tails(self, out: torch.Tensor, out_ch0: int) -> List["ops.Tail"]:
        """This 1x1x1 ConvBR (<= 8 output channels) applied to the x0.5 trilinear down-sampling of the producer's output, as one or
        two down-sampling tails (4 output channels each) of the kernel that produces its input; `out` is at half resolution."""
        if self._geometry() != 1 or self.conv.out_channels > 8:
            raise ValueError("only 1x1x1 ConvBR with <= 8 output channels can be fused as down-sampling tails")
        wk, scale, shift = self.prepared()
        tails = []
        for c0 in range(0, self.conv.out_channels, 4):
            c1 = min(c0 + 4, self.conv.out_channels)
            tails.append(ops.Tail(wk[c0:c1], None if scale is None else scale[c0:c1], None if shift is None else shift[c0:c1],
                                  self.relu, out, out_ch0 + c0, down=True))
        return tails

    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, out_ch0: int = 0,
                resample_to: Optional[Sequence[int]] = None, tails: Optional[Sequence["ops.Tail"]] = None,
                store_main: bool = True, out_dtype: Optional[torch.dtype] = None, x_g4: bool = False) -> torch.Tensor:
        """`out`/`out_ch0` write into a channel slice of a wider buffer.  `x_g4` (3x3x3 form of the fused executor only): x is a
        channel-group-interleaved buffer (ops.conv3d_k3).  `resample_to` (1x1x1 only) first resamples x
        trilinearly (align_corners=True) to that size inside the same kernel — the reference's
        `conv(F.interpolate(x, size, mode='trilinear', align_corners=True))`.  The 2-D flavour accepts [B,C,H,W] (or an
        already depth-1 5-D view) and returns the same rank.  `out_dtype=torch.float32` (Cout <= 2 form only) keeps the result
        in fp32 although x is bf16: the head's `mat`."""
        if self.autograd_mode(x):
            if out is not None or tails:
                raise RuntimeError("rag_amd ConvBR: fused destinations/tails belong to the inference executor")
            return self._forward_autograd(x, resample_to)
        k = self._geometry()
        wk, scale, shift = self.prepared()
        cout = self.conv.out_channels
        if x_g4 and (k != 3 or self._small() or resample_to is not None or x.dim() != 5):
            raise RuntimeError("rag_amd ConvBR: a G4 input is taken by the 3x3x3 form only")
        if k == -3:
            return ops.conv2d_k3_strided(x if x.dim() == 4 else x[:, :, 0], wk, scale, shift, self.relu, self.conv.stride[0])
        squeeze = x.dim() == 4
        if squeeze:
            x = x.unsqueeze(2)
        # mixed storage (MatchingNet._run_chain): a bf16 input into an fp32 buffer is taken by the resample + 1x1x1 launch only
        # (an input already at the output size is its identity case)
        mixed = out is not None and out.dtype != x.dtype
        if mixed:
            if k != 1 or x.dtype != torch.bfloat16 or out.dtype != torch.float32:
                raise RuntimeError("rag_amd ConvBR: only a 1x1x1 conv crosses from bf16 storage to an fp32 destination")
            resample_to = tuple(x.shape[2:]) if resample_to is None else resample_to
        elif resample_to is not None and tuple(resample_to) == tuple(x.shape[2:]):
            resample_to = None
        if resample_to is not None:
            if k != 1:
                raise NotImplementedError("ConvBR: fused resample is built for the 1x1x1 form only")
            if out is None:
                out = torch.empty((x.shape[0], cout) + tuple(int(v) for v in resample_to), device=x.device, dtype=x.dtype)
            if not mixed and _volume(resample_to) > _volume(x.shape[2:]) and cout <= x.shape[1]:
                # upsampling: mix the channels (and fold the BatchNorm) on the SMALL volume, then interpolate the Cout maps and
                # apply the ReLU — both steps are affine and the taps sum to one, so only the rounding order differs
                low = torch.empty((x.shape[0], cout) + tuple(x.shape[2:]), device=x.device, dtype=x.dtype)
                ops.conv3d_k1(x, wk, scale, shift, False, low)
                ops.trilinear3d_act(low, resample_to, True, self.relu, out, out_ch0)
            else:
                ops.conv3d_k1_resample(x, resample_to, True, wk, scale, shift, self.relu, out, out_ch0)
            return out[:, :, 0] if squeeze else out
        if out is None:
            odt = out_dtype if (out_dtype is not None and k == 3 and self._small()) else x.dtype
            out = torch.empty((x.shape[0], cout) + tuple(x.shape[2:]), device=x.device, dtype=odt)
        if k == 3 and self._small():
            ops.conv3d_k3_small(x, wk, scale, shift, self.relu, out, out_ch0)
        elif k == 3:
            groups = [out_ch0 + 4 * g for g in range(ops.packed_groups(cout))]
            ops.conv3d_k3(x, wk, cout, scale, shift, self.relu, out, groups, tails=tails, store_main=store_main, x_g4=x_g4)
        else:
            ops.conv3d_k1(x, wk, scale, shift, self.relu, out, out_ch0)
        return out[:, :, 0] if squeeze else out


class ConvBR_3d(_ConvBR):
    """src/automl/operations_3d.py:31-47 — same ctor `(C_in, C_out, kernel_size, stride, padding, bn=True, relu=True)`."""
    NDIM = 3


class ConvBR_2d(_ConvBR):
    """src/automl/operations_2d.py:31-47 (Feature Net) on the same kernels over a depth-1 volume."""
    NDIM = 2


class Identity_2d(nn.Module):
    """src/automl/operations_2d.py Identity_2d."""

    def forward(self, x):
        return x


# src/automl/operations_3d.py:5-8 (stride is always 1 on the hot path)
def _skip(C, stride):
    if stride != 1:
        raise NotImplementedError("skip_connect_3d with stride != 1 is dead code in the reference (FactorizedReduce typo)")
    return Identity_3d()


OPS_3d = {
    "skip_connect_3d": _skip,
    "3d_conv_3x3": lambda C, stride: ConvBR_3d(C, C, 3, stride, 1),
}


# src/automl/genotypes_2d.py:10-12, operations_2d.py:5-8
PRIMITIVES = ["skip_connect_2d", "conv_3x3"]


def _skip_2d(C, stride):
    if stride != 1:
        raise NotImplementedError("skip_connect_2d with stride != 1 is dead code in the reference")
    return Identity_2d()


OPS_2d = {"skip_connect_2d": _skip_2d, "conv_3x3": lambda C, stride: ConvBR_2d(C, C, 3, stride, 1)}


class DisparityRegression(nn.Module):
    """src/models/rag_model.py:18-29: out[b,y,x] = sum_d x[b,d,y,x] * d."""

    def __init__(self, maxdisp):
        super().__init__()
        self.maxdisp = maxdisp

    def forward(self, x):
        assert x.is_contiguous() is True
        if ag.needs_grad(x):
            return ag.DispRegFn.apply(x, self.maxdisp)
        return ops.disparity_regression(x, self.maxdisp)


class Disp(nn.Module):
    """src/models/rag_model.py:32-44, fused: trilinear x3 (align_corners=False) -> Softmin -> regression."""

    def __init__(self, maxdisp=192):
        super().__init__()
        self.maxdisp = maxdisp
        self.softmax = nn.Softmin(dim=1)                            # kept for attribute parity; unused
        self.disparity = DisparityRegression(maxdisp=self.maxdisp)  # idem

    def forward(self, x):
        if ag.needs_grad(x):
            return ag.DispFn.apply(x, self.maxdisp)
        return ops.disp_softargmin(x, self.maxdisp)


class _Cell(nn.Module):
    """Shared executor of Cell_3d (rag_model.py:114-177) and Cell_2d (:47-111) on HIP kernels; tensors are 5-D
    (the 2-D cell runs on depth-1 volumes: scale_dimension(1, s) == 1, and trilinear with one plane is bilinear).

    forward(prev_prev_input, prev_input) -> (prev_input, concat).  The 1x1x1 preprocess convs
    write s0|s1 into one buffer, every selected op writes/accumulates straight into its
    channel slice of the concat buffer (no `sum`, no `torch.cat` passes), and sibling
    3x3x3 convs that read the same state run as one launch.  Op/branch pairing is
    positional like the reference (ops created in genotype-row order, consumed in
    ascending-branch visit order: SURVEY.md §8 A6).
    """

    def __init__(self, steps, block_multiplier, prev_prev_fmultiplier, prev_filter_multiplier, genotype,
                 filter_multiplier, downup_sample):
        super().__init__()
        self.genotype = genotype
        self.C_in = block_multiplier * filter_multiplier
        self.C_out = filter_multiplier
        self.C_prev = int(block_multiplier * prev_filter_multiplier)
        self.C_prev_prev = int(block_multiplier * prev_prev_fmultiplier)
        self.downup_sample = downup_sample
        self.pre_preprocess = self.CONV(self.C_prev_prev, self.C_out, 1, 1, 0)
        self.preprocess = self.CONV(self.C_prev, self.C_out, 1, 1, 0)
        self._steps = steps
        self.block_multiplier = block_multiplier
        self._ops = nn.ModuleList()
        if downup_sample == -1:
            self.scale = 0.5
        elif downup_sample == 1:
            self.scale = 2
        for x in self._rows():
            self._ops.append(self.OPS[self.PRIMS[x[1]]](self.C_out, stride=1))
        self._fused_cache: Dict[tuple, tuple] = {}

    def _rows(self):
        raise NotImplementedError

    def scale_dimension(self, dim, scale):
        return int((float(dim) - 1.0) * scale + 1.0) if dim % 2 == 1 else int(float(dim) * scale)

    def _contributions(self) -> Dict[int, List[Tuple[int, nn.Module]]]:
        """new-state index -> [(source state j, op module)] in the reference's visit order."""
        selected = set(int(v) for v in np.asarray(self._rows())[:, 0])
        contribs: Dict[int, List[Tuple[int, nn.Module]]] = {}
        offset, n_states, ops_index = 0, 2, 0
        for _ in range(self._steps):
            lst = []
            for j in range(n_states):
                if offset + j in selected:
                    lst.append((j, self._ops[ops_index]))
                    ops_index += 1
            contribs[n_states] = lst
            offset += n_states
            n_states += 1
        return contribs

    def _fused(self, mods: Sequence[ConvBR_3d]):
        """Concatenated packed weights / scale / shift of sibling convs (cached on their stamps)."""
        key = tuple(id(m) for m in mods)
        stamps = tuple(m.stamp() for m in mods)
        hit = self._fused_cache.get(key)
        if hit is None or hit[0] != stamps:
            prep = [m.prepared() for m in mods]
            if len(mods) == 1:
                fused = prep[0]
            else:
                # sibling convs as one stacked convolution: pack the concatenated raw weights (the packed layout has sections
                # that do not concatenate); BN scale / shift simply stack
                with torch.no_grad():
                    wk = ops.conv3d_k3_pack(torch.cat([m.conv.weight.detach() for m in mods]))
                fused = (wk, torch.cat([p[1] for p in prep]), torch.cat([p[2] for p in prep]))
            hit = (stamps, fused)
            self._fused_cache[key] = hit
        return hit[1]

    def _convbrs(self):
        """the ConvBR units of this cell (cached: the module tree of a cell never changes after construction)."""
        units = self.__dict__.get("_convbr_cache")
        if units is None:
            units = [m for m in self.modules() if isinstance(m, _ConvBR)]
            self.__dict__["_convbr_cache"] = units
        return units

    def autograd_mode(self, *inputs) -> bool:
        units = self._convbrs()
        if torch.is_grad_enabled():
            if any(t is not None and t.requires_grad for t in inputs):
                return True
            if any(m.conv.weight.requires_grad or m.bn.weight.requires_grad or m.bn.bias.requires_grad for m in units):
                return True
        return any(m.use_bn and m.bn.training for m in units)

    def forward(self, prev_prev_input, prev_input):
        run = self._run_autograd if self.autograd_mode(prev_prev_input, prev_input) else (lambda a, b: self._run(a, b)[0])
        if prev_input.dim() == 4:     # 2-D cell: run on depth-1 volumes
            return prev_input, run(prev_prev_input.unsqueeze(2), prev_input.unsqueeze(2)).squeeze(2)
        return prev_input, run(prev_prev_input, prev_input)

    def _run_autograd(self, prev_prev_input, prev_input):
        """The reference's forward node by node (rag_model.py:143-177) on the autograd Functions."""
        s0, s1 = prev_prev_input, prev_input
        if self.downup_sample != 0:
            s1 = ag.resample(s1, self.out_size(s1.shape[2:]), True)
        s0 = ag.resample(s0, s1.shape[2:], True)
        if s0.shape[1] != self.C_out:
            s0 = self.pre_preprocess(s0)
        s1 = self.preprocess(s1)
        states = [s0, s1]
        contribs = self._contributions()
        n_states = 2 + self._steps
        pending: Dict[int, list] = {k: [] for k in contribs}
        for k, lst in contribs.items():
            if not lst:
                raise ValueError("Cell_3d: a step with no selected branch (the reference fails in torch.cat here too)")
        # Sources in ascending order: state j is complete once every source < j has been applied (targets are always
        # later states).  The conv branches leaving one state run as ONE stacked convolution (ag.ConvBRGroupFn).
        for j in range(n_states):
            if j >= 2:
                acc = pending[j][0]
                for h in pending[j][1:]:
                    acc = ag.AddFn.apply(acc, h)
                states.append(acc)
            out_ops = [(k, op) for k, lst in contribs.items() for (src, op) in lst if src == j]
            convs = [(k, op) for (k, op) in out_ops if isinstance(op, _ConvBR)]
            grouped = (len(convs) > 1 and all(op._geometry() == 3 and op.use_bn and op.relu for _k, op in convs)
                       and states[j].dtype == torch.float32)
            if grouped:
                params = [p for _k, op in convs for p in (op.conv.weight, op.bn.weight, op.bn.bias)]
                # every target already holding exactly one contribution: hand it to the group as that unit's residual, the sum
                # then comes out of the BatchNorm + ReLU pass (no add launch, no extra tensor)
                fuse = all(len(pending[k]) == 1 and pending[k][0].dtype == torch.float32 for k, _op in convs)
                res = [pending[k][0] for k, _op in convs] if fuse else []
                outs = dict(zip([k for k, _op in convs],
                                ag.ConvBRGroupFn.apply(states[j], tuple(op for _k, op in convs), *params, *res)))
                if fuse:
                    for k, _op in convs:
                        pending[k].clear()
            for k, op in out_ops:
                pending[k].append(outs[k] if (grouped and isinstance(op, _ConvBR)) else op(states[j]))
        return torch.cat(states[-self.block_multiplier:], dim=1)

    def out_size(self, prev_size: Sequence[int]) -> Tuple[int, int, int]:
        """Spatial size this cell works at (and outputs) given the size of prev_input."""
        if self.downup_sample == 0:
            return tuple(int(v) for v in prev_size)
        return tuple(self.scale_dimension(int(v), self.scale) for v in prev_size)

    def dual_plan(self, s0_channels: int, s0_from_tail: bool = True) -> bool:
        """True when ONE dual launch produces every new state of this cell (what `_run` calls single_dual): both inputs feed conv
        branches into all of them and nothing else does — a property of the genotype (and of whether s0 goes through pre_preprocess)."""
        C = self.C_out
        contribs = self._contributions()
        conv_from = {j: [(k, op) for k, lst in contribs.items() for (src, op) in lst if src == j and isinstance(op, _ConvBR)] for j in (0, 1)}
        return (bool(conv_from[0]) and (s0_from_tail or s0_channels != C) and [k for k, _ in conv_from[0]] == [k for k, _ in conv_from[1]]
                and all(len(contribs[k]) == 2 for k, _ in conv_from[0]) and len(conv_from[0]) == self._steps
                and self.block_multiplier == self._steps)

    def _run(self, prev_prev_input, prev_input, pre: Optional[torch.Tensor] = None, pre_has=(False, False),
             tails: Optional[Sequence["ops.Tail"]] = None, store_main: bool = True, size: Optional[Sequence[int]] = None,
             pre_g4: bool = False, dtype: Optional[torch.dtype] = None):
        """forward() plus the cross-cell fusion hooks used by MatchingNet.matching (`dtype`: this cell's storage type when it is not
        its inputs' — mixed storage, MatchingNet._run_chain):
        `pre`/`pre_has`: the [B, 2C, ...] buffer in which the producers of the inputs have already written s0 (ch 0..C)
        and/or s1 (ch C..2C) as fused tails; `tails`/`store_main`: consumer 1x1x1 convs to compute in THIS cell's final
        conv launch (possible when one dual launch produces all new states).  Returns (concat or None, tails_applied)."""
        C = self.C_out
        if C % 4 != 0 or self.block_multiplier > self._steps:
            raise NotImplementedError("rag_amd.Cell_3d: filter_multiplier must be a multiple of 4 and "
                                      "block_multiplier <= steps (true for every cell the reference builds)")
        s0, s1 = prev_prev_input, prev_input
        # the trilinear resamples (rag_model.py:146-153) are fused into the 1x1x1 preprocess convs that consume them
        if size is None:
            size = self.out_size(s1.shape[2:])
        size = tuple(int(v) for v in size)
        if not pre_has[0] and s0.shape[1] == C and tuple(s0.shape[2:]) != size:
            s0 = ops.trilinear3d(s0, size, True)     # no pre_preprocess to fuse into (never the case in Network)
        D, H, W = size
        if (D == 1 and pre is None and not tails and store_main and s0.shape[1] != C and s1.dtype == torch.float32 and s0.dtype == torch.float32
                and ops.get_conv_precision() == "f16x3" and self.block_multiplier == self._steps):
            # A Cell_2d in which every new state is conv(s0) + conv(s1) (the all-conv genotype): ONE launch — the two 1x1 ConvBRs and
            # their bilinear resamples run in the staging of the dual 3x3 launch (ragmi_cell2d_fwd); s0 / s1 are never written
            contribs = self._contributions()
            from_ = {j: [(k, op) for k, lst in contribs.items() for (src, op) in lst if src == j and isinstance(op, _ConvBR)] for j in (0, 1)}
            if (len(from_[0]) == self._steps and [k for k, _ in from_[0]] == [k for k, _ in from_[1]]
                    and all(len(contribs[k]) == 2 for k, _ in from_[0])
                    and ops.cell2d_supported(C, s0.shape[1], s1.shape[1], C * self._steps, H, W)):
                cat = torch.empty((s1.shape[0], self.block_multiplier * C, D, H, W), device=s1.device, dtype=s1.dtype)
                pa, sa, ha = self._fused([op for _k, op in from_[0]])
                pb, sb, hb = self._fused([op for _k, op in from_[1]])
                first = 2 + self._steps - self.block_multiplier
                groups = [(k - first) * C + 4 * g for k, _op in from_[0] for g in range(C // 4)]
                ops.cell2d(s0, self.pre_preprocess.prepared() + (self.pre_preprocess.relu,), s1,
                           self.preprocess.prepared() + (self.preprocess.relu,), C, pa, sa, ha, pb, sb, hb, C * self._steps, True, cat, groups)
                return cat, False
        if pre_g4 and not (pre is not None and pre_has[0] and pre_has[1]):
            raise RuntimeError("rag_amd.Cell_3d: a G4 input buffer must arrive complete from its producers' fused tails")
        if pre is None:
            B, dev = s1.shape[0], s1.device
            pre = torch.empty((B, 2 * C, D, H, W), device=dev, dtype=dtype or s1.dtype)
        B, dev, adt = pre.shape[0], pre.device, pre.dtype
        n_states = 2 + self._steps
        first_cat = n_states - self.block_multiplier             # first state that lands in the concat buffer
        tails_applied = False
        contribs = self._contributions()
        conv_from = {j: [(k, op) for k, lst in contribs.items() for (src, op) in lst
                         if src == j and isinstance(op, _ConvBR)] for j in (0, 1)}
        # one dual launch produces every new state <=> both inputs feed conv branches into all of them and nothing else does
        single_dual = (bool(conv_from[0]) and (pre_has[0] or s0.shape[1] != C) and [k for k, _ in conv_from[0]] == [k for k, _ in conv_from[1]]
                       and all(len(contribs[k]) == 2 for k, _ in conv_from[0]) and len(conv_from[0]) == self._steps
                       and self.block_multiplier == self._steps)
        use_tails = bool(tails) and single_dual and C * self._steps <= 16
        drop_main = use_tails and not store_main
        cat = (pre if drop_main else   # placeholder pointer: nothing is stored when the output is only consumed by tails
               torch.empty((B, self.block_multiplier * C, D, H, W), device=dev, dtype=adt))
        scratch = (torch.empty((B, (first_cat - 2) * C, D, H, W), device=dev, dtype=adt)
                   if first_cat > 2 else None)

        # (buffer, first channel) of every state
        where: List[Tuple[torch.Tensor, int]] = []
        # an UP-sampled input runs conv-first through the ConvBR forward (channel mix on the small volume); everything else —
        # down-sampling or an input already at the cell's size — shares one paired launch (measured: splitting an
        # identity + down-sampling pair into two launches costs 43 us instead of 27)
        grows = any(s is not None and _volume(size) > _volume(s.shape[2:]) and C <= s.shape[1] for s in (s0, s1))
        if not pre_has[0] and not pre_has[1] and s0.shape[1] != C and not grows and s0.dtype == s1.dtype == adt:
            # both 1x1x1 convs (each with its own fused resample) as ONE launch
            w0, sc0, sh0 = self.pre_preprocess.prepared()
            w1, sc1, sh1 = self.preprocess.prepared()
            ops.conv3d_k1_resample_pair([(s0, w0, sc0, sh0, self.pre_preprocess.relu, 0),
                                         (s1, w1, sc1, sh1, self.preprocess.relu, C)], size, pre)
            where.append((pre, 0))
        else:
            if pre_has[0]:
                where.append((pre, 0))             # already written by the producer of prev_prev_input (fused tail)
            elif s0.shape[1] != C:
                self.pre_preprocess(s0, out=pre, out_ch0=0, resample_to=size)
                where.append((pre, 0))
            else:
                where.append((s0.contiguous(), 0))
            if not pre_has[1]:
                self.preprocess(s1, out=pre, out_ch0=C, resample_to=size)
        where.append((pre, C))
        for k in range(2, n_states):
            where.append((cat, (k - first_cat) * C) if k >= first_cat else (scratch, (k - 2) * C))

        written = {k: False for k in contribs}
        pending_id = {k: [j for (j, op) in lst if not isinstance(op, _ConvBR)] for k, lst in contribs.items()}
        for k, lst in contribs.items():
            if not lst:
                raise ValueError("Cell_3d: a step with no selected branch (the reference fails in torch.cat here too)")

        def finalize(k: int) -> None:
            buf, ch = where[k]
            ids = pending_id[k]
            while ids:
                if not written[k]:
                    if len(ids) >= 2:
                        (ba, ca), (bb, cb) = where[ids[0]], where[ids[1]]
                        ops.add(ba, ca, bb, cb, buf, ch, C)
                        del ids[:2]
                    else:
                        bs, cs = where[ids.pop(0)]
                        torch.mul(bs[:, cs:cs + C], 1, out=buf[:, ch:ch + C])   # lone identity: a copy KERNEL (no memcpy node when captured)
                    written[k] = True
                else:
                    bs, cs = where[ids.pop(0)]
                    ops.add(buf, ch, bs, cs, buf, ch, C)

        # Fast path: the conv branches from s0 and from s1 feed the same new states and nothing else does
        # (e.g. the all-conv genotype): ONE dual-input launch computes relu(bn(conv(s0))) + relu(bn(conv(s1)))
        # for all of them, so the running sum never goes through HBM.
        done = set()
        if (conv_from[0] and where[0][0] is pre and [k for k, _ in conv_from[0]] == [k for k, _ in conv_from[1]]
                and all(len(contribs[k]) == 2 and not pending_id[k] for k, _ in conv_from[0])
                and len({id(where[k][0]) for k, _ in conv_from[0]}) == 1):
            pa, sa, ha = self._fused([op for _k, op in conv_from[0]])
            pb, sb, hb = self._fused([op for _k, op in conv_from[1]])
            groups = [where[k][1] + 4 * g for k, _op in conv_from[0] for g in range(C // 4)]
            ops.conv3d_k3_dual(pre, C, pa, sa, ha, pb, sb, hb, C * len(conv_from[0]), True,
                               where[conv_from[0][0][0]][0], groups,
                               tails=tails if use_tails else None, store_main=not drop_main, x_g4=pre_g4)
            tails_applied = use_tails
            for j in (0, 1):
                for k, op in conv_from[j]:
                    written[k] = True
                    done.add((j, id(op)))
        elif pre_g4:
            raise RuntimeError("rag_amd.Cell_3d: a G4 input buffer was planned for a cell that does not run as one dual launch")

        for j in range(n_states):
            if j >= 2:
                finalize(j)
            parts: Dict[object, list] = {}
            for k, lst in contribs.items():
                for (src, op) in lst:
                    if src != j or not isinstance(op, _ConvBR) or (j, id(op)) in done:
                        continue
                    if written[k]:
                        res = where[k]                                   # running sum: accumulate in place
                    else:
                        ready = [i for i in pending_id[k] if i <= j]     # identity partner already complete
                        if ready:
                            pending_id[k].remove(ready[0])
                            res = where[ready[0]]
                        else:
                            res = None
                    written[k] = True
                    parts.setdefault(None if res is None else id(res[0]), []).append((k, op, res))
            if not parts:
                continue
            xbuf, xch = where[j]
            x = xbuf[:, xch:xch + C]
            for items in parts.values():
                mods = [op for (_k, op, _r) in items]
                packed, scale, shift = self._fused(mods)
                out_groups, res_groups = [], []
                for (k, _op, res) in items:
                    out_groups += [where[k][1] + 4 * g for g in range(C // 4)]
                    if res is not None:
                        res_groups += [res[1] + 4 * g for g in range(C // 4)]
                res_buf = items[0][2][0] if items[0][2] is not None else None
                # all destinations of one launch live in one buffer except when a scratch state is involved
                by_buf: Dict[int, list] = {}
                for idx, (k, _op, _res) in enumerate(items):
                    by_buf.setdefault(id(where[k][0]), []).append(idx)
                if len(by_buf) == 1:
                    ops.conv3d_k3(x, packed, C * len(mods), scale, shift, True, where[items[0][0]][0], out_groups,
                                  res_buf, res_groups if res_buf is not None else None)
                else:
                    for idxs in by_buf.values():
                        sub = [items[i] for i in idxs]
                        p2, s2, h2 = self._fused([op for (_k, op, _r) in sub])
                        og = [where[k][1] + 4 * g for (k, _o, _r) in sub for g in range(C // 4)]
                        rg = [r[1] + 4 * g for (_k, _o, r) in sub if r is not None for g in range(C // 4)]
                        ops.conv3d_k3(x, p2, C * len(sub), s2, h2, True, where[sub[0][0]][0], og,
                                      res_buf, rg if res_buf is not None else None)
        for k in contribs:
            finalize(k)
        if tails_applied and drop_main:
            return None, True          # the concat exists only inside the kernel: its consumers were the tails
        return cat, tails_applied


class Cell_3d(_Cell):
    """src/models/rag_model.py:114-177: forward(prev_prev_input, prev_input) -> (prev_input, concat); ops from genotype.reduce."""
    CONV, OPS, PRIMS = ConvBR_3d, OPS_3d, PRIMITIVES_3D

    def _rows(self):
        return self.genotype.reduce


class Cell_2d(_Cell):
    """src/models/rag_model.py:47-111 (Feature Net cell; ops from genotype.normal) on the same executor."""
    CONV, OPS, PRIMS = ConvBR_2d, OPS_2d, PRIMITIVES

    def _rows(self):
        return self.genotype.normal


# Matching-Net macro architecture, src/models/rag_model.py:238-261:
# (prev_prev_fmultiplier, prev_filter_multiplier, filter_multiplier, downup_sample)
_CELL3D_ARCH = ((4, 4, 4, 0), (4, 4, 4, 0), (4, 4, 4, 0), (4, 4, 8, -1),
                (4, 8, 16, -1), (8, 16, 8, 1), (16, 8, 16, -1), (8, 16, 16, 0))


class MatchingNet(nn.Module):
    """The Matching-Net half of the reference `Network` (src/models/rag_model.py:230-275, 325-387).

    forward(left_fea[B,C,h,w], right_fea[B,C,h,w], task_arch=None) -> disp[B,3h,3w] wraps the three
    hot pieces the reference runs inline: cost-volume loop (:375-383) -> matching() (:325-366) ->
    Disp (:32-44).  Sub-module names (stem3d0, stem3d1, cells_3d, last_3_3d, last_6_3d, last_12_3d,
    disp) and therefore state_dict keys are the reference's; units are nn.ModuleLists indexed by
    task_arch[name][0] exactly like Network.matching.  `maxdisp` is a ctor argument (the reference
    hard-codes 192, rag_model.py:274).
    """

    def __init__(self, genotype=ALL_CONV_GENOTYPE, maxdisp: int = 192):
        super().__init__()
        self._init_matching(genotype, maxdisp)

    def _init_matching(self, genotype, maxdisp):
        self._step = 3
        self._block_multiplier = 3
        self._filter_multiplier = 4
        self._num_layers_3d = 8
        initial_fm = self._filter_multiplier * self._block_multiplier
        if not hasattr(self, "length"):
            self.length, self.arch_init = {}, {}
        for name in ("stem_3d0", "stem_3d1", "last_3_3d", "last_6_3d", "last_12_3d"):
            self.length[name] = 1
            self.arch_init[name] = [0]
        self.cells_3d = nn.ModuleList()
        self.stem3d0 = nn.ModuleList([ConvBR_3d(initial_fm * 2, initial_fm, 3, stride=1, padding=1)])
        self.stem3d1 = nn.ModuleList([ConvBR_3d(initial_fm, initial_fm, 3, stride=1, padding=1)])
        for i in range(self._num_layers_3d):
            self.cells_3d.append(nn.ModuleList([self._new_cell_3d(i, genotype)]))
            self.arch_init["cell_3d" + str(i)] = [0]
            self.length["cell_3d" + str(i)] = 1
        self.last_3_3d = nn.ModuleList([ConvBR_3d(initial_fm, 1, 3, 1, 1, bn=False, relu=False)])
        self.last_6_3d = nn.ModuleList([ConvBR_3d(initial_fm * 2, initial_fm, 1, 1, 0)])
        self.last_12_3d = nn.ModuleList([ConvBR_3d(initial_fm * 4, initial_fm * 2, 1, 1, 0)])
        self.maxdisp = maxdisp
        self.disp = Disp(self.maxdisp)

    def _new_cell_3d(self, i: int, genotype) -> Cell_3d:
        pp, p, fm, du = _CELL3D_ARCH[i]
        return Cell_3d(self._step, self._block_multiplier, pp, p, genotype, fm, du)

    # -- rag_model.py:325-366
    def matching(self, x, task_arch, path=None, features=None):
        """`features=(left_fea, right_fea)` with x=None: the cost volume is not built; stem3d0 consumes the feature maps
        directly (ragmi_costvol_stem_fwd) — inference only, same result as matching(cost_volume(...))."""
        def unit(name):
            return task_arch[name][0] if task_arch is not None else None

        cells = []
        for i, cell in enumerate(self.cells_3d):
            arch_cell = None
            if task_arch is not None:
                arch_cell = task_arch["cell_3d" + str(i)][0]
            elif path is not None:
                arch_cell = path[i + 1]
            cells.append(cell[arch_cell])
        last = self._run_chain(x, self.stem3d0[unit("stem_3d0")], self.stem3d1[unit("stem_3d1")], cells, features)
        return self._head(self._vol_size(x, features), last, unit("last_3_3d"), unit("last_6_3d"), unit("last_12_3d"))

    # -- rag_model.py:663-685
    def search_matching(self, x, selected_ops, t, features=None):
        cells = [cell[selected_ops[i + 10]] for i, cell in enumerate(self.cells_3d)]
        last = self._run_chain(x, self.stem3d0[selected_ops[8]], self.stem3d1[selected_ops[9]], cells, features)
        return self._head(self._vol_size(x, features), last, t, t, t)

    def _vol_size(self, x, features):
        if x is not None:
            return tuple(x.shape[2:])
        return (int(self.maxdisp / 3),) + tuple(features[0].shape[2:])

    def _run_chain(self, x, stem0, stem1, cells, features=None):
        """stem3d0 -> stem3d1 -> cells (rag_model.py:341-351) with cross-module fusion: the 1x1x1 pre_preprocess /
        preprocess conv of a cell that needs no resampling is computed in the epilogue of the kernel that PRODUCES its
        input (a "tail"), and a tensor consumed only by tails is never written to HBM.  Tensors: T[-2] = stem0 output,
        T[-1] = stem1 output, T[i] = output of cell i; cell i reads T[i-2] (prev_prev) and T[i-1] (prev)."""
        train = (stem0.autograd_mode(*(features if x is None else (x,))) or stem1.autograd_mode()
                 or any(c.autograd_mode() for c in cells))
        if x is None and (train or not stem0.costvol_fusable(features[0].shape[1])):
            x = self.cost_volume(*features)
        if train:
            out = (stem0(x),)                     # training: the reference's graph node by node (rag_model.py:341-351)
            out = (out[0], stem1(out[0]))
            for c in cells:
                out = c(out[0], out[1])
            return out[-1]
        n = len(cells)
        vol = self._vol_size(x, features)
        sizes = {-2: vol, -1: vol}
        for i, c in enumerate(cells):
            sizes[i] = c.out_size(sizes[i - 1])
        ref = x if x is not None else features[0]
        B, dev, adt = ref.shape[0], ref.device, ref.dtype
        # Storage type per cell (round 5, mixed storage of BASELINE configs[2]): under bf16 storage only the FULL-RESOLUTION tensors stay
        # bf16 — a cell that works below the cost volume's resolution (at most 1/8 of its voxels), and every cell behind one, keeps fp32
        # (ops.set_bf16_deep_fp32: the level-12 cells and the head carry most of the bf16 error, the deep levels ~4 % of the bytes;
        # tests/analysis_bf16_stage_epe.py).  cdt[j] = dtype of cell j's s0|s1 buffer
        # and of its output; the stems' outputs keep `adt`.  The edges that cross are bf16 -> fp32 only: down-sampling tails
        # (RAGMI_TAIL_F32) and the resample + 1x1x1 launch (RAGMI_OUT_F32).
        cdt: Dict[int, torch.dtype] = {-2: adt, -1: adt}
        for j in range(n):
            deep = adt == torch.bfloat16 and ops.bf16_deep_fp32_enabled() and (
                _volume(sizes[j]) * ops.bf16_deep_fp32_ratio() <= _volume(vol) or cdt[j - 1] == torch.float32 or cdt[j - 2] == torch.float32)
            cdt[j] = torch.float32 if deep else adt

        def down_ok(i, j):
            """cell j's 1x1x1 conv on T[i] as DOWN-SAMPLING tails of T[i]'s producer: cell j works at exactly half of T[i]'s size,
            the source pairs of that x0.5 resampling are aligned, and the producer is a level-3 dual launch on the z-marching
            split-operand kernel (fp32 storage under the default precision, or bf16 storage) — the only form that takes them"""
            if i < 0 or (cdt[i] == torch.float32 and ops.get_conv_precision() != "f16x3"):
                return False
            if cdt[j] != cdt[i] and not (cdt[i] == torch.bfloat16 and cdt[j] == torch.float32):
                return False          # (a bf16 producer may write fp32 down-sampling tails: RAGMI_TAIL_F32; nothing else crosses)
            src, dst, prod = sizes[i], sizes[j], cells[i]
            if tuple(2 * v for v in dst) != tuple(src) or not ops.down2_tail_supported(*src):
                return False
            cj = cells[j]
            if cj.C_out > 8 or cj.C_out % 4 != 0:
                return False
            # the producer: one dual launch with all its new states (what `_run` calls single_dual), 12 output channels, on the x3 form
            if prod.downup_sample != 0 or prod.C_out * prod._steps > 16 or prod.C_out != 4:
                return False
            return ops.conv3d_k3_uses_x3(2 * prod.C_out, prod.C_out * prod._steps, B, *src, nset=2, ntail=1, dtype=cdt[i])

        def fusable(i):
            """consumers of T[i] that can ride on its producer: [(cell index j, role 0 = pre_preprocess / 1 = preprocess, down)]"""
            out = []
            j = i + 1
            same = lambda jj: cdt[jj] == cdt[i]  # noqa: E731  (full-resolution tails store the producer's own type)
            if 0 <= j < n and cells[j].downup_sample == 0 and cells[j].C_out <= 4 and cells[j].C_out % 4 == 0 and same(j):
                out.append((j, 1, False))
            elif 0 <= j < n and cells[j].downup_sample == -1 and down_ok(i, j):
                out.append((j, 1, True))
            j = i + 2
            # (the cell between producer and consumer keeps the size: for j == 0 that "cell" is stem3d1 — same size by construction —
            # not cells[-1], the LAST cell)
            mid_same = j - 1 < 0 or (j - 1 < n and cells[j - 1].downup_sample == 0)
            if (0 <= j < n and cells[j].downup_sample == 0 and mid_same and cells[j].C_out <= 4
                    and cells[j].C_out % 4 == 0 and cells[j].C_prev_prev != cells[j].C_out and same(j)):
                out.append((j, 0, False))
            elif (0 <= j < n and cells[j].downup_sample == -1 and mid_same and cells[j].C_prev_prev != cells[j].C_out
                  and down_ok(i, j)):
                out.append((j, 0, True))
            return out

        def all_consumers(i):
            return [(j, r) for (j, r) in ((i + 1, 1), (i + 2, 0)) if 0 <= j < n]

        pre: Dict[int, torch.Tensor] = {}
        has: Dict[int, List[bool]] = {}

        # --- G4 plan (round 5): which of the private level-3 tensors are stored channel-group-interleaved ([B][C/4][D][H][W][4],
        # include/rag_amd.h) instead of as channel planes.  pre[j] (cell j's s0|s1 buffer) can be G4 when cell j runs as ONE dual launch
        # on the kernel that reads G4 (ops.conv3d_k3_g4_caps bit 0) and BOTH halves arrive as full-resolution fused tails from producers
        # that can write G4 (bit 1; stem3d0 folded with the cost volume always can).  The full-resolution tails of one producer launch
        # share a layout, so candidates are withdrawn until every producer is consistent.  T[-2] (stem3d0 -> stem3d1) can be G4 when
        # nothing but stem3d1 and fused tails reads it.  Everything else, and every module boundary, stays channel planes.
        def n_tails(i):
            """(full-resolution, down-sampling) tail counts of T[i]'s producer launch"""
            f = fusable(i)
            return (sum(1 for (_j, _r, d) in f if not d), sum((cells[j].C_out + 3) // 4 for (j, _r, d) in f if d))

        def out_channels(i):
            return (stem0 if i == -2 else stem1).conv.out_channels if i < 0 else cells[i].block_multiplier * cells[i].C_out

        def writes_g4(i):
            """T[i]'s producer applies its fused tails and can write them G4"""
            nt, nd = n_tails(i)
            if i == -2:
                return x is None or bool(ops.conv3d_k3_g4_caps(stem0.conv.in_channels, out_channels(-2), B, *vol, nset=1, ntail=nt, dtype=adt) & 2)
            if i == -1:
                return bool(ops.conv3d_k3_g4_caps(out_channels(-2), out_channels(-1), B, *vol, nset=1, ntail=nt, dtype=adt) & 2)
            c = cells[i]
            return (c.dual_plan(out_channels(i - 2)) and c.C_out * c._steps <= 16
                    and bool(ops.conv3d_k3_g4_caps(2 * c.C_out, c.C_out * c._steps, B, *sizes[i], nset=2, ntail=nt, ndown=nd, dtype=cdt[i]) & 2))

        g4: Dict[int, bool] = {}
        g4_ok = ((adt == torch.float32 and ops.get_conv_precision() == "f16x3") or adt == torch.bfloat16) and ops.g4_enabled()
        for j in range(n):
            c = cells[j]
            nt, nd = n_tails(j)
            g4[j] = (g4_ok and (j, 1, False) in fusable(j - 1) and (j, 0, False) in fusable(j - 2) and c.dual_plan(out_channels(j - 2))
                     and bool(ops.conv3d_k3_g4_caps(2 * c.C_out, c.C_out * c._steps, B, *sizes[j], nset=2, ntail=nt if c.C_out * c._steps <= 16 else 0,
                                                    ndown=nd if c.C_out * c._steps <= 16 else 0, dtype=cdt[j]) & 1)
                     and writes_g4(j - 1) and writes_g4(j - 2))
        changed = True
        while changed:          # one layout per producer launch
            changed = False
            for i in range(-2, n):
                js = [j for (j, _r, d) in fusable(i) if not d]
                if len({g4[j] for j in js}) > 1:
                    for j in js:
                        g4[j] = False
                    changed = True
        # stem3d0's own output: read by stem3d1 (3x3x3) and by cell 0's fused pre_preprocess tail only
        t2_g4 = (g4_ok and x is None and n >= 1 and (0, 0, False) in fusable(-2)
                 and bool(ops.conv3d_k3_g4_caps(out_channels(-2), out_channels(-1), B, *vol, nset=1, ntail=n_tails(-1)[0], dtype=adt) & 1))

        self.last_g4_plan = {"pre": dict(g4), "stem0_out": bool(t2_g4)}      # (what the tests and tools read)

        def tails_for(i):
            """[(consumer cell j, role, [Tail, ...])]: one full-resolution tail, or the one / two down-sampling tails of a consumer
            that works one level down"""
            specs = []
            for (j, role, down) in fusable(i):
                if j not in pre:
                    pre[j] = torch.empty((B, 2 * cells[j].C_out) + sizes[j], device=dev, dtype=cdt[j])
                    has[j] = [False, False]
                mod = cells[j].preprocess if role == 1 else cells[j].pre_preprocess
                ch0 = cells[j].C_out if role == 1 else 0
                specs.append((j, role, mod.as_down_tails(pre[j], ch0) if down else [mod.as_tail(pre[j], ch0, g4=g4[j])]))
            return specs

        def flat(specs):
            return [t for (_j, _r, ts) in specs for t in ts] or None

        def settle(i, specs, applied, tensor):
            """mark fused consumers as done, or run them as plain 1x1x1 launches if the producer could not fuse them"""
            for (j, role, ts) in specs:
                if not applied:
                    if g4[j]:
                        raise RuntimeError("rag_amd.MatchingNet: a producer planned to write a G4 tail did not apply its tails")
                    mod = cells[j].preprocess if role == 1 else cells[j].pre_preprocess
                    mod(tensor, out=pre[j], out_ch0=ts[0].out_ch0, resample_to=sizes[j] if ts[0].down else None)
                has[j][role] = True

        T: Dict[int, Optional[torch.Tensor]] = {}
        specs = tails_for(-2)
        specs1 = tails_for(-1)
        need_main = len(specs1) < len(all_consumers(-1)) or n == 0
        keep1 = need_main or not specs1
        # Round 5: stem3d0 and stem3d1 as ONE call whose second kernel expands stem3d0's output from the variant planes in its own
        # staging (ops.costvol_stem_conv3d): when nothing but stem3d1 and fused tails reads T[-2], that 164 MB tensor is never written.
        fuse_stems = (x is None and ops.stem_fusion_enabled() and n >= 1
                      and ((adt == torch.float32 and ops.get_conv_precision() == "f16x3") or adt == torch.bfloat16)
                      and (0, 0, False) in fusable(-2) and stem0.conv.out_channels == 12 and stem1._geometry() == 3 and not stem1._small()
                      and ops.costvol_stem_conv3d_supported(features[0].shape[1], 12, stem1.conv.out_channels, B, *vol,
                                                            ntail=len(flat(specs1) or []), dtype=adt))
        self.last_g4_plan["stems_fused"] = bool(fuse_stems)
        if fuse_stems:
            _w0, scale0, shift0 = stem0.prepared()
            wk1, scale1, shift1 = stem1.prepared()
            cout1 = stem1.conv.out_channels
            # cell 0's pre_preprocess (the one tail on stem3d0's output) in the four idle rows of stem3d1's 12-channel matrix product
            t0s = flat(specs)
            rows = ops.stem_tail_rows_enabled() and cout1 == 12 and t0s is not None and len(t0s) == 1 and t0s[0].weight2d.shape[0] == 4
            if rows:
                tmod = cells[0].pre_preprocess
                key = (stem1.stamp(), tmod.stamp())
                hit = getattr(stem1, "_rows_cache", None)
                if hit is None or hit[0] != key:
                    with torch.no_grad():
                        hit = (key, ops.conv3d_k3_pack(ops.stem_tail_rows_weight(stem1.conv.weight.detach(), t0s[0].weight2d)))
                    stem1._rows_cache = hit
                wk1 = hit[1]
            self.last_g4_plan["stem_tail_rows"] = bool(rows)
            out1 = torch.empty((B, cout1) + vol, device=dev, dtype=adt) if keep1 else None
            ops.costvol_stem_conv3d(features[0], features[1], self.maxdisp, stem0.costvol_variants(), 12, scale0, shift0, stem0.relu, flat(specs),
                                    wk1, cout1, scale1, shift1, stem1.relu, out1, [4 * g for g in range(ops.packed_groups(cout1))],
                                    tails=flat(specs1), store_main=keep1, tail0_rows=bool(rows))
            T[-2] = None
            settle(-2, specs, True, None)
        else:
            # stem3d0: its output also feeds stem3d1 (3x3x3), so it is materialised
            if x is None:
                T[-2] = stem0.forward_costvol(features[0], features[1], self.maxdisp, tails=flat(specs), out_g4=t2_g4)
            else:
                T[-2] = stem0(x, tails=flat(specs))
            settle(-2, specs, True, T[-2])
            # stem3d1
            out1 = stem1(T[-2], tails=flat(specs1), store_main=keep1, x_g4=t2_g4)
        T[-1] = out1 if keep1 else None
        settle(-1, specs1, True, out1)
        def shared_pre(i):
            """Cells i and i+1 both resample T[i-1] (the `prev` of one, the `prev_prev` of the other) to the SAME size: cell i's two
            1x1x1 convs and cell i+1's pre_preprocess as ONE launch — T[i-1] is gathered by both in the same launch (the second gather
            hits the L2 the first one filled) and cell i+1 keeps a plain 1x1x1 launch for its other input."""
            j = i + 1
            if j >= n or any(has.get(i, (False, False))) or has.get(j, [False, False])[0] or sizes[i] != sizes[j]:
                return
            s0, s1 = T.get(i - 2), T.get(i - 1)
            ci, cj = cells[i], cells[j]
            if s0 is None or s1 is None or s0.shape[1] == ci.C_out or s1.shape[1] == cj.C_out or tuple(s1.shape[2:]) == sizes[i]:
                return
            if len({s0.dtype, s1.dtype, cdt[i], cdt[j]}) > 1:
                return        # (mixed storage: the crossing launches stay with their cells)
            if any(_volume(sizes[i]) > _volume(t.shape[2:]) for t in (s0, s1)):
                return        # an up-sampling input runs conv-first (two launches): stays with its cell
            for k in (i, j):
                if k not in pre:
                    pre[k] = torch.empty((B, 2 * cells[k].C_out) + sizes[k], device=dev, dtype=cdt[k])
                    has[k] = [False, False]
            ops.conv3d_k1_resample_multi([(s0,) + ci.pre_preprocess.prepared() + (ci.pre_preprocess.relu, pre[i], 0),
                                          (s1,) + ci.preprocess.prepared() + (ci.preprocess.relu, pre[i], ci.C_out),
                                          (s1,) + cj.pre_preprocess.prepared() + (cj.pre_preprocess.relu, pre[j], 0)], sizes[i])
            has[i] = [True, True]
            has[j][0] = True

        for i, c in enumerate(cells):
            shared_pre(i)
            specs = tails_for(i)
            need_main = i == n - 1 or len(specs) < len(all_consumers(i))   # the head reads the last cell's output
            cat, applied = c._run(T[i - 2], T[i - 1], pre=pre.get(i), pre_has=tuple(has.get(i, (False, False))),
                                  tails=flat(specs), store_main=need_main, size=sizes[i], pre_g4=g4[i], dtype=cdt[i])
            settle(i, specs, applied, cat)
            T[i] = cat
            T.pop(i - 2, None)
        return T[n - 1]

    def _head(self, vol, last_output, i3, i6, i12):
        d, h, w = vol
        # `mat` — the [B,1,d,h,w] cost the soft-argmin reads — is always stored in fp32, also under bf16 activation storage: with
        # |cost| ~ 1e4 a bf16 rounding of it alone moved the disparity by 0.04-0.09 px (tests/analysis_bf16_stage_epe.py), and the
        # tensor is 1/12 of one level-3 activation
        f32 = torch.float32
        if last_output.size()[3] == h:
            return self.last_3_3d[i3](last_output, out_dtype=f32)
        heads = (self.last_3_3d[i3], self.last_6_3d[i6], self.last_12_3d[i12])
        up = ag.resample if any(m.autograd_mode(last_output) for m in heads) else ops.trilinear3d
        def up_last3(y):
            """upsample_6 + last_3_3d (rag_model.py:357-365): ONE kernel when the upsampling is an exact factor 2 (always, for the
            sizes the reference accepts) — the 12-channel full-resolution tensor is never written."""
            m3 = self.last_3_3d[i3]
            if (up is ops.trilinear3d and m3._small() and m3.conv.out_channels == 1 and tuple(y.shape[2:]) == (d // 2, h // 2, w // 2)
                    and (d, h, w) == (2 * (d // 2), 2 * (h // 2), 2 * (w // 2)) and ops.upconv3d_c1_supported(y.shape[1], d // 2, h // 2, w // 2)):
                wk, scale, shift = m3.prepared()             # the raw [1, C, 3, 3, 3] weight (VALU forms read it as is)
                return ops.upconv3d_c1(y, wk, scale, shift, m3.relu, out_dtype=f32)
            return m3(up(y, (d, h, w), True), out_dtype=f32)

        if last_output.size()[3] == h // 2:
            return up_last3(self.last_6_3d[i6](last_output))
        if last_output.size()[3] == h // 4:
            # upsample_12 is fused into last_6_3d's 1x1x1 kernel (conv-first); upsample_6 is fused into last_3_3d's kernel
            m12 = self.last_12_3d[i12]
            if (last_output.dtype == torch.bfloat16 and ops.bf16_head_fp32_enabled() and not m12.autograd_mode(last_output)
                    and m12._geometry() == 1):
                # bf16 storage: the head keeps fp32 from its first 1x1x1 conv on (the crossing launch: RAGMI_BF16 | RAGMI_OUT_F32)
                y12 = m12(last_output, out=torch.empty((last_output.shape[0], m12.conv.out_channels) + tuple(last_output.shape[2:]),
                                                       device=last_output.device, dtype=f32))
            else:
                m6 = self.last_6_3d[i6]
                half = (d // 2, h // 2, w // 2)
                if (up is ops.trilinear3d and m12._geometry() == 1 and m6._geometry() == 1 and ops.chain_k1_enabled()
                        and _volume(half) > _volume(last_output.shape[2:]) and m6.conv.out_channels <= m6.conv.in_channels
                        and ops.conv3d_k1_chain_supported(m12.conv.in_channels, m12.conv.out_channels, m6.conv.out_channels)):
                    # last_12_3d and the channel mix of last_6_3d (conv-first, as ConvBR.forward runs an up-sampling 1x1x1) as ONE launch
                    w1, s1, h1 = m12.prepared()
                    w2, s2, h2 = m6.prepared()
                    low = torch.empty((last_output.shape[0], m6.conv.out_channels) + tuple(last_output.shape[2:]), device=last_output.device,
                                      dtype=last_output.dtype)
                    ops.conv3d_k1_chain(last_output, w1, s1, h1, m12.relu, w2, s2, h2, False, low)
                    y = torch.empty((last_output.shape[0], m6.conv.out_channels) + half, device=last_output.device, dtype=last_output.dtype)
                    ops.trilinear3d_act(low, half, True, m6.relu, y, 0)
                    return up_last3(y)
                y12 = m12(last_output)
            y = self.last_6_3d[i6](y12, resample_to=(d // 2, h // 2, w // 2))
            return up_last3(y)
        # the reference reaches `return mat` with mat unbound here (UnboundLocalError)
        raise ValueError("MatchingNet: feature height must be a multiple of 4 (input H a multiple of 12)")

    def cost_volume(self, left_fea, right_fea):
        """The inline loop of rag_model.py:375-383 as one kernel."""
        if ag.needs_grad(left_fea, right_fea):
            return ag.CostVolFn.apply(left_fea, right_fea, self.maxdisp)
        return ops.costvol(left_fea, right_fea, self.maxdisp)

    def forward(self, left_fea, right_fea, task_arch=None):
        if task_arch is None:
            task_arch = self.arch_init
        if left_fea.shape != right_fea.shape or left_fea.dim() != 4:
            raise ValueError("MatchingNet: left/right features must both be [B, C, h, w]")
        # the cost volume (rag_model.py:375-383) is folded into stem3d0 (ragmi_costvol_stem_fwd); matching(cost_volume(..))
        # remains available and gives the same result
        cost = self.matching(None, task_arch, None, features=(left_fea, right_fea))
        return self.disp(cost)
