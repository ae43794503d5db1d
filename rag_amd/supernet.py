"""The MdeNAS supernet of the growth loop's cell search, on the HIP kernels.

Reference: `BasicNetwork` (src/automl/mdenas_basicmodel.py:48-134) = `AutoFeature` (automl/build_model_2d.py:155-240)
-> the inline cost-volume loop (:83-91) -> `AutoMatching` (automl/build_model_3d.py:155-275) -> `Disp`; every edge of
every cell is a `MixedOp` holding one module per primitive, and a forward pass runs the ONE op per edge named by
`fea_ops` / `mat_ops` (build_model_3d.py:11-24, sampled per step by mdenas_search.py).  Same leaf computations as the
grown network — ConvBR (1x1, 3x3), identity, (bi/tri)linear resize, sum, concat — so the modules below only wire
`rag_amd.modules` / `rag_amd.autograd`: class names, constructor signatures, attribute names and therefore state_dict
keys follow the reference (a reference supernet checkpoint loads strictly), and both inference and the search's
training step (mdenas_search.py:164-173) run on the HIP kernels.  2-D cells run on depth-1 volumes like `Cell_2d`.
"""
from __future__ import annotations

from copy import deepcopy
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import autograd as ag
from . import ops
from .modules import (OPS_2d, OPS_3d, PRIMITIVES, PRIMITIVES_3D, ConvBR_2d, ConvBR_3d, Disp, Genotype, MatchingNet)


def _resize(x: torch.Tensor, size: Sequence[int]) -> torch.Tensor:
    """F.interpolate(x, size, mode='(bi|tri)linear', align_corners=True) on a 5-D (depth-1 for 2-D) tensor."""
    size = tuple(int(v) for v in size)
    if tuple(x.shape[2:]) == size:
        return x
    return ag.resample(x, size, True) if ag.needs_grad(x) else ops.trilinear3d(x, size, True)


def _sum(parts):
    """sum(new_states) (build_model_3d.py:138)."""
    acc = parts[0]
    for h in parts[1:]:
        if ag.needs_grad(acc, h):
            acc = ag.AddFn.apply(acc, h)
        else:
            a, b = acc.contiguous(), h.contiguous()
            acc = ops.add(a, 0, b, 0, torch.empty_like(a), 0, a.shape[1])
    return acc


class _MixedOp(nn.Module):
    """build_model_3d.py:11-24 / build_model_2d.py:12-25: one module per primitive, forward(x, selected_op)."""
    OPS, PRIMS = OPS_3d, PRIMITIVES_3D

    def __init__(self, C, stride):
        super().__init__()
        self._ops = nn.ModuleList()
        for primitive in self.PRIMS:
            self._ops.append(self.OPS[primitive](C, stride))       # no 'pool' primitive exists in either list

    def forward(self, x, selected_op):
        return self._ops[int(selected_op)](x)


class MixedOp_3d(_MixedOp):
    OPS, PRIMS = OPS_3d, PRIMITIVES_3D


class MixedOp_2d(_MixedOp):
    OPS, PRIMS = OPS_2d, PRIMITIVES


class _SuperCell(nn.Module):
    """build_model_3d.py:26-152 (2-D twin: build_model_2d.py:27-152).  forward(s0, s1_down, s1_same, s1_up, n_alphas)
    returns the list of concat features, one per non-None s1 variant (always one in AutoMatching / AutoFeature)."""
    CONV, MIXED = ConvBR_3d, MixedOp_3d

    def __init__(self, steps, block_multiplier, prev_prev_fmultiplier, prev_fmultiplier_down, prev_fmultiplier_same,
                 prev_fmultiplier_up, filter_multiplier):
        super().__init__()
        self.C_in = block_multiplier * filter_multiplier
        self.C_out = filter_multiplier
        self.C_prev_prev = int(prev_prev_fmultiplier * block_multiplier)
        self._prev_fmultiplier_same = prev_fmultiplier_same
        if prev_fmultiplier_down is not None:
            self.C_prev_down = int(prev_fmultiplier_down * block_multiplier)
            self.preprocess_down = self.CONV(self.C_prev_down, self.C_out, 1, 1, 0)
        if prev_fmultiplier_same is not None:
            self.C_prev_same = int(prev_fmultiplier_same * block_multiplier)
            self.preprocess_same = self.CONV(self.C_prev_same, self.C_out, 1, 1, 0)
        if prev_fmultiplier_up is not None:
            self.C_prev_up = int(prev_fmultiplier_up * block_multiplier)
            self.preprocess_up = self.CONV(self.C_prev_up, self.C_out, 1, 1, 0)
        if prev_prev_fmultiplier != -1:
            self.pre_preprocess = self.CONV(self.C_prev_prev, self.C_out, 1, 1, 0)
        self._steps = steps
        self.block_multiplier = block_multiplier
        self._ops = nn.ModuleList()
        for i in range(self._steps):
            for j in range(2 + i):
                self._ops.append(None if (prev_prev_fmultiplier == -1 and j == 0) else self.MIXED(self.C_out, 1))

    def scale_dimension(self, dim, scale):
        assert isinstance(dim, int)
        return int((float(dim) - 1.0) * scale + 1.0) if dim % 2 else int(dim * scale)

    def prev_feature_resize(self, prev_feature, mode):
        scale = 0.5 if mode == "down" else 2
        return _resize(prev_feature, [self.scale_dimension(int(v), scale) for v in prev_feature.shape[2:]])

    def forward(self, s0, s1_down, s1_same, s1_up, n_alphas):
        variants = []
        if s1_down is not None:
            variants.append(self.preprocess_down(self.prev_feature_resize(s1_down, "down")))
        if s1_same is not None:
            variants.append(self.preprocess_same(s1_same))
        if s1_up is not None:
            variants.append(self.preprocess_up(self.prev_feature_resize(s1_up, "up")))
        if s0 is not None:
            s0 = _resize(s0, variants[-1].shape[2:])          # the reference uses the LAST assigned size (:88-99)
            if s0.shape[1] != self.C_out:
                s0 = self.pre_preprocess(s0)
        final_concates = []
        for s1 in variants:
            states = [s0, s1]                                   # s0 None: the reference's placeholder 0, never consumed
            offset = 0
            for _i in range(self._steps):
                new_states = []
                for j, h in enumerate(states):
                    op = self._ops[offset + j]
                    if op is None:
                        continue
                    new_states.append(op(h, n_alphas[offset + j]))
                offset += len(states)
                states.append(_sum(new_states))
            final_concates.append(torch.cat(states[-self.block_multiplier:], dim=1))
        return final_concates


class SuperCell_3d(_SuperCell):
    CONV, MIXED = ConvBR_3d, MixedOp_3d


class SuperCell_2d(_SuperCell):
    """2-D cell on depth-1 volumes: scale_dimension(1, s) == 1 keeps the depth, trilinear on one plane is bilinear."""
    CONV, MIXED = ConvBR_2d, MixedOp_2d


def _head(mod, last, d, h, w):
    """the shared upsample chain of AutoMatching / AutoFeature (build_model_3d.py:259-274)."""
    if last.shape[3] == h:
        return mod.last_3(last)
    if last.shape[3] == h // 2:
        return mod.last_3(_resize(mod.last_6(last), (d, h, w)))
    if last.shape[3] == h // 4:
        y = _resize(mod.last_12(last), (max(d // 2, 1), h // 2, w // 2))
        return mod.last_3(_resize(mod.last_6(y), (d, h, w)))
    if last.shape[3] == h // 8:
        y = _resize(mod.last_24(last), (max(d // 4, 1), h // 4, w // 4))
        y = _resize(mod.last_12(y), (max(d // 2, 1), h // 2, w // 2))
        return mod.last_3(_resize(mod.last_6(y), (d, h, w)))
    raise ValueError("supernet head: unsupported size (H and W must be multiples of 12)")   # the reference: unbound `mat`


class AutoMatching(nn.Module):
    """automl/build_model_3d.py:155-275; network architecture levels [0,0,0,1,2,1,2,2]."""

    def __init__(self, num_layers=8, filter_multiplier=4, block_multiplier=3, step=3, cell=SuperCell_3d):
        super().__init__()
        self.cells = nn.ModuleList()
        self._step, self._num_layers = step, num_layers
        self._block_multiplier, self._filter_multiplier = block_multiplier, filter_multiplier
        f = int(filter_multiplier)
        self._num_end = f * block_multiplier
        self.stem0 = ConvBR_3d(self._num_end * 2, self._num_end, 3, stride=1, padding=1)
        # (prev_prev, down, same, up, filter) per layer, build_model_3d.py:172-211
        spec = [(-1, None, f, None, f), (f, None, f, f * 2, f), (f, None, f, f * 2, f), (f, f, f * 2, f * 4, f * 2),
                (f, f * 2, f * 4, f * 8, f * 4), (f * 2, f, f * 2, f * 4, f * 2), (f * 4, f * 2, f * 4, f * 8, f * 4),
                (f * 2, f * 2, f * 4, f * 8, f * 4)]
        for i in range(num_layers):
            self.cells.append(cell(step, block_multiplier, *spec[min(i, 7)]))
        self.last_3 = ConvBR_3d(self._num_end, 1, 3, 1, 1, bn=False, relu=False)
        self.last_6 = ConvBR_3d(self._num_end * 2, self._num_end, 1, 1, 0)
        self.last_12 = ConvBR_3d(self._num_end * 4, self._num_end * 2, 1, 1, 0)
        self.last_24 = ConvBR_3d(self._num_end * 8, self._num_end * 4, 1, 1, 0)

    def forward(self, x, n_alphas):
        stem = self.stem0(x)
        c = self.cells
        l3, = c[0](None, None, stem, None, n_alphas)
        l3_1, = c[1](stem, None, l3, None, n_alphas)
        l3_2, = c[2](l3, None, l3_1, None, n_alphas)
        l6, = c[3](l3_1, l3_2, None, None, n_alphas)
        l12, = c[4](l3_2, l6, None, None, n_alphas)
        l6, = c[5](l6, None, None, l12, n_alphas)
        l12_1, = c[6](l12, l6, None, None, n_alphas)
        l12_2, = c[7](l6, None, l12_1, None, n_alphas)
        d, h, w = x.shape[2:]
        return _head(self, l12_2, d, h, w)


class AutoFeature(nn.Module):
    """automl/build_model_2d.py:155-240; levels [1,0,1,0]; input [B,3,H,W] -> features [B,12,H/3,W/3]."""

    def __init__(self, num_layers=4, filter_multiplier=4, block_multiplier=3, step=3, cell=SuperCell_2d):
        super().__init__()
        self.cells = nn.ModuleList()
        self._num_layers, self._step = num_layers, step
        self._block_multiplier, self._filter_multiplier = block_multiplier, filter_multiplier
        f = int(filter_multiplier)
        half = int(f / 2)
        self._num_end = f * block_multiplier
        self.stem0 = ConvBR_2d(3, half * block_multiplier, 3, stride=1, padding=1)
        self.stem1 = ConvBR_2d(half * block_multiplier, half * block_multiplier, 3, stride=3, padding=1)
        self.stem2 = ConvBR_2d(half * block_multiplier, f * block_multiplier, 3, stride=1, padding=1)
        spec = [(-1, f, None, None, f * 2), (f, None, f, f * 2, f), (f * 2, f, f * 2, f * 4, f * 2), (f, None, f, f * 2, f)]
        for i in range(num_layers):
            self.cells.append(cell(step, block_multiplier, *spec[i]))
        self.last_3 = ConvBR_2d(self._num_end, self._num_end, 1, 1, 0, bn=False, relu=False)
        self.last_6 = ConvBR_2d(self._num_end * 2, self._num_end, 1, 1, 0)
        self.last_12 = ConvBR_2d(self._num_end * 4, self._num_end * 2, 1, 1, 0)
        self.last_24 = ConvBR_2d(self._num_end * 8, self._num_end * 4, 1, 1, 0)

    def forward(self, x, n_alphas):
        stem2 = self.stem2(self.stem1(self.stem0(x))).unsqueeze(2)        # depth-1 volume from here on
        c = self.cells
        l6, = c[0](None, stem2, None, None, n_alphas)
        l3_1, = c[1](stem2, None, None, l6, n_alphas)
        l6_1, = c[2](l6, l3_1, None, None, n_alphas)
        l3_2, = c[3](l3_1, None, None, l6_1, n_alphas)
        _one, h, w = stem2.shape[2:]
        return _head(self, l3_2, 1, h, w)[:, :, 0]


class BasicNetwork(nn.Module):
    """automl/mdenas_basicmodel.py:48-134: forward(left, right, fea_ops, mat_ops) -> disp; `p` holds the per-edge op
    probabilities the MdeNAS search updates, `genotype()` reads the searched cell out of them."""

    def __init__(self, steps=3, multiplier=4, stem_multiplier=3, device="cuda:0", maxdisp: int = 192):
        super().__init__()
        self._steps, self._multiplier, self.device = steps, multiplier, device
        self.num_ops = len(PRIMITIVES)
        self.num_edges = sum(1 for i in range(self._steps) for n in range(2 + i))
        self.feature = AutoFeature()
        self.matching = AutoMatching()
        self.maxdisp = maxdisp                  # the reference hard-codes 192 (mdenas_basicmodel.py:63)
        self.disp = Disp(self.maxdisp)
        self.p = None
        self._initialize_p()

    def new(self):
        model_new = BasicNetwork(device=self.device, maxdisp=self.maxdisp).to(self.device)
        model_new.p = deepcopy(self.probability())
        return model_new

    def forward(self, left, right, fea_ops, mat_ops):
        x = self.feature(left, fea_ops)
        y = self.feature(right, fea_ops)
        cost = MatchingNet.cost_volume(self, x.contiguous(), y.contiguous())      # the loop of :83-91 as one kernel
        cost = self.matching(cost, mat_ops)
        return self.disp(cost)

    def _initialize_p(self):
        k, n = self.num_edges, self.num_ops
        self.p = {"normal": torch.full((k, n), 1 / n), "reduce": torch.full((k, n), 1 / n)}

    def probability(self):
        return self.p

    def genotype(self) -> Genotype:
        """The discrete architecture the probabilities currently favour (mdenas_basicmodel.py:98-133): per step, the two incoming
        edges whose strongest non-`none` operation is most probable, each with its most probable operation."""
        def strongest_two(prob: np.ndarray) -> np.ndarray:
            rows, first = [], 0
            for width in range(2, 2 + self._steps):            # step k chooses among its k + 2 candidate edges
                block = prob[first:first + width]
                strength = block[:, 1:].max(axis=1)            # column 0 is the `none` operation
                keep = first + np.argsort(-strength, kind="stable")[:2]
                rows.extend([int(e), int(prob[e].argmax())] for e in keep)
                first += width
            return np.asarray(rows)

        return Genotype(normal=strongest_two(F.softmax(self.p["normal"], dim=-1).numpy()), normal_concat=None,
                        reduce=strongest_two(F.softmax(self.p["reduce"], dim=-1).numpy()), reduce_concat=None)
