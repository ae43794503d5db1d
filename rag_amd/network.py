"""`Network` — the reference's grown stereo model (src/models/rag_model.py:181-845) with the Matching-Net
half running on the HIP kernels of ``rag_amd``.

What `approaches/rag.py` (Appr) drives is kept one-to-one: ``forward(left, right, t, task_arch, path)``,
``search_forward(left, right, t, selected_ops)``, ``feature`` / ``matching`` / ``search_*``, the growth API
``expand(t, genotype, device)`` / ``select(t)`` / ``get_new_model`` / ``get_param`` / ``modify_param`` and the
attributes ``p, length, arch_init, new_models, model_to_train`` plus every per-layer ``nn.ModuleList`` name,
hence the checkpoint layout of ``run.py:194-196``.

The 2-D Feature Net (`Cell_2d`, `ConvBR_2d`; ≈1 % of the FLOPs) runs on the same HIP
kernels as the Matching Net: a 2-D conv is the 3-D kernel over a depth-1 volume, the stride-3 stem has its own
kernel (`ragmi_conv2d_k3_strided_fwd`), bilinear resampling is the trilinear kernel with one plane.  So
`Network.forward(left, right)` is HIP end to end; PyTorch only allocates.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn as nn

from .modules import Cell_2d, Cell_3d, ConvBR_2d, ConvBR_3d, MatchingNet, _ConvBR  # noqa: F401

# Feature-Net macro architecture, rag_model.py:207-219: (prev_prev_fm, prev_fm, filter_multiplier, downup)
_CELL2D_ARCH = ((4, 4, 8, -1), (4, 8, 4, 1), (8, 4, 8, -1), (4, 8, 4, 1))


class Network(MatchingNet):
    """src/models/rag_model.py:181-845.  Unit lists grow with `expand` and shrink with `select`;
    `task_arch[name][0]` picks the unit of each layer for a task."""

    def __init__(self, genotype, device, maxdisp: int = 192):
        nn.Module.__init__(self)
        self.device = device
        self._num_layers_2d = 4
        self._K_multiplier = 2
        self._genotype0 = genotype
        self.length: Dict[str, int] = {}
        self.arch_init: Dict[str, List[int]] = {}
        # feature net (HIP, depth-1 volumes)
        self.cells_2d = nn.ModuleList()
        self.stem2d0 = nn.ModuleList([self._new_unit("stem_2d0", genotype)])
        self.stem2d1 = nn.ModuleList([self._new_unit("stem_2d1", genotype)])
        self.stem2d2 = nn.ModuleList([self._new_unit("stem_2d2", genotype)])
        for name in ("stem_2d0", "stem_2d1", "stem_2d2", "last_3_2d"):
            self.length[name], self.arch_init[name] = 1, [0]
        for i in range(self._num_layers_2d):
            self.cells_2d.append(nn.ModuleList([self._new_unit(f"cell_2d{i}", genotype)]))
            self.length[f"cell_2d{i}"], self.arch_init[f"cell_2d{i}"] = 1, [0]
        self.last_3_2d = nn.ModuleList([self._new_unit("last_3_2d", genotype)])
        # matching net (HIP)
        self._init_matching(genotype, maxdisp)
        self.act_dtype = torch.float32   # torch.bfloat16: Matching-Net activations stored as bf16 (BASELINE config 3)
        self.p = None               # per-layer unit probabilities during search_t
        self.new_models = None
        self.model_to_train = None

    # ------------------------------------------------------------------ unit bookkeeping
    # order of self.p: 3 stems 2d, 4 cells 2d, last_3_2d, 2 stems 3d, 8 cells 3d   (rag_model.py:403-498)
    def _p_layers(self) -> List[str]:
        return (["stem_2d0", "stem_2d1", "stem_2d2"] + [f"cell_2d{i}" for i in range(self._num_layers_2d)] +
                ["last_3_2d", "stem_3d0", "stem_3d1"] + [f"cell_3d{i}" for i in range(self._num_layers_3d)])

    def _units(self, name: str) -> nn.ModuleList:
        if name.startswith("cell_2d"):
            return self.cells_2d[int(name[7:])]
        if name.startswith("cell_3d"):
            return self.cells_3d[int(name[7:])]
        return getattr(self, {"stem_2d0": "stem2d0", "stem_2d1": "stem2d1", "stem_2d2": "stem2d2", "stem_3d0": "stem3d0",
                              "stem_3d1": "stem3d1"}.get(name, name))

    def _new_unit(self, name: str, genotype) -> nn.Module:
        fm = 12  # initial_fm = _filter_multiplier * _block_multiplier (rag_model.py:194)
        if name == "stem_2d0":
            return ConvBR_2d(3, fm // 2, 3, stride=1, padding=1)
        if name == "stem_2d1":
            return ConvBR_2d(fm // 2, fm, 3, stride=3, padding=1)
        if name == "stem_2d2":
            return ConvBR_2d(fm, fm, 3, stride=1, padding=1)
        if name == "last_3_2d":
            return ConvBR_2d(fm, fm, 1, 1, 0, bn=False, relu=False)
        if name.startswith("cell_2d"):
            pp, p, f, du = _CELL2D_ARCH[int(name[7:])]
            return Cell_2d(3, 3, pp, p, genotype, f, du)
        if name == "stem_3d0":
            return ConvBR_3d(fm * 2, fm, 3, stride=1, padding=1)
        if name == "stem_3d1":
            return ConvBR_3d(fm, fm, 3, stride=1, padding=1)
        if name.startswith("cell_3d"):
            return self._new_cell_3d(int(name[7:]), genotype)
        if name == "last_3_3d":
            return ConvBR_3d(fm, 1, 3, 1, 1, bn=False, relu=False)
        if name == "last_6_3d":
            return ConvBR_3d(fm * 2, fm, 1, 1, 0)
        if name == "last_12_3d":
            return ConvBR_3d(fm * 4, fm * 2, 1, 1, 0)
        raise KeyError(name)

    # ------------------------------------------------------------------ forward paths
    def feature(self, x, task_arch, path):                      # rag_model.py:285-323
        def unit(name):
            return task_arch[name][0] if task_arch is not None else None

        stem0 = self.stem2d0[unit("stem_2d0")](x)
        stem1 = self.stem2d1[unit("stem_2d1")](stem0)
        stem2 = self.stem2d2[unit("stem_2d2")](stem1)
        out = (stem1, stem2)
        for i, cell in enumerate(self.cells_2d):
            arch_cell = task_arch[f"cell_2d{i}"][0] if task_arch is not None else (path[i + 1] if path is not None else None)
            out = cell[arch_cell](out[0], out[1])
        if out[-1].size()[2] != stem2.size()[2]:
            raise ValueError("Network.feature: H and W must be multiples of 12 (the reference prints 'this is a bug' here)")
        return self.last_3_2d[unit("last_3_2d")](out[-1])

    def search_feature(self, x, selected_ops):                  # rag_model.py:641-660
        stem0 = self.stem2d0[selected_ops[0]](x)
        stem1 = self.stem2d1[selected_ops[1]](stem0)
        stem2 = self.stem2d2[selected_ops[2]](stem1)
        out = (stem1, stem2)
        for i, cell in enumerate(self.cells_2d):
            out = cell[selected_ops[i + 3]](out[0], out[1])
        if out[-1].size()[2] != stem2.size()[2]:
            raise ValueError("Network.search_feature: H and W must be multiples of 12")
        return self.last_3_2d[selected_ops[7]](out[-1])

    def _training_graph(self, *inputs) -> bool:
        """True when the call must build the reference's autograd graph / use batch statistics (rag.py:155-219)."""
        units = [m for m in self.modules() if isinstance(m, _ConvBR)] if self.training or torch.is_grad_enabled() else None
        if torch.is_grad_enabled() and (any(t.requires_grad for t in inputs) or any(p.requires_grad for p in self.parameters())):
            return True
        if units is None:
            # eval() on the root puts every unit in eval; a unit switched back to train() by hand is still honoured below
            units = self.__dict__.get("_convbr_cache")
            if units is None or self.__dict__.get("_convbr_count") != sum(len(self._units(n)) for n in self._p_layers()):
                units = [m for m in self.modules() if isinstance(m, _ConvBR)]
                self.__dict__["_convbr_cache"] = units
                self.__dict__["_convbr_count"] = sum(len(self._units(n)) for n in self._p_layers())
        return any(m.use_bn and m.bn.training for m in units)

    def _features(self, left, right, feature):
        if self._training_graph(left, right):
            # two passes like the reference: train-mode BatchNorm statistics are per view (rag_model.py:371-372)
            return feature(left), feature(right)
        # inference: both views share the Feature-Net weights and results are batch-independent -> one batched pass
        B = left.shape[0]
        fea = feature(torch.cat([left, right])).to(self.act_dtype)
        return fea[:B].contiguous(), fea[B:].contiguous()

    def forward(self, left, right, t, task_arch=None, path=None):   # rag_model.py:369-387
        lf, rf = self._features(left, right, lambda x: self.feature(x, task_arch, path))
        cost = self.matching(None, task_arch, path, features=(lf, rf))     # cost volume folded into stem3d0
        return self.disp(cost)

    def search_forward(self, left, right, t, selected_ops):        # rag_model.py:688-706
        lf, rf = self._features(left, right, lambda x: self.search_feature(x, selected_ops))
        cost = self.search_matching(None, selected_ops, t, features=(lf, rf))
        return self.disp(cost)

    # ------------------------------------------------------------------ growth API
    def expand(self, t, genotype, device="cuda"):               # rag_model.py:391-522
        """Add one candidate unit per layer (built from `genotype`) and one head triple for task t; reset `p` so
        that every existing unit is K times as likely as the new one."""
        self.p = []
        for name in self._p_layers():
            self._units(name).append(self._new_unit(name, genotype).to(device))
            n_old = self.length[name]
            temp = torch.full((n_old + 1,), 1 / (self._K_multiplier * n_old + 1))
            temp[:n_old] *= self._K_multiplier
            self.p.append(temp)
        for name in ("last_3_3d", "last_6_3d", "last_12_3d"):
            self._units(name).append(self._new_unit(name, genotype).to(device))
        self.get_new_model(t=t)

    def get_new_model(self, t):                                 # rag_model.py:525-551
        new_models = {name: [self.length[name]] for name in self._p_layers()}
        for name in ("last_3_3d", "last_6_3d", "last_12_3d"):
            new_models[name] = [t]
        self.new_models = new_models

    def _modules_of(self, models) -> List[nn.Module]:
        order = (["stem_2d0", "stem_2d1", "stem_2d2", "last_3_2d", "stem_3d0", "stem_3d1", "last_3_3d", "last_6_3d",
                  "last_12_3d"] + [f"cell_2d{i}" for i in range(self._num_layers_2d)] +
                 [f"cell_3d{i}" for i in range(self._num_layers_3d)])
        return [self._units(name)[idx] for name in order if name in models for idx in models[name]]

    def get_param(self, models):                                # rag_model.py:553-595
        return [{"params": m.parameters()} for m in self._modules_of(models)]

    def modify_param(self, models, requires_grad=True):         # rag_model.py:597-638
        for m in self._modules_of(models):
            for param in m.parameters():
                param.requires_grad = requires_grad

    def select(self, t):                                        # rag_model.py:709-845
        """Keep, per layer, the most probable unit: reuse an old one (drop the candidate) or keep the candidate
        (it becomes a unit to train).  Returns the task's architecture dict."""
        model_to_train: Dict[str, list] = {}
        best_archi: Dict[str, list] = {}
        for k, name in enumerate(self._p_layers()):
            _v, idx = torch.max(self.p[k], dim=0)
            c = self.length[name]
            model_to_train[name], best_archi[name] = [], []
            if idx == c:
                best_archi[name].append(c)
                model_to_train[name].append(c)
            else:
                best_archi[name].append(idx)
                del self._units(name)[c]
            self.length[name] = len(self._units(name))
        for name in ("last_3_3d", "last_6_3d", "last_12_3d"):
            model_to_train[name], best_archi[name] = [t], [t]
        self.model_to_train = model_to_train
        return best_archi
