"""Checkpoint / task-architecture round trip and a multi-task inference harness.

The reference saves `{'task', 'model': state_dict, 'optimizer'}` per task (run.py:194-196) but never persists the
per-task architectures `Appr.archis` (rag.py:93-97, 229) nor the genotype each grown unit was built from
(rag_model.py:391-522), so a grown model cannot be reloaded or served.  Here:

  * `save_checkpoint` writes the reference's three keys unchanged (the reference can still read the file) plus
    `archis`, `genotypes` (rows per grown unit) and `maxdisp`;
  * `load_checkpoint` rebuilds the grown `nn.ModuleList`s from the state_dict key names (unit counts per layer, head
    count per task), checks every cell's conv/identity pattern against its genotype, loads strictly, and returns
    `(Network, archis)`; a reference checkpoint loads too if the caller supplies the genotypes / archis it lacks;
  * `MultiTaskStereo` serves `forward(left, right, t)` with `archis[t]` (the task index comes from the caller: no
    Scene Router).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
import torch.nn as nn

from .modules import _ConvBR, Genotype
from .network import Network

_HEADS = ("last_3_3d", "last_6_3d", "last_12_3d")


def _rows(genotype) -> Dict[str, list]:
    return {"normal": np.asarray(genotype.normal).tolist(), "reduce": np.asarray(genotype.reduce).tolist()}


def _genotype(rows) -> Genotype:
    if isinstance(rows, Genotype) or hasattr(rows, "reduce"):
        return rows
    return Genotype(normal=np.asarray(rows["normal"]), normal_concat=None, reduce=np.asarray(rows["reduce"]), reduce_concat=None)


def unit_genotypes(net: Network) -> Dict[str, List[dict]]:
    """genotype rows of every cell unit, keyed by layer name ('cell_2d0'.. 'cell_3d7'), unit order = ModuleList order."""
    out = {}
    for name in net._p_layers():
        if name.startswith("cell_"):
            out[name] = [_rows(u.genotype) for u in net._units(name)]
    return out


def save_checkpoint(path, net: Network, archis: Sequence[dict], task: int, optimizer=None) -> dict:
    """run.py:194-196's file plus what is needed to rebuild the grown model."""
    data = {"task": int(task), "model": net.state_dict(), "optimizer": optimizer.state_dict() if optimizer is not None else None,
            "archis": [{k: [int(v) for v in vs] for k, vs in a.items()} for a in archis],
            "genotypes": unit_genotypes(net), "maxdisp": int(net.maxdisp), "format": "rag_amd/1"}
    if path is not None:
        torch.save(data, path)
    return data


def _unit_counts(keys) -> Dict[str, int]:
    """ModuleList lengths from key names: 'stem2d0.<k>.', 'cells_3d.<i>.<k>.', 'last_3_3d.<t>.' ..."""
    counts: Dict[str, int] = {}
    attr_to_layer = {"stem2d0": "stem_2d0", "stem2d1": "stem_2d1", "stem2d2": "stem_2d2", "last_3_2d": "last_3_2d",
                     "stem3d0": "stem_3d0", "stem3d1": "stem_3d1", "last_3_3d": "last_3_3d", "last_6_3d": "last_6_3d",
                     "last_12_3d": "last_12_3d"}
    for k in keys:
        m = re.match(r"(cells_2d|cells_3d)\.(\d+)\.(\d+)\.", k)
        if m:
            name = ("cell_2d" if m.group(1) == "cells_2d" else "cell_3d") + m.group(2)
            counts[name] = max(counts.get(name, 0), int(m.group(3)) + 1)
            continue
        m = re.match(r"(\w+)\.(\d+)\.", k)
        if m and m.group(1) in attr_to_layer:
            name = attr_to_layer[m.group(1)]
            counts[name] = max(counts.get(name, 0), int(m.group(2)) + 1)
    return counts


def _check_cell_pattern(unit, prefix: str, keys: set, name: str, idx: int) -> None:
    """The state_dict tells which positional ops are convolutions; it must agree with the genotype the unit was built from."""
    for n, op in enumerate(unit._ops):
        has = f"{prefix}_ops.{n}.conv.weight" in keys
        if has != isinstance(op, _ConvBR):
            raise ValueError(f"load_checkpoint: {name} unit {idx}: op {n} is {'a conv' if has else 'an identity'} in the checkpoint "
                             f"but {'an identity' if has else 'a conv'} in the supplied genotype — wrong genotype for this unit")


def load_checkpoint(src: Union[str, dict], device="cuda", genotypes=None, archis: Optional[Sequence[dict]] = None,
                    allow_pickle: bool = False):
    """-> (Network in eval mode on `device`, archis).  `genotypes`: a Genotype for every unit, or {layer: [Genotype/rows per
    unit]}; needed only for reference checkpoints (ours carry them).  `archis`: likewise.
    The file is read with `weights_only=True` (tensors, ints, lists and dicts — everything save_checkpoint and the reference's
    run.py:194-196 write, torch.optim.SGD state included); `allow_pickle=True` opts into full unpickling for a legacy file
    that holds arbitrary objects — it executes code from the file, so only for files you trust."""
    data = torch.load(src, map_location="cpu", weights_only=not allow_pickle) if not isinstance(src, dict) else src
    sd = data["model"]
    keys = set(sd.keys())
    counts = _unit_counts(keys)
    stored = data.get("genotypes")
    if genotypes is None and stored is None:
        raise ValueError("load_checkpoint: this checkpoint has no genotypes (a reference run.py checkpoint); pass genotypes=")

    def geno(name: str, idx: int) -> Genotype:
        src_g = genotypes if genotypes is not None else stored
        if isinstance(src_g, dict):
            return _genotype(src_g[name][idx])
        return _genotype(src_g)

    first = geno("cell_3d0", 0)
    net = Network(first, "cpu", maxdisp=int(data.get("maxdisp", 192)))
    for name in net._p_layers() + list(_HEADS):
        units = net._units(name)
        want = counts.get(name, 1)
        if name.startswith("cell_"):
            units[0] = net._new_unit(name, geno(name, 0))
        for idx in range(1, want):
            units.append(net._new_unit(name, geno(name, idx) if name.startswith("cell_") else first))
        if name not in _HEADS:
            net.length[name] = want
    for name in net._p_layers():
        if name.startswith("cell_"):
            attr = "cells_2d" if name.startswith("cell_2d") else "cells_3d"
            for idx, unit in enumerate(net._units(name)):
                _check_cell_pattern(unit, f"{attr}.{name[7:]}.{idx}.", keys, name, idx)
    net.load_state_dict(sd, strict=True)
    out_archis = archis if archis is not None else data.get("archis")
    if out_archis is None:
        out_archis = [net.arch_init]
    for t, a in enumerate(out_archis):
        for name, (k, *_rest) in a.items():
            if int(k) >= len(net._units(name)):
                raise ValueError(f"load_checkpoint: archis[{t}][{name!r}] = {k} but the checkpoint has {len(net._units(name))} unit(s)")
    return net.to(device).eval(), [dict(a) for a in out_archis]


class MultiTaskStereo(nn.Module):
    """Serve a grown model: forward(left, right, t) = Network.forward(left, right, t, archis[t]) (rag.py:416)."""

    def __init__(self, net: Network, archis: Sequence[dict]):
        super().__init__()
        self.net = net
        self.archis = [dict(a) for a in archis]

    @property
    def n_tasks(self) -> int:
        return len(self.archis)

    def forward(self, left, right, t: int):
        if not 0 <= t < len(self.archis):
            raise IndexError(f"MultiTaskStereo: task {t} not in [0, {len(self.archis)})")
        return self.net(left, right, t, self.archis[t])
