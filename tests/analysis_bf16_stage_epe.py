#!/usr/bin/env python3
"""Where does the bf16-storage error of BASELINE configs[2] come from?  CPU-only analysis on the oracle (test infrastructure):
the fp32 forward is re-run with the tensors the HIP executor STORES rounded to bf16, one stage group at a time.

Rounding points of the fused executor (rag_amd/modules.py): stem outputs, each cell's s0|s1 (`pre`) and its output (`cat`: the
sums are rounded once), the head's 1x1x1 / upsampled tensors, and `mat` (the [B,1,d,h,w] cost the soft-argmin reads).

usage: python tests/analysis_bf16_stage_epe.py [H W maxdisp]      (defaults 192 384 96; not collected by pytest)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import matching_oracle as O  # noqa: E402


def bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def run(lf, rf, sd, rows, maxdisp, stages):
    """stages: set of {'input', 'stems', 'l3', 'l6', 'l12', 'head', 'mat'} whose stored tensors are rounded to bf16."""
    level = {0: "l3", 1: "l3", 2: "l3", 3: "l6", 4: "l12", 5: "l6", 6: "l12", 7: "l12"}
    orig_conv, orig_cell, orig_interp = O.conv_br_3d, O.cell_3d, F.interpolate

    def conv(x, sd_, prefix, **kw):
        y = orig_conv(x, sd_, prefix, **kw)
        grp = ("stems" if prefix.startswith("stem3d") else
               "mat" if prefix.startswith("last_3_3d") else
               "head" if prefix.startswith("last_") else
               level[int(prefix.split(".")[1])] if ("preprocess" in prefix) else None)     # _ops outputs live in registers
        return bf(y) if grp in stages else y

    def cell(pp, p, sd_, prefix, rows_, fm, du, training=False):
        prev, cat = orig_cell(pp, p, sd_, prefix, rows_, fm, du, training)
        return prev, (bf(cat) if level[int(prefix.split(".")[1])] in stages else cat)

    in_head = {"on": False}

    def interp(x, size, mode="trilinear", align_corners=None):
        y = orig_interp(x, size, mode=mode, align_corners=align_corners)
        return bf(y) if ("head" in stages and in_head["on"] and x.shape[1] > 1) else y

    O.conv_br_3d, O.cell_3d = conv, cell
    try:
        if "input" in stages:
            lf, rf = bf(lf), bf(rf)
        cost = O.cost_volume(lf, rf, maxdisp)
        in_head["on"] = True
        F.interpolate = interp
        mat = O.matching(cost, sd, rows, None, False)
        F.interpolate = orig_interp
        return O.disp_head(mat, maxdisp), mat
    finally:
        O.conv_br_3d, O.cell_3d, F.interpolate = orig_conv, orig_cell, orig_interp


def main():
    H, W, maxdisp = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (192, 384, 96)
    rows = O.ALL_CONV
    torch.set_num_threads(8)
    for seed, f in ((0, 1.0), (4, 1.0), (0, 1e-3)):       # (f: last_3_3d's weight scaled — |cost| ~ 10-100, a trained net's cost scale)
        sd = O.random_matching_state_dict(rows, seed=seed)
        sd["last_3_3d.0.conv.weight"] = sd["last_3_3d.0.conv.weight"] * f
        g = torch.Generator().manual_seed(1234)
        lf, rf = torch.randn((1, 12, H // 3, W // 3), generator=g), torch.randn((1, 12, H // 3, W // 3), generator=g)
        ref, mat = run(lf, rf, sd, rows, maxdisp, set())
        print(f"seed {seed}, last_3_3d weight x {f:g}: {H}x{W} D={maxdisp}; |mat| max {float(mat.abs().max()):.3g}, std {float(mat.std()):.3g}")
        everything = {"input", "stems", "l3", "l6", "l12", "head", "mat"}
        for name, st in [("all stored tensors bf16", everything), ("all but mat", everything - {"mat"}),
                         ("all but mat and head", everything - {"mat", "head"}),
                         ("only mat", {"mat"}), ("only head", {"head"}), ("only input", {"input"}), ("only stems", {"stems"}),
                         ("only level-3 cells", {"l3"}), ("only level-6 cells", {"l6"}), ("only level-12 cells", {"l12"}),
                         ("MIXED: input + stems + level 3", {"input", "stems", "l3"}),
                         ("MIXED: ... + level 6", {"input", "stems", "l3", "l6"})]:
            out, _ = run(lf, rf, sd, rows, maxdisp, st)
            print(f"   {name:28s} EPE vs fp32 = {O.epe(out, ref):.4e} px")


if __name__ == "__main__":
    main()
