#!/usr/bin/env python3
"""How accurate must the split-operand convolution be for the EPE gate?  CPU-only analysis on the oracle (test infrastructure; not
collected by pytest).  Every 3x3x3 convolution the HIP executor runs on the 16-bit matrix cores (stem3d1 and all cell ops; stem3d0
and last_3_3d stay fp32 there) is replaced by an emulation of  hi*hi + hi*lo + lo*hi  with the operands split

    bf16x3: hi = bf16(x), lo = bf16(x - hi)                       (RAGMI_F32X3 today: ~2^-17 per product)
    fp16x3: hi = fp16(x * 2^-s), lo = fp16(x * 2^-s - hi)         (11-bit parts: ~2^-23 per product; s from the tensor's absmax)

accumulated exactly (fp64), so only the split error is measured.  Reported: EPE of each variant and of the plain fp32 oracle
against the fp64 evaluation of the same network, over weight seeds.

usage: python tests/analysis_split_precision.py [H W maxdisp]   (default 192 384 96)"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import matching_oracle as O  # noqa: E402


def split_bf16(t):
    hi = t.to(torch.bfloat16).to(torch.float32)
    lo = (t - hi).to(torch.bfloat16).to(torch.float32)
    return hi, lo, 1.0


def split_fp16(t, target=2.0 ** 14):
    amax = float(t.abs().max())
    s = 1.0 if amax == 0 else 2.0 ** torch.floor(torch.log2(torch.tensor(target / amax))).item()
    ts = t * s
    hi = ts.to(torch.float16).to(torch.float32)
    lo = (ts - hi).to(torch.float16).to(torch.float32)
    assert torch.isfinite(hi).all()
    return hi, lo, s


def emulated(mode):
    split = {"bf16x3": split_bf16, "fp16x3": split_fp16}[mode]
    orig = F.conv3d

    def conv3d(x, w, bias=None, stride=1, padding=0, *a, **kw):
        if w.shape[2:] != (3, 3, 3) or x.dtype != torch.float32 or w.shape[0] == 1 or w.shape[1] == 24:
            return orig(x, w, bias, stride, padding, *a, **kw)          # 1x1x1, stem3d0 (24 in) and last_3_3d (1 out): fp32
        xh, xl, sx = split(x)
        wh, wl, sw = split(w)
        d = torch.float64
        y = orig(xh.to(d), wh.to(d), None, stride, padding) + orig(xh.to(d), wl.to(d), None, stride, padding) + \
            orig(xl.to(d), wh.to(d), None, stride, padding)
        return (y / (sx * sw)).to(torch.float32)
    return conv3d


def main():
    H, W, maxdisp = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (192, 384, 96)
    torch.set_num_threads(8)
    for name, rows in (("all-conv", O.ALL_CONV), ("all-skip", O.ALL_SKIP)):
        for seed in (0, 1, 2):
            sd = O.random_matching_state_dict(rows, seed=seed)
            g = torch.Generator().manual_seed(1234 + seed)
            lf, rf = torch.randn((1, 12, H // 3, W // 3), generator=g), torch.randn((1, 12, H // 3, W // 3), generator=g)
            ref64 = O.matching_net_forward(lf.double(), rf.double(), {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()},
                                           rows, maxdisp)
            out32 = O.matching_net_forward(lf, rf, sd, rows, maxdisp)
            res = {"fp32 oracle": O.epe(out32, ref64)}
            orig = F.conv3d
            for mode in ("bf16x3", "fp16x3"):
                F.conv3d = emulated(mode)
                try:
                    out = O.matching_net_forward(lf, rf, sd, rows, maxdisp)
                finally:
                    F.conv3d = orig
                res[mode] = O.epe(out, ref64)
                res[mode + " vs fp32 oracle"] = O.epe(out, out32)
            print(f"{name} seed {seed} {H}x{W} D={maxdisp}: " + "; ".join(f"{k} {v:.3e}" for k, v in res.items()), flush=True)


if __name__ == "__main__":
    main()
