"""Randomised shapes through the fused executor's round-5 switches (G4 layout, fused stems, stem3d0's tail in the product's idle rows,
mixed bf16 storage): with the switches off the forward is rounds 1-4's; with them on it must give the same bits (G4, fused stems) or
the same disparities to the documented class (tail rows).  Not part of the test suite: python tools/fuzz_fused.py [cases] [seed]"""
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rag_amd as ra  # noqa: E402
from oracle import matching_oracle as O  # noqa: E402

DEV = "cuda:0"
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
import numpy as np  # noqa: E402
GENOTYPES = (("all-conv", O.ALL_CONV), ("all-conv", O.ALL_CONV), ("all-skip", O.ALL_SKIP),
             ("mixed", np.array([[0, 1], [1, 0], [3, 0], [2, 1], [8, 1], [6, 0]])))      # (SURVEY 8 A6's unsorted probe rows)
bad = 0
for case in range(n_cases):
    gname, rows = rng.choice(GENOTYPES)
    B = rng.choice((1, 1, 2, 3))
    h = 4 * rng.randint(10, 40)
    w = 4 * rng.randint(14, 60)
    maxdisp = 12 * rng.randint(3, 12)
    dt = rng.choice((torch.float32, torch.bfloat16))
    sd = O.random_matching_state_dict(rows, seed=case)
    net = ra.MatchingNet(ra.Genotype(rows, None, rows, None), maxdisp=maxdisp)
    net.load_state_dict(sd, strict=True)
    net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(1000 + case)
    lf, rf = (torch.randn((B, 12, h, w), generator=g).to(DEV).to(dt) for _ in range(2))
    outs = {}
    try:
        for name, g4, fuse, trows in (("off", False, False, False), ("g4", True, False, False), ("g4+fused", True, True, False), ("all", True, True, True)):
            ra.ops.set_g4(g4); ra.ops.set_stem_fusion(fuse); ra.ops.set_stem_tail_rows(trows)
            with torch.no_grad():
                outs[name] = net(lf, rf)
            plan = dict(net.last_g4_plan)
        with torch.no_grad():
            rep = net(lf, rf)
    finally:
        ra.ops.set_g4(True); ra.ops.set_stem_fusion(True); ra.ops.set_stem_tail_rows(True)
    ok_bits = torch.equal(outs["g4"], outs["off"]) and torch.equal(outs["g4+fused"], outs["off"]) and torch.equal(rep, outs["all"])
    e = O.epe(outs["all"].float().cpu(), outs["off"].float().cpu())
    fin = bool(torch.isfinite(outs["all"]).all())
    if dt == torch.float32:
        ok_e = e < 5e-4
    else:
        # bf16 storage: the rows tail is formed from the ROUNDED activations — one more 8-bit rounding on one of cell 0's inputs; judge it
        # against the fp32 build, as the bf16 gates do
        with torch.no_grad():
            d32 = net(lf.float(), rf.float()).cpu()
        e_all, e_off = O.epe(outs["all"].float().cpu(), d32), O.epe(outs["off"].float().cpu(), d32)
        ok_e = e_all <= 1.5 * e_off + 5e-3
        e = e_all - e_off
    status = "ok" if (ok_bits and fin and ok_e) else "FAIL"
    bad += status != "ok"
    print(f"case {case:2d}: {gname:8s} B={B} {h}x{w} D={maxdisp // 3} {str(dt).split('.')[-1]:8s} fused={plan.get('stems_fused')} rows={plan.get('stem_tail_rows')} "
          f"g4={[plan['pre'].get(j) for j in (0, 1, 2)]}: bits {'same' if ok_bits else 'DIFFER'}, rows EPE {e:+.2e} px ({'vs off' if dt == torch.float32 else 'change of the EPE vs the fp32 build'}) -> {status}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
