"""EPE (vs the fp32 build, full configs[2] size, two cost scales) of the bf16-storage modes: python tools/bf16_modes.py"""
import sys, torch, time
sys.path.insert(0, '.')
import rag_amd as ra
from oracle import matching_oracle as O
DEV = 'cuda:0'; BF = torch.bfloat16
rows = O.ALL_CONV
for f in (1.0, 1e-3):
    sd = O.random_matching_state_dict(rows, seed=0)
    sd["last_3_3d.0.conv.weight"] = sd["last_3_3d.0.conv.weight"] * f
    net = ra.MatchingNet(ra.ALL_CONV_GENOTYPE, maxdisp=192); net.load_state_dict(sd); net = net.to(DEV).eval()
    g = torch.Generator().manual_seed(1234)
    lf, rf = torch.randn((1, 12, 128, 416), generator=g).to(DEV), torch.randn((1, 12, 128, 416), generator=g).to(DEV)
    with torch.no_grad():
        d32 = net(lf, rf).cpu()
        for name, deep, head in (("mixed (deep fp32)", True, True), ("all cells bf16, head fp32", False, True), ("everything bf16", False, False)):
            ra.ops.set_bf16_deep_fp32(deep); ra.ops.set_bf16_head_fp32(head)
            d16 = net(lf.to(BF), rf.to(BF)).cpu()
            print(f"last_3 x {f:g}: {name:28s} EPE vs the fp32 build {O.epe(d16, d32):.4e} px")
