"""Times the plane-stationary split-operand convolution (conv3d_x3p.hip) on the level-3 shapes through the C ABI; with a
-DRAGMI_X3P_STAMPS build of the library also prints the in-kernel phase stamps (cycles per step of wave 1 of every workgroup)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
dev = 'cuda:0'
dbg = torch.zeros(8, dtype=torch.int64, device=dev)
os.environ["RAGMI_X3P_DBG"] = hex(dbg.data_ptr())
import rag_amd
from rag_amd import ops
B, D, H, W = 1, 64, 128, 416
torch.manual_seed(0)
x = torch.randn((B, 8, D, H, W), device=dev)
wa = torch.randn((12, 4, 3, 3, 3), device=dev) * 0.1
wb = torch.randn((12, 4, 3, 3, 3), device=dev) * 0.1
pa, pb = ops.conv3d_k3_pack(wa), ops.conv3d_k3_pack(wb)
sc = torch.ones(12, device=dev); sh = torch.zeros(12, device=dev)
out = torch.empty((B, 12, D, H, W), device=dev)
x12 = torch.randn((B, 12, D, H, W), device=dev); w12 = torch.randn((12, 12, 3, 3, 3), device=dev) * 0.1; p12 = ops.conv3d_k3_pack(w12)
def dual(): ops.conv3d_k3_dual(x, 4, pa, sc, sh, pb, sc, sh, 12, True, out)
def stem(): ops.conv3d_k3(x12, p12, 12, sc, sh, True, out)
for name, fn in (("dual", dual), ("stem1", stem)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dbg.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print(name, os.environ.get("RAGMI_X3P_DIAG", "0"), f"{e0.elapsed_time(e1)/20*1e3:.1f} us")
    t = dbg.cpu().tolist()
    if sum(t):
        nwg = 256; steps = 20 * 72 * 832 / 256 / 3.25 if False else None
        tot = sum(t[:4])
        print("   stamps (share of wave time): barrier wait %.1f%%, reads+verdict+overflow check %.1f%%, first half (dz2 + ninth tap + staging) %.1f%%, second half (dz1, dz0 + epilogue) %.1f%%; total %.3g cycles over all workgroups"
              % (100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot, tot))
