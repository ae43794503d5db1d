"""Timing aid for the paired 1x1x1 + resample launches of the level-6 / level-12 cells (RAGMI_K1R_NCO forces the slab width)."""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = "cuda:0"
ops = rag_amd.ops
cases = {"c4": ((12, (64, 128, 416)), (24, (32, 64, 208)), 16, (16, 32, 104)),
         "c6": ((48, (16, 32, 104)), (24, (32, 64, 208)), 16, (16, 32, 104)),
         "c3": ((12, (64, 128, 416)), (12, (64, 128, 416)), 8, (32, 64, 208))}
for name, (a, b, cout, size) in cases.items():
    g = torch.Generator().manual_seed(1)
    specs = []
    for k, (cin, shp) in enumerate((a, b)):
        x = torch.randn((1, cin) + shp, generator=g).to(dev)
        w = (torch.randn((cout, cin), generator=g) * 0.1).to(dev)
        sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
        specs.append((x, w, sc, sh, True, k * cout))
    out = torch.empty((1, 2 * cout) + size, device=dev)
    for _ in range(3):
        ops.conv3d_k1_resample_pair(specs, size, out)
    torch.cuda.synchronize()
    s = torch.cuda.Stream()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        with torch.cuda.graph(gr, capture_error_mode="thread_local"):
            for _ in range(20):
                ops.conv3d_k1_resample_pair(specs, size, out)
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gr.replay()
        e1.record()
        torch.cuda.synchronize()
    print(f"{name} NCO={os.environ.get('RAGMI_K1R_NCO', 'auto')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
