"""In-kernel stamps of the level-3 z-marching launches: where a wave's cycles go inside one plane step, and the
clock the chip actually holds under this kernel.

    bash tools/build_variant.sh diag rag_amd/csrc/conv3d_x3.hip "-DRAGMI_DIAG"
    RAG_AMD_LIB=rag_amd/lib/librag_amd_diag.so RAGMI_X3_DIAG=32 python tools/x3_stamps.py [dual|stem1] > gpurun_out/x3_stamps.txt

The profiling build's kernel sums, per wave, the s_memtime cycles between the phase edges of every plane step
(conv3d_x3.hip: X3_STAMP) and stamps s_memtime / s_memrealtime (100 MHz) at its first and last instruction; the launch measured is
the last one of >= 2 s of back-to-back launches on random data (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "dual"
dev = "cuda:0"
ops = rag_amd.ops
lib = rag_amd.load_library()
g = torch.Generator().manual_seed(1)
shape = (1, 64, 128, 416)
r = lambda *s: (torch.randn(s, generator=g) * 0.1).to(dev)  # noqa: E731
tails_out = torch.empty((1, 8) + shape[1:], device=dev)


def tails():
    return [ops.Tail(r(4, 12), r(4).abs() + 0.5, r(4), True, tails_out, 4 * k) for k in range(2)]


WORDS = 16
buf = torch.zeros((4096 * 8 * WORDS,), dtype=torch.int64, device=dev)
assert lib.ragmi_diag_x3_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
if hasattr(lib, "ragmi_diag_x3q_stamp_buffer"):
    assert lib.ragmi_diag_x3q_stamp_buffer(ctypes.c_void_p(buf.data_ptr())) == 0
y = torch.empty((1, 12) + shape[1:], device=dev)
with ops.conv_precision("f16x3"):
    if which == "stem1":
        x = torch.randn((1, 12) + shape[1:], generator=g).to(dev)
        pk = ops.conv3d_k3_pack(r(12, 12, 3, 3, 3))
        sc, sh, tl = r(12).abs() + 0.5, r(12), tails()
        run = lambda: ops.conv3d_k3(x, pk, 12, sc, sh, True, y, None, tails=tl, store_main=False)  # noqa: E731
    else:
        x8 = torch.randn((1, 8) + shape[1:], generator=g).to(dev)
        pa, pb = ops.conv3d_k3_pack(r(12, 4, 3, 3, 3)), ops.conv3d_k3_pack(r(12, 4, 3, 3, 3))
        sa, ha, sb, hb, tl = r(12).abs() + 0.5, r(12), r(12).abs() + 0.5, r(12), tails()
        run = lambda: ops.conv3d_k3_dual(x8, 4, pa, sa, ha, pb, sb, hb, 12, True, y, tails=tl)  # noqa: E731
    run()
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.5:
        for _ in range(50):
            run()
        torch.cuda.synchronize()
        n += 50
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    buf.zero_()
    torch.cuda.synchronize()
    for _ in range(20):
        run()
    ev0.record()
    run()
    ev1.record()
    torch.cuda.synchronize()
us = ev0.elapsed_time(ev1) * 1e3
d = buf.cpu().numpy().reshape(-1, WORDS)
d = d[d[:, 7] > 0]
steps = d[:, 7].astype(np.float64)
life = (d[:, 9] - d[:, 8]).astype(np.float64)
real = (d[:, 11] - d[:, 10]).astype(np.float64)
clock = life / real * 100.0          # MHz
names = ["barrier 1 (wait)", "note_overflow + commit (split, LDS writes)", "barrier 2 (wait)", "restart check + prefetch issue",
         "MFMA block (LDS reads, 48-66 + MFMAs)", "epilogue (BN, ReLU, stores, tails)", "outside the plane steps (ring start, decode)"]
print(f"launch: {which}, {len(d)} waves stamped, {n + 21} launches in front, last launch {us:.1f} us (HIP events, stamped build)")
print(f"clock held (d s_memtime / d s_memrealtime x 100 MHz): median {np.median(clock):.0f} MHz, p10 {np.percentile(clock, 10):.0f}, p90 {np.percentile(clock, 90):.0f}")
print(f"wave lifetime: median {np.median(life):.0f} cycles = {np.median(real) / 100:.1f} us; plane steps per wave: median {np.median(steps):.0f} (min {steps.min():.0f}, max {steps.max():.0f})")
tot = d[:, :7].sum(axis=1).astype(np.float64)
print(f"{'phase':52s} {'cycles/step (median over waves)':>32s} {'share of wave lifetime':>24s}")
for k, nm in enumerate(names):
    per = d[:, k] / steps
    print(f"{nm:52s} {np.median(per):32.0f} {100 * np.median(d[:, k] / life):23.1f}%")
print(f"{'sum of stamped phases / lifetime':52s} {'':32s} {100 * np.median(tot / life):23.1f}%")
print(f"cycles per plane step incl. its share of the start-up: {np.median(life / steps):.0f}; in-loop only: {np.median(d[:, :6].sum(axis=1) / steps):.0f}")
for w in range(8):
    sel = d[w::8]
    print(f"  wave {w}: barrier1 {np.median(sel[:, 0] / sel[:, 7]):6.0f} commit {np.median(sel[:, 1] / sel[:, 7]):6.0f} barrier2 {np.median(sel[:, 2] / sel[:, 7]):6.0f} "
          f"prefetch {np.median(sel[:, 3] / sel[:, 7]):6.0f} mfma {np.median(sel[:, 4] / sel[:, 7]):6.0f} epilogue {np.median(sel[:, 5] / sel[:, 7]):6.0f}")
