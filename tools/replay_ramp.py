"""Per-replay duration of the headline hipGraph right after a device synchronize (is the first replay of a timed region slower?).
    python tools/replay_ramp.py"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import rag_amd  # noqa: E402

dev = torch.device("cuda:0")
net = bench.build_net(dev)
g = torch.Generator().manual_seed(1234)
lf = torch.randn((1, bench.FEA_C, bench.H // 3, bench.W // 3), generator=g).to(dev)
rf = torch.randn((1, bench.FEA_C, bench.H // 3, bench.W // 3), generator=g).to(dev)


def step():
    with torch.no_grad():
        return net(lf, rf)


for _ in range(3):
    step()
torch.cuda.synchronize()
graph, _ = bench.try_capture(step)
for idle_ms in (0, 5, 50):
    torch.cuda.synchronize()
    time.sleep(idle_ms * 1e-3)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(30):
        graph.replay()
        ev[i + 1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    d = [ev[i].elapsed_time(ev[i + 1]) for i in range(30)]
    print(f"idle {idle_ms} ms: enqueue {t_enq * 1e3:.2f} ms, wall {wall * 1e3:.2f} ms for 30 replays ({wall / 30 * 1e3:.4f} ms/step); per replay (ms): "
          + " ".join(f"{x:.3f}" for x in d[:8]) + " ... " + " ".join(f"{x:.3f}" for x in d[-3:]))
