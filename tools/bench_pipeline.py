"""End-to-end throughput with the Feature Net of pair n+1 overlapped with the Matching Net of pair n (two HIP streams inside one
captured graph), against the single-stream end-to-end pass and the Matching-Net-only pass.
    python tools/bench_pipeline.py"""
import os
import sys
import time
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rag_amd  # noqa: E402

dev = torch.device("cuda:0")
H, W, MAXDISP = 384, 1248, 192
torch.manual_seed(0)
net = rag_amd.Network(rag_amd.ALL_CONV_GENOTYPE, dev, maxdisp=MAXDISP).to(dev).eval()
g = torch.Generator().manual_seed(1234)
left = torch.randn((1, 3, H, W), generator=g).to(dev)
right = torch.randn((1, 3, H, W), generator=g).to(dev)


def features():
    return net._features(left, right, lambda x: net.feature(x, net.arch_init, None))


def matching(lf, rf):
    return net.disp(net.matching(None, net.arch_init, None, features=(lf, rf)))


def timed(graph, n=30):
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        graph.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


with torch.no_grad():
    lf, rf = features()
    out = matching(lf, rf)
    torch.cuda.synchronize()
    ref = out.clone()
    # (a) single stream: features then matching
    ga = torch.cuda.CUDAGraph()
    with torch.cuda.graph(ga):
        a_lf, a_rf = features()
        a_out = matching(a_lf, a_rf)
    # (b) matching only
    gb = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gb):
        b_out = matching(lf, rf)
    # (c) pipelined: matching(pair n: features held from the previous replay) || features(pair n+1) on a side stream
    hold_l, hold_r = lf.clone(), rf.clone()
    side = torch.cuda.Stream()
    gc = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gc):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            n_lf, n_rf = features()
        c_out = matching(hold_l, hold_r)
        main.wait_stream(side)
        hold_l.copy_(n_lf); hold_r.copy_(n_rf)          # hand the next pair's features over (kernel copies, 2 x 2.5 MB)
    ta, tb, tc = timed(ga), timed(gb), timed(gc)
    torch.cuda.synchronize()
    print(f"single stream end to end {ta:.4f} ms ({1e3 / ta:.1f} maps/s); matching only {tb:.4f} ms ({1e3 / tb:.1f}); "
          f"pipelined (features of pair n+1 beside matching of pair n) {tc:.4f} ms per pair ({1e3 / tc:.1f} maps/s)")
    print("outputs equal:", bool(torch.equal(a_out, ref)), bool(torch.equal(c_out, ref)))
