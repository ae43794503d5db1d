#!/usr/bin/env python3
"""bench.py — disparity maps/sec of the Matching-Net forward (left_fea, right_fea) -> disp.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 either launched by
torch.distributed.run with one rank per GPU, or — WORLD_SIZE unset — this script starts its own N ranks (launch_ranks: N child
processes, started before anything touches the GPU; rank 0's line is relayed).  A "step" is one pass of the hot path
(cost volume -> 3-D conv aggregation -> soft-argmin) over one batch of synthetic stereo
features already resident in HBM.  Workload at every N: BASELINE.json configs[1] per GPU
(B=1, 384x1248, D=192, fp32, all-conv genotype, seeded weights with randomised BN —
SURVEY.md §8(d)); pairs are independent, so ranks shard the batch with NO data-path
collective (weak scaling).  Rank 0 prints ONE JSON line on stdout.

Extra objects in that line:
  roofline      the dominant kernel of the step (largest share of the per-kernel HIP-event time; today the split-operand (RAGMI_F32X3)
                convolution of the level-3 cells, conv3d_x3.hip).  `achieved` = its ALGORITHMIC bytes (input once +
                output once) or flops per launch / its mean launch duration, priced against the LARGER
                of its two floors (HBM 8 TB/s, or the dense MFMA peak of the form it issues).  The timed region replays ONE
                hipGraph, inside which single kernels cannot be bracketed, so the per-kernel durations come from an eager pass
                of the same kernels on the same stream right after the timed region (`avg_launch_us_source`); rocprofv3's
                average for the same kernel agrees (profiles/).  `traffic` = HBM bytes per launch from the committed PMC
                summary named in `traffic_source` (separate --pmc passes of this command; not measured in this run).
  cpu_baseline  the CPU oracle (a port of the reference's ATen op sequence) timed on the host cores of this box on ONE
                pair of the same workload (rank 0, N=1 only): median of 3 runs, thread count = min(16, os.cpu_count())
                (the 1-GPU box's CPU share; stated in `cores`).
  strict_fp32   (fp32 runs, N=1) the same workload with every contraction on the fp32-input MFMA forms
                (ops.set_conv_precision("fp32")): value_fp32_mfma and its EPE, so the record carries both arithmetic contracts.
  epe_bf16_vs_fp32  (bf16 runs) EPE of the bf16-storage output against the fp32 build on the same pair.
  configs       (default fp32 B=1 run on one GPU) the OTHER BASELINE.json configurations, each a few untimed-by-the-headline graph
                replays after the timed region, so that the driver's one line observes every configuration:
                config2_bf16_b8 (configs[2]: B=8 bf16 storage; maps/s, EPE vs the fp32 build and vs the CPU oracle on pair 0),
                config3_480x960_b8 (configs[3]: one GPU's shard of the 64-pair DrivingStereo batch; maps/s),
                config4_train (configs[4]: training step, B=4 at 192x384; ms/step, pairs/s, conv_precision, roofline of its
                dominant kernel, cpu_baseline = the oracle's forward + backward of ONE pair timed once),
                all_skip (SURVEY 8(d)'s lower bound: the all-skip genotype at the headline size; maps/s + EPE vs the oracle).
  library       the shared object the product path loaded (RAG_AMD_LIB can redirect it: A/B tooling), graph_nodes: node census
                of the timed hipGraph (a graph holding memcpy / memset nodes is not replayed: DESIGN.md 4.4).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, MAXDISP, FEA_C = 384, 1248, 192, 12
PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md, BF16 dense
PEAK_HBM_GBS = 8000.0


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_net(device, seed=0, genotype=None):
    """Benchmark protocol of SURVEY.md §8(d): reference init (Kaiming fan_out, done by the module
    ctor), then BN gamma~U(.5,1.5), beta~N(0,.1), running_mean~N(0,.1), running_var~U(.5,1.5)."""
    import rag_amd
    torch.manual_seed(seed)
    net = rag_amd.MatchingNet(genotype if genotype is not None else rag_amd.ALL_CONV_GENOTYPE, maxdisp=MAXDISP)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm3d):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    return net.to(device).eval()


class K3Profiler:
    """Brackets every ragmi_conv3d_k3_fwd call with HIP events on the launch stream and
    attributes it to the kernel instantiation the library reports (ragmi_conv3d_k3_plan)."""

    def __init__(self, ops):
        self.ops, self.records, self.enabled = ops, [], False
        self._orig = ops.conv3d_k3

        def wrapped(x, packed, cout, *a, **kw):
            if not self.enabled:
                return self._orig(x, packed, cout, *a, **kw)
            B, Cin, D, Hh, Ww = x.shape
            log_tx, rows, groups = ops.conv3d_k3_plan(cout, B, D, Hh, Ww)
            res = a[5] if len(a) > 5 else kw.get("res")
            if ops.conv3d_k3_uses_x3(Cin, cout, B, D, Hh, Ww, 1, res is not None, len(kw.get("tails") or []), x.dtype):
                groups, log_tx, rows = [(Cin + 3) // 4], "x3", 0          # f16x3 kernel: conv3d_x3_kernel<channel groups, sets>
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._orig(x, packed, cout, *a, **kw)
            e1.record()
            self.records.append((e0, e1, (tuple(groups), log_tx, rows, 1), 2.0 * 27 * Cin * cout * B * D * Hh * Ww,
                                 4.0 * (Cin + cout) * B * D * Hh * Ww, len(groups)))
            return out

        ops.conv3d_k3 = wrapped
        self._orig_dual = ops.conv3d_k3_dual

        def wrapped_dual(x, cin_a, pa, sa, ha, pb, sb, hb, cout, *a, **kw):
            if not self.enabled:
                return self._orig_dual(x, cin_a, pa, sa, ha, pb, sb, hb, cout, *a, **kw)
            B, Cin, D, Hh, Ww = x.shape
            log_tx, rows, groups = ops.conv3d_k3_plan(cout, B, D, Hh, Ww, 2)
            res = a[3] if len(a) > 3 else kw.get("res")
            if ops.conv3d_k3_uses_x3(Cin, cout, B, D, Hh, Ww, 2, res is not None, len(kw.get("tails") or []), x.dtype):
                groups, log_tx, rows = [(Cin + 3) // 4], "x3", 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._orig_dual(x, cin_a, pa, sa, ha, pb, sb, hb, cout, *a, **kw)
            e1.record()
            self.records.append((e0, e1, (tuple(groups), log_tx, rows, 2), 2.0 * 27 * Cin * cout * B * D * Hh * Ww,
                                 4.0 * (Cin + cout) * B * D * Hh * Ww, len(groups)))
            return out

        ops.conv3d_k3_dual = wrapped_dual

    def summary(self):
        by = {}
        for e0, e1, key, flops, nbytes, nl in self.records:
            d = by.setdefault(key, [0.0, 0.0, 0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += flops
            d[2] += nbytes
            d[3] += nl
        return by


def roofline_of(by, args, dt, graphed, steps=None, f32_dtype_is_split=True):
    """`roofline` object of the dominant 3x3x3 convolution kernel of a profiled pass (K3Profiler.summary()): the kernel with the largest
    share of the per-kernel HIP-event time, priced against the LARGER of its two floors (HBM: algorithmic bytes / 8 TB/s; matrix
    cores: the MFMA flops it must issue / dense peak of the form it issues).  dt = seconds of the `steps` timed steps."""
    steps = steps or args.steps
    dom_key = max(by, key=lambda k: by[k][0]) if by else None
    if dom_key is None:
        return None
    secs, flops, nbytes, nlaunch = by[dom_key]
    groups, log_tx, rows, nset = dom_key
    ach = flops / secs * 1e-12
    x3 = log_tx == "x3"
    if x3 and groups[0] == 2 and nset == 2 and args.dtype != "bf16":
        kname = "conv3d_x3q_kernel<2,"       # the level-3 dual cells (conv3d_x3q.hip): every tail / layout instantiation
    else:
        kname = (f"conv3d_x3_kernel<{groups[0]}, {nset}>" if x3 else f"conv3d_k3_kernel<{groups[0]}, {log_tx}, {rows}, {nset}, 2, 0>")
    peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
    # which roof bounds this kernel: the larger of its two floors per launch — HBM: algorithmic bytes / 8 TB/s;
    # matrix cores: the MFMA flops it must ISSUE / dense peak (the split form issues 3 16-bit MFMAs per fp32
    # product, 2 with bf16 activation storage; row / K padding not counted)
    issue = (2.0 if args.dtype == "bf16" else 3.0) if x3 else 1.0
    t_hbm = (nbytes / nlaunch) / (PEAK_HBM_GBS * 1e9)
    t_mfma = issue * (flops / nlaunch) / (peak * 1e12)
    traffic, traffic_src = pmc_traffic_bytes(kname)
    common = {"traffic": traffic,
              "traffic_source": (f"{traffic_src} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, committed; "
                                 "not measured in this run)") if traffic_src else None,
              "algorithmic_bytes_per_launch": nbytes / nlaunch,
              "launches_per_step": nlaunch // steps, "avg_launch_us": round(secs / nlaunch * 1e6, 2),
              "avg_launch_us_source": ("HIP events around each launch in an eager pass of the same kernels on the same stream right "
                                       "after the timed hipGraph replays (a launch inside the replayed graph is 10-20 % shorter: "
                                       "profiles/r05*_warm_timeline.txt)" if graphed else
                                       "HIP events around each launch inside the timed region (eager launches)"),
              "flops_per_launch": flops / nlaunch, "share_of_step": round(secs / steps / (dt / steps), 3),
              "floor_us": {"hbm": round(t_hbm * 1e6, 1), "mfma": round(t_mfma * 1e6, 1)}}
    if t_hbm >= t_mfma:
        gbs = nbytes / secs * 1e-9
        roofline = {"kernel": kname + " *>" if kname.endswith(",") else kname, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), **common,
                    "mfma": {"achieved_tflops": round(ach, 2), "peak": peak, "frac": round(ach / peak, 4)}}
    else:
        roofline = {"kernel": kname + " *>" if kname.endswith(",") else kname, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), **common}
    if x3:
        roofline["note"] = ("fp32 convolution on the 16-bit matrix cores (scaled fp16 halves: hi*hi + hi*lo + lo*hi, fp32 accumulate). Its HBM "
                            "floor (input once + output once at 8 TB/s) is above its MFMA floor (3 16x16x32 MFMAs per product at "
                            "the 2.5 PFLOP/s dense peak), so HBM is the roof it is priced against; `mfma` = ALGORITHMIC fp32 "
                            f"flops / time against the bf16 dense peak ({ach / PEAK_FP32_MFMA_TFLOPS:.2f} of the fp32 matrix peak 157.3). "
                            "Neither roof binds it, nor does instruction issue: in-kernel stamps, the clock it holds (2.1 GHz) and "
                            "switch-off builds are in profiles/r05_x3_stamps.md (matrix pipe ~40 % busy; chains of short dependent phases "
                            "per workgroup, two workgroups per CU overlapping 1.39x).")
    return roofline


def dual_launches_in_graph_us(step, vol, repeats=20, replays=3):
    """Launch times of the level-3 DUAL convolution launches of `step` as they run INSIDE a replayed hipGraph: one eager pass records the
    arguments of every conv3d_k3_dual call on a volume of size `vol`; each recorded launch is then captured `repeats` times into a graph
    of its own (same buffers, same stream discipline as the timed graph) and `replays` back-to-back replays are bracketed by HIP events on
    the launch stream.  Per-kernel events cannot be recorded inside a captured graph on this runtime (external events are refused), and
    events around EAGER launches read 10-20 % long (host gaps, idle clocks) — this is the faithful figure.  Returns [us per launch]."""
    import rag_amd
    ops = rag_amd.ops
    calls = []
    orig = ops.conv3d_k3_dual

    def spy(x, *a, **k):
        if tuple(x.shape[2:]) == tuple(vol):
            calls.append((x, a, k))
        return orig(x, *a, **k)

    ops.conv3d_k3_dual = spy
    try:
        step()
        torch.cuda.synchronize()
    finally:
        ops.conv3d_k3_dual = orig
    out = []
    for (x, a, k) in calls:
        def many():
            for _ in range(repeats):
                orig(x, *a, **k)
        graph, _ = try_capture(many)
        if graph is None:
            return []
        for _ in range(2):
            graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(replays):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) * 1e3 / (repeats * replays))
    return out


def pmc_traffic_bytes(kernel_name: str):
    """(HBM bytes per launch of `kernel_name`, source file) from the newest committed rocprofv3 PMC summary
    (profiles/r*_pmc_summary.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes of this same bench command).
    gfx950 correction per MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B -> doubled; WRITE_SIZE is exact.
    (None, None) if no summary holds the kernel."""
    import glob
    import re
    base = re.sub(r"[<(].*", "", kernel_name)
    targs = kernel_name[len(base):].strip("<>").replace(" ", "")
    # newest first: the round number, then the letter suffix of the tag (r04a .. r04z, r04aa ..: a longer suffix is a later one)
    def age(q):
        tag = os.path.basename(q).split("_")[0]
        m = re.match(r"r(\d+)(.*)", tag)
        return (int(m.group(1)), len(m.group(2)), m.group(2)) if m else (-1, 0, tag)
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), key=age, reverse=True):
        try:
            with open(path) as f:
                table = json.load(f)
            # the kernel's instantiations differ in a leading storage type and trailing template flags (fused tails or not):
            # call-weighted mean over those whose template arguments contain ours
            recs = [v for k, v in table.items() if k.startswith(base + "<") and targs in k.replace(" ", "")]
            calls = sum(r["calls"] for r in recs)
            if calls:
                return (int(sum((2.0 * r["fetch_kib"] + r["write_kib"]) * 1024 * r["calls"] for r in recs) / calls),
                        os.path.relpath(path, ROOT))
        except (OSError, ValueError, KeyError):
            continue
    return None, None


GPU_ERROR = []      # messages of synchronizes that failed inside an error handler: a sticky GPU error (fault, abort)


def safe_sync():
    """torch.cuda.synchronize() for error handlers: after a sticky GPU error the synchronize itself raises again — swallowed here,
    so that a failed rider leg is recorded under its key and never costs the headline its line.  Such a failure is NOT a green run:
    it is remembered (GPU_ERROR), the JSON line carries "gpu_error" and the process exits non-zero after printing it."""
    try:
        torch.cuda.synchronize()
    except Exception as exc:  # noqa: BLE001
        GPU_ERROR.append(f"{type(exc).__name__}: {exc}")
        log(f"bench: synchronize in an error handler failed too ({type(exc).__name__}: {exc})")


def try_capture(fn, census_out=None):
    """Capture fn() into a hipGraph; (graph, result) or (None, None) if the capture is refused.  Thread-local error mode:
    other threads of the process (RCCL's watchdog polls events) must not invalidate the capture.  A failed capture
    leaves the bench on eager launches instead of killing the run.  The captured graph's nodes are counted
    (rag_amd.train.graph_census): a graph holding memcpy / memset nodes is not replay-safe on this runtime when null-stream
    copies (.cpu(), .item()) run between replays, so such a capture is dropped for eager launches too."""
    from rag_amd.train import graph_census
    graph = torch.cuda.CUDAGraph(keep_graph=True)
    try:
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            out = fn()
        census = graph_census(graph)
        if census_out is not None:
            census_out.update(census)
        if census["memcpy"] or census["memset"]:
            log(f"bench: captured graph holds {census['memcpy']} memcpy / {census['memset']} memset node(s); eager launches instead")
            torch.cuda.synchronize()
            return None, None
        graph.instantiate()
        graph.replay()
        torch.cuda.synchronize()
        return graph, out
    except Exception as exc:  # noqa: BLE001
        log(f"bench: hipGraph capture failed ({type(exc).__name__}: {exc}); falling back to eager launches")
        safe_sync()
        return None, None


# Untimed replays of a freshly captured hipGraph in front of every timed region.  tools/replay_ramp.py (round 4): after the idle time of
# a capture (or any synchronize) the first replays of the headline graph take 1.19-1.22 ms and converge to the steady 1.04 ms only
# after ~25 of them (clock / power management ramp; the same with 0, 5 or 50 ms of idle time in front) — a constant ~1.1 ms per timed
# region, 5 % of a 20-step region and none of a production stream's.  They are warm-up in the contract's sense: untimed, and reported
# in the line (`warmup_detail`).
RAMP_REPLAYS = 30


def ramp(graph, n=RAMP_REPLAYS):
    if graph is not None:
        for _ in range(n):
            graph.replay()


def measure(step, steps, use_graph=True, warm=2):
    """(seconds per step, output of the measured launches, 'hipGraph' | 'eager'): `warm` eager passes, capture, `steps` replays
    bracketed by device syncs.  Used by the legs that ride along after the headline's timed region."""
    for _ in range(max(warm, 1)):
        out = step()
    torch.cuda.synchronize()
    graph = None
    if use_graph:
        graph, cap = try_capture(step)
        if graph is not None:
            out = cap
    ramp(graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        if graph is not None:
            graph.replay()
        else:
            out = step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out, ("hipGraph" if graph is not None else "eager")


def shard_range(n_items: int, world: int, rank: int):
    """Contiguous [lo, hi) slice of `n_items` independent stereo pairs owned by `rank` (GPU g gets pairs
    [g*B/N, (g+1)*B/N)); sizes differ by at most one."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def timed_region(step, steps: int, warmup: int, dist=None, sync=lambda: None, device="cpu"):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + device sync on both sides;
    returns the MAX elapsed seconds over ranks (the only collective: one scalar all-reduce for timing)."""
    def barrier():
        if dist is not None:
            dist.barrier()
        sync()

    for _ in range(warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def end_to_end(device, steps, seed=0, use_graph=True):
    """(left, right) images -> disparity through rag_amd.Network: Feature Net + Matching Net, all HIP.
    Reported beside the headline Matching-Net metric ('separately reported, end-to-end')."""
    import rag_amd
    torch.manual_seed(seed)
    net = rag_amd.Network(rag_amd.ALL_CONV_GENOTYPE, device, maxdisp=MAXDISP)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm2d)):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
    net = net.to(device).eval()
    g = torch.Generator().manual_seed(1234)
    left = torch.randn((1, 3, H, W), generator=g).to(device)
    right = torch.randn((1, 3, H, W), generator=g).to(device)
    with torch.no_grad():
        for _ in range(2):
            net(left, right, 0, net.arch_init)
        torch.cuda.synchronize()
        graph = None
        if use_graph:
            graph, _ = try_capture(lambda: net(left, right, 0, net.arch_init))
            use_graph = graph is not None
        ramp(graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            if graph is not None:
                graph.replay()
            else:
                net(left, right, 0, net.arch_init)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(1.0 / dt, 3), "unit": "disparity maps/s", "ms_per_pair": round(dt * 1e3, 4),
            "what": f"(left,right)[1,3,{H},{W}] -> disp, Feature Net + Matching Net on HIP, B=1 fp32, "
                    f"{'hipGraph' if use_graph else 'eager'}"}


def randomize_bn(net, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, (torch.nn.BatchNorm3d, torch.nn.BatchNorm2d)):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)


TRAIN_H, TRAIN_W, TRAIN_B = 192, 384, 4     # reference train crop and per-GPU batch (stereo_dataset.py:59, run_rag.sh)


class TrainProfiler:
    """HIP events around every 3x3x3 weight-gradient and forward / data-gradient convolution call of one eager training step
    (the two kernel families that carry the step's MFMA work), keyed by (kind, Cin, Cout, B, D, H, W)."""

    def __init__(self, ops):
        self.ops, self.records, self.enabled = ops, [], False
        self._wgrad, self._k3 = ops.conv3d_k3_wgrad, ops.conv3d_k3

        def wgrad(x, g, cout, *a, **kw):
            if not self.enabled:
                return self._wgrad(x, g, cout, *a, **kw)
            B, Cin, D, Hh, Ww = x.shape
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._wgrad(x, g, cout, *a, **kw)
            e1.record()
            self.records.append((("conv3d_k3_wgrad", Cin, cout, B, D, Hh, Ww), e0, e1))
            return out

        def k3(x, packed, cout, *a, **kw):
            if not self.enabled:
                return self._k3(x, packed, cout, *a, **kw)
            B, Cin, D, Hh, Ww = x.shape
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self._k3(x, packed, cout, *a, **kw)
            e1.record()
            self.records.append((("conv3d_k3 (forward / data gradient)", Cin, cout, B, D, Hh, Ww), e0, e1))
            return out

        ops.conv3d_k3_wgrad, ops.conv3d_k3 = wgrad, k3

    def restore(self):
        self.ops.conv3d_k3_wgrad, self.ops.conv3d_k3 = self._wgrad, self._k3

    def dominant(self):
        by = {}
        for key, e0, e1 in self.records:
            d = by.setdefault(key, [0.0, 0])
            d[0] += e0.elapsed_time(e1) * 1e-3
            d[1] += 1
        if not by:
            return None
        key = max(by, key=lambda k: by[k][0])
        return key, by[key][0], by[key][1], sum(v[0] for v in by.values())


def train_cpu_baseline(seed=0):
    """Oracle leg of the training configuration (checker code, allowed here only): forward + smooth-L1 + backward of ONE pair of
    the same workload (Matching Net at 192x384, D=192, all-conv, train-mode BN) through the CPU oracle + PyTorch autograd, timed once."""
    from oracle import matching_oracle as O
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    rows = O.ALL_CONV
    sd = O.random_matching_state_dict(rows, seed=seed)
    g = torch.Generator().manual_seed(77)
    small = [torch.randn((1, FEA_C, 16, 32), generator=g) for _ in range(2)]
    O.train_step(small[0], small[1], torch.rand((1, 48, 96), generator=g) * 50, sd, rows, 48)          # thread-pool warm-up
    lf = torch.randn((1, FEA_C, TRAIN_H // 3, TRAIN_W // 3), generator=g)
    rf = torch.randn((1, FEA_C, TRAIN_H // 3, TRAIN_W // 3), generator=g)
    gt = torch.rand((1, TRAIN_H, TRAIN_W), generator=g) * 200
    t0 = time.perf_counter()
    O.train_step(lf, rf, gt, sd, rows, MAXDISP)
    dt = time.perf_counter() - t0
    return {"value": 1.0 / dt, "unit": "stereo pairs/s", "cores": cores, "kind": "port",
            "sample": f"ONE forward + backward of 1 pair ({TRAIN_H}x{TRAIN_W}, D={MAXDISP}, Matching Net from features, train-mode BN, "
                      f"smooth-L1) through the CPU oracle + PyTorch autograd after a small-shape warm-up ({dt:.2f} s); the GPU step "
                      "also runs the Feature Net, the all-reduce, clip and SGD"}


def train_leg(device, dist, rank, n_gpus, B, steps, warmup, use_graph, precision, with_cpu=False):
    """BASELINE configs[4]: one data-parallel training step per `step` — images -> Feature Net -> cost volume ->
    Matching Net -> Disp -> masked smooth-L1 -> backward -> flat-bucket gradient all-reduce (RCCL) -> clip -> SGD,
    forward and backward on the HIP kernels (rag_amd.autograd).  All units trainable (task 0 of the growth loop).
    Returns the JSON line (rank 0) or None."""
    import rag_amd
    from rag_amd.train import GradBucket, GraphedTrainStep, exchange_and_update, forward_backward, make_optimizer, train_step
    torch.manual_seed(0)                                   # identical replicas
    net = rag_amd.Network(rag_amd.ALL_CONV_GENOTYPE, device, maxdisp=MAXDISP)
    randomize_bn(net, 1)
    net = net.to(device).train()
    bucket = GradBucket(net.parameters())
    opt = make_optimizer(net.parameters(), bucket=bucket)
    g = torch.Generator().manual_seed(1234 + rank)         # each replica its own shard of the global batch
    left = torch.randn((B, 3, TRAIN_H, TRAIN_W), generator=g).to(device)
    right = torch.randn((B, 3, TRAIN_H, TRAIN_W), generator=g).to(device)
    gt = (torch.rand((B, TRAIN_H, TRAIN_W), generator=g) * 200).to(device)
    losses = []
    graphed = None
    if use_graph:
        try:
            graphed = GraphedTrainStep(net, opt, bucket, left, right, gt, clip=5.0, dist=dist, precision=precision)
        except Exception as exc:  # noqa: BLE001
            log(f"bench: hipGraph capture of the training step failed ({type(exc).__name__}: {exc}); eager launches")
            safe_sync()

    def step():
        if graphed is not None:
            losses.append(graphed() * 1.0)                 # a kernel, not a memcpy node's cousin on the null stream
        else:
            losses.append(train_step(net, opt, bucket, left, right, gt, clip=5.0, dist=dist, precision=precision))

    dt = timed_region(step, steps, warmup, dist, torch.cuda.synchronize, device)
    if rank != 0:
        return None
    # rank 0, untimed extra step, eager: phase split + per-call HIP events of the convolution families (roofline of the dominant one)
    prof = TrainProfiler(rag_amd.ops)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    prof.enabled = True
    ev[0].record()
    forward_backward(net, bucket, left, right, gt, precision=precision)
    ev[1].record()
    prof.enabled = False
    exchange_and_update(opt, bucket, clip=5.0, dist=dist)
    ev[2].record()
    torch.cuda.synchronize()
    prof.restore()
    phases = {"forward_backward_ms_eager": round(ev[0].elapsed_time(ev[1]), 3), "allreduce_clip_sgd_ms": round(ev[1].elapsed_time(ev[2]), 3)}
    roofline = None
    dom = prof.dominant()
    ms = dt / steps * 1e3
    if dom is not None:
        (kind, Cin, Cout, Bk, D, Hh, Ww), secs, nlaunch, conv_secs = dom
        vox = float(Bk) * D * Hh * Ww
        flops, nbytes = 2.0 * 27 * Cin * Cout * vox, 4.0 * (Cin + Cout) * vox
        x3 = kind.startswith("conv3d_k3 (") and precision == "f16x3" and rag_amd.ops.conv3d_k3_uses_x3(Cin, Cout, Bk, D, Hh, Ww)
        peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
        t_hbm, t_mfma = nbytes / (PEAK_HBM_GBS * 1e9), (3.0 if x3 else 1.0) * flops / (peak * 1e12)
        per = secs / nlaunch
        common = {"kernel": f"{kind}, Cin={Cin} Cout={Cout} on [{Bk},{D},{Hh},{Ww}] voxels" + (" (conv3d_k3_wgrad_kernel)" if "wgrad" in kind else ""),
                  "traffic": None, "algorithmic_bytes_per_launch": nbytes, "flops_per_launch": flops, "launches_per_step": nlaunch,
                  "avg_launch_us": round(per * 1e6, 2),
                  "avg_launch_us_source": "HIP events around each call in one eager step after the timed region (same kernels, same stream)",
                  "share_of_step": round(secs / (dt / steps), 3), "conv_families_share_of_step": round(conv_secs / (dt / steps), 3),
                  "floor_us": {"hbm": round(t_hbm * 1e6, 1), "mfma": round(t_mfma * 1e6, 1)}}
        if t_hbm >= t_mfma:
            roofline = {"bound": "hbm", "achieved": round(nbytes / per * 1e-9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(nbytes / per * 1e-9 / PEAK_HBM_GBS, 4), **common}
        else:
            roofline = {"bound": "mfma", "achieved": round(flops / per * 1e-12, 2), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(flops / per * 1e-12 / peak, 4), **common}
    cpu = train_cpu_baseline() if (with_cpu and n_gpus == 1) else None
    return {
        "metric": "training stereo pairs/sec at 192x384 D=192 (fwd+bwd+grad all-reduce+SGD step)",
        "value": round(n_gpus * B * steps / dt, 3), "unit": "stereo pairs/s",
        "n_gpus": n_gpus, "steps": steps, "warmup": warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[4]: training step, {B} pairs/GPU at {TRAIN_H}x{TRAIN_W}, D={MAXDISP}, all-conv genotype, "
                               "all units trainable, train-mode BN, SGD(1e-3, 0.9, wd 3e-3), clip 5",
                   "global_batch": n_gpus * B, "parallelism": f"dp{n_gpus}", "collective": "one flat fp32 gradient bucket all-reduce/step",
                   "grad_bucket_bytes": int(bucket.flat.numel() * 4),
                   "ranks_seen": dist.get_world_size() if dist is not None else 1,
                   "dist_backend": dist.get_backend() if dist is not None else None,
                   "conv_precision": precision,
                   "launch": "forward+backward as one hipGraph, exchange/clip/SGD eager" if graphed is not None else "eager"},
        "phases_rank0": phases, "loss_first_last": [round(float(losses[0]), 5), round(float(losses[-1]), 5)],
        "roofline": roofline, "cpu_baseline": cpu,
    }


def cpu_baseline(net, lf, rf):
    """Oracle leg (checker code, allowed here only): one pair of the same workload on host cores."""
    from oracle import matching_oracle as O
    cores = min(16, os.cpu_count() or 1)   # the 1-GPU box's CPU share (SURVEY 8(d) says os.cpu_count(): the box reports the host's)
    torch.set_num_threads(cores)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    rows = O.ALL_CONV
    O.matching_net_forward(lf[:, :, :16, :32].cpu().contiguous(), rf[:, :, :16, :32].cpu().contiguous(), sd, rows, 48)  # thread-pool warm-up
    lc, rc = lf[:1].cpu(), rf[:1].cpu()
    runs = 3                                  # ~15 s of CPU work: a bounded sample of the same workload (one pair per run)
    times = []
    for _ in range(runs):
        t0 = time.perf_counter()
        ref = O.matching_net_forward(lc, rc, sd, rows, MAXDISP)
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[runs // 2]
    return {"value": 1.0 / dt, "unit": "disparity maps/s", "cores": cores, "kind": "port",
            "sample": f"median of {runs} timed runs of 1 pair B=1 {H}x{W} D={MAXDISP} fp32 after a small-shape warm-up ({dt:.2f} s per pair)"}, ref


def other_configs(net, lf, rf, out_f32, ref, device, use_graph, dist):
    """The OTHER BASELINE.json configurations, measured after the headline's timed region (few replays each) so that the one line
    the driver records observes every configuration.  `ref`: the CPU oracle's disparity of the headline pair (or None)."""
    import rag_amd
    from oracle import matching_oracle as O
    res = {}
    h, w = lf.shape[2:]
    g = torch.Generator().manual_seed(4321)

    def leg(name, fn):
        """run one configuration; a failure is recorded under its key and does not stop the others"""
        try:
            fn()
        except Exception as exc:  # noqa: BLE001
            log(f"bench: leg {name} failed ({type(exc).__name__}: {exc})")
            res[name] = {"error": f"{type(exc).__name__}: {exc}"}
            safe_sync()

    def fwd(n_, a_, b_):
        def step():
            with torch.no_grad():
                return n_(a_, b_)
        return step

    # configs[2]: B=8, bf16 activation storage / fp32 accumulate; pair 0 = the headline pair
    def _config2_bf16_b8():
        lf8 = torch.cat([lf, torch.randn((7, FEA_C, h, w), generator=g).to(device)])
        rf8 = torch.cat([rf, torch.randn((7, FEA_C, h, w), generator=g).to(device)])
        dt, out, launch = measure(fwd(net, lf8.bfloat16(), rf8.bfloat16()), 5, use_graph)
        res["config2_bf16_b8"] = {
            "value": round(8 / dt, 3), "unit": "disparity maps/s", "ms_per_step": round(dt * 1e3, 4), "steps": 5, "launch": launch,
            "workload": f"BASELINE configs[2]: 8 stereo pairs/GPU/step, {H}x{W}, D={MAXDISP}, bf16 storage / f32 accumulate, all-conv genotype",
            "epe_bf16_vs_fp32_px": float((out[:1].double() - out_f32[:1].double()).abs().flatten(1).mean(dim=1).mean()),
            "epe_gpu_vs_cpu_px": O.epe(out[:1].float().cpu(), ref) if ref is not None else None,
            "epe_gate_px": 0.05, "storage": "bf16 for the full-resolution (level-3) tensors, fp32 from the first cell below that resolution on (mixed storage, DESIGN.md 4.4)",
            "epe_note": "seeded random weights drive |cost| to 1e4-1e5: softmin is nearly an argmin (DESIGN.md 4.2)"}
        log(f"  configs[2] bf16 B=8: {res['config2_bf16_b8']['value']} maps/s, EPE vs fp32 build {res['config2_bf16_b8']['epe_bf16_vs_fp32_px']:.3e}")
        del out, lf8, rf8

    leg("config2_bf16_b8", _config2_bf16_b8)
    # configs[3]: one GPU's shard (B=8) of the 64-pair DrivingStereo batch at the reference's eval pad 480x960 (stereo_dataset.py:95-96)
    def _config3_480x960_b8():
        h3, w3 = 480 // 3, 960 // 3
        a3 = torch.randn((8, FEA_C, h3, w3), generator=g).to(device)
        b3 = torch.randn((8, FEA_C, h3, w3), generator=g).to(device)
        dt, out, launch = measure(fwd(net, a3, b3), 5, use_graph)
        res["config3_480x960_b8"] = {
            "value": round(8 / dt, 3), "unit": "disparity maps/s", "ms_per_step": round(dt * 1e3, 4), "steps": 5, "launch": launch,
            "workload": f"BASELINE configs[3]: one rank's shard of the 64-pair batch: 8 stereo pairs/GPU/step, 480x960, D={MAXDISP}, f32, all-conv genotype"}
        log(f"  configs[3] 480x960 B=8: {res['config3_480x960_b8']['value']} maps/s")
        del out, a3, b3

    leg("config3_480x960_b8", _config3_480x960_b8)
    # the all-skip genotype (what an untrained BasicNetwork.genotype() returns) as the lower bound, headline size
    def _all_skip():
        skip = build_net(device, genotype=rag_amd.modules.ALL_SKIP_GENOTYPE)
        dt, out, launch = measure(fwd(skip, lf, rf), 10, use_graph)
        epe_skip = None
        if ref is not None:
            sd = {k: v.detach().cpu() for k, v in skip.state_dict().items()}
            t0 = time.perf_counter()
            ref_skip = O.matching_net_forward(lf[:1].cpu(), rf[:1].cpu(), sd, O.ALL_SKIP, MAXDISP)
            cpu_s = time.perf_counter() - t0
            epe_skip = O.epe(out[:1].float().cpu(), ref_skip)
        res["all_skip"] = {
            "value": round(1 / dt, 3), "unit": "disparity maps/s", "ms_per_step": round(dt * 1e3, 4), "steps": 10, "launch": launch,
            "workload": f"all-skip genotype (identity branches only in every cell), 1 stereo pair, {H}x{W}, D={MAXDISP}, f32",
            "epe_gpu_vs_cpu_px": epe_skip, "cpu_oracle_s_per_pair": round(cpu_s, 2) if ref is not None else None}
        log(f"  all-skip genotype: {res['all_skip']['value']} maps/s, EPE {epe_skip}")
        del out, skip
        torch.cuda.empty_cache()

    leg("all_skip", _all_skip)
    # configs[4]: the training step (B=4 at 192x384), its dominant kernel's roofline and the oracle's fwd+bwd beside it
    def _config4_train():
        line = train_leg(device, dist, 0, 1, TRAIN_B, 5, 2, use_graph, "fp32", with_cpu=ref is not None)
        res["config4_train"] = {k: line[k] for k in ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "phases_rank0", "loss_first_last",
                                                     "roofline", "cpu_baseline")}
        res["config4_train"]["workload"] = line["config"]["workload"]
        res["config4_train"]["conv_precision"] = line["config"]["conv_precision"]
        res["config4_train"]["launch"] = line["config"]["launch"]
        log(f"  configs[4] training step: {line['ms_per_step']} ms/step, {line['value']} pairs/s ({line['config']['conv_precision']})")

    leg("config4_train", _config4_train)
    return res


def rank_env(args_gpus: int):
    """Rank plumbing of one process: (world, rank, local_rank, n_gpus, seed) from the environment a launcher sets
    (torch.distributed.run, or launch_ranks below).  Touches no GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = world if world > 1 else 1
    return world, rank, local_rank, n_gpus, 1234 + rank


def free_port() -> int:
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def launch_ranks(n: int, argv) -> int:
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this process — which has made NO HIP call and never
    touches the GPU — starts N fresh child processes of this script, one rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    the way torch.distributed.run sets them), waits for all of them, relays rank 0's JSON line on stdout and returns non-zero if any
    child failed.  No exec: the children are ordinary subprocesses."""
    import subprocess
    import tempfile
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    with tempfile.TemporaryFile("w+") as out0:
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        # wait for all; a rank that dies takes the job with it (its peers would wait for it in a collective until the timeout):
        # the survivors — exactly the processes started above — are terminated after a short grace period
        failed_at = None
        while any(q.poll() is None for q in procs):
            if failed_at is None and any(q.poll() not in (None, 0) for q in procs):
                failed_at = time.monotonic()
            if failed_at is not None and time.monotonic() - failed_at > 5.0:
                for q in procs:
                    if q.poll() is None:
                        q.kill()
            time.sleep(0.1)
        codes = [q.returncode for q in procs]
        out0.seek(0)
        lines = [ln for ln in out0.read().splitlines() if ln.strip()]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if lines and not bad:
        print(lines[-1], flush=True)
        return 0
    log(f"bench: launcher: ranks failed (rank, exit code): {bad}; rank 0 printed {len(lines)} line(s)")
    if lines:
        print(lines[-1], flush=True)
    return 1


def dry_ranks(args_gpus: int) -> None:
    """--dry-ranks: the rank plumbing of main() without a GPU — every rank reports (rank, world, local rank -> device, seed), the
    ranks meet in a gloo group (barrier + gather), rank 0 prints one JSON line.  What tests/test_bench_sharding.py runs through
    the launcher above."""
    import torch.distributed as dist
    world, rank, local_rank, n_gpus, seed = rank_env(args_gpus)
    if os.environ.get("RAGMI_BENCH_DRY_FAIL_RANK") == str(rank):      # test hook: a rank that dies before the group forms
        sys.exit(7)
    me = {"rank": rank, "world": world, "local_rank": local_rank, "device": f"cuda:{local_rank}", "seed": seed}
    everyone = [me]
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        dist.barrier()
    if rank == 0:
        print(json.dumps({"dry_ranks": True, "n_gpus": n_gpus, "ranks_seen": len(everyone), "dist_backend": "gloo" if world > 1 else None,
                          "ranks": everyone}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="stereo pairs per GPU per step (configs[1]: 1)")
    ap.add_argument("--graph", type=int, default=None,
                    help="1 (default): replay the step as a captured hipGraph; 0: eager launches")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="activation storage: f32 (configs[1], default) or bf16 storage / fp32 accumulate (configs[2])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--train", action="store_true", help="BASELINE configs[4]: time the data-parallel training step instead")
    ap.add_argument("--train-precision", choices=["fp32", "f16x3"], default="fp32",
                    help="arithmetic of the training step's 3x3x3 convolutions: fp32 (default, the reference's class) or f16x3 (opt-in)")
    ap.add_argument("--no-configs", action="store_true", help="skip the legs of the other BASELINE configurations (configs object)")
    ap.add_argument("--hw", default=None, help="HxW of the stereo pairs, multiples of 12 (default 384x1248 = configs[1]; configs[3]: "
                                               "480x960 with --batch 8)")
    ap.add_argument("--precision", choices=["f16x3", "fp32"], default=None,
                    help="arithmetic contract of the 3x3x3 convolutions of the TIMED path (default: the library's default, f16x3); fp32 = "
                         "the strict leg as the headline (profiling passes of the strict path: tools/collect_profiles.sh)")
    ap.add_argument("--dry-ranks", action="store_true", help="rank plumbing only (no GPU): every rank reports rank / world / device / seed")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher: be the launcher (before anything touches the GPU: `import torch` above makes no HIP call)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    if args.dry_ranks:
        dry_ranks(args.gpus)
        return
    if args.graph is None:
        args.graph = 1
    global H, W
    if args.hw:
        H, W = (int(v) for v in args.hw.lower().split("x"))
        if H % 12 or W % 12:
            ap.error("--hw: H and W must be multiples of 12 (rag_model.py:317-323)")

    world, rank, local_rank, n_gpus, seed = rank_env(args.gpus)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # everything (eager launches, hipGraph replays, timing events) on ONE explicit stream rather than the legacy null stream
    torch.cuda.set_stream(torch.cuda.Stream(device))

    import rag_amd
    rag_amd.load_library()          # fail loudly if the HIP extension is missing
    if args.precision:
        rag_amd.ops.set_conv_precision(args.precision)
    if args.train:
        line = train_leg(device, dist, rank, n_gpus, args.batch if args.batch > 1 else TRAIN_B, args.steps, args.warmup, bool(args.graph),
                         args.train_precision, with_cpu=not args.no_cpu_baseline)
        if line is not None:
            line["library"] = rag_amd.lib_path()
            print(json.dumps(line), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    net = build_net(device)
    B, h, w = args.batch, H // 3, W // 3
    g = torch.Generator().manual_seed(seed)
    act = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    lf = torch.randn((B, FEA_C, h, w), generator=g).to(device).to(act)
    rf = torch.randn((B, FEA_C, h, w), generator=g).to(device).to(act)

    prof = K3Profiler(rag_amd.ops)

    def step():
        with torch.no_grad():
            return net(lf, rf)

    for _ in range(max(args.warmup, 1)):
        out = step()
    torch.cuda.synchronize()

    graph = None
    graph_nodes = {}
    if args.graph:
        graph, captured = try_capture(step, graph_nodes)
        if graph is not None:
            out = captured

    holder = {}

    def timed_step():
        if graph is not None:
            graph.replay()
        else:
            holder["out"] = step()

    prof.enabled = graph is None and rank == 0
    # (the W warm-up steps ran above, eagerly: one-time costs; the replays below are the clock ramp, RAMP_REPLAYS above)
    dt = timed_region(timed_step, args.steps, RAMP_REPLAYS if graph is not None else 0, dist, torch.cuda.synchronize, device)
    prof.enabled = False
    out = holder.get("out", out)

    if graph is not None and rank == 0:   # per-kernel events need eager launches: same kernels, extra pass
        prof.enabled = True
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        prof.enabled = False

    if rank == 0:
        torch.cuda.synchronize()
        by = prof.summary()
        roofline = roofline_of(by, args, dt, graph is not None)
        if graph is not None and roofline and roofline["kernel"].startswith("conv3d_x3") and "<2," in roofline["kernel"]:
            # the dominant kernel = the level-3 dual launches: their times INSIDE a replayed graph (dual_launches_in_graph_us) replace the
            # eager-pass figure in `achieved` / `frac`; the eager one stays beside it
            try:
                with torch.no_grad():
                    us = dual_launches_in_graph_us(step, (MAXDISP // 3, H // 3, W // 3))
            except Exception as exc:      # noqa: BLE001 (a measurement aid: never fatal)
                log(f"bench: in-graph launch timing failed ({type(exc).__name__}: {exc})")
                us = []
            if us and len(us) == roofline["launches_per_step"]:
                avg = sum(us) / len(us)
                roofline["avg_launch_us_eager"] = roofline["avg_launch_us"]
                roofline["avg_launch_us"] = round(avg, 2)
                roofline["launch_us_in_graph"] = [round(u, 1) for u in us]
                roofline["avg_launch_us_source"] = ("HIP events on the launch stream around back-to-back replays of a captured hipGraph holding 20 launches "
                                                    "of each of this forward's level-3 dual launches with their own buffers (per-kernel events cannot be "
                                                    "recorded inside a graph on this runtime; events around eager launches — avg_launch_us_eager — read "
                                                    "10-20 % long); rocprofv3's averages of the replayed graph agree: profiles/r05*_step_timeline.txt")
                if roofline["bound"] == "hbm":
                    gbs = roofline["algorithmic_bytes_per_launch"] / (avg * 1e-6) * 1e-9
                    roofline["achieved"], roofline["frac"] = round(gbs, 1), round(gbs / PEAK_HBM_GBS, 4)
                    tf = roofline["flops_per_launch"] / (avg * 1e-6) * 1e-12
                    roofline["mfma"]["achieved_tflops"], roofline["mfma"]["frac"] = round(tf, 2), round(tf / roofline["mfma"]["peak"], 4)
                roofline["share_of_step"] = round(sum(us) * 1e-6 / (dt / args.steps), 3)
        if by:
            for k, (s_, f_, b_, n_) in sorted(by.items(), key=lambda kv: -kv[1][0]):
                log(f"  conv3d {'x3 (split operands) channel groups' if k[1] == 'x3' else 'k3 G'}={k[0]} tx={k[1]} R={k[2]} nset={k[3]}: {n_ // args.steps} launches/step, {s_ / args.steps * 1e3:.3f} ms/step, "
                    f"{f_ / s_ * 1e-12:.1f} TFLOP/s, {b_ / s_ * 1e-9:.0f} GB/s (in+out)")
        cpu = None
        if n_gpus == 1 and not args.no_cpu_baseline:
            cpu, ref = cpu_baseline(net, lf.float(), rf.float())
            from oracle import matching_oracle as O
            epe = O.epe(out[:1].float().cpu(), ref)
            log(f"  EPE of the timed GPU path vs the CPU oracle on the same pair: {epe:.3e} px")
            cpu["epe_gpu_vs_cpu_px"] = epe
        strict = None
        if n_gpus == 1 and args.dtype == "f32" and rag_amd.ops.get_conv_precision() == "f16x3":
            # the same workload with every contraction on the fp32-input MFMA forms: the record carries both arithmetic contracts
            with rag_amd.ops.conv_precision("fp32"):
                for _ in range(2):
                    out32 = step()
                torch.cuda.synchronize()
                g32, cap32 = try_capture(step) if args.graph else (None, None)
                n32 = min(args.steps, 10)
                ramp(g32)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n32):
                    if g32 is not None:
                        g32.replay()
                    else:
                        out32 = step()
                torch.cuda.synchronize()
                dt32 = (time.perf_counter() - t0) / n32
                out32 = cap32 if g32 is not None else out32
                # its own roofline: per-kernel HIP events of an eager pass of the same kernels under the strict contract
                prof.records.clear()
                prof.enabled = True
                for _ in range(n32):
                    step()
                torch.cuda.synchronize()
                prof.enabled = False
                strict_roofline = roofline_of(prof.summary(), args, dt32 * n32, g32 is not None, steps=n32)
            strict = {"value_fp32_mfma": round(B / dt32, 3), "ms_per_step": round(dt32 * 1e3, 4), "steps": n32,
                      "what": "same workload, ops.set_conv_precision('fp32'): every 3x3x3 contraction on v_mfma_f32_4x4x1 (exact fmaf chains)",
                      "roofline": strict_roofline}
            if cpu is not None:
                strict["epe_gpu_vs_cpu_px"] = O.epe(out32[:1].float().cpu(), ref)
            log(f"  strict fp32 MFMA: {strict['value_fp32_mfma']} maps/s, EPE {strict.get('epe_gpu_vs_cpu_px')}")
        epe_bf16_vs_fp32 = None
        if n_gpus == 1 and args.dtype == "bf16":
            with torch.no_grad():
                d32 = net(lf[:1].float(), rf[:1].float())
            epe_bf16_vs_fp32 = float((out[:1].double() - d32.double()).abs().flatten(1).mean(dim=1).mean())
            log(f"  EPE bf16 storage vs the fp32 build on the same pair: {epe_bf16_vs_fp32:.3e} px")
        e2e = None
        if n_gpus == 1 and args.dtype == "f32":
            try:      # a rider: it must never cost the headline its line
                e2e = end_to_end(device, min(args.steps, 10), use_graph=bool(args.graph))
            except Exception as exc:  # noqa: BLE001
                log(f"bench: end_to_end leg failed ({type(exc).__name__}: {exc})")
                e2e = {"error": f"{type(exc).__name__}: {exc}"}
                safe_sync()
        if e2e and "value" in e2e:
            log(f"  end-to-end (images -> disparity, Feature Net + Matching Net): {e2e['value']} maps/s ({e2e['ms_per_pair']} ms/pair)")
        configs = None
        if (n_gpus == 1 and args.dtype == "f32" and B == 1 and (H, W) == (384, 1248) and not args.no_configs):
            try:      # the riders must never cost the headline its line
                configs = other_configs(net, lf, rf, out, ref if cpu is not None else None, device, bool(args.graph), dist)
            except Exception as exc:  # noqa: BLE001
                log(f"bench: the legs of the other configurations failed ({type(exc).__name__}: {exc})")
                configs = {"error": f"{type(exc).__name__}: {exc}"}
                safe_sync()
        ms = dt / args.steps * 1e3
        line = {
            "metric": f"disparity maps/sec at {H}x{W} D=192 (Matching-Net forward)",
            "value": round(n_gpus * B * args.steps / dt, 3), "unit": "disparity maps/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32 (RAGMI_F32X3: level-3/6/12 3x3x3 convolutions with fp32 operands split into power-of-two-scaled FP16 hi+lo halves on "
                      "the 16-bit matrix cores, hi*hi + hi*lo + lo*hi, fp32 accumulate: fp32-class accuracy, bound in include/rag_amd.h; "
                      "strict_fp32 holds the RAGMI_F32 number)"
                      if (args.dtype == "f32" and rag_amd.ops.get_conv_precision() == "f16x3") else
                      "f32" if args.dtype == "f32" else "bf16 storage of the full-resolution tensors (deep levels fp32) / f32 accumulate"), "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{3 if (H, W) == (480, 960) else 1 if args.dtype == 'f32' else 2}]: {B} stereo pair(s)/GPU/step, {H}x{W}, D={MAXDISP}, {args.dtype}, "
                                   "all-conv genotype, (left_fea,right_fea)->disp, inputs resident in HBM",
                       "global_batch": n_gpus * B, "sharding": "batch split, no collective",
                       "launch": "hipGraph" if graph is not None else "eager",
                       "warmup_detail": (f"{max(args.warmup, 1)} eager steps (one-time costs), capture + 1 replay, then {RAMP_REPLAYS} replays (clock ramp: "
                                         "tools/replay_ramp.py) — all untimed, in front of the barrier + synchronize that opens the timed region"
                                         if graph is not None else f"{max(args.warmup, 1)} eager steps"),
                       "ranks_seen": dist.get_world_size() if dist is not None else 1,
                       "dist_backend": dist.get_backend() if dist is not None else None},
            "roofline": roofline, "cpu_baseline": cpu, "strict_fp32": strict, "epe_bf16_vs_fp32": epe_bf16_vs_fp32, "end_to_end": e2e,
            "configs": configs, "library": rag_amd.lib_path(), "graph_nodes": graph_nodes or None,
        }
        if GPU_ERROR:
            line["gpu_error"] = GPU_ERROR
        print(json.dumps(line), flush=True)
        if GPU_ERROR:
            sys.exit(3)      # a sticky GPU error in a rider leg must not pass as a successful run
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
