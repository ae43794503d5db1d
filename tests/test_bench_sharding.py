"""Multi-rank logic of bench.py on CPU: 2 processes, gloo backend.  The data path has no collective (pairs are
independent); what is distributed is the batch split and the max-over-ranks timing."""
import os
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_shard_range_partitions_pairs():
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def step():                      # rank 1 is the straggler: the job time is ITS time
        calls.append(1)
        time.sleep(0.02 * (1 + 2 * rank))

    dt = bench.timed_region(step, steps=5, warmup=2, dist=dist)
    # sharded "forward": each rank owns a slice of 5 independent items; gathering restores the full batch
    lo, hi = bench.shard_range(5, world, rank)
    x = torch.arange(5.0)
    mine = (x[lo:hi] * 2 + 1)
    parts = [None] * world
    dist.all_gather_object(parts, mine)
    if rank == 0:
        torch.save({"dt": dt, "calls": len(calls), "gathered": torch.cat(parts)}, out)
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["calls"] == 7                                   # W + K steps exactly
    assert 5 * 0.06 * 0.9 <= r["dt"] <= 5 * 0.06 * 3         # max over ranks = the straggler's 5 x 60 ms
    assert torch.equal(r["gathered"], torch.arange(5.0) * 2 + 1)


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    import torch.nn as nn
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rag_amd.train import GradBucket, make_optimizer
    torch.manual_seed(0)                       # identical replicas
    model = nn.Sequential(nn.Linear(5, 7), nn.ReLU(), nn.Linear(7, 3))
    for p in model[0].parameters():            # a frozen ("reused") unit stays out of the bucket
        p.requires_grad = False
    bucket = GradBucket(model.parameters())
    opt = make_optimizer(model.parameters(), lr=0.1, momentum=0.0, weight_decay=0.0)
    x = torch.randn((4, 5), generator=torch.Generator().manual_seed(100 + rank))
    bucket.zero()
    model(x).square().sum().backward()
    local = bucket.flat.clone()
    bucket.all_reduce_mean(dist)
    norm = bucket.clip_(1e9)
    opt.step()
    q.put((rank, local.numpy(), bucket.flat.clone().numpy(), float(norm), [p.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_bucket_allreduce_two_ranks_gloo():
    """Config 5's gradient exchange: one flat bucket, sum then / N, identical parameters afterwards on both ranks."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, l0, a0, n0, p0), (_, l1, a1, n1, p1) = res
    assert l0.shape == (7 * 3 + 3,)                               # only the trainable layer is in the bucket
    np.testing.assert_allclose(a0, (l0 + l1) / 2, rtol=1e-6, atol=1e-7)
    np.testing.assert_array_equal(a0, a1)
    assert n0 == n1
    for u, v in zip(p0, p1):
        np.testing.assert_array_equal(u, v)


def _run_bench(*argv, env=None):
    import subprocess
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=e, capture_output=True, text=True, timeout=300)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher (WORLD_SIZE unset): the parent starts 2 children before touching any GPU, each
    goes through main()'s rank plumbing (RANK / LOCAL_RANK -> device, seed 1234 + rank), they meet in a process group, and rank 0's
    JSON line is relayed with ranks_seen == 2.  --dry-ranks: the plumbing without a GPU (gloo)."""
    import json
    r = _run_bench("--gpus", "2", "--dry-ranks")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["dry_ranks"] and d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["dist_backend"] == "gloo"
    assert [q["rank"] for q in d["ranks"]] == [0, 1]
    assert [q["device"] for q in d["ranks"]] == ["cuda:0", "cuda:1"]
    assert [q["seed"] for q in d["ranks"]] == [1234, 1235]
    assert all(q["world"] == 2 for q in d["ranks"])


def test_bench_one_rank_needs_no_launcher():
    import json
    r = _run_bench("--gpus", "1", "--dry-ranks")
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and d["dist_backend"] is None and d["ranks"][0]["device"] == "cuda:0"


def test_bench_under_an_external_launcher_does_not_relaunch():
    """With WORLD_SIZE set (torch.distributed.run's environment) the script is a rank, never a launcher."""
    import json
    r = _run_bench("--gpus", "2", "--dry-ranks", env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ranks_seen"] == 1 and d["ranks"][0]["world"] == 1


def test_bench_launcher_reports_a_failed_rank():
    """A child that dies makes the launcher exit non-zero (rank 1 exits before the group forms; rank 0's rendezvous then fails too)."""
    r = _run_bench("--gpus", "2", "--dry-ranks", env={"RAGMI_BENCH_DRY_FAIL_RANK": "1", "TORCH_DIST_INIT_BARRIER": "0"})
    assert r.returncode != 0
