"""Multi-rank logic of bench.py on CPU: 2 processes, gloo backend.  The data path has no collective (pairs are
independent); what is distributed is the batch split and the max-over-ranks timing."""
import os
import sys
import time

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
