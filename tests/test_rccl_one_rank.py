"""RCCL on hardware with the rank count a one-GPU box has: the training step's flat-bucket gradient
all-reduce (rag_amd.train.GradBucket.all_reduce_mean, reference loop approaches/rag.py:204-216) in a 1-rank `nccl` group.

The child (tests/rccl_child.py) is run to completion by conftest.py at collection time, before this pytest process initialises
the GPU, so it is a fresh process in every sense and never shares the GPU with the parent; this test only reads its verdict."""
import json

import pytest

import conftest


@pytest.mark.gpu
def test_rccl_one_rank_allreduce_equals_no_dist():
    child = conftest.RCCL_CHILD
    assert child is not None, "conftest did not start the RCCL child (no GPU visible at collection time?)"
    proc, out_path, log_path = child
    rc = proc.returncode
    assert rc is not None, "conftest did not wait for the RCCL child"
    with open(log_path) as f:
        log = f.read()
    try:
        with open(out_path) as f:
            verdict = json.load(f)
    except OSError:
        pytest.fail(f"RCCL child wrote no verdict (exit code {rc}); its output:\n{log[-4000:]}")
    print("RCCL child verdict:", json.dumps({k: v for k, v in verdict.items() if k != "error"}))
    assert "error" not in verdict, verdict["error"] + "\n" + log[-2000:]
    assert verdict["dist_backend"] == "nccl" and verdict["ranks_seen"] == 1
    assert verdict["probe_ok"]
    assert verdict["dist_vs_nodist_maxdiff"] <= verdict["tolerance"], verdict     # within 10x the step's own run-to-run noise
    assert verdict["momentum_maxdiff"] <= 1e-5
    assert len(verdict["losses_dist"]) == 4
    assert all(abs(a - b) <= 1e-5 * max(1.0, abs(b)) for a, b in zip(verdict["losses_dist"], verdict["losses_nodist"]))
    assert verdict["graph_nodes"]["memcpy"] == 0 and verdict["graph_nodes"]["memset"] == 0
    assert verdict["ok"] and rc == 0
