import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


RCCL_CHILD = None     # (finished Popen, verdict path, log path) of tests/rccl_child.py, run to completion before this process touches the GPU


def _start_rccl_child(config, items):
    """tests/test_rccl_one_rank.py needs a FRESH process for its 1-rank nccl group.  It runs HERE, to completion, before this
    process makes its first HIP call (torch.cuda.is_available() below): the two processes never use the GPU at the same time,
    so a fault or a hang belongs to exactly one of them and the parent's timing tests are not perturbed.  The GPU's presence
    is read from the device node, not from torch (device_count() may initialise HIP on some ROCm builds)."""
    global RCCL_CHILD
    markexpr = config.getoption("markexpr", "") or ""
    wanted = any(item.nodeid.startswith("tests/test_rccl_one_rank.py") or "test_rccl_one_rank" in item.nodeid for item in items)
    if RCCL_CHILD is not None or not wanted or "not gpu" in markexpr or not os.path.exists("/dev/kfd"):
        return
    import subprocess
    import tempfile
    tmp = tempfile.mkdtemp(prefix="ragmi_rccl_")
    out_path, log_path = os.path.join(tmp, "verdict.json"), os.path.join(tmp, "child.log")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    with open(log_path, "w") as logf:
        proc = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rccl_child.py"), out_path], stdout=logf,
                                stderr=subprocess.STDOUT, env=env, cwd=ROOT)
        try:
            proc.wait(timeout=900)
        except subprocess.TimeoutExpired:
            proc.kill()          # exactly the process started above
            proc.wait()
    RCCL_CHILD = (proc, out_path, log_path)


def pytest_collection_modifyitems(config, items):
    _start_rccl_child(config, items)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def split_sd(arrays, prefix="sd::"):
    """state_dict entries of a fixture as torch tensors (keys stripped of `prefix`)."""
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in arrays.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    return load_golden
