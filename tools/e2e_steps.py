"""How the end-to-end leg's number depends on the length of its timed region (bench.end_to_end): python tools/e2e_steps.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
for steps in (10, 10, 30, 100, 30, 10):
    r = bench.end_to_end(dev, steps)
    print(steps, r["value"], r["ms_per_pair"], flush=True)
