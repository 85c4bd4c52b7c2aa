#!/usr/bin/env python3
"""A/B of g2048_step variants (boards per lane) in ONE process, interleaved rounds, hipGraph of K launches
each, HIP events around each replay (cdna_hip_programming.md rule 24). Prints us per launch (median/min)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ge.import_package()
from g2048 import ops

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K, ROUNDS = 100, 12
dev = torch.device("cuda", 0)
boards = ops.synth_boards(n, device=dev)
actions = ops.synth_actions(n, device=dev)
out = torch.empty_like(boards)
scores = torch.zeros(n, dtype=torch.int32, device=dev)
reward = torch.empty(n, dtype=torch.float32, device=dev)
flags = torch.empty(n, dtype=torch.uint8, device=dev)
variants = [1, 5, 2, 3]     # low 2 bits: boards per lane 1 / 2 / 4; bit 2: direction by per-lane selects (round-1 form)
graphs = {}
for v in variants:
    for t in range(3):
        ops.step(boards, actions, scores, 1, t, out=out, reward=reward, flags=flags, tune=v)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for t in range(K):
                ops.step(boards, actions, scores, 1, t, out=out, reward=reward, flags=flags, tune=v)
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    graphs[v] = g
res = {v: [] for v in variants}
for r in range(ROUNDS):
    for v in variants:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); graphs[v].replay(); b.record()
        torch.cuda.synchronize()
        res[v].append(a.elapsed_time(b) * 1e3 / K)
for v in variants:
    x = np.array(res[v][2:])
    us = np.median(x)
    desc = "boards/lane %d%s" % ([0, 1, 2, 4][v & 3], ", select-direction" if v & 4 else "")
    print("variant %2d (%s): median %.2f us  min %.2f us  -> %.2f TB/s algorithmic, %.1f Gsteps/s" % (
        v, desc, us, x.min(), n * 46 / us / 1e6, n / us / 1e3))
