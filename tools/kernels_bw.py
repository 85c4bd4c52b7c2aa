#!/usr/bin/env python3
"""Achieved algorithmic GB/s of every secondary kernel (hipGraph of 50 launches, HIP events), at 1 Mi and 8 Mi boards."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

ge.import_package()
from g2048 import ops, _lib as L

dev = torch.device("cuda", 0)


def timed(fn, K=50):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for _ in range(K):
                fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3 / K)
    return best


for n in (1 << 20, 1 << 23):
    b = ops.synth_boards(n, device=dev)
    a = ops.synth_actions(n, device=dev)
    tiles = ops.unpack(b)
    obs = torch.empty((n, 16), dtype=torch.float32, device=dev)
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    ev = torch.empty(n, dtype=torch.float64, device=dev)
    pk = torch.empty_like(b)
    sc = torch.zeros(n, dtype=torch.int32, device=dev)
    probs = torch.softmax(torch.randn(n, 4, device=dev), 1)
    act = torch.empty(n, dtype=torch.uint8, device=dev); pa = torch.empty(n, dtype=torch.float32, device=dev)
    rows = [
        ("obs_f32 (R16 W64)", 80, lambda: ops.obs(b, out=obs)),
        ("unpack_i32 (R16 W64)", 80, lambda: ops.unpack(b, out=tiles)),
        ("pack_i32 (R64 W16)", 80, lambda: ops.pack(tiles, out=pk)),
        ("valid_moves env (R16 W1)", 17, lambda: ops.valid_moves(b, False, out=mask)),
        ("valid_moves agent (R16 W1)", 17, lambda: ops.valid_moves(b, True, out=mask)),
        ("eval fast (R16 W8)", 24, lambda: ops.evaluate(b, L.EVAL_FAST, out=ev)),
        ("eval full (R16 W8)", 24, lambda: ops.evaluate(b, L.EVAL_FULL, out=ev)),
        ("eval ppo heuristic (R16 W8)", 24, lambda: ops.evaluate(b, L.EVAL_PPO_HEURISTIC, out=ev)),
        ("eval ppo shaping (R16 W8)", 24, lambda: ops.evaluate(b, L.EVAL_PPO_SHAPING, out=ev)),
        ("reset (W16+4)", 20, lambda: ops.reset(n, 1, 0, 0, boards=pk, scores=sc)),
        ("synth_boards (W16)", 16, lambda: ops.synth_boards(n, out=pk)),
        ("sample_actions (R16+1 W1+4)", 22, lambda: ops.sample_actions(probs, mask, actions=act, prob=pa)),
        ("step f32 (R21 W25)", 46, lambda: ops.step(b, a, sc, 1, 0, out=pk, reward=pa, flags=mask)),
    ]
    print("n = %d boards" % n)
    for name, bytes_per, fn in rows:
        us = timed(fn)
        print("  %-30s %8.2f us  %7.0f GB/s algorithmic  (%.2f of 8 TB/s)" % (name, us, n * bytes_per / us / 1e3, n * bytes_per / us / 1e3 / 8000))
