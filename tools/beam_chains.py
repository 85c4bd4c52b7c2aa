"""Experiment: the benchmark's 4096 beam-search games as C independent sub-batches on C streams (chain c's call k+1 behind chain c's call
k only), against one launch per call -- does one sub-batch's tail (the last searches of a launch run on thinly occupied SIMDs)
overlap the other's start?  python tools/beam_chains.py [games]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from g2048 import ops  # noqa: E402
import bench  # noqa: E402

bench.torch = torch
SEED = 0x2048
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
roots = bench.beam_roots(ops, n, 0, dev)
CALLS = 20


def run(C):
    per = (n + C - 1) // C
    slices = [(lo, min(lo + per, n)) for lo in range(0, n, per)]
    streams = [torch.cuda.Stream(device=dev) for _ in slices]
    hist = [ops.BeamHistory(dev) for _ in slices]
    parts = [roots[lo:hi].contiguous() for lo, hi in slices]
    cur = torch.cuda.current_stream(dev)

    def burst(first):
        outs = []
        for s in streams:
            s.wait_stream(cur)
        for k in range(CALLS):
            for c, (lo, hi) in enumerate(slices):
                with torch.cuda.stream(streams[c]):
                    outs.append(ops.beam_get_action(parts[c], 20, 30, seed=SEED, step_index=first + k, game_id_base=lo, want_expanded=True,
                                                    history=hist[c])[2])
        for s in streams:
            cur.wait_stream(s)
        return outs
    burst(0)
    torch.cuda.synchronize()
    best, acts = 0.0, None
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        outs = burst(100 + 100 * rep)
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3
        best = max(best, sum(int(o.sum().item()) for o in outs) / sec)
    total = sum(int(o.sum().item()) for o in outs)
    print("%d games as %d sub-batch(es) on %d stream(s): %.3e expansions/s, %.1f us per batch decision" % (n, C, C, best, 1e6 * total / best / CALLS))


for C in (1, 2, 4, 1, 2):
    run(C)
