"""One 1 Mi-board env.step as C independent sub-batch chains (VecGame2048(chains=C)'s launch form), measured two ways.

    python tools/chains_rate.py [n] [K]

A chain is a contiguous slice of the boards; step t+1 of chain c is ordered only behind step t of chain c, so one chain's
launch head / drain can overlap another chain's arithmetic. Per C in {1, 2, 4, 8}:
  graph   K steps captured as ONE hipGraph with C parallel branches (fork / join on side streams), replayed; event pair around
          a replay queued behind another replay
  queued  the same launches enqueued eagerly on C streams behind a gate (a spinning kernel on each stream, so that the host's
          launch rate is not in the figure), event pairs per stream from gate release to the last launch
Every form's outputs are compared with the single-launch result (boards, scores, rewards, flags: bit for bit).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from g2048 import ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
SEED = 0x2048
dev = torch.device("cuda", 0)
boards = ops.synth_boards(n, seed=SEED, device=dev)
actions = ops.synth_actions(n, seed=SEED, device=dev)


def buffers():
    return (torch.empty_like(boards), torch.zeros(n, dtype=torch.int32, device=dev),
            torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.uint8, device=dev))


ref = buffers()
for t in range(K):
    ops.step(boards, actions, ref[1], SEED, t, 0, out=ref[0], reward=ref[2], flags=ref[3])
torch.cuda.synchronize()


def slices(C):
    per = -(-n // C)
    per = (per + 255) // 256 * 256          # whole blocks per chain
    return [(lo, min(lo + per, n)) for lo in range(0, n, per)]


def prepared(C, bufs):
    out, scores, reward, flags = bufs
    return [ops.PreparedStep(boards[lo:hi], actions[lo:hi], scores[lo:hi], SEED, lo, out=out[lo:hi], reward=reward[lo:hi],
                             flags=flags[lo:hi]) for lo, hi in slices(C)]


def check(bufs, what):
    ok = all(torch.equal(a, b) for a, b in zip(bufs, ref))
    if not ok:
        print("   !!! %s differs from the single launch" % what)
    return ok


def graph_form(C, chain_major=False):
    bufs = buffers()
    calls = prepared(C, bufs)
    torch.cuda.synchronize()
    main = torch.cuda.Stream(device=dev)
    sides = [torch.cuda.Stream(device=dev) for _ in range(C - 1)]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(main):
        with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
            for s in sides:
                s.wait_stream(main)
            if chain_major:
                for c, call in enumerate(calls):
                    for t in range(K):
                        call(t, (main if c == 0 else sides[c - 1]).cuda_stream)
            else:
                for t in range(K):                      # launch order: step-major, the chains interleaved
                    for c, call in enumerate(calls):
                        call(t, (main if c == 0 else sides[c - 1]).cuda_stream)
            for s in sides:
                main.wait_stream(s)
    torch.cuda.synchronize()
    res = []
    for r in range(14):
        bufs[1].zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g.replay()
        bufs[1].zero_()
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 1e3 / K)
    ok = check(bufs, "graph C=%d" % C)
    x = np.array(res[2:])
    return np.median(x), x.min(), ok


def linear_graphs_form(C):
    """One linear hipGraph per chain (K launches of that chain), each replayed on its own stream; fork / join by events."""
    bufs = buffers()
    calls = prepared(C, bufs)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(C)]
    graphs = []
    for c, call in enumerate(calls):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(streams[c]):
            with torch.cuda.graph(g, stream=streams[c], capture_error_mode="thread_local"):
                for t in range(K):
                    call(t, streams[c].cuda_stream)
        graphs.append(g)
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream(dev)

    def replay_all():
        fork = torch.cuda.Event()
        fork.record(cur)
        for c, g in enumerate(graphs):
            streams[c].wait_event(fork)
            with torch.cuda.stream(streams[c]):
                g.replay()
        for s in streams:
            cur.wait_stream(s)
    res, wall = [], []
    import time
    for r in range(14):
        bufs[1].zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        replay_all()
        bufs[1].zero_()
        a.record()
        replay_all()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) * 1e3 / K)
        bufs[1].zero_()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        replay_all()
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) * 1e6 / K)
    ok = check(bufs, "linear graphs C=%d" % C)
    x = np.array(res[2:])
    return np.median(x), x.min(), ok, float(np.median(wall[2:]))


def queued_form(C):
    bufs = buffers()
    calls = prepared(C, bufs)
    streams = [torch.cuda.Stream(device=dev) for _ in range(C)]
    res = []
    for r in range(8):
        bufs[1].zero_()
        torch.cuda.synchronize()
        gate = torch.cuda.Event()
        e0 = [torch.cuda.Event(enable_timing=True) for _ in range(C)]
        e1 = [torch.cuda.Event(enable_timing=True) for _ in range(C)]
        with torch.cuda.stream(streams[0]):
            torch.cuda._sleep(int(2.4e9 * 0.004))       # ~4 ms: the host enqueues everything below meanwhile
            gate.record()
        for c, s in enumerate(streams):
            s.wait_event(gate)
            e0[c].record(s)
        for t in range(K):
            for c, call in enumerate(calls):
                call(t, streams[c].cuda_stream)
        for c, s in enumerate(streams):
            e1[c].record(s)
        torch.cuda.synchronize()
        # all chains start at the gate; the job ends with the slowest chain
        res.append(max(e0[0].elapsed_time(e1[c]) for c in range(C)) * 1e3 / K)
    ok = check(bufs, "queued C=%d" % C)
    x = np.array(res[1:])
    return np.median(x), x.min(), ok


print("%d boards, K = %d steps per measurement; us per whole-batch step (median / min)" % (n, K))
for C in (1, 2, 3, 4, 8):
    gm, gmin, ok1 = graph_form(C)
    qm, qmin, ok2 = queued_form(C)
    cm, cmin, ok3 = graph_form(C, chain_major=True)
    lm, lmin, ok4, lwall = linear_graphs_form(C)
    print("chains %d: branch graph chain-major %.2f / %.2f us   one linear graph per chain on its own stream %.2f / %.2f us (wall incl. sync %.2f)  equal: %s" % (
        C, cm, cmin, lm, lmin, lwall, ok3 and ok4))
    print("chains %d: graph %.2f / %.2f us   queued streams %.2f / %.2f us   equal to single launch: %s  -> frac %.3f (graph median)" % (
        C, gm, gmin, qm, qmin, ok1 and ok2, n * 46 / gm / 1e6 / 8.0))
