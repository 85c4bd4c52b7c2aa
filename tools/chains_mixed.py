"""Experiment: one chain replayed from a linear hipGraph (one host call for its K launches), the other host-paced with plain launches
-- 4.8 us of host time per plain launch makes the one-thread alternating form (2 launches per step = 9.6 us) as slow as the GPU
itself (~9.8 us per step in steady state); with one chain in a graph the host enqueues ONE launch per step. Order: chain 0's first
plain launch (the GPU starts at once), chain 1's graph, chain 0's other launches.
python tools/chains_mixed.py [K]     (G2048_LIB=<other build> for A/B)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from g2048 import ops  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n = 1 << 20
dev = torch.device("cuda", 0)
boards = ops.synth_boards(n, seed=0x2048, device=dev)
actions = ops.synth_actions(n, seed=0x2048, device=dev)
out = torch.empty_like(boards)
scores = torch.zeros(n, dtype=torch.int32, device=dev)
reward = torch.empty(n, dtype=torch.float32, device=dev)
flags = torch.empty(n, dtype=torch.uint8, device=dev)
sc = ops.StepChains(n, 2, dev)
calls = [ops.PreparedStep(boards[lo:hi], actions[lo:hi], scores[lo:hi], 0x2048, lo, out=out[lo:hi], reward=reward[lo:hi], flags=flags[lo:hi])
         for lo, hi in sc.bounds]
sc.keep_alive(boards, actions, out, scores, reward, flags)


def alternating():
    sc.fork()
    lanes = [(call, sc.stream(c).cuda_stream) for c, call in enumerate(calls)]
    for t in range(K):
        for call, sp in lanes:
            call(t, sp)
    sc.join()


# chain 1's K launches as a linear graph (captured on a stream of its own, replayed on chain 1's stream)
cap = torch.cuda.Stream(device=dev)
g1 = torch.cuda.CUDAGraph()
with torch.cuda.stream(cap):
    with torch.cuda.graph(g1, stream=cap, capture_error_mode="thread_local"):
        for t in range(K):
            calls[1](t, cap.cuda_stream)
torch.cuda.synchronize()


def mixed():
    sc.fork()
    sp0 = sc.stream(0).cuda_stream
    calls[0](0, sp0)
    with torch.cuda.stream(sc.stream(1)):
        g1.replay()
    for t in range(1, K):
        calls[0](t, sp0)
    sc.join()


# both chains' steps P..K-1 as linear graphs, steps 0..P-1 as plain alternating launches that keep the GPU busy while the graphs
# are being launched
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
tails = []
for c in range(2):
    cs = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(cs):
        with torch.cuda.graph(g, stream=cs, capture_error_mode="thread_local"):
            for t in range(P, K):
                calls[c](t, cs.cuda_stream)
    tails.append(g)
torch.cuda.synchronize()


def prefix_then_graphs():
    sc.fork()
    lanes = [(call, sc.stream(c).cuda_stream) for c, call in enumerate(calls)]
    for t in range(P):
        for call, sp in lanes:
            call(t, sp)
    for c in range(2):
        with torch.cuda.stream(sc.stream(c)):
            tails[c].replay()
    sc.join()


def measure(name, run):
    run()
    torch.cuda.synchronize()
    walls, evs, hosts = [], [], []
    for rep in range(14):
        scores.zero_()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record()
        run()
        b.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        walls.append((t2 - t0) * 1e6 / K)
        hosts.append((t1 - t0) * 1e6 / K)
        evs.append(a.elapsed_time(b) * 1e3 / K)
    w, e, h = np.array(walls[2:]), np.array(evs[2:]), np.array(hosts[2:])
    print("%-44s wall %.2f / %.2f us per step (median / min), event pair %.2f / %.2f, host enqueue time %.2f per step" % (
        name, np.median(w), w.min(), np.median(e), e.min(), np.median(h)))
    return scores.clone()


print("%s: K = %d steps of 1,048,576 boards as two chains from an idle stream, one host thread" % (os.path.basename(os.environ.get("G2048_LIB", "libg2048_hip.so")), K))
for rnd in range(3):
    ref = measure("both chains plain launches, alternating", alternating)
    got = measure("chain 1 from a linear hipGraph, chain 0 plain", mixed)
    assert torch.equal(ref, got), "scores differ"
    got = measure("%d plain steps, then one linear graph per chain" % P, prefix_then_graphs)
    assert torch.equal(ref, got), "scores differ"
