"""Experiment: the order in which ONE host thread enqueues the two chains' launches. The bench form alternates A B A B ...
(~4.6 us of host time per launch, the queues switch every launch); chains are independent, so any interleaving is legal: G
launches of one chain, then G of the other (A A B B ..., A A A A B B B B ...). Measures host enqueue time, wall clock and event pair
per step for K steps from an idle stream, every form checked against the alternating one.
python tools/chains_interleave.py [K] [G ...]     (G2048_LIB=<other build> for A/B)"""
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
GS = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 5, 10, 20]
n = 1 << 20
dev = torch.device("cuda", 0)
boards = ops.synth_boards(n, seed=0x2048, device=dev)
actions = ops.synth_actions(n, seed=0x2048, device=dev)
out = torch.empty_like(boards)
scores = torch.zeros(n, dtype=torch.int32, device=dev)
reward = torch.empty(n, dtype=torch.float32, device=dev)
flags = torch.empty(n, dtype=torch.uint8, device=dev)
sc = ops.StepChains(n, 2, dev)
TUNE = int(os.environ.get("TUNE", "0"))          # 0: the library's choice (one board per lane at this size), 2: two boards per lane
calls = [ops.PreparedStep(boards[lo:hi], actions[lo:hi], scores[lo:hi], 0x2048, lo, out=out[lo:hi], reward=reward[lo:hi], flags=flags[lo:hi], tune=TUNE)
         for lo, hi in sc.bounds]
sc.keep_alive(boards, actions, out, scores, reward, flags)


def runner(G):
    def run():
        sc.fork()
        lanes = [(call, sc.stream(c).cuda_stream) for c, call in enumerate(calls)]
        for t0 in range(0, K, G):
            for call, sp in lanes:
                for t in range(t0, min(t0 + G, K)):
                    call(t, sp)
        sc.join()
    return run


def measure(G):
    run = runner(G)
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
    print("G = %2d launches of a chain in a row: wall %.2f / %.2f us per step (median / min), event pair %.2f / %.2f, host enqueue time %.2f per step" % (
        G, np.median(w), w.min(), np.median(e), e.min(), np.median(h)))
    return scores.clone()


print("%s: K = %d steps of 1,048,576 boards as two chains from an idle stream, one host thread" % (os.path.basename(os.environ.get("G2048_LIB", "libg2048_hip.so")), K))
for rnd in range(3):
    ref = None
    for G in GS:
        got = measure(G)
        ref = got if ref is None else ref
        assert torch.equal(ref, got), "scores differ for G = %d" % G
