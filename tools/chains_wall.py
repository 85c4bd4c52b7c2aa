"""K steps of configs[1] from an IDLE stream, as the bench's timed region runs them: wall clock (perf_counter around launch +
synchronize) and the event pair on the launch stream, for the launch forms: one hipGraph of single launches; one hipGraph with C
parallel branches; eager launches on C streams (ops.PreparedStep, host-paced).   python tools/chains_wall.py [K]"""
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


def form(C, graph):
    sc = ops.StepChains(n, C, dev)
    calls = [ops.PreparedStep(boards[lo:hi], actions[lo:hi], scores[lo:hi], 0x2048, lo, out=out[lo:hi], reward=reward[lo:hi], flags=flags[lo:hi])
             for lo, hi in sc.bounds]

    def run():
        sc.fork()
        for t in range(K):
            for c, call in enumerate(calls):
                call(t, sc.stream(c).cuda_stream)
        sc.join()
    run()
    torch.cuda.synchronize()
    g = None
    if graph:
        side = torch.cuda.Stream(device=dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                run()
        torch.cuda.synchronize()
    walls, evs, hosts = [], [], []
    for rep in range(12):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a.record()
        if g is not None:
            g.replay()
        else:
            run()
        b.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        walls.append((t2 - t0) * 1e6 / K)
        hosts.append((t1 - t0) * 1e6 / K)
        evs.append(a.elapsed_time(b) * 1e3 / K)
    w, e, h = np.array(walls[2:]), np.array(evs[2:]), np.array(hosts[2:])
    print("chains %d %-26s wall %.2f / %.2f us per step (median / min), event pair %.2f / %.2f, host enqueue time %.2f per step" % (
        C, "hipGraph" if graph else "eager (PreparedStep)", np.median(w), w.min(), np.median(e), e.min(), np.median(h)))


print("K = %d steps of 1,048,576 boards from an idle stream" % K)
for C in (1, 2):
    form(C, True)
    form(C, False)
