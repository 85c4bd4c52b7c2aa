"""Experiment: the two-chain step with one HOST THREAD per chain. The one-thread form pays ~4.6 us of host time per launch
alternating between two queues (9.2 us per step against ~10 us of GPU time: the host paces the GPU); here each chain's K launches
come from a thread of its own (ctypes releases the GIL for the duration of the call), started together by an event.
python tools/chains_threads.py [K]     (G2048_LIB=<other build> for A/B)"""
import os
import sys
import threading
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


class Worker(threading.Thread):
    """Launches chain 1's K steps whenever `go` is set; `done` when they are all enqueued."""

    def __init__(self):
        super().__init__(daemon=True)
        self.go, self.done, self.stop = threading.Event(), threading.Event(), False
        self.stream_ptr = None

    def run(self):
        torch.cuda.set_device(dev)
        while True:
            self.go.wait()
            self.go.clear()
            if self.stop:
                return
            call, sp = calls[1], self.stream_ptr
            for t in range(K):
                call(t, sp)
            self.done.set()


w = Worker()
w.start()


def run_threads():
    sc.fork()
    w.stream_ptr = sc.stream(1).cuda_stream
    w.go.set()
    call, sp = calls[0], sc.stream(0).cuda_stream
    for t in range(K):
        call(t, sp)
    w.done.wait()
    w.done.clear()
    sc.join()


def run_one_thread():
    sc.fork()
    lanes = [(call, sc.stream(c).cuda_stream) for c, call in enumerate(calls)]
    for t in range(K):
        for call, sp in lanes:
            call(t, sp)
    sc.join()


def measure(name, run):
    run()
    torch.cuda.synchronize()
    want = scores.clone()
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
    wv, e, h = np.array(walls[2:]), np.array(evs[2:]), np.array(hosts[2:])
    print("%-34s wall %.2f / %.2f us per step (median / min), event pair %.2f / %.2f, host enqueue time %.2f per step" % (
        name, np.median(wv), wv.min(), np.median(e), e.min(), np.median(h)))
    return want


print("%s: K = %d steps of 1,048,576 boards as two chains, from an idle stream" % (os.path.basename(os.environ.get("G2048_LIB", "libg2048_hip.so")), K))
for rnd in range(3):
    measure("one host thread (bench form)", run_one_thread)
    ref = scores.clone()
    measure("one host thread per chain", run_threads)
    assert torch.equal(ref, scores), "the threaded form's scores differ"
w.stop = True
w.go.set()
