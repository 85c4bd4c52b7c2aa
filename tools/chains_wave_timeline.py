"""When do the wavefronts of two independent sub-batch chains run? The profiler-free timeline of VecGame2048(chains=2)'s launch
form: a -DG2048_INSTRUMENT=4 build writes the start tick, end tick (100 MHz wall clock, shared by all launches) and SIMD of every
wavefront over lanes 0..2 of its f32 reward; here every (step, chain) launch of a K-step hipGraph gets a reward buffer of its own,
so one replay yields the interval of every launch and the wavefronts resident per chain over time.

    tools/build_ab.sh stiming -DG2048_INSTRUMENT=4
    G2048_LIB=build_ab/libg2048_stiming.so python3 tools/chains_wave_timeline.py [chains] [K]
"""
import os
import sys

os.environ["G2048_ALLOW_INSTRUMENTED"] = "1"
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from g2048 import ops, _lib  # noqa: E402

assert _lib.lib().g2048_build_flags() & 4, "needs a -DG2048_INSTRUMENT=4 build (G2048_LIB=build_ab/libg2048_stiming.so)"
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = 1 << 20
dev = torch.device("cuda", 0)
boards = ops.synth_boards(n, seed=0x2048, device=dev)
actions = ops.synth_actions(n, seed=0x2048, device=dev)
out = torch.empty_like(boards)
scores = torch.zeros(n, dtype=torch.int32, device=dev)
flags = torch.empty(n, dtype=torch.uint8, device=dev)
reward = torch.zeros((K, n), dtype=torch.float32, device=dev)
per = (-(-n // C) + 255) // 256 * 256
bounds = [(lo, min(lo + per, n)) for lo in range(0, n, per)]
calls = [[ops.PreparedStep(boards[lo:hi], actions[lo:hi], scores[lo:hi], 0x2048, lo, out=out[lo:hi], reward=reward[t, lo:hi], flags=flags[lo:hi])
          for lo, hi in bounds] for t in range(K)]
torch.cuda.synchronize()
main = torch.cuda.Stream(device=dev)
sides = [torch.cuda.Stream(device=dev) for _ in bounds[1:]]
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    with torch.cuda.graph(g, stream=main, capture_error_mode="thread_local"):
        for s in sides:
            s.wait_stream(main)
        for t in range(K):
            for c in range(len(bounds)):
                calls[t][c](t, (main if c == 0 else sides[c - 1]).cuda_stream)
        for s in sides:
            main.wait_stream(s)
torch.cuda.synchronize()
IDLE = "idle" in sys.argv       # replay from an idle stream (as bench.py's timed region does) instead of behind another replay
EAGER = "eager" in sys.argv     # plain launches on the chains' streams instead of the graph


def eager():
    for s in sides:
        s.wait_stream(torch.cuda.current_stream(dev))
    for t in range(K):
        for c in range(len(bounds)):
            calls[t][c](t, (torch.cuda.current_stream(dev) if c == 0 else sides[c - 1]).cuda_stream)
    for s in sides:
        torch.cuda.current_stream(dev).wait_stream(s)


for rep in range(3):
    if not IDLE:
        g.replay()
    torch.cuda.synchronize() if IDLE else None
    eager() if EAGER else g.replay()      # the second of two back-to-back replays is the one looked at (its ticks overwrite the first's)
    torch.cuda.synchronize()
w = reward.view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
iv = {}
base = None
for t in range(K):
    for c, (lo, hi) in enumerate(bounds):
        x = w[t, lo:hi].reshape(-1, 64)
        iv[(t, c)] = (x[:, 0].copy(), x[:, 1].copy(), x[:, 2].copy())
        m = x[:, 0].min()
        base = m if base is None else min(base, m)
print("%d chain(s) of %s boards, %s of %d steps, %s; times in us from the first wavefront's start" % (
    len(bounds), [hi - lo for lo, hi in bounds], "eager launches" if EAGER else "one hipGraph", K,
    "from an idle stream" if IDLE else "second of two back-to-back replays"))
print(" step chain   first start   last end   (launch interval)   wavefronts  SIMDs")
ends = []
for t in range(K):
    for c in range(len(bounds)):
        s, e, simd = iv[(t, c)]
        a, b = (s.min() - base) * 0.01, (e.max() - base) * 0.01
        ends.append(b)
        if t < 6 or t >= K - 2:
            print("  %3d  %3d   %10.2f %10.2f   %8.2f us            %6d  %5d" % (t, c, a, b, b - a, len(s), len(np.unique(simd))))
total = max(ends)
print("whole graph: first wavefront start -> last wavefront end %.2f us = %.2f us per 1 Mi-board step" % (total, total / K))
# launches in flight over time (a launch = [first wavefront start, last wavefront end])
edges = sorted([((iv[k][0].min() - base) * 0.01, 1) for k in iv] + [((iv[k][1].max() - base) * 0.01, -1) for k in iv])
depth, tp, in1, in2 = 0, 0.0, 0.0, 0.0
for tt, d in edges:
    if depth >= 1:
        in1 += tt - tp
    if depth >= 2:
        in2 += tt - tp
    depth += d
    tp = tt
print("launches in flight (wavefronts of it on the chip): >= 1 for %.1f %% of the graph's time, >= 2 for %.1f %%" % (100 * in1 / total, 100 * in2 / total))
# wavefronts resident per SIMD, by chain, every 0.5 us over steps 2..4
t_lo = (iv[(2, 0)][0].min() - base) * 0.01
t_hi = (iv[(min(4, K - 1), len(bounds) - 1)][1].max() - base) * 0.01
grid = np.arange(t_lo, t_hi, 0.5)
for c in range(len(bounds)):
    S = np.concatenate([iv[(t, c)][0] for t in range(K)]).astype(np.float64)
    E = np.concatenate([iv[(t, c)][1] for t in range(K)]).astype(np.float64)
    S, E = (S - base) * 0.01, (E - base) * 0.01
    print("chain %d wavefronts resident per SIMD at t = %.1f, +0.5, ... us: %s" % (
        c, t_lo, " ".join("%.1f" % (np.sum((S <= x) & (E > x)) / 1024.0) for x in grid)))
