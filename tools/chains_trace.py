"""Workload for `rocprofv3 --kernel-trace`: configs[1]'s env.step as one hipGraph of K steps, once as single launches (1 chain)
and once as two independent sub-batch chains on parallel branches; each graph replayed R times. tools/chains_timeline.py reads
the kernel trace and prints the intervals.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 $ROOT/tools/chains_trace.py [K] [R]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from g2048 import ops, VecGame2048  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
R = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n = 1 << 20
dev = torch.device("cuda", 0)
boards = ops.synth_boards(n, seed=0x2048, device=dev)
actions = ops.synth_actions(n, seed=0x2048, device=dev)
for chains in (1, 2):
    env = VecGame2048(n, device=dev, seed=0x2048, chains=chains)
    env.load(boards)
    env.step(actions)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for t in range(K):
                env.step(actions, join=False)
            env.join()
    torch.cuda.synchronize()
    for r in range(R):
        g.replay()
    torch.cuda.synchronize()
    print("chains=%d: %d replays of a %d-step graph" % (chains, R, K))

# the same two chains as plain launches on two streams, queued behind a gate (no hipGraph): does a kernel tracer keep two
# QUEUES concurrent when the launches do not come from a graph?
for chains in (1, 2):
    sc = ops.StepChains(n, chains, dev)
    out = torch.empty_like(boards)
    scores = torch.zeros(n, dtype=torch.int32, device=dev)
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    flags = torch.empty(n, dtype=torch.uint8, device=dev)
    calls = [ops.PreparedStep(boards[lo:hi], actions[lo:hi], scores[lo:hi], 0x2048, lo, out=out[lo:hi], reward=reward[lo:hi], flags=flags[lo:hi])
             for lo, hi in sc.bounds]
    torch.cuda.synchronize()
    for r in range(R):
        torch.cuda._sleep(int(2.4e9 * 0.003))      # gate: the host queues everything below while the GPU spins
        sc.fork()
        for t in range(K):
            for c, call in enumerate(calls):
                call(t, sc.stream(c).cuda_stream)
        sc.join()
        torch.cuda.synchronize()
    print("chains=%d: %d bursts of %d eager steps behind a gate" % (chains, R, K))
