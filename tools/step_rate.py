"""g2048_step alone: us per 1 Mi-board launch (hipGraph of 100 launches, median / min of 12 replays). G2048_LIB=<other build> for A/B."""
import os, sys
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
for t in range(3):
    ops.step(boards, actions, scores, 1, t, out=out, reward=reward, flags=flags)
torch.cuda.synchronize()
side = torch.cuda.Stream(device=dev)
side.wait_stream(torch.cuda.current_stream(dev))
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        for t in range(K):
            ops.step(boards, actions, scores, 1, t, out=out, reward=reward, flags=flags)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize()
res = []
for r in range(ROUNDS):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); g.replay(); b.record()
    torch.cuda.synchronize()
    res.append(a.elapsed_time(b) * 1e3 / K)
x = np.array(res[2:])
print("%s: %d boards, median %.2f us  min %.2f us per launch -> %.2f TB/s algorithmic" % (
    os.path.basename(os.environ.get("G2048_LIB", "libg2048_hip.so")), n, np.median(x), x.min(), n * 46 / np.median(x) / 1e6))
