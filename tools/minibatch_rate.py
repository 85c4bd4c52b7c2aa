"""g2048_minibatch_gather alone: microseconds per launch and GB/s of its algorithmic bytes (per sample: 64 B observation row + 16 B next board + 1 + 4 + 4 + 1 B
scalars read at RANDOM transition indices; 64 + 64 + 8 + 4 + 4 + 4 B written in order) over config 4's buffer of 8,388,608 transitions.
python3 tools/minibatch_rate.py [batch ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
dev = torch.device("cuda")
n = 65536 * 128
boards = ops.synth_boards(n, seed=1, device=dev)
obs = ops.obs(boards)
nxt = ops.synth_boards(n, seed=2, device=dev)
acts = ops.synth_actions(n, seed=1, device=dev)
logp = torch.zeros(n, device=dev); rew = torch.zeros(n, device=dev); flags = torch.zeros(n, dtype=torch.uint8, device=dev)
BYTES = 64 + 16 + 1 + 4 + 4 + 1 + 64 + 64 + 8 + 4 + 4 + 4
for B in [int(x) for x in sys.argv[1:]] or [4096, 65536, 1 << 20, 1 << 23]:
    for w in range(3):
        ops.minibatch_gather(obs, acts, logp, rew, nxt, flags, B, 7, w)
    torch.cuda.synchronize()
    best = None
    for rep in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for k in range(10):
            ops.minibatch_gather(obs, acts, logp, rew, nxt, flags, B, 7, 10 + k)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        best = ms if best is None else min(best, ms)
    print("%8d samples of %d transitions: %.2f us per launch, %.0f GB/s of %d B per sample (%.3f of 8 TB/s)" % (B, n, best * 1e3, B * BYTES / best / 1e6, BYTES,
          B * BYTES / best / 1e6 / 8000), flush=True)
