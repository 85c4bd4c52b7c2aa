"""Beam leg of bench.py alone: expansions/s of g2048_beam_get_action on the benchmark's root set (4096 games, w=20, d=30).
G2048_LIB=<other build> for A/B comparisons."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
SEED = 0x2048
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
import bench
bench.torch = torch
roots = bench.beam_roots(ops, n, 0, dev)
BAL = os.environ.get("NO_BALANCE") is None
for w in range(3):
    ops.beam_get_action(roots, 20, 30, seed=SEED, step_index=w, want_expanded=True, balanced_order=BAL)
torch.cuda.synchronize()
best = 0
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    outs = []
    e0.record()
    for k in range(20):
        outs.append(ops.beam_get_action(roots, 20, 30, seed=SEED, step_index=10 + k, want_expanded=True, balanced_order=BAL)[2])
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3
    best = max(best, torch.stack(outs).sum().item() / sec)
    us = sec / 20 * 1e6
print(("balanced " if BAL else "caller order ") + "%s: %d games %.3e expansions/s (%.1f us per call)" % (os.path.basename(os.environ.get("G2048_LIB", "libg2048_hip.so")), n, best, us))
