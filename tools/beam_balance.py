"""How unevenly does the beam batch load the SIMDs?  Blocks b, b + 1024, b + 2048, ... of a 64-thread-block launch share a
SIMD on MI355X (tools/ubench/placement.hip), so a launch lasts as long as the heaviest group of games. Per-game cost proxy:
children expanded. Prints max / mean of the per-SIMD sums for the bench root order, a random order, and a sorted 'snake'
deal; and times the kernel with the roots permuted on the host accordingly (game ids permuted along: same decisions)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops

SEED, N, W, D = 0x2048, 4096, 20, 30
dev = torch.device("cuda")
import bench
bench.torch = torch
roots = bench.beam_roots(ops, N, 0, dev)


def run(order, reps=20):
    r = roots[order].contiguous()
    for _ in range(2):
        ops.beam_get_action(r, W, D, seed=SEED, step_index=1, want_expanded=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tot = 0
    outs = []
    for k in range(reps):
        a, p, e = ops.beam_get_action(r, W, D, seed=SEED, step_index=10 + k, want_expanded=True)
        outs.append(e)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3
    exp = torch.stack(outs).sum().item()
    return sec / reps * 1e6, exp / sec, outs[0].cpu().numpy().astype(np.int64)


def imbalance(cost_in_block_order, groups=1024):
    c = cost_in_block_order.reshape(-1, groups).sum(axis=0)
    return c.max() / c.mean()


ident = torch.arange(N, device=dev)
us, rate, cost = run(ident)
print("bench order:    %.1f us per launch, %.3e expansions/s; per-SIMD cost max/mean %.3f; per-game cost mean %.0f sd %.0f min %d max %d"
      % (us, rate, imbalance(cost), cost.mean(), cost.std(), cost.min(), cost.max()))
rng = np.random.default_rng(1)
perm = rng.permutation(N)
us, rate, c2 = run(torch.from_numpy(perm).to(dev))
print("random order:   %.1f us, %.3e; max/mean %.3f" % (us, rate, imbalance(c2)))
# snake deal of the games sorted by measured cost (a perfect predictor -- an upper bound for any root-derived estimate)
srt = np.argsort(-cost, kind="stable")
order = np.empty(N, dtype=np.int64)
for p in range(N // 1024):
    seg = srt[p * 1024:(p + 1) * 1024]
    order[p * 1024:(p + 1) * 1024] = seg if p % 2 == 0 else seg[::-1]
us, rate, c3 = run(torch.from_numpy(order).to(dev))
print("snake by cost:  %.1f us, %.3e; max/mean %.3f" % (us, rate, imbalance(c3)))
# a root-derived predictor: the depth class (empty cells of the root) then the empty count
b = roots.cpu().numpy()
empty = (b == 0).sum(axis=1)
depth = np.where(empty <= 4, 25, np.where(empty >= 10, 10, 30))
key = depth * 100 + (16 - empty)
srt = np.argsort(-key, kind="stable")
for p in range(N // 1024):
    seg = srt[p * 1024:(p + 1) * 1024]
    order[p * 1024:(p + 1) * 1024] = seg if p % 2 == 0 else seg[::-1]
us, rate, c4 = run(torch.from_numpy(order).to(dev))
print("snake by depth class: %.1f us, %.3e; max/mean %.3f" % (us, rate, imbalance(c4)))
for dcls in (10, 25, 30):
    m = depth == dcls
    print("  depth %d: %d games, cost mean %.0f sd %.0f" % (dcls, m.sum(), cost[m].mean() if m.any() else 0, cost[m].std() if m.any() else 0))
