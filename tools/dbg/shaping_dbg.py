import sys, os
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge
g2048 = ge.import_package()
from oracle import oracle as O
from g2048 import ops
from test_gpu_rollout import TinyTransformerPolicy, oracle_remember_of_rollout
DEV = "cuda:0"
torch.manual_seed(1)
n, T = 2048, 96
pol = TinyTransformerPolicy().to(DEV).eval()
rc = g2048.RolloutCollector(n, T, pol, device=DEV, seed=11, id_base=5, shaping=True, seen_capacity_log2=8)
R = O.Remember()
b, sc = O.reset_batch(n, seed=11, epoch=0, id_base=5)
for c in range(3):
    res = rc.collect()
    acts = res["actions"].cpu().numpy()
    want = np.empty((T, n)); wnov = np.empty((T, n), bool)
    nbs = np.empty((T, n, 16), np.uint8); sts = np.empty((T, n, 16), np.uint8)
    for t in range(T):
        nb, _, r, fl = O.step_batch(b, acts[t], sc.copy(), seed=11, step_index=c * T + t, id_base=5, opts=0)
        want[t], wnov[t] = R.batch(b, nb, r)
        nbs[t] = nb; sts[t] = b
        b, sc, _, _ = O.step_batch(b, acts[t], sc, seed=11, step_index=c * T + t, id_base=5, opts=1)
    got = res["shaping"].cpu().numpy()
    bad = np.argwhere(got != want)
    print("collect", c, "mismatches", len(bad), "cap_log2", rc.seen.capacity_log2, "count", len(rc.seen), R.n_seen,
          "highest", int(rc.seen.highest.item()), R.highest_tile_seen)
    print("  next_boards equal:", np.array_equal(rc.next_boards.cpu().numpy(), nbs), " state_max equal:",
          np.array_equal(rc.state_maxcode.cpu().numpy(), sts.max(axis=2)))
    for t, e in bad[:10]:
        print("  ", t, e, got[t, e], want[t, e], got[t, e] - want[t, e], "oracle novel", wnov[t, e])
