"""Sorting networks against the counting loop over complete games for a range of beam widths / depths (every decision of every
game, helpers on): final boards, scores, move counts, invalid moves and expansions must agree.
usage: python tools/soak_network.py [games]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
from g2048.vec import VecGame2048

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda")
bad = 0
for w, d, fd in ((20, 30, False), (17, 30, False), (24, 20, False), (32, 30, False), (28, 12, False), (12, 30, False), (16, 30, False),
                 (5, 30, False), (20, 30, True), (32, 6, True), (10, 15, False), (20, 3, False), (31, 4, False)):
    res = []
    t0 = time.time()
    for rbc in (False, True):
        env = VecGame2048(n, device=dev, seed=900 + w)
        r = ops.play_games(env.boards, env.scores, w, d, 5000, 512, 1024, 900 + w, 0, fd, False, rank_by_counting=rbc)
        res.append((env.boards.cpu(), env.scores.cpu(), r["moves"].cpu(), r["invalid_moves"].cpu(), r["expanded"].cpu()))
    same = all(torch.equal(x, y) for x, y in zip(*res))
    bad += not same
    print("w=%3d d=%2d fixed_down=%d: %d games, %9d decisions, same: %s  (%.1f s)" % (w, d, fd, n, int(res[0][2].sum()), same, time.time() - t0), flush=True)
print("soak done, configurations that differ:", bad)
sys.exit(1 if bad else 0)
