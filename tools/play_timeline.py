"""The evaluation launch game by game (a -DG2048_INSTRUMENT=2 build writes, into the milestone record, when a game started / ended,
at which move and when it registered for helpers, how many searches its owner ran after that, how many helper results it took
and how many arrived late):  G2048_LIB=build_ab/libg2048_ptiming.so python3 tools/play_timeline.py [games]"""
import os, sys
os.environ["G2048_ALLOW_INSTRUMENTED"] = "1"      # this tool reads the clock ticks a -DG2048_INSTRUMENT=2 build writes over real outputs
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops, _lib
assert _lib.lib().g2048_build_flags() & 2, "needs a -DG2048_INSTRUMENT=2 build (tools/build_ab.sh NAME -DG2048_INSTRUMENT=2; G2048_LIB=build_ab/libg2048_NAME.so): a product build's outputs are results, not clock ticks"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dev = torch.device("cuda")
for rep in range(2):
    boards, scores = ops.reset(n, 2025, 0, 0, device=dev)
    r = ops.play_games(boards, scores, 20, 30, max_moves=5000, seed=2025)
    torch.cuda.synchronize()
m = r["milestone_move"].cpu().numpy().astype(np.int64)
moves = r["moves"].cpu().numpy(); inv = r["invalid_moves"].cpu().numpy()
u = lambda x: x & 0xffffffff
t0 = u(m[:, 0]); t1 = u(m[:, 1]); base = t0.min()
start = (t0 - base) * 0.01e-3; end = (t1 - base) * 0.01e-3                      # ms
reg_t = m[:, 2]; reg_at = np.where(reg_t >= 0, (u(m[:, 3]) - base) * 0.01e-3, np.nan)
searches, hits, late = m[:, 4], m[:, 5], m[:, 6]
print("games %d: launch %.1f ms; games started within %.2f ms (late starters: %d after 1 ms, last start %.1f ms)" % (
    n, end.max(), np.percentile(start, 50), int((start > 1.0).sum()), start.max()))
for q in (10, 25, 50, 75, 90, 95, 99, 100):
    print("  %3d %% of the games over at %.1f ms" % (q, np.percentile(end, q)))
cap = moves >= 5000
print("capped games: %d; end %.1f .. %.1f ms (mean %.1f); start %.2f .. %.1f ms" % (cap.sum(), end[cap].min(), end[cap].max(), end[cap].mean(),
      start[cap].min(), start[cap].max()))
rc = cap & (reg_t >= 0)
print("  registered at move %d .. %d (median %d), at %.1f .. %.1f ms (median %.1f)" % (reg_t[rc].min(), reg_t[rc].max(), np.median(reg_t[rc]),
      np.nanmin(reg_at[rc]), np.nanmax(reg_at[rc]), np.nanmedian(reg_at[rc])))
after = 5000 - reg_t[rc]; dur = end[rc] - reg_at[rc]
print("  after registration: %.0f moves in %.1f ms (medians) = %.1f us per move; owner searches %.0f, helper results taken %.0f, late %.0f"
      " -> %.2f moves per owner search, %.1f us per owner search" % (np.median(after), np.median(dur), np.median(dur / after) * 1e3,
      np.median(searches[rc]), np.median(hits[rc]), np.median(late[rc]), np.median(after / np.maximum(searches[rc], 1)),
      np.median(dur / np.maximum(searches[rc], 1)) * 1e3))
pre = reg_at[rc] - start[rc]
print("  before registration: %.0f moves in %.1f ms (medians) = %.1f us per move" % (np.median(reg_t[rc]), np.median(pre), np.median(pre / np.maximum(reg_t[rc], 1)) * 1e3))
worst = np.argsort(-end)[:8]
for g in worst:
    print("  latest game %5d: start %.2f end %.1f ms, moves %d invalid %d, registered at move %d (%.1f ms), searches %d hits %d late %d" % (
        g, start[g], end[g], moves[g], inv[g], reg_t[g], reg_at[g], searches[g], hits[g], late[g]))
other = ~cap
print("other games: mean %.0f moves, end mean %.1f ms; %.1f us per move overall" % (moves[other].mean(), end[other].mean(),
      ((end[other] - start[other]) / np.maximum(moves[other], 1)).mean() * 1e3))
