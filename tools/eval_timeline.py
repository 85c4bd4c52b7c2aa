"""Where the evaluation's time goes: 4096 games (w=20, d=30) with the move cap at 250 .. 5000, with and without helpers."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
import g2048
g2048.evaluate_beam_search(256, 20, 30, seed=3, max_moves=50)
r = g2048.evaluate_beam_search(4096, 20, 30, seed=2025)
m = np.array(r["moves"]); inv = np.array(r["invalid_moves"])
print("full: %.4f s; moves mean %.0f; games with > 1000/1500/2000/2500/3000/4000 moves: %s" % (r["elapsed_s"], m.mean(), [(m > x).sum() for x in (1000, 1500, 2000, 2500, 3000, 4000)]))
print("capped %d; of the uncapped: max %d, 99th pct %d" % ((m >= 5000).sum(), m[m < 5000].max(), np.percentile(m[m < 5000], 99)))
for cap in (250, 500, 1000, 1500, 2000, 2500, 3000, 4000, 5000):
    t = min(g2048.evaluate_beam_search(4096, 20, 30, seed=2025, max_moves=cap)["elapsed_s"] for _ in range(2))
    t1 = min(g2048.evaluate_beam_search(4096, 20, 30, seed=2025, max_moves=cap, one_phase=True)["elapsed_s"] for _ in range(1))
    print("cap %4d: helpers %.4f s   one wavefront %.4f s   decisions %d" % (cap, t, t1, np.minimum(m, cap).sum()), flush=True)
