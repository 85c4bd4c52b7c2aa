"""Helper wavefronts on small batches and short caps (1, 16, 100, 256 games; 400 / 5000 moves), default and wide tunings."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
import g2048
g2048.evaluate_beam_search(64, 20, 30, seed=3, max_moves=50)
for n in (1, 16, 100, 256):
    for cap in (400, 5000):
        for tune in (None, "", "2048,100000,16,60", "2048,100000,16,200"):
            tuning = tuple(int(x) for x in tune.split(",")) if tune else None
            best = min(g2048.evaluate_beam_search(n, 20, 30, seed=3, max_moves=cap, one_phase=tune is None, tuning=tuning)["elapsed_s"] for _ in range(2))
            print("n=%4d cap=%4d %-22s %.4f s" % (n, cap, "one wavefront" if tune is None else "helpers " + tune, best), flush=True)
