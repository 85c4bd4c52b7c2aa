"""A/B of the evaluation driver's speculative helpers (g2048_play_games): same games, time per variant.
usage: python tools/eval_tail_ab.py [games] [width] [depth] ["helpers,games_left,stuck,wait_us;..."]   (explicit tunings go through g2048_play_games_tuned)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
import g2048

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
w = int(sys.argv[2]) if len(sys.argv) > 2 else 20
d = int(sys.argv[3]) if len(sys.argv) > 3 else 30
g2048.evaluate_beam_search(256, w, d, seed=3, max_moves=50)
torch.cuda.synchronize()
base = None
variants = [("one wavefront per game", None), ("helpers, defaults", "")]
for spec in (sys.argv[4].split(";") if len(sys.argv) > 4 else []):
    variants.append((spec, spec))
for name, tune in variants:
    tuning = tuple(int(x) for x in tune.split(",")) if tune else None      # "helpers,games_left,stuck,wait_us"
    best = None
    for rep in range(2):
        r = g2048.evaluate_beam_search(n, w, d, seed=2025, one_phase=tune is None, tuning=tuning)
        best = r["elapsed_s"] if best is None else min(best, r["elapsed_s"])
    sig = (tuple(r["scores"]), tuple(r["moves"]), tuple(r["invalid_moves"]), tuple(r["highest_tiles"]), r["total_expansions"])
    if base is None:
        base = sig
    print("%-36s %.4f s  same games: %s  moves %d invalid %d capped %d" % (
        name, best, sig == base, r["total_moves"], sum(r["invalid_moves"]), sum(1 for m in r["moves"] if m >= 5000)), flush=True)
