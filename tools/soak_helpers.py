"""Soak test of the evaluation driver's helper wavefronts: random batch sizes / widths / depths / caps / seeds / helper tunings,
every output of g2048_play_games compared with the one-wavefront-per-game run of the same games.
usage: python tools/soak_helpers.py [seconds]"""
import os, sys, time, random
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
import g2048

KEYS = ("scores", "highest_tiles", "moves", "valid_moves", "invalid_moves", "milestones", "best_games", "total_expansions",
        "unfinished")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
rnd = random.Random(20260104)
t0 = time.time()
runs = bad = 0
last = t0
while time.time() - t0 < budget:
    n = rnd.choice([1, 2, 3, 7, 33, 100, 257, 600, 1500, 4096])
    w = rnd.choice([1, 4, 8, 16, 17, 20, 32, 33, 64, 100])
    d = rnd.choice([1, 3, 6, 10, 20, 30])
    cap = rnd.choice([50, 300, 1000, 5000])
    if n * cap * w * d > 4096 * 5000 * 20 * 30 // 3:
        cap = max(50, cap // 10)
    kw = dict(num_games=n, beam_width=w, search_depth=d, seed=rnd.getrandbits(40), max_moves=cap,
              game_id_base=rnd.getrandbits(35), fixed_down=rnd.random() < 0.25)
    tune = rnd.choice([None, None, "%d,%d,%d,%d" % (rnd.choice([0, 8, 64, 512, 2048]), rnd.choice([0, 16, 256, 100000]),
                                                    rnd.choice([1, 4, 16, 64]), rnd.choice([0, 5, 60, 300]))])
    r0 = g2048.evaluate_beam_search(one_phase=True, **kw)
    r1 = g2048.evaluate_beam_search(tuning=tuple(int(x) for x in tune.split(",")) if tune else None, **kw)
    ok = all(r0[k] == r1[k] for k in KEYS) and np.array_equal(r0["final_boards"], r1["final_boards"])
    runs += 1
    if not ok:
        bad += 1
        print("MISMATCH", kw, tune, [k for k in KEYS if r0[k] != r1[k]], flush=True)
    if time.time() - last > 20:
        last = time.time()
        print("%4d runs, %d mismatches, %.0f s" % (runs, bad, last - t0), flush=True)
print("soak done: %d runs, %d mismatches" % (runs, bad))
sys.exit(1 if bad else 0)
