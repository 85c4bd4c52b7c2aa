#!/usr/bin/env python3
"""Reproduce the reference's beam-search evaluation (run_evaluation.py / report.md) on the GPU.

    python examples/evaluate_beam.py --games 4096 --width 20 --depth 30 [--out overall_results.json] [--save-dir DIR]

All games are played to completion together (one wavefront per game for the search, one lane per board for the
env step); prints the summary the reference prints and optionally writes its overall_results.json schema. With --save-dir the
run also keeps what evaluate_beam_search.py / train.py write per game: game_N_data.json (board / score / max-tile histories) for every
game that reached 2048 and BeamSearchAgent_best_moveset_tile_T.txt for the best game -- rebuilt on the device from the one action
byte per move the fused kernel records."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2048-using-reinforcement-learning_amd"))
import g2048
from g2048.evaluate import save_overall_results

ap = argparse.ArgumentParser()
ap.add_argument("--games", type=int, default=1024)
ap.add_argument("--width", type=int, default=20)
ap.add_argument("--depth", type=int, default=30)
ap.add_argument("--seed", type=int, default=2025)
ap.add_argument("--max-moves", type=int, default=5000)
ap.add_argument("--out", default=None)
ap.add_argument("--save-dir", default=None)
a = ap.parse_args()
res = g2048.evaluate_beam_search(a.games, a.width, a.depth, seed=a.seed, max_moves=a.max_moves,
                                 histories="high_tile" if a.save_dir else None)
s = res["summary"]
print("==== EVALUATION SUMMARY ====")
print("Highest tile reached: %d" % s["highest_tile"])
print("Best score: %d" % s["best_score"])
print("Average score: %.1f" % s["average_score"])
print("Average highest tile: %.1f" % s["average_highest_tile"])
print("Games reaching >= 2048: %.1f%%   hit the %d-move cap: %d" % (100 * s["rate_2048_or_more"], a.max_moves, s["hit_move_cap"]))
print("Highest tile distribution:", json.dumps(s["tile_distribution_pct"]))
print("%d games, %d moves, %.2f s  (%.3g moves/s, %.3g node expansions/s)" % (
    a.games, res["total_moves"], res["elapsed_s"], s["moves_per_s"], s["expansions_per_s"]))
if a.out:
    print("wrote", save_overall_results(res, a.out))
if a.save_dir:
    os.makedirs(a.save_dir, exist_ok=True)
    for i, game in sorted(res["games"].items()):                      # evaluate_beam_search.py:170-196: the games that reached 2048
        g2048.save_game_data(game, os.path.join(a.save_dir, "game_%d_data.json" % (i + 1)))
    best = res["best_game_idx"]
    if best in res["games"]:                                          # train.py:140-142
        g2048.save_moveset(res["games"][best], os.path.join(a.save_dir, "BeamSearchAgent_best_moveset_tile_%d.txt"
                                                             % res["games"][best]["highest_tile"]))
    print("wrote %d game_N_data.json files to %s" % (len(res["games"]), a.save_dir))
