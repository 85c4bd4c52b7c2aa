import sys, time, torch
sys.path.insert(0, "/root/repo")
import __graft_entry__ as ge
ge.import_package()
import g2048
for n in (256, 1024, 4096, 8192):
    g2048.evaluate_beam_search(n, 20, 30, seed=3, max_moves=50)
    torch.cuda.synchronize()
    r = g2048.evaluate_beam_search(n, 20, 30, seed=3, max_moves=400)
    print("games %5d: 400 moves in %.4f s -> %.1f us per decision round (moves %d)" % (n, r["elapsed_s"], r["elapsed_s"] / 400 * 1e6, r["total_moves"]))
