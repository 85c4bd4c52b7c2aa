"""Complete games at the reference's evaluation configuration (width 20, depth 30, 5000-move cap): the GPU evaluation against
the oracle playing the same games one by one on the host cores (a process pool; the oracle is test infrastructure).
usage: python tools/oracle_full_games.py [games] [processes]"""
import os, sys, time
from multiprocessing import get_context
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SEED, W, D, CAP, BASE = 2025, 20, 30, 5000, 0


def oracle_game(gid):
    from oracle import oracle as O
    O.set_num_threads(1)
    k0, k1 = O.rng_keys(SEED, O.DOM_RESET, 0)
    b = O.env_reset(O.rng_draw(k0, k1, gid, 0), O.rng_draw(k0, k1, gid, 1))
    score = moves = invalid = 0
    done = False
    while not done and moves < CAP:
        a = O.beam_get_action(b, -1, W, D, seed=SEED, step_index=moves, game_id=gid)["action"]
        s0, s1 = O.rng_keys(SEED, O.DOM_STEP, moves)
        b, score, r, done, v, hi = O.env_step(b, score, a, O.rng_draw(s0, s1, gid, 0))
        invalid += int(not v)
        moves += 1
    return gid, int(score), moves, invalid, [int(x) for x in b.reshape(-1)]


def oracle_game_with_history(gid):
    """oracle_game plus what evaluate_beam_search.run_game also returns (evaluate_beam_search.py:44-50, :72-75): the action of
    every move, the board before every move and after the last one (uint8 log2 codes, (moves + 1, 16)) and the scores."""
    import numpy as np
    from oracle import oracle as O
    O.set_num_threads(1)
    k0, k1 = O.rng_keys(SEED, O.DOM_RESET, 0)
    b = O.env_reset(O.rng_draw(k0, k1, gid, 0), O.rng_draw(k0, k1, gid, 1))
    score = moves = invalid = 0
    done = False
    actions, boards, scores = [], [O.pack(b[None, :])[0].copy()], [0]
    while not done and moves < CAP:
        a = O.beam_get_action(b, -1, W, D, seed=SEED, step_index=moves, game_id=gid)["action"]
        s0, s1 = O.rng_keys(SEED, O.DOM_STEP, moves)
        b, score, r, done, v, hi = O.env_step(b, score, a, O.rng_draw(s0, s1, gid, 0))
        invalid += int(not v)
        moves += 1
        actions.append(int(a)); boards.append(O.pack(b[None, :])[0].copy()); scores.append(int(score))
    return (gid, int(score), moves, invalid, [int(x) for x in b.reshape(-1)], bytes(actions), np.stack(boards).astype(np.uint8),
            np.asarray(scores, dtype=np.int64))


if __name__ == "__main__":
    import numpy as np
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    procs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    t0 = time.time()
    with get_context("spawn").Pool(procs) as pool:
        job = pool.map_async(oracle_game_with_history, range(BASE, BASE + n), chunksize=1)
        import torch
        import __graft_entry__ as ge
        ge.import_package()
        import g2048
        res = g2048.evaluate_beam_search(n, W, D, seed=SEED, max_moves=CAP, game_id_base=BASE, histories="all")
        print("GPU: %d games, %d moves, %.3f s (action stream recorded, every game replayed into its histories)"
              % (n, res["total_moves"], res["elapsed_s"]), flush=True)
        while not job.ready():
            job.wait(30)
            print("  oracle still playing, %.0f s" % (time.time() - t0), flush=True)
        ref = job.get()
    bad = bad_hist = 0
    for gid, score, moves, invalid, board, actions, codes, scores in ref:
        g = gid - BASE
        ok = (res["scores"][g] == score and res["moves"][g] == moves and res["invalid_moves"][g] == invalid and
              [int(x) for x in res["final_boards"][g].reshape(-1)] == board)
        bad += not ok
        if not ok:
            print("MISMATCH game", gid, (res["scores"][g], res["moves"][g]), (score, moves))
        # f1: the move-set the fused kernel recorded and the histories replayed from it (evaluate_beam_search.py:44-50, :72-75)
        game = res["games"][g]
        want = np.where(codes > 0, 1 << codes.astype(np.int64), 0).reshape(moves + 1, 4, 4)
        okh = (game["moveset"] == list(actions) and game["scores_history"] == [int(x) for x in scores] and
               np.array_equal(np.stack(game["board_history"]), want) and
               game["max_tiles_history"] == [int(x) for x in want.reshape(moves + 1, 16).max(axis=1)])
        bad_hist += not okh
        if not okh:
            print("HISTORY MISMATCH game", gid)
    print("oracle: %d games in %.0f s on %d processes; mismatching games: %d; games whose move-set or replayed histories differ: %d; "
          "capped games %d; moves %d" % (n, time.time() - t0, procs, bad, bad_hist, sum(1 for r in ref if r[2] >= CAP), sum(r[2] for r in ref)))
    sys.exit(1 if (bad or bad_hist) else 0)
