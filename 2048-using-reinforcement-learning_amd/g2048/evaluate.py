"""Batched evaluation driver: many beam-search games played to completion, resident on the GPU.

Device-resident version of the reference's evaluation loops --
`run_evaluation.evaluate_beam_search` (run_evaluation.py:48-130) and
`evaluate_beam_search.run_game / run_evaluation` (evaluate_beam_search.py:16-217): every game does
`action = agent.get_action(state)` (no caller mask, so the agent's own validity check applies,
run_evaluation.py:62) then `env.step(action)` until done or the move cap (5000, run_evaluation.py:67).
Here all games advance together: one `g2048_beam_get_action` launch (one wavefront per game) and one
`g2048_step` launch per move, no host round trip except an "all finished?" poll every `check_every` moves.
Finished games stay in the batch as no-ops (their counters are frozen), so ids -- and therefore every random
draw -- never change: game g at move t always uses draws (seed, BEAM|STEP, t, game_id_base + g).

The result dict has the keys of both reference drivers (scores, highest_tiles, moves, valid_moves,
invalid_moves, milestones, best_games, final_boards, best_board, best_score, best_game_idx) and
`save_overall_results` writes the reference's overall_results.json schema (evaluate_beam_search.py:198-214).

Per-game histories (`histories=`): the reference's `run_game` also returns `board_history`, `max_tiles_history` and
`scores_history` (evaluate_beam_search.py:44-50, :72-75, :88-97), writes them to `game_{i}_data.json` for the games that
reached 2048 (:185-196), and `train.py` keeps every episode's move-set (`moveset`, :51, :67) and writes the best ones to
`*_best_moveset_tile_N.txt` (:140-142). The fused kernel keeps ONE byte per move (the action); since every spawn is a
counter-based draw keyed by (seed, move, game id), replaying those bytes on the device (`g2048_replay_games`) rebuilds every
intermediate board, score and max tile of the games asked for -- `results["games"][i]` has the reference's `run_game` keys,
`save_game_data` / `save_moveset` write the reference's two file formats.
"""
import json
import time
import warnings

import torch

from . import _lib as L
from . import ops
from .vec import VecGame2048

MILESTONES = (64, 128, 256, 512, 1024, 2048, 4096, 8192)      # evaluate_beam_search.py:42-43 (fixed in the kernel)


def evaluate_beam_search(num_games=4096, beam_width=20, search_depth=30, seed=0x2048, max_moves=5000,
                         device="cuda", game_id_base=0, check_every=64, early_game_threshold=512,
                         mid_game_threshold=1024, fixed_down=False, use_graph=True, fused=True, one_phase=False, _table_only=False,
                         tuning=None, histories=None):
    """fused=True (default): every game is played start to finish by its own wavefront in one kernel launch
    (`g2048_play_games`; helper wavefronts of that launch pre-compute decisions for the last games, one_phase=True turns
    them off -- same games). fused=False: the step-by-step loop (one beam launch + one step launch + bookkeeping per move for
    the whole batch; with use_graph=True a captured hipGraph of one move is replayed). All three produce identical games.
    tuning = (helpers, games_left, stuck, wait_us): explicit helper-wavefront parameters (ops.play_games; measurements, tests).
    histories: None, or which games also get their per-move histories (results["games"][i]: board_history, scores_history,
    max_tiles_history, moveset, ... -- the dict evaluate_beam_search.run_game returns): "best5" (results["best_games"]),
    "high_tile" (every game that reached 2048, the ones the reference writes game_N_data.json for), "all", or an iterable of
    game indices. Fused driver only: the kernel records one action byte per move and the asked games are replayed on the device."""
    dev = torch.device(device)
    n = int(num_games)
    if histories is not None and not fused:
        raise ValueError("g2048.evaluate_beam_search: histories need the fused driver (fused=True)")
    t_start = time.perf_counter()
    env = VecGame2048(n, device=dev, seed=seed, id_base=game_id_base)
    boards0 = env.boards.clone() if histories is not None else None
    actions = None
    alive = torch.ones(n, dtype=torch.uint8, device=dev)
    moves = torch.zeros(n, dtype=torch.int32, device=dev)
    valid_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    invalid_cnt = torch.zeros(n, dtype=torch.int32, device=dev)
    ms_move = torch.full((n, len(MILESTONES)), -1, dtype=torch.int32, device=dev)
    expanded_sum = torch.zeros(n, dtype=torch.int64, device=dev)
    t = 0
    graph = None
    if fused:
        res = ops.play_games(env.boards, env.scores, beam_width, search_depth, max_moves, early_game_threshold,
                             mid_game_threshold, seed, game_id_base, fixed_down, one_phase, tuning=tuning,
                             want_actions=histories is not None)
        alive, moves, valid_cnt, invalid_cnt = res["alive"], res["moves"], res["valid_moves"], res["invalid_moves"]
        ms_move, expanded_sum = res["milestone_move"], res["expanded"]
        actions = res.get("actions")
        t = max_moves
    elif use_graph:
        # One move = [keys_advance, beam, step (in place), track], all reading their RNG keys / move index from a device
        # key block, captured once into a hipGraph and replayed per move: no per-move host work besides the replay.
        kb = ops.KeyBlock(seed, 0, dev)
        out = (torch.empty(n, dtype=torch.uint8, device=dev), torch.empty(n, dtype=torch.float32, device=dev),
               torch.empty(n, dtype=torch.int32, device=dev))

        def one_move():
            kb.advance()
            ops.beam_get_action(env.boards, beam_width, search_depth, None, early_game_threshold, mid_game_threshold,
                                game_id_base=game_id_base, fixed_down=fixed_down, keyblock=kb, out=out)
            ops.step(env.boards, out[0], env.scores, 0, 0, game_id_base, out=env.boards, reward=env.reward, flags=env.flags,
                     keyblock=kb)
            ops.track_episodes(env.flags, alive, moves, valid_cnt, invalid_cnt, ms_move, 0, out[2], expanded_sum, keyblock=kb)
        # The warm-up move below really executes (capture does not), so the state is snapshotted first -- on the current
        # stream, BEFORE the side stream is made to wait for it, so the snapshot is ordered ahead of the warm-up -- and
        # restored afterwards whether or not the capture succeeded: the loop always starts from move 0.
        tracked = (env.boards, env.scores, alive, moves, valid_cnt, invalid_cnt, ms_move, expanded_sum, kb.counter)
        state = [x.clone() for x in tracked]
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                one_move()                      # warm-up outside capture
                side.synchronize()
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    one_move()
        except Exception as exc:                # noqa: BLE001 -- capture unavailable: plain launches, same games
            warnings.warn("g2048.evaluate_beam_search: hipGraph capture failed (%s: %s); falling back to one launch "
                          "sequence per move" % (type(exc).__name__, exc), RuntimeWarning, stacklevel=2)
            graph = None
            torch.cuda.synchronize(dev)
        cur.wait_stream(side)
        for dst, src in zip(tracked, state):
            dst.copy_(src)                      # rewind to move 0
    while t < max_moves:
        if graph is not None:
            graph.replay()
        else:
            actions, _, expanded = ops.beam_get_action(env.boards, beam_width, search_depth, None, early_game_threshold,
                                                       mid_game_threshold, seed, t, game_id_base, fixed_down,
                                                       want_expanded=True)
            env.step(actions)
            # evaluate_beam_search.py:42-64 for every game, one kernel
            ops.track_episodes(env.flags, alive, moves, valid_cnt, invalid_cnt, ms_move, t, expanded, expanded_sum)
        t += 1
        if t % check_every == 0 and not bool(alive.any()):
            break
    expanded_total = expanded_sum.sum()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t_start

    # one int64 row per game: score, moves, valid, invalid, alive, expanded, 8 milestone moves, 16 tiles
    table = torch.cat([env.scores.to(torch.int64)[:, None], moves.to(torch.int64)[:, None], valid_cnt.to(torch.int64)[:, None],
                       invalid_cnt.to(torch.int64)[:, None], alive.to(torch.int64)[:, None], expanded_sum[:, None],
                       ms_move.to(torch.int64), ops.unpack(env.boards).to(torch.int64).reshape(n, 16)], dim=1)
    if _table_only:
        return table.cpu().numpy()
    results = results_from_table(table.cpu().numpy(), elapsed, beam_width, search_depth, seed, max_moves)
    if histories is not None:
        results["games"] = game_histories(results, histories, boards0, actions, moves, seed, game_id_base)
    return results


def _select_games(results, which):
    n = len(results["scores"])
    if isinstance(which, str):
        if which == "best5":
            return list(results["best_games"])
        if which == "high_tile":                        # evaluate_beam_search.py:170: game_result['highest_tile'] >= 2048
            return [i for i, t in enumerate(results["highest_tiles"]) if t >= 2048]
        if which == "all":
            return list(range(n))
        raise ValueError("histories must be 'best5', 'high_tile', 'all' or an iterable of game indices")
    sel = [int(i) for i in which]
    if any(i < 0 or i >= n for i in sel):
        raise ValueError("histories: game index out of range")
    return sel


def game_histories(results, which, boards0, actions, moves, seed, game_id_base=0):
    """{game index: the dict evaluate_beam_search.run_game returns (evaluate_beam_search.py:86-97)} for the selected games,
    rebuilt on the device from their action bytes. boards0: the start boards of ALL games (uint8 (n,16)), actions: uint8
    (n, max_moves) as ops.play_games(want_actions=True) returned it, moves: int32 (n,)."""
    sel = _select_games(results, which)
    if not sel:
        return {}
    dev = boards0.device
    idx = torch.tensor(sel, dtype=torch.int64, device=dev)
    longest = int(moves.index_select(0, idx).max().item())
    bh, shist, fh = ops.replay_games(boards0.index_select(0, idx).contiguous(),
                                     actions.index_select(0, idx)[:, :max(longest, 1)].contiguous(),
                                     moves.index_select(0, idx).contiguous(), seed, game_ids=idx + int(game_id_base), longest=longest)
    k, hist = bh.shape[0], bh.shape[1]
    tiles = ops.unpack(bh.view(k * hist, 16)).view(k, hist, 4, 4).cpu().numpy()
    shist, fh = shist.cpu().numpy(), fh.cpu().numpy()
    acts = actions.index_select(0, idx)[:, :max(longest, 1)].cpu().numpy()
    out = {}
    for row, g in enumerate(sel):
        m = results["moves"][g]
        boards = tiles[row, :m + 1]
        ms = {t: None for t in MILESTONES}
        for t, v in results["milestones_by_game"][g].items():
            ms[t] = v
        out[g] = {
            "score": results["scores"][g], "highest_tile": results["highest_tiles"][g], "moves": m,
            "valid_moves": results["valid_moves"][g], "invalid_moves": results["invalid_moves"][g],
            "milestones": ms,
            "board_history": [b.copy() for b in boards],                       # :45 + :73: start, then the state after every move
            "max_tiles_history": [int(b.max()) for b in boards],               # :48 + :74
            "scores_history": [int(x) for x in shist[row, :m + 1]],            # :49 + :75 (starts at 0)
            "final_board": boards[m].copy(),
            "moveset": [int(a) for a in acts[row, :m]],                        # train.py:51,67
            "valid_history": [bool(f & L.FLAG_VALID) for f in fh[row, :m]],
        }
    return out


def save_moveset(game, path):
    """train.py:140-142: the move-set of one game (`results["games"][i]`, or any list of actions) as the reference writes
    `*_best_moveset_tile_N.txt` -- the actions joined by commas, no newline."""
    moveset = game["moveset"] if isinstance(game, dict) else game
    with open(path, "w") as f:
        f.write(",".join(map(str, moveset)))
    return path


def save_game_data(game, path):
    """evaluate_beam_search.py:185-196: game_{i}_data.json of one game -- the run_game dict with arrays as lists."""
    import numpy as np
    keys = ("score", "highest_tile", "moves", "valid_moves", "invalid_moves", "milestones", "board_history",
            "max_tiles_history", "scores_history", "final_board")
    out = {k: (game[k].tolist() if isinstance(game[k], np.ndarray) else game[k]) for k in keys}
    out["board_history"] = [b.tolist() if isinstance(b, np.ndarray) else b for b in game["board_history"]]
    out["milestones"] = {str(k): v for k, v in game["milestones"].items()}        # (json.dump turns the int keys into these strings)
    with open(path, "w") as f:
        json.dump(out, f)
    return path


TABLE_COLUMNS = 6 + len(MILESTONES) + 16


def results_from_table(table, elapsed, beam_width, search_depth, seed, max_moves):
    """The result dict from the per-game table (rows in global game order; columns as built in evaluate_beam_search)."""
    n = table.shape[0]
    scores = table[:, 0]
    final_boards = table[:, 14:30].reshape(n, 4, 4).astype("int32")
    highest = final_boards.reshape(n, 16).max(axis=1) if n else table[:, 0]
    ms_host = table[:, 6:14]
    order = sorted(range(n), key=lambda i: scores[i], reverse=True)       # stable, like the reference's top-5 update
    best = int(order[0]) if n else 0
    results = {
        "scores": [int(s) for s in scores],
        "highest_tiles": [int(h) for h in highest],
        "moves": [int(m) for m in table[:, 1]],
        "valid_moves": [int(m) for m in table[:, 2]],
        "invalid_moves": [int(m) for m in table[:, 3]],
        "milestones": {m: [int(v) for v in ms_host[:, k] if v >= 0] for k, m in enumerate(MILESTONES)},
        "milestones_by_game": [{m: int(ms_host[g, k]) for k, m in enumerate(MILESTONES) if ms_host[g, k] >= 0} for g in range(n)],
        "best_games": [int(i) for i in order[:5]],
        "final_boards": final_boards,
        "best_board": final_boards[best].copy() if n else None,
        "best_score": int(scores[best]) if n else 0,
        "best_game_idx": best,
        "unfinished": int(table[:, 4].sum()),
        "total_moves": int(table[:, 1].sum()),
        "total_expansions": int(table[:, 5].sum()),
        "elapsed_s": elapsed,
        "parameters": {"beam_width": beam_width, "search_depth": search_depth, "num_games": n, "seed": seed,
                       "max_moves": max_moves},
    }
    results["summary"] = summarize(results)
    return results


def evaluate_beam_search_sharded(num_games=4096, beam_width=20, search_depth=30, seed=0x2048, max_moves=5000, device=None,
                                 game_id_base=0, **kw):
    """The evaluation over all ranks of the default process group (one process per GPU, torchrun environment): games are
    independent and every draw is keyed by the global game id, so rank r plays the contiguous range g2048.dist.shard gives
    it with no communication, and one all-gather of the per-game table (30 int64 per game) at the end gives every rank the
    result of the whole evaluation -- identical to evaluate_beam_search(num_games) on one GPU. elapsed_s = slowest rank."""
    from . import dist as gdist
    if kw.get("histories") is not None:
        raise ValueError("evaluate_beam_search_sharded gathers the per-game table only; replay histories per rank with evaluate_beam_search")
    w, r, lr = gdist.world()
    dev = torch.device(device) if device is not None else torch.device("cuda", lr)
    lo, hi = gdist.shard(num_games, r, w)
    t0 = time.perf_counter()
    part = evaluate_beam_search(hi - lo, beam_width, search_depth, seed=seed, max_moves=max_moves, device=dev,
                                game_id_base=game_id_base + lo, _table_only=True, **kw)
    table = gdist.all_gather_rows(torch.from_numpy(part).to(dev))
    elapsed = gdist.max_over_ranks(time.perf_counter() - t0, dev)
    res = results_from_table(table.cpu().numpy(), elapsed, beam_width, search_depth, seed, max_moves)
    res["parameters"]["world_size"] = w
    return res


def summarize(results):
    """The numbers the reference prints / reports (run_evaluation.py:110-128, report.md)."""
    n = max(len(results["scores"]), 1)
    tiles = results["highest_tiles"]
    dist = {}
    for tile in tiles:
        dist[tile] = dist.get(tile, 0) + 1
    return {
        "games": len(results["scores"]),
        "highest_tile": max(tiles) if tiles else 0,
        "best_score": max(results["scores"]) if tiles else 0,
        "average_score": sum(results["scores"]) / n,
        "average_highest_tile": sum(tiles) / n,
        "rate_2048_or_more": sum(1 for x in tiles if x >= 2048) / n,
        "tile_distribution_pct": {int(k): 100.0 * v / n for k, v in sorted(dist.items())},
        "hit_move_cap": results.get("unfinished", 0),
        "moves_per_s": results["total_moves"] / results["elapsed_s"] if results.get("elapsed_s") else None,
        "expansions_per_s": results["total_expansions"] / results["elapsed_s"] if results.get("elapsed_s") else None,
    }


def save_overall_results(results, path):
    """overall_results.json with the reference's keys (evaluate_beam_search.py:198-214)."""
    p = results["parameters"]
    out = {
        "scores": results["scores"], "highest_tiles": results["highest_tiles"], "moves": results["moves"],
        "valid_moves": results["valid_moves"], "invalid_moves": results["invalid_moves"],
        "milestones": {str(k): v for k, v in results["milestones"].items()},
        "best_games": results["best_games"],
        "parameters": {"beam_width": p["beam_width"], "search_depth": p["search_depth"], "num_games": p["num_games"]},
    }
    with open(path, "w") as f:
        json.dump(out, f, indent=4)
    return path
