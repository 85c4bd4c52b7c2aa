"""f1: the batched evaluation driver -- games played to completion on the GPU -- against the
oracle playing the same games one by one with the same draw schedule (exact), and against the reference's
published quality numbers (report.md) statistically."""
import json
import warnings

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
MILESTONES = (64, 128, 256, 512, 1024, 2048, 4096, 8192)


@pytest.fixture(scope="module")
def g2048():
    import __graft_entry__ as ge
    ge.ensure_built()
    return ge.import_package()


def oracle_game(O, seed, gid, width, depth, max_moves):
    k0, k1 = O.rng_keys(seed, O.DOM_RESET, 0)
    b = O.env_reset(O.rng_draw(k0, k1, gid, 0), O.rng_draw(k0, k1, gid, 1))
    score, moves, valid_n, invalid_n = 0, 0, 0, 0
    ms = {m: None for m in MILESTONES}
    done = False
    while not done and moves < max_moves:
        a = O.beam_get_action(b, -1, width, depth, seed=seed, step_index=moves, game_id=gid)["action"]
        s0, s1 = O.rng_keys(seed, O.DOM_STEP, moves)
        b, score, r, done, v, hi = O.env_step(b, score, a, O.rng_draw(s0, s1, gid, 0))
        for m in MILESTONES:
            if b.max() >= m and ms[m] is None:
                ms[m] = moves
        valid_n += int(v); invalid_n += int(not v)
        moves += 1
    return dict(score=score, moves=moves, valid=valid_n, invalid=invalid_n, board=b.reshape(4, 4), ms=ms)


def test_evaluation_driver_exact_vs_oracle(g2048, oracle, tmp_path):
    n, w, d, cap, seed = 48, 3, 4, 300, 0xBEEF
    res = g2048.evaluate_beam_search(n, w, d, seed=seed, max_moves=cap, game_id_base=500, check_every=16)
    for g in range(n):
        ref = oracle_game(oracle, seed, 500 + g, w, d, cap)
        assert res["scores"][g] == ref["score"] and res["moves"][g] == ref["moves"], g
        assert res["valid_moves"][g] == ref["valid"] and res["invalid_moves"][g] == ref["invalid"], g
        assert np.array_equal(res["final_boards"][g], ref["board"]), g
        assert res["highest_tiles"][g] == ref["board"].max()
    assert res["best_games"] == sorted(range(n), key=lambda i: res["scores"][i], reverse=True)[:5]
    assert res["best_score"] == max(res["scores"]) and res["best_game_idx"] == res["best_games"][0]
    from g2048.evaluate import save_overall_results
    p = save_overall_results(res, str(tmp_path / "overall_results.json"))
    js = json.load(open(p))
    assert set(js) == {"scores", "highest_tiles", "moves", "valid_moves", "invalid_moves", "milestones", "best_games",
                       "parameters"}
    assert js["parameters"] == {"beam_width": w, "search_depth": d, "num_games": n}
    assert set(js["milestones"]) == {str(m) for m in MILESTONES}


def test_milestones_exact(g2048, oracle):
    n, w, d, cap, seed = 12, 4, 6, 400, 77
    res = g2048.evaluate_beam_search(n, w, d, seed=seed, max_moves=cap, check_every=8)
    refs = [oracle_game(oracle, seed, g, w, d, cap) for g in range(n)]
    for m in MILESTONES:
        assert res["milestones"][m] == [r["ms"][m] for r in refs if r["ms"][m] is not None], m


def test_quality_matches_reference_report(g2048):
    """report.md (reference, beam w=20 d=30, 100 games): 35% of games reach >= 2048, average score 18,945.6, average
    highest tile 1,315.8, ~4% hit the 5000-move cap. 1024 games here. Band for each statistic: the published value
    +- (4 standard errors of this 1024-game sample + 2 standard errors of the reference's own 100-game sample, both
    from this sample's per-game spread) -- the published numbers are themselves one draw of a 100-game experiment."""
    res = g2048.evaluate_beam_search(1024, 20, 30, seed=2025, max_moves=5000)
    s = res["summary"]
    print("quality:", json.dumps({k: v for k, v in s.items() if k != "tile_distribution_pct"}), s["tile_distribution_pct"])
    n = 1024

    def band(sd):
        return 4.0 * sd / np.sqrt(n) + 2.0 * sd / np.sqrt(100.0)

    p_ref = 0.35
    assert abs(s["rate_2048_or_more"] - p_ref) <= band(np.sqrt(p_ref * (1 - p_ref))), s["rate_2048_or_more"]
    assert abs(s["average_score"] - 18945.6) <= band(np.std(res["scores"])), s["average_score"]
    assert abs(s["average_highest_tile"] - 1315.8) <= band(np.std(res["highest_tiles"])), s["average_highest_tile"]
    assert s["highest_tile"] >= 2048
    cap_rate = s["hit_move_cap"] / n               # report.md: "~4%" of 100 games
    assert abs(cap_rate - 0.04) <= band(np.sqrt(0.04 * 0.96)), cap_rate


def test_dyn_entry_points_equal_scalar_forms(g2048):
    """The *_dyn entry points (keys from the device key block) == the scalar forms at step_index = counter."""
    from g2048 import ops
    dev_ = "cuda:0"
    n, seed = 30000, 0xABCDEF
    b = ops.synth_boards(n, seed=3, device=dev_)
    a = ops.synth_actions(n, seed=3, step_index=0, device=dev_)
    kb = ops.KeyBlock(seed, start=41, device=dev_)
    for t in (41, 42, 43):
        kb.advance()
        sc1 = torch.zeros(n, dtype=torch.int32, device=dev_); sc2 = torch.zeros_like(sc1)
        o1, r1, f1 = ops.step(b, a, sc1, seed, t, id_base=7, reward_f64=True, auto_reset=True)
        o2, r2, f2 = ops.step(b, a, sc2, 0, 0, id_base=7, reward_f64=True, auto_reset=True, keyblock=kb)
        assert bool((o1 == o2).all()) and bool((f1 == f2).all()) and bool((sc1 == sc2).all())
        assert np.array_equal(r1.cpu().numpy(), r2.cpu().numpy(), equal_nan=True)
        a1, p1, e1 = ops.beam_get_action(b[:512], 20, 30, seed=seed, step_index=t, game_id_base=5, want_expanded=True)
        a2, p2, e2 = ops.beam_get_action(b[:512], 20, 30, game_id_base=5, want_expanded=True, keyblock=kb)
        assert bool((a1 == a2).all()) and bool((p1 == p2).all()) and bool((e1 == e2).all())
        probs = torch.softmax(torch.randn(n, 4, device=dev_), 1)
        m = ops.valid_moves(b)
        s1, q1 = ops.sample_actions(probs, m, seed=seed, step_index=t, id_base=9)
        s2, q2 = ops.sample_actions(probs, m, id_base=9, keyblock=kb)
        assert bool((s1 == s2).all()) and bool((q1 == q2).all())
    assert int(kb.counter.item()) == 44 and int(kb.words[8].item()) == 43


def test_graph_replayed_driver_equals_plain_driver(g2048):
    """fused=False on both sides: the hipGraph-replayed move loop against plain per-move launches."""
    kw = dict(num_games=96, beam_width=6, search_depth=8, seed=99, max_moves=500, check_every=16, fused=False)
    r0 = g2048.evaluate_beam_search(use_graph=False, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("error")              # a capture fallback would make this test vacuous
        r1 = g2048.evaluate_beam_search(use_graph=True, **kw)
    for k in ("scores", "highest_tiles", "moves", "valid_moves", "invalid_moves", "milestones", "best_games", "total_expansions"):
        assert r0[k] == r1[k], k
    assert np.array_equal(r0["final_boards"], r1["final_boards"])


def test_graph_capture_failure_falls_back_to_the_same_games(g2048, monkeypatch):
    """A failed capture happens AFTER the warm-up move has run: the driver must warn, rewind to move 0 and play the
    same games with plain launches."""
    kw = dict(num_games=64, beam_width=5, search_depth=6, seed=5, max_moves=300, check_every=16, fused=False)
    r0 = g2048.evaluate_beam_search(use_graph=False, **kw)

    class Boom:
        def __init__(self, *a, **k):
            raise RuntimeError("capture refused (test)")

    monkeypatch.setattr(torch.cuda, "graph", Boom)
    with pytest.warns(RuntimeWarning, match="capture failed"):
        r1 = g2048.evaluate_beam_search(use_graph=True, **kw)
    for k in ("scores", "highest_tiles", "moves", "valid_moves", "invalid_moves", "milestones", "total_expansions"):
        assert r0[k] == r1[k], k
    assert np.array_equal(r0["final_boards"], r1["final_boards"])


def test_fused_play_games_equals_stepwise_drivers(g2048):
    """g2048_play_games (each wavefront plays its whole game in one launch) == the step-by-step drivers."""
    for w, d in ((6, 8), (20, 30)):
        kw = dict(num_games=64 if w == 20 else 160, beam_width=w, search_depth=d, seed=31 + w, max_moves=700 if w == 6 else 5000,
                  check_every=32, game_id_base=(1 << 36) + 5)
        r0 = g2048.evaluate_beam_search(fused=False, use_graph=False, **kw)
        r1 = g2048.evaluate_beam_search(fused=True, **kw)
        for k in ("scores", "highest_tiles", "moves", "valid_moves", "invalid_moves", "milestones", "best_games",
                  "total_expansions", "unfinished"):
            assert r0[k] == r1[k], (w, k)
        assert np.array_equal(r0["final_boards"], r1["final_boards"])


PLAY_KEYS = ("scores", "highest_tiles", "moves", "valid_moves", "invalid_moves", "milestones", "best_games",
             "total_expansions", "unfinished")


@pytest.mark.parametrize("n,w,d,cap,fixed_down", [
    (1, 20, 12, 900, False),        # one game: four helpers from the first move
    (37, 8, 6, 900, False),         # one 64-child pass per level
    (300, 20, 10, 1500, False),     # two passes; 300 > 256 so the late registration path (few games left) runs too
    (24, 40, 4, 400, False),        # four passes
    (6, 100, 3, 150, False),        # eight passes
    (64, 20, 8, 700, True),         # true DOWN: no stuck games, the helpers serve valid-move successors only
])
def test_speculative_helpers_play_the_same_games(g2048, n, w, d, cap, fixed_down):
    """g2048_play_games with helper wavefronts (decisions for the next moves' possible roots searched ahead of time)
    == every game on its one wavefront (G2048_PLAY_ONE_PHASE): helpers may only change the time."""
    kw = dict(num_games=n, beam_width=w, search_depth=d, seed=4242 + n, max_moves=cap, game_id_base=(1 << 33) + 9,
              fixed_down=fixed_down)
    r0 = g2048.evaluate_beam_search(one_phase=True, **kw)
    for rep in range(2):                                   # helper timing differs run to run, the games may not
        r1 = g2048.evaluate_beam_search(**kw)
        for k in PLAY_KEYS:
            assert r0[k] == r1[k], (k, rep)
        assert np.array_equal(r0["final_boards"], r1["final_boards"])
    if not fixed_down and n >= 37:
        assert sum(r0["invalid_moves"]) > 0                # the repeated-root chain was exercised


def test_speculative_helpers_tuning_extremes(g2048, monkeypatch):
    """No helper at all, helpers that are always late (0 us wait) and eager registration give the same games (explicit
    tuning through g2048_play_games_tuned; out-of-range values are clamped; the library reads no environment variable)."""
    kw = dict(num_games=96, beam_width=12, search_depth=8, seed=99, max_moves=1200)
    r0 = g2048.evaluate_beam_search(one_phase=True, **kw)
    monkeypatch.setenv("G2048_PLAY_TUNE", "0,0,0,0")      # round 2's hook: must be ignored now
    for tune in ((0, 256, 16, 60), (384, 256, 16, 0), (64, 1000000, 1, 200), (2048, 0, 1000000, 60), (1 << 31, 7, 0, 1 << 31)):
        r1 = g2048.evaluate_beam_search(tuning=tune, **kw)
        for k in PLAY_KEYS:
            assert r0[k] == r1[k], (tune, k)
        assert np.array_equal(r0["final_boards"], r1["final_boards"])


def test_play_games_workspace_entry_points(g2048):
    """g2048_play_games_ws with caller scratch (what ops.play_games uses), with no scratch (= no helpers), with too
    little (refused), and g2048_play_games allocating its own: all the same games."""
    from g2048 import ops, _lib as L
    dev = torch.device("cuda")
    n, w, d, cap, seed = 40, 12, 6, 500, 606

    def fresh():
        b, s = ops.reset(n, seed, 0, 0, device=dev)
        out = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(3)]
        ms = torch.full((n, 8), -1, dtype=torch.int32, device=dev)
        ex = torch.zeros(n, dtype=torch.int64, device=dev)
        al = torch.zeros(n, dtype=torch.uint8, device=dev)
        ac = torch.zeros((n, cap), dtype=torch.uint8, device=dev)      # (the library fills it with 0xFF itself)
        return b, s, out, ms, ex, al, ac

    def args(t):
        b, s, out, ms, ex, al, ac = t
        return (b.data_ptr(), s.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), ms.data_ptr(), ex.data_ptr(),
                al.data_ptr(), ac.data_ptr(), w, d, 512, 1024, cap, L.u64(seed), L.u64(0), n, 0)

    need = int(L.lib().g2048_play_games_workspace(n))
    assert need >= n * 8 * 64 and int(L.lib().g2048_play_games_workspace(0)) == 0
    assert int(L.lib().g2048_play_games_workspace((1 << 16) + 1)) == 0          # beyond 65,536 games: no helpers
    results = []
    for mode in ("ws", "null", "own"):
        t = fresh()
        if mode == "ws":
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            L.call(dev, L.lib().g2048_play_games_ws, *args(t), ws.data_ptr(), need, L.stream_ptr(dev))
        elif mode == "null":
            L.call(dev, L.lib().g2048_play_games_ws, *args(t), None, 0, L.stream_ptr(dev))
        else:
            L.call(dev, L.lib().g2048_play_games, *args(t), L.stream_ptr(dev))
        torch.cuda.synchronize()
        b, s, out, ms, ex, al, ac = t
        results.append([x.cpu() for x in (b, s, *out, ms, ex, al, ac)])
    for r in results[1:]:
        assert all(torch.equal(x, y) for x, y in zip(results[0], r))
    moves, ac = results[0][2], results[0][-1]                       # the action stream: 0..3 up to the game's end, 0xFF after it
    col = torch.arange(cap)[None, :]
    assert bool((ac[col < moves[:, None]] <= 3).all()) and bool((ac[col >= moves[:, None]] == 0xFF).all())
    t = fresh()
    small = torch.empty(need - 64, dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="workspace"):
        L.call(dev, L.lib().g2048_play_games_ws, *args(t), small.data_ptr(), need - 64, L.stream_ptr(dev))
    with pytest.raises(RuntimeError, match="workspace"):
        L.call(dev, L.lib().g2048_play_games_ws, *args(t), small.data_ptr() + 16, need, L.stream_ptr(dev))      # misaligned


def test_complete_evaluations_network_vs_counting(g2048):
    """Millions of decisions: 1536 complete games (w=20, d=30) ranked by the sorting networks and by the counting loop end
    in the same boards, scores and move counts (this comparison is what caught a rare mis-ranking of scores a few ulp
    apart; tests/test_gpu_beam.py holds that decision)."""
    from g2048 import ops
    from g2048.vec import VecGame2048
    total = 0
    for n, w, d in ((1536, 20, 30), (1024, 12, 10), (512, 32, 8)):        # two passes with a tail row / one pass / up to 128 children
        res = []
        for rbc in (False, True):
            env = VecGame2048(n, device=torch.device("cuda"), seed=2025 + w)
            r = ops.play_games(env.boards, env.scores, w, d, 5000, 512, 1024, 2025 + w, 0, False, False, rank_by_counting=rbc)
            res.append((env.boards.cpu(), env.scores.cpu(), r["moves"].cpu(), r["invalid_moves"].cpu(), r["expanded"].cpu()))
        assert all(torch.equal(x, y) for x, y in zip(*res)), (n, w, d)
        total += int(res[0][2].sum())
    assert total > 2_500_000


def test_reference_configuration_vs_oracle_400_moves(g2048, oracle):
    """The reference's evaluation configuration itself (width 20, depth 30) against the oracle playing game by game: 24
    games, the first 400 moves of each (about ten thousand decisions through both sorting networks and the helpers)."""
    n, w, d, cap, seed = 24, 20, 30, 400, 2025
    res = g2048.evaluate_beam_search(n, w, d, seed=seed, max_moves=cap, game_id_base=460)
    for g in range(n):
        ref = oracle_game(oracle, seed, 460 + g, w, d, cap)
        assert res["scores"][g] == ref["score"] and res["moves"][g] == ref["moves"], g
        assert res["invalid_moves"][g] == ref["invalid"] and np.array_equal(res["final_boards"][g], ref["board"]), g


def test_complete_games_vs_oracle_reference_configuration(g2048):
    """96 complete games at width 20 / depth 30 / 5000-move cap against the oracle playing them one by one on the host cores
    (16 worker processes that never touch the GPU; tools/oracle_full_games.py ran 2048 games the same way in round 2)."""
    import os
    import sys
    from multiprocessing import get_context
    from conftest import REPO
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import oracle_full_games as T
    n = 96
    with get_context("spawn").Pool(16) as pool:
        job = pool.map_async(T.oracle_game_with_history, range(n), chunksize=1)
        res = g2048.evaluate_beam_search(n, T.W, T.D, seed=T.SEED, max_moves=T.CAP, game_id_base=0, histories="all")
        plain = g2048.evaluate_beam_search(n, T.W, T.D, seed=T.SEED, max_moves=T.CAP, game_id_base=0)      # no action stream asked for
        ref = job.get(timeout=280)
    for k in PLAY_KEYS:
        assert res[k] == plain[k], k
    assert set(res["games"]) == set(range(n))
    for gid, score, moves, invalid, board, actions, codes, scores in ref:
        assert res["scores"][gid] == score and res["moves"][gid] == moves and res["invalid_moves"][gid] == invalid, gid
        assert [int(x) for x in res["final_boards"][gid].reshape(-1)] == board, gid
        # f1, the rest of run_game's result (evaluate_beam_search.py:44-50, :72-75): the move-set the fused kernel recorded and
        # the per-move histories replayed from it, against the oracle's own record of the same game
        game = res["games"][gid]
        assert game["moveset"] == list(actions), gid
        assert len(game["board_history"]) == moves + 1 == len(game["scores_history"]) == len(game["max_tiles_history"]), gid
        want = np.where(codes > 0, 1 << codes.astype(np.int64), 0).reshape(moves + 1, 4, 4)
        assert np.array_equal(np.stack(game["board_history"]), want), gid
        assert game["scores_history"] == [int(x) for x in scores], gid
        assert game["max_tiles_history"] == [int(x) for x in want.reshape(moves + 1, 16).max(axis=1)], gid
        assert game["scores_history"][0] == 0 and game["scores_history"][-1] == score and np.array_equal(game["final_board"], want[-1])
        assert game["milestones"] == {m: (res["milestones_by_game"][gid].get(m)) for m in MILESTONES}


def test_action_stream_does_not_depend_on_helpers_or_tuning(g2048, tmp_path):
    """The recorded move-sets (and so the replayed histories) are the same with helper wavefronts, without them, with the
    tuning forced to its extremes and with the counting ranking; the two file formats of the reference are written from them:
    train.py:140-142 (*_best_moveset_tile_N.txt) and evaluate_beam_search.py:185-196 (game_N_data.json)."""
    from g2048 import ops, save_moveset, save_game_data
    from g2048.vec import VecGame2048
    n, w, d, cap, seed = 80, 20, 8, 900, 777
    streams = []
    for kw in (dict(one_phase=True), dict(), dict(tuning=(0, 256, 16, 60)), dict(tuning=(64, 1000000, 1, 200)),
               dict(tuning=(2048, 0, 1000000, 0)), dict(rank_by_counting=True)):
        env = VecGame2048(n, device=torch.device("cuda"), seed=seed, id_base=17)
        r = ops.play_games(env.boards, env.scores, w, d, cap, seed=seed, game_id_base=17, want_actions=True, **kw)
        streams.append((r["actions"].cpu(), r["moves"].cpu(), env.boards.cpu()))
    for a, m, b in streams[1:]:
        assert torch.equal(a, streams[0][0]) and torch.equal(m, streams[0][1]) and torch.equal(b, streams[0][2])
    a, m = streams[0][0], streams[0][1]
    col = torch.arange(cap)[None, :]
    assert bool((a[col < m[:, None]] <= 3).all()) and bool((a[col >= m[:, None]] == 0xFF).all())
    res = g2048.evaluate_beam_search(n, w, d, seed=seed, max_moves=cap, game_id_base=17, histories="best5")
    assert sorted(res["games"]) == sorted(res["best_games"])
    g = res["best_games"][0]
    game = res["games"][g]
    assert game["moveset"] == [int(x) for x in a[g, :int(m[g])]]
    p = save_moveset(game, str(tmp_path / ("BeamSearchAgent_best_moveset_tile_%d.txt" % game["highest_tile"])))
    txt = open(p).read()
    assert txt == ",".join(str(x) for x in game["moveset"]) and not txt.endswith("\n")
    js = json.load(open(save_game_data(game, str(tmp_path / ("game_%d_data.json" % (g + 1))))))
    assert set(js) == {"score", "highest_tile", "moves", "valid_moves", "invalid_moves", "milestones", "board_history",
                       "max_tiles_history", "scores_history", "final_board"}
    assert js["moves"] == game["moves"] and len(js["board_history"]) == game["moves"] + 1 and js["final_board"] == js["board_history"][-1]
    assert set(js["milestones"]) == {str(t) for t in MILESTONES}
    # a replay of a recorded move-set through the public step API ends in the same board (the reference's move-set files are
    # exactly such streams: include/g2048.h, g2048_step_many)
    acts = torch.tensor(game["moveset"], dtype=torch.uint8, device="cuda")[:, None].contiguous()
    b0, s0 = ops.reset(1, seed, 0, 17 + g, device=torch.device("cuda"))
    out, fl, _, _, _ = ops.step_many(b0, s0, seed, 0, len(game["moveset"]), 17 + g, actions=acts)
    assert np.array_equal(ops.unpack(out).cpu().numpy().reshape(4, 4), game["final_board"]) and int(s0.item()) == game["score"]
    hi = g2048.evaluate_beam_search(n, w, d, seed=seed, max_moves=cap, game_id_base=17, histories="high_tile")
    assert sorted(hi["games"]) == [i for i, t in enumerate(hi["highest_tiles"]) if t >= 2048]
    with pytest.raises(ValueError):
        g2048.evaluate_beam_search(4, 3, 3, max_moves=10, fused=False, histories="all")


def test_device_plan_reports_what_a_launch_will_use():
    """g2048_device_plan: the occupancy-derived numbers behind a launch (helper cap = a quarter of the evaluation kernel's
    resident blocks, default helpers within it; a beam batch runs with issue priority iff it is resident at once)."""
    import __graft_entry__ as ge
    ge.import_package()
    from g2048 import ops
    p = ops.device_plan(20, 4096)
    assert p["compute_units"] > 0 and p["play_resident_blocks_per_cu"] > 0
    assert p["helper_cap"] == max(p["compute_units"] * p["play_resident_blocks_per_cu"] // 4, 1)
    assert p["default_helpers"] == min(8 * 4096, max(4096 // 2, 1024), p["helper_cap"])
    assert p["beam_resident_blocks"] % p["compute_units"] == 0 and p["beam_resident_blocks"] >= 4 * p["compute_units"]
    assert p["beam_issue_priority"] == (4096 <= p["beam_resident_blocks"])
    big = ops.device_plan(20, p["beam_resident_blocks"] + 1)
    assert not big["beam_issue_priority"]
    assert ops.device_plan(128, 64)["play_resident_blocks_per_cu"] <= p["play_resident_blocks_per_cu"]


def test_replay_of_every_game_at_full_size(g2048):
    """BASELINE config 3's size: 4096 complete games (w=20, d=30), the action stream of ALL of them replayed on the device.
    Size-independent properties: entry moves[g] of a game's replayed history is the board / score the fused kernel ended with,
    the flags byte of its last move says done unless the game hit the cap, scores never decrease, the tile sum grows by the
    spawned 2 or 4 exactly on valid moves, and the counts of valid / invalid flags equal the kernel's counters."""
    from g2048 import ops, _lib as L
    from g2048.vec import VecGame2048
    n, cap, seed = 4096, 5000, 2025
    env = VecGame2048(n, device=torch.device("cuda"), seed=seed)
    b0 = env.boards.clone()
    r = ops.play_games(env.boards, env.scores, 20, 30, cap, seed=seed, want_actions=True)
    moves = r["moves"]
    bh, sh, fh = ops.replay_games(b0, r["actions"], moves, seed)
    idx = moves.to(torch.int64)
    rows = torch.arange(n, device=bh.device)
    assert torch.equal(bh[rows, idx], env.boards) and torch.equal(sh[rows, idx], env.scores)
    last = fh[rows, (idx - 1).clamp(min=0)]
    done = (last & L.FLAG_DONE).bool()
    assert torch.equal(done, r["alive"] == 0)
    t = torch.arange(fh.shape[1], device=bh.device)[None, :]
    live = t < idx[:, None]
    valid = ((fh & L.FLAG_VALID) != 0) & live
    assert torch.equal(valid.sum(dim=1).to(torch.int32), r["valid_moves"])
    assert torch.equal((live & ~valid).sum(dim=1).to(torch.int32), r["invalid_moves"])
    assert bool((sh[:, 1:] >= sh[:, :-1])[live[:, :-1]].all())
    tiles = torch.where(bh > 0, torch.ones_like(bh, dtype=torch.int32) << bh.to(torch.int32), torch.zeros_like(bh, dtype=torch.int32)).sum(dim=2)
    diff = (tiles[:, 1:] - tiles[:, :-1])[:, :fh.shape[1] - 1]
    v = valid[:, :fh.shape[1] - 1]
    lv = live[:, :fh.shape[1] - 1]
    assert bool(((diff == 2) | (diff == 4))[v].all()) and bool((diff == 0)[lv & ~v].all())
    assert int(r["alive"].sum()) > 0 and int((moves == cap).sum()) >= int(r["alive"].sum())


def test_complete_games_of_the_real_reference(g2048):
    """tests/golden/games.npz (the REAL reference's env + agent in run_game's loop, evaluate_beam_search.py:29-98): the fused
    evaluation plays the same games -- counters, final boards -- and the recorded move-sets and replayed histories are the
    reference's moveset / board_history / scores_history / max_tiles_history / milestones, move by move."""
    from conftest import load_golden, tiles_of
    g = load_golden("games.npz")
    seed, meta = int(g["seed"]), g["meta"]
    k = 0
    while k < meta.shape[0]:
        w, d, cap, gid0 = (int(x) for x in meta[k, :4])
        group = [j for j in range(meta.shape[0]) if tuple(meta[j, :3]) == tuple(meta[k, :3])]
        assert [int(meta[j, 3]) for j in group] == list(range(gid0, gid0 + len(group)))
        res = g2048.evaluate_beam_search(len(group), w, d, seed=seed, max_moves=cap, game_id_base=gid0, histories="all")
        for i, j in enumerate(group):
            _, _, _, gid, moves, valid_n, invalid_n, score, done = (int(x) for x in meta[j])
            assert (res["moves"][i], res["valid_moves"][i], res["invalid_moves"][i], res["scores"][i]) == (moves, valid_n, invalid_n, score), j
            game = res["games"][i]
            assert game["moveset"] == g["g%d_moveset" % j].tolist(), j
            assert np.array_equal(np.stack(game["board_history"]).reshape(moves + 1, 16), tiles_of(g["g%d_boards" % j])), j
            assert game["scores_history"] == g["g%d_scores" % j].tolist() and game["max_tiles_history"] == g["g%d_max_tiles" % j].tolist(), j
            want_ms = {m: (int(v) if v >= 0 else None) for m, v in zip(MILESTONES, g["g%d_milestones" % j])}
            assert game["milestones"] == want_ms, j
            assert not done or res["final_boards"][i].min() > 0                   # a finished game ends on a full board
        k = group[-1] + 1


@pytest.mark.timeout(400, method="thread")
@pytest.mark.parametrize("n,w,d,cap", [(9000, 4, 3, 60), (20000, 3, 3, 50), (65536, 2, 2, 40), (65537, 2, 2, 40)])
def test_helpers_between_4096_games_and_the_cut_off(g2048, n, w, d, cap):
    """The fused evaluation (run_evaluation.py:48-69 per game) at the batch sizes nothing else runs: launches far larger than
    the chip holds at once (owners that start late, helper blocks dispatched last) up to 65,536 games, the last size with
    helper wavefronts, and 65,537, the first without (g2048_play_games_workspace = 0 there). Tiny searches and a short move
    cap keep it to seconds. With helpers == G2048_PLAY_ONE_PHASE: final boards, scores, every counter, milestone and action
    byte."""
    from g2048 import ops, _lib as L
    dev = torch.device("cuda")
    ws_bytes = int(L.lib().g2048_play_games_workspace(n))
    assert (ws_bytes > 0) == (n <= 65536)
    runs = []
    for one_phase in (True, False):
        b, s = ops.reset(n, 515, 0, 3 << 34, device=dev)
        r = ops.play_games(b, s, w, d, max_moves=cap, seed=515, game_id_base=3 << 34, one_phase=one_phase, want_actions=True)
        torch.cuda.synchronize()
        runs.append((b, s, r))
    (b0, s0, r0), (b1, s1, r1) = runs
    assert torch.equal(b0, b1) and torch.equal(s0, s1)
    for k in ("moves", "valid_moves", "invalid_moves", "milestone_move", "expanded", "alive", "actions"):
        assert torch.equal(r0[k], r1[k]), k
    moves = r0["moves"]
    assert int(moves.max()) == cap and int(moves.min()) >= 1 and int(r0["expanded"].sum()) > n        # games were really played
