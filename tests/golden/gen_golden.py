#!/usr/bin/env python3
"""Golden-vector generator: runs the REFERENCE itself and records inputs -> outputs.

Usage (build container only; /root/reference never travels to the GPU box):

    python tests/golden/gen_golden.py /root/reference

It imports the reference's environment/game_2048.py, agents/beam_search_agent.py
and agents/ppo_agent.py read-only via sys.path (nothing is copied), replaces the
three `random` entry points the hot path draws from (random.choice / random.random
/ random.randint) with a recorded stream of 32-bit draws `h`, and writes the
fixtures in this directory (*.npz; data only). The mapping from a draw to a spawn
decision is the build's own (DESIGN.md "RNG"):

    index = ((h >> 16) * n) >> 16        is4 = (h & 0xFFFF) >= 58982

so a fixture row is fully explicit: (board, action, h) -> what the reference did.
Boards are stored as uint8 log2 codes (0 = empty) to keep the files small.

The draws themselves come from the oracle's counter RNG (oracle/oracle.py) purely
as a convenient deterministic source; nothing in a fixture depends on how `h` was
produced, except rng_pin.npz which pins the RNG's own outputs (a regression pin,
not a reference-derived vector).
"""
import os
import random
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
from oracle import oracle as O  # noqa: E402  (draw source + draw->decision mapping only)

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, REF)
from environment.game_2048 import Game2048Env  # noqa: E402
import agents.beam_search_agent as bsa  # noqa: E402
from agents.beam_search_agent import BeamSearchAgent  # noqa: E402

warnings.filterwarnings("ignore")
SEED = 0x2048


# ----------------------------------------------------------------------------
class DrawStream:
    """Feeds recorded 32-bit draws to the reference through the `random` module."""

    def __init__(self):
        self.source = None      # callable() -> next h
        self.cur = 0
        self.consumed = 0

    def _next(self):
        self.cur = int(self.source())
        self.consumed += 1
        return self.cur

    def choice(self, seq):                      # env add_new_tile (:64), agent fallback (:128)
        return seq[O.draw_index(self._next(), len(seq))]

    def randint(self, a, b):                    # agent _add_random_tile (:265)
        return a + O.draw_index(self._next(), b - a + 1)

    def rand(self):                             # the 2-or-4 draw that follows (:67 / :269)
        return 0.95 if O.draw_is4(self.cur) else 0.05


STREAM = DrawStream()
_default_ctr = [0]


def _default_source():              # constructor-time reset() draws nobody records
    _default_ctr[0] += 1
    return (_default_ctr[0] * 2654435761) & 0xFFFFFFFF


STREAM.source = _default_source
random.choice = STREAM.choice
random.randint = STREAM.randint
random.random = STREAM.rand


def list_source(values):
    it = iter(values)

    def src():
        try:
            return next(it)
        except StopIteration:       # only constructor-time reset() draws ever land here
            return _default_source()
    return src


def codes_of(tiles):
    t = np.asarray(tiles, dtype=np.int64).reshape(-1)
    c = np.zeros(t.shape, dtype=np.uint8)
    nz = t > 0
    c[nz] = np.round(np.log2(t[nz])).astype(np.uint8)
    assert np.array_equal(np.where(c > 0, 1 << c.astype(np.int64), 0), t)
    return c


def tiles_of(codes):
    c = np.asarray(codes, dtype=np.int64)
    return np.where(c > 0, 1 << c, 0).astype(np.int32)


def set_env(env, tiles, score=0):
    env.board = np.array(tiles, dtype=np.int32).reshape(4, 4).copy()
    env.score = score
    env.game_over = False
    env.highest_tile = np.max(env.board)


def hashed(domain, index, ident, ctr=0):
    k0, k1 = O.rng_keys(SEED, domain, index)
    return O.rng_draw(k0, k1, ident, ctr)


# ------------------------------------------------------------ board pools ---
def random_code_boards(rng, n, p_empty, max_code):
    b = rng.integers(1, max_code + 1, size=(n, 16)).astype(np.uint8)
    b[rng.random((n, 16)) < p_empty] = 0
    return b


def selfplay_boards(n_games, policy_rng, greedy=False, tag=0):
    """States visited by the reference env under random (or greedy-merge) play."""
    out = []
    env = Game2048Env()
    for g in range(n_games):
        t = [0]
        STREAM.source = lambda: hashed(O.DOM_STEP, t[0], 1000 * tag + g, 7)
        env.reset()
        for step in range(4000):
            t[0] = step + 1
            out.append(codes_of(env.board))
            vm = env.get_valid_moves()
            if not any(vm):
                break
            if greedy:
                best, best_gain = None, -1
                for a in range(4):
                    if vm[a]:
                        nb, gain = O.env_move(env.board.reshape(-1), a)
                        key = gain * 16 + int((nb == 0).sum())
                        if key > best_gain:
                            best, best_gain = a, key
                a = best
            else:
                a = int(policy_rng.integers(0, 4))
            env.step(a)
    return np.array(out, dtype=np.uint8)


def edge_boards():
    e = []
    e.append(np.zeros(16, np.uint8))                                   # all empty
    e.append(np.array([1, 2, 1, 2, 2, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2, 1], np.uint8))   # dead board
    e.append(np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16], np.uint8))
    e.append(np.array([17, 17, 16, 16, 1, 1, 1, 1, 2, 2, 2, 0, 3, 0, 3, 0], np.uint8))  # big merges
    e.append(np.array([1, 1, 1, 1] * 4, np.uint8))                      # everything merges
    e.append(np.array([2, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], np.uint8))
    e.append(np.array([1, 0, 0, 0] + [0] * 12, np.uint8))
    e.append(np.array([0] * 15 + [1], np.uint8))
    e.append(np.array([1, 2, 3, 4, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0], np.uint8))   # LEFT/RIGHT invalid
    e.append(np.array([1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0], np.uint8))   # UP/DOWN invalid... col
    e.append(np.array([1, 2, 1, 2, 2, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2, 2], np.uint8))   # one merge left
    e.append(np.array([1, 2, 1, 2, 2, 1, 2, 1, 1, 2, 1, 2, 2, 1, 2, 0], np.uint8))   # one empty
    e.append(np.array([11, 10, 9, 8, 4, 5, 6, 7, 3, 2, 1, 1, 0, 0, 0, 0], np.uint8))  # snake-ish late
    e.append(np.array([9, 8, 7, 6, 2, 3, 4, 5, 1, 1, 0, 0, 0, 0, 0, 0], np.uint8))   # mid
    e.append(np.array([10, 9, 8, 7, 3, 4, 5, 6, 2, 1, 2, 1, 1, 2, 1, 2], np.uint8))  # late, full
    return np.array(e, dtype=np.uint8)


# ----------------------------------------------------------------- fixtures -
def gen_row_slide():
    """All 18^4 code rows through the reference's _move_left (exhaustive)."""
    env = Game2048Env()
    codes = np.array(np.meshgrid(*[np.arange(18)] * 4, indexing="ij")).reshape(4, -1).T.astype(np.uint8)
    out = np.zeros_like(codes)
    gain = np.zeros(codes.shape[0], dtype=np.int32)
    for i in range(0, codes.shape[0], 4):
        rows = codes[i:i + 4]
        set_env(env, tiles_of(rows).reshape(16), 0)
        env._move_left()
        res = codes_of(env.board).reshape(4, 4)
        out[i:i + 4] = res
        # per-row gain: re-run each row alone
        for r in range(4):
            t = np.zeros((4, 4), np.int32)
            t[0] = tiles_of(rows[r])
            set_env(env, t.reshape(16), 0)
            env._move_left()
            gain[i + r] = int(env.score)
            assert np.array_equal(codes_of(env.board).reshape(4, 4)[0], res[r])
    np.savez_compressed(os.path.join(HERE, "row_slide.npz"), out=out, gain=gain)
    print("row_slide", codes.shape[0])


def gen_step_transitions(pool):
    env = Game2048Env()
    n = pool.shape[0]
    rng = np.random.default_rng(11)
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    scores_in = rng.integers(0, 50000, size=n).astype(np.int32)
    hs = np.array([hashed(O.DOM_STEP, 5, i) for i in range(n)], dtype=np.uint32)
    board_out = np.zeros((n, 16), np.uint8)
    score_out = np.zeros(n, np.int32)
    reward = np.zeros(n, np.float64)
    done = np.zeros(n, np.uint8)
    valid = np.zeros(n, np.uint8)
    highest = np.zeros(n, np.int32)
    consumed = np.zeros(n, np.uint8)
    for i in range(n):
        set_env(env, tiles_of(pool[i]), int(scores_in[i]))
        STREAM.source = list_source([int(hs[i])])
        STREAM.consumed = 0
        st, r, d, info = env.step(int(actions[i]))
        board_out[i] = codes_of(st)
        score_out[i] = int(info["score"])
        reward[i] = float(r)
        done[i] = bool(d)
        valid[i] = bool(info["valid_move"])
        highest[i] = int(info["highest_tile"])
        consumed[i] = STREAM.consumed
        assert isinstance(r, (float, np.floating))
    np.savez_compressed(os.path.join(HERE, "step_transitions.npz"), board_in=pool, action=actions,
                        score_in=scores_in, h=hs, board_out=board_out, score_out=score_out, reward=reward,
                        done=done, valid=valid, highest_tile=highest, consumed=consumed)
    print("step_transitions", n, "valid", int(valid.sum()), "done", int(done.sum()),
          "nan", int(np.isnan(reward).sum()))


def gen_valid_and_agent_moves(pool):
    env = Game2048Env()
    agent = BeamSearchAgent()
    n = pool.shape[0]
    env_mask = np.zeros(n, np.uint8)
    agent_mask = np.zeros(n, np.uint8)
    mv_board = np.zeros((n, 4, 16), np.uint8)
    mv_score = np.zeros((n, 4), np.int32)
    mv_valid = np.zeros((n, 4), np.uint8)
    env_board = np.zeros((n, 4, 16), np.uint8)
    env_gain = np.zeros((n, 4), np.int32)
    for i in range(n):
        t = tiles_of(pool[i])
        set_env(env, t, 0)
        vm = env.get_valid_moves()
        env_mask[i] = sum(int(v) << a for a, v in enumerate(vm))
        am = agent._check_valid_moves(t.reshape(4, 4).copy())
        agent_mask[i] = sum(int(v) << a for a, v in enumerate(am))
        for a in range(4):
            nb, sc, v = agent._make_move(t.reshape(4, 4).copy(), a)
            mv_board[i, a] = codes_of(nb)
            mv_score[i, a] = int(sc)
            mv_valid[i, a] = bool(v)
            set_env(env, t, 0)
            env._execute_move(a)
            env_board[i, a] = codes_of(env.board)
            env_gain[i, a] = int(env.score)
    np.savez_compressed(os.path.join(HERE, "moves.npz"), board=pool, env_mask=env_mask, agent_mask=agent_mask,
                        agent_board=mv_board, agent_score=mv_score, agent_valid=mv_valid,
                        env_board=env_board, env_gain=env_gain)
    print("moves", n, "env!=agent mask:", int((env_mask != agent_mask).sum()))


def gen_eval_scores(pool):
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        from agents.ppo_agent import PPOAgent
        ppo = PPOAgent()
    agent = BeamSearchAgent()
    n = pool.shape[0]
    fast = np.zeros(n, np.float64)
    full = np.zeros((n, 3), np.float64)
    heur = np.zeros(n, np.float64)
    mono = np.zeros((n, 4), np.float64)
    norm = np.zeros((n, 16), np.float32)
    shaping = np.zeros(n, np.float64)
    phase_of = np.zeros(n, np.uint8)
    for i in range(n):
        t = tiles_of(pool[i])
        g = t.reshape(4, 4)
        fast[i] = agent._fast_evaluate(g.copy(), "early")
        for p, name in enumerate(("early", "mid", "late")):
            full[i, p] = agent._evaluate_state(g.copy(), name)
        phase_of[i] = ("early", "mid", "late").index(agent._determine_game_phase(np.max(g)))
        heur[i] = ppo.evaluate_heuristic(t.copy())
        for k, (rd, cd) in enumerate(((1, 1), (1, -1), (-1, 1), (-1, -1))):
            mono[i, k] = ppo.monotonicity(g.copy(), rd, cd)
        # PPOAgent.remember (ppo_agent.py:234-269) with its stateful terms neutralised: no new-highest-tile
        # bonus (highest_tile_seen huge), no novelty (hash pre-seeded), no regression (state == next_state);
        # what is stored is 0 + 0.1*sum(log2 top-4) + 0.3*evaluate_heuristic(next_state)
        ppo.highest_tile_seen = 1 << 30
        ppo.seen_states.add(hash(t.tobytes()))
        with contextlib.redirect_stdout(io.StringIO()):
            ppo.remember(t.copy(), 0, 0.0, 0.0, t.copy(), False)
        shaping[i] = float(ppo.memory.buffer[-1][3])
        ppo.memory.clear()
        ns = ppo.normalize_state(t.copy())
        assert ns.dtype == np.float32
        norm[i] = ns
    np.savez_compressed(os.path.join(HERE, "eval_scores.npz"), board=pool, fast=fast, full=full, phase=phase_of,
                        ppo_heuristic=heur, monotonicity=mono, normalize=norm, ppo_shaping=shaping)
    print("eval_scores", n)


def gen_beam(roots, with_mask):
    """(root, width, depth, game_id) -> action, prob, draws consumed, per-level top-k scores."""
    configs = [(20, 30), (10, 15), (15, 20), (3, 4)]
    rows = []
    trace_rows = []
    TL = 30
    levels = []

    real_sorted = sorted

    def tracing_sorted(it, key=None, reverse=False):
        res = real_sorted(it, key=key, reverse=reverse)
        levels.append([float(x["score"]) for x in res])
        return res

    bsa.sorted = tracing_sorted
    env = Game2048Env()             # built here: its constructor-time reset() must not eat recorded draws
    gid = 0
    t0 = time.time()
    for ci, (w, d) in enumerate(configs):
        agent = BeamSearchAgent(beam_width=w, search_depth=d)
        sub = roots if (w, d) == (20, 30) else roots[:: 3]
        for ri in range(sub.shape[0]):
            t = tiles_of(sub[ri])
            k0, k1 = O.rng_keys(SEED, O.DOM_BEAM, 0)
            ctr = [0]

            def src():
                h = O.rng_draw(k0, k1, gid, ctr[0])
                ctr[0] += 1
                return h

            STREAM.source = src
            STREAM.consumed = 0
            del levels[:]
            mask = -1
            vm = None
            if with_mask and (ri % 2 == 1):
                set_env(env, t, 0)
                vm = env.get_valid_moves()
                mask = sum(int(v) << a for a, v in enumerate(vm))
            a, p = agent.get_action(t.copy(), vm)
            rows.append((w, d, gid, mask, int(a), float(p), STREAM.consumed, len(levels)))
            tr = np.full((TL, 20), np.nan)
            cnt = np.zeros(TL, np.int32)
            for li, sc in enumerate(levels[:TL]):
                top = sc[:w]
                cnt[li] = len(top)
                tr[li, :len(top)] = top
            trace_rows.append((sub[ri], tr, cnt))
            gid += 1
        print("  beam config", (w, d), "done, elapsed %.1fs" % (time.time() - t0))
    del bsa.sorted
    rows = np.array(rows, dtype=np.float64)
    np.savez_compressed(
        os.path.join(HERE, "beam_decisions.npz"),
        root=np.array([r[0] for r in trace_rows], np.uint8),
        width=rows[:, 0].astype(np.int32), depth=rows[:, 1].astype(np.int32),
        game_id=rows[:, 2].astype(np.int64), mask=rows[:, 3].astype(np.int32),
        action=rows[:, 4].astype(np.int32), prob=rows[:, 5].astype(np.float32),
        consumed=rows[:, 6].astype(np.int32), n_levels=rows[:, 7].astype(np.int32),
        trace_scores=np.array([r[1] for r in trace_rows], np.float64),
        trace_counts=np.array([r[2] for r in trace_rows], np.int32),
        seed=np.uint64(SEED), step_index=np.uint64(0))
    print("beam_decisions", rows.shape[0], "actions hist", np.bincount(rows[:, 4].astype(int), minlength=4),
          "prob!=1:", int((rows[:, 5] != 1.0).sum()))


def gen_beam_masks(roots):
    """get_action with ARBITRARY caller masks, including ones that disagree with the agent's own validity:
    reaches the random fallback (beam_search_agent.py:126-128: random.choice(valid_actions), prob 0.5)."""
    rng = np.random.default_rng(123)
    agent = BeamSearchAgent(beam_width=5, search_depth=6)
    rows = []
    for ri in range(roots.shape[0]):
        t = tiles_of(roots[ri])
        for rep in range(3):
            mask = int(rng.integers(0, 16))
            vm = [bool((mask >> a) & 1) for a in range(4)]
            gid = 5000 + 3 * ri + rep
            k0, k1 = O.rng_keys(SEED, O.DOM_BEAM, 1)
            ctr = [0]

            def src():
                h = O.rng_draw(k0, k1, gid, ctr[0])
                ctr[0] += 1
                return h
            STREAM.source = src
            STREAM.consumed = 0
            a, p = agent.get_action(t.copy(), vm)
            rows.append((ri, mask, gid, int(a), float(p), STREAM.consumed))
    rows = np.array(rows, dtype=np.float64)
    fallback = int(((rows[:, 4] == 0.5) & (rows[:, 1] != 0)).sum())
    np.savez_compressed(os.path.join(HERE, "beam_masks.npz"), root=roots, root_index=rows[:, 0].astype(np.int32),
                        mask=rows[:, 1].astype(np.int32), game_id=rows[:, 2].astype(np.int64),
                        action=rows[:, 3].astype(np.int32), prob=rows[:, 4].astype(np.float32),
                        consumed=rows[:, 5].astype(np.int32), seed=np.uint64(SEED), step_index=np.uint64(1))
    print("beam_masks", rows.shape[0], "random-fallback decisions", fallback)


def gen_episodes():
    """Full seeded episodes (reset spawns included) + the 1000-step config-1 trace with auto-reset."""
    env = Game2048Env()
    eps = []
    for e in range(4):
        t = [0]
        resets = [hashed(O.DOM_RESET, 0, e, 0), hashed(O.DOM_RESET, 0, e, 1)]
        STREAM.source = list_source(resets)
        st = env.reset()
        rec = dict(reset_h=np.array(resets, np.uint32), board0=codes_of(st), action=[], h=[], board=[], reward=[],
                   done=[], score=[], valid=[])
        for step in range(5000):
            a = hashed(O.DOM_SYNTH_ACTION, step, e) >> 30
            h = hashed(O.DOM_STEP, step, e)
            STREAM.source = list_source([h])
            st, r, d, info = env.step(int(a))
            rec["action"].append(a); rec["h"].append(h); rec["board"].append(codes_of(st))
            rec["reward"].append(float(r)); rec["done"].append(bool(d)); rec["score"].append(int(info["score"]))
            rec["valid"].append(bool(info["valid_move"]))
            if d:
                break
        eps.append(rec)
        print("  episode", e, "len", len(rec["action"]), "score", rec["score"][-1])
    out = {}
    for e, rec in enumerate(eps):
        out["ep%d_reset_h" % e] = rec["reset_h"]
        out["ep%d_board0" % e] = rec["board0"]
        out["ep%d_action" % e] = np.array(rec["action"], np.uint8)
        out["ep%d_h" % e] = np.array(rec["h"], np.uint32)
        out["ep%d_board" % e] = np.array(rec["board"], np.uint8)
        out["ep%d_reward" % e] = np.array(rec["reward"], np.float64)
        out["ep%d_done" % e] = np.array(rec["done"], np.uint8)
        out["ep%d_valid" % e] = np.array(rec["valid"], np.uint8)
        out["ep%d_score" % e] = np.array(rec["score"], np.int32)
    # config 1: one board, 1000 random-action steps, auto-reset on done.
    # Draw schedule = the product's VecGame2048 schedule for board id 0:
    #   reset(epoch 0): DOM_RESET idx 0 ctr 0/1; step t: DOM_STEP idx t; auto-reset after a
    #   terminal step t: DOM_EPISODE idx t ctr 0/1; action t: DOM_SYNTH_ACTION idx t (>> 30).
    STREAM.source = list_source([hashed(O.DOM_RESET, 0, 0, 0), hashed(O.DOM_RESET, 0, 0, 1)])
    st = env.reset()
    c1 = dict(board0=codes_of(st), board=[], reward=[], done=[], score=[], action=[])
    for t in range(1000):
        a = hashed(O.DOM_SYNTH_ACTION, t, 0) >> 30
        STREAM.source = list_source([hashed(O.DOM_STEP, t, 0)])
        st, r, d, info = env.step(int(a))
        c1["action"].append(a); c1["reward"].append(float(r)); c1["done"].append(bool(d))
        c1["score"].append(int(info["score"]))
        if d:
            STREAM.source = list_source([hashed(O.DOM_EPISODE, t, 0, 0), hashed(O.DOM_EPISODE, t, 0, 1)])
            st = env.reset()
        c1["board"].append(codes_of(st))       # post-auto-reset board, like the product's boards_out
    out["c1_board0"] = c1["board0"]
    out["c1_action"] = np.array(c1["action"], np.uint8)
    out["c1_board"] = np.array(c1["board"], np.uint8)
    out["c1_reward"] = np.array(c1["reward"], np.float64)
    out["c1_done"] = np.array(c1["done"], np.uint8)
    out["c1_score"] = np.array(c1["score"], np.int32)
    out["seed"] = np.uint64(SEED)
    np.savez_compressed(os.path.join(HERE, "episodes.npz"), **out)
    print("episodes: c1 dones", int(np.sum(c1["done"])))


def gen_simulate(pool):
    """Game2048Env.simulate_move (game_2048.py:341-387): all (cell, tile) successors with reward and done, for
    env.highest_tile both equal to max(state) (as inside an episode) and larger (milestone branch fires)."""
    env = Game2048Env()
    n = pool.shape[0]
    rng = np.random.default_rng(31)
    actions = rng.integers(0, 4, size=n).astype(np.uint8)
    hi_code = np.zeros(n, np.uint8)
    count = np.zeros(n, np.uint8)
    succ = np.zeros((n, 32, 16), np.uint8)
    reward = np.zeros((n, 32), np.float64)
    done = np.zeros((n, 32), np.uint8)
    for i in range(n):
        t = tiles_of(pool[i])
        mx = int(pool[i].max())
        hc = mx if i % 3 else min(mx + 1 + (i % 7), 17)         # every third row: highest_tile above the board's max
        hi_code[i] = hc
        set_env(env, np.zeros(16, np.int32), 123)                # the env's own board / score must come back untouched
        env.board[0, 0] = 2
        env.highest_tile = np.int32(1 << hc) if hc else np.int32(0)
        res = env.simulate_move(t.copy(), int(actions[i]))
        assert env.score == 123 and env.board[0, 0] == 2
        count[i] = len(res)
        for k, (ns, r, d) in enumerate(res):
            succ[i, k] = codes_of(ns)
            reward[i, k] = float(r)
            done[i, k] = bool(d)
    np.savez_compressed(os.path.join(HERE, "simulate_move.npz"), board=pool, action=actions, highest_code=hi_code,
                        count=count, succ=succ, reward=reward, done=done)
    print("simulate_move", n, "non-empty", int((count > 0).sum()), "max successors", int(count.max()),
          "done successors", int(done.sum()))


def gen_rng_pin():
    rows = []
    for seed in (0, 1, SEED, 2**63 + 12345, 2**64 - 1):
        for dom in (1, 2, 3, 4, 5, 6):
            for idx in (0, 1, 999, 2**40 + 7):
                k0, k1 = O.rng_keys(seed, dom, idx)
                for ident in (0, 1, 2**20 + 3, 2**32 + 5, 2**63 + 11):
                    for ctr in (0, 1, 77):
                        rows.append((seed, dom, idx, ident, ctr, k0, k1, O.rng_draw(k0, k1, ident, ctr)))
    a = np.array(rows, dtype=np.uint64)
    np.savez_compressed(os.path.join(HERE, "rng_pin.npz"), rows=a)
    print("rng_pin", a.shape[0])


def gen_step_noop(pool):
    """Game2048Env.step with action values outside 0..3: _execute_move (:97-114) has no branch for them, so nothing moves."""
    env = Game2048Env()
    rng = np.random.default_rng(404)
    dead = np.array([[(a if (r + c) % 2 == 0 else b) for r in range(4) for c in range(4)] for a in range(1, 6) for b in range(6, 10)], np.uint8)
    sel = np.concatenate([pool[rng.choice(pool.shape[0], 400, replace=False)], dead])      # + 20 finished (checkerboard) boards
    n = sel.shape[0]
    actions = rng.choice(np.array([4, 5, 7, 100, 255], np.uint8), size=n)
    scores_in = rng.integers(0, 50000, size=n).astype(np.int32)
    board_out = np.zeros((n, 16), np.uint8); score_out = np.zeros(n, np.int32); reward = np.zeros(n, np.float64)
    done = np.zeros(n, np.uint8); valid = np.zeros(n, np.uint8); consumed = np.zeros(n, np.uint8)
    for i in range(n):
        set_env(env, tiles_of(sel[i]), int(scores_in[i]))
        STREAM.source = list_source([hashed(O.DOM_STEP, 6, i)])
        STREAM.consumed = 0
        st, r, d, info = env.step(int(actions[i]))
        board_out[i] = codes_of(st); score_out[i] = int(info["score"]); reward[i] = float(r)
        done[i] = bool(d); valid[i] = bool(info["valid_move"]); consumed[i] = STREAM.consumed
    assert not valid.any() and not consumed.any() and np.array_equal(board_out, sel)
    np.savez_compressed(os.path.join(HERE, "step_noop.npz"), board_in=sel, action=actions, score_in=scores_in, board_out=board_out,
                        score_out=score_out, reward=reward, done=done, valid=valid)
    print("step_noop", n, "done", int(done.sum()), "nan", int(np.isnan(reward).sum()))


def gen_remember():
    """PPOAgent.remember (agents/ppo_agent.py:234-269) run SEQUENTIALLY by the reference over an ordered list of
    transitions, all of its terms live (highest_tile_seen and seen_states start fresh, as a new agent's do):
      part 1  a vectorised rollout in (step, env) order: 24 reference envs x 150 steps, auto-reset when an episode ends; envs
              2k and 2k+1 are twins (same draws, same actions) for their first 6 steps, so the novelty term sees repeats;
      part 2  320 transitions between unrelated boards (max tile falling, rising by several doublings, huge tiles),
              which exercises the regression term the reference's own episode loop never reaches.
    Stored: state / next_state codes, the reward passed in, the reward remember() stored, and the agent's final state."""
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        from agents.ppo_agent import PPOAgent
        ppo = PPOAgent()
    assert ppo.highest_tile_seen == 2 and len(ppo.seen_states) == 0
    n_env, n_steps, twin_steps = 24, 150, 6
    envs = [Game2048Env() for _ in range(n_env)]

    def gid_at(e, t):           # twins share a draw / action identity for the first steps
        return 9000 + (e - (e & 1) if t < twin_steps else e)

    states = []
    for e, env in enumerate(envs):
        g = gid_at(e, 0)
        STREAM.source = list_source([hashed(O.DOM_RESET, 0, g, 0), hashed(O.DOM_RESET, 0, g, 1)])
        states.append(env.reset())
    S, NS, RIN, ROUT, DONE = [], [], [], [], []

    def feed(state, next_state, reward, done):
        with contextlib.redirect_stdout(io.StringIO()):
            ppo.remember(state.copy(), 0, 0.0, reward, next_state.copy(), done)
        S.append(codes_of(state)); NS.append(codes_of(next_state)); RIN.append(float(reward))
        ROUT.append(float(ppo.memory.buffer[-1][3])); DONE.append(bool(done))
        ppo.memory.buffer.clear()

    for t in range(n_steps):
        for e, env in enumerate(envs):
            g = gid_at(e, t)
            a = int(hashed(O.DOM_SYNTH_ACTION, t, g) >> 30)
            STREAM.source = list_source([hashed(O.DOM_STEP, t, g)])
            ns, r, d, info = env.step(a)
            feed(states[e], ns, r, d)
            if d:
                STREAM.source = list_source([hashed(O.DOM_EPISODE, t, g, 0), hashed(O.DOM_EPISODE, t, g, 1)])
                ns = env.reset()
            states[e] = ns
    n_rollout = len(S)
    rng = np.random.default_rng(4242)
    pool = np.concatenate([random_code_boards(rng, 200, 0.3, 11), random_code_boards(rng, 60, 0.5, 17),
                           random_code_boards(rng, 60, 0.0, 4)]).astype(np.uint8)
    pool[pool.max(axis=1) == 0, 0] = 1
    for i in range(320):
        a, b = pool[rng.integers(pool.shape[0])], pool[rng.integers(pool.shape[0])]
        if i % 5 == 0:
            b = a                      # an exact repeat of an earlier / the same board
        feed(tiles_of(a), tiles_of(b), float(rng.normal() * 3.0), False)
    rout, rin = np.array(ROUT), np.array(RIN)
    print("remember", len(S), "transitions; novel", int(len(ppo.seen_states)), "highest", int(ppo.highest_tile_seen),
          "dones", int(np.sum(DONE)))
    np.savez_compressed(os.path.join(HERE, "remember.npz"), state=np.array(S, np.uint8), next_state=np.array(NS, np.uint8),
                        reward_in=rin, reward_out=rout, done=np.array(DONE, np.uint8), n_rollout=np.int64(n_rollout),
                        n_env=np.int64(n_env), final_highest_tile=np.int64(ppo.highest_tile_seen),
                        final_seen=np.int64(len(ppo.seen_states)))


def gen_simulate_sampled():
    """agents/hybrid.py:578-629 -- the simulate_move the hybrid agent patches onto its own copy of the env -- for a pool of
    (board, action): the list of (next_state, weighted reward, done) it returns. random.sample is replaced by the recorded
    draws (pick j = the idx(h_j, n - j)-th empty cell not picked before, the product's mapping)."""
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        import agents.hybrid as hyb
    env = hyb.Game2048Env()
    rng = np.random.default_rng(777)
    pool = np.concatenate([random_code_boards(rng, 500, 0.30, 11), random_code_boards(rng, 200, 0.05, 4),
                           random_code_boards(rng, 150, 0.0, 3), random_code_boards(rng, 150, 0.8, 17),
                           random_code_boards(rng, 100, 0.5, 2)]).astype(np.uint8)
    pool[pool.max(axis=1) == 0, 5] = 1
    n = pool.shape[0]
    actions = (np.arange(n) % 4).astype(np.uint8)
    H = np.zeros((n, 3), np.uint32)
    succ = np.zeros((n, 8, 16), np.uint8)
    rew = np.zeros((n, 8), np.float64)
    done = np.zeros((n, 8), np.uint8)
    count = np.zeros(n, np.uint8)
    cur = {}

    def sample(population, k):
        picks = O.sample_picks(cur["h"], len(population))
        assert len(picks) == k
        return [population[i] for i in picks]
    real_sample = random.sample
    random.sample = sample
    try:
        for i in range(n):
            H[i] = [hashed(O.DOM_SIMULATE, 3, i, j) for j in range(3)]
            cur["h"] = H[i]
            res = env.simulate_move(tiles_of(pool[i]).reshape(4, 4), int(actions[i]))
            count[i] = len(res)
            for k, (st, r, d) in enumerate(res):
                succ[i, k] = codes_of(st); rew[i, k] = float(r); done[i, k] = bool(d)
    finally:
        random.sample = real_sample
    print("simulate_sampled", n, "counts", np.bincount(count, minlength=7))
    np.savez_compressed(os.path.join(HERE, "simulate_sampled.npz"), board=pool, action=actions, h=H, succ=succ, reward=rew,
                        done=done, count=count, seed=np.uint64(SEED), step_index=np.uint64(3))


def gen_games():
    """Complete games in the shape of the reference's run_game (evaluate_beam_search.py:29-98) and train.py's episode loop
    (:48-75), played by the REAL Game2048Env + BeamSearchAgent with the product's draw schedule: reset (SEED, RESET, 0, game id,
    0/1); move t: the search's draws (SEED, BEAM, t, game id, j) in the order the agent consumes them, the env's spawn (SEED,
    STEP, t, game id, 0). Recorded per game: what run_game returns beyond the counters -- board_history, scores_history,
    max_tiles_history, milestones -- and train.py's moveset. Small beams keep the reference's Python search affordable (three games at
    width 3 / depth 4 and four greedy ones run to their end or the 5000-move cap, three at width 4 / depth 5 are cut at 250 moves); two
    games at the evaluation configuration (width 20, depth 30) are cut at 30 moves, and (round 5) three more at that configuration are
    played to their END under the evaluation's 5000-move cap (run_evaluation.py:48-69; ~0.13 s per decision in the reference's Python,
    2-4 minutes per game) -- the reference's own headline configuration pinned end to end."""
    env = Game2048Env()
    out, meta = {}, []
    t0 = time.time()
    configs = [(3, 4, 5000, 3), (1, 1, 5000, 4), (4, 5, 250, 3), (20, 30, 30, 2), (20, 30, 5000, 3)]      # width, depth, move cap, games
    gid = 7000
    k = 0
    for w, d, cap, ngames in configs:
        agent = BeamSearchAgent(beam_width=w, search_depth=d)
        for _ in range(ngames):
            STREAM.source = list_source([hashed(O.DOM_RESET, 0, gid, 0), hashed(O.DOM_RESET, 0, gid, 1)])
            state = env.reset()
            done, moves, valid_n, invalid_n = False, 0, 0, 0
            milestones = {m: -1 for m in (64, 128, 256, 512, 1024, 2048, 4096, 8192)}
            boards, scores, maxt, moveset = [codes_of(state)], [0], [int(np.max(state))], []
            while not done and moves < cap:                               # evaluate_beam_search.py:52
                k0, k1 = O.rng_keys(SEED, O.DOM_BEAM, moves)
                ctr = [0]

                def src():
                    h = O.rng_draw(k0, k1, gid, ctr[0])
                    ctr[0] += 1
                    return h
                STREAM.source = src
                action, _ = agent.get_action(state)                       # :54 (no caller mask)
                STREAM.source = list_source([hashed(O.DOM_STEP, moves, gid)])
                state, reward, done, info = env.step(action)              # :58
                mt = int(np.max(state))
                for m in milestones:                                      # :60-64
                    if mt >= m and milestones[m] < 0:
                        milestones[m] = moves
                valid_n += int(bool(info["valid_move"])); invalid_n += int(not info["valid_move"])
                boards.append(codes_of(state)); maxt.append(mt); scores.append(int(info["score"])); moveset.append(int(action))
                moves += 1
            out["g%d_boards" % k] = np.array(boards, np.uint8)
            out["g%d_scores" % k] = np.array(scores, np.int64)
            out["g%d_max_tiles" % k] = np.array(maxt, np.int64)
            out["g%d_moveset" % k] = np.array(moveset, np.uint8)
            out["g%d_milestones" % k] = np.array([milestones[m] for m in (64, 128, 256, 512, 1024, 2048, 4096, 8192)], np.int32)
            meta.append((w, d, cap, gid, moves, valid_n, invalid_n, int(info["score"]), int(done)))
            print("  game", k, (w, d), "moves", moves, "score", int(info["score"]), "max", maxt[-1], "done", done, "%.1fs" % (time.time() - t0))
            gid += 1; k += 1
    out["meta"] = np.array(meta, np.int64)      # width, depth, cap, game id, moves, valid, invalid, score, done
    out["seed"] = np.uint64(SEED)
    np.savez_compressed(os.path.join(HERE, "games.npz"), **out)


def gen_pattern():
    """Game2048Env._evaluate_pattern (environment/game_2048.py:313-339; nobody in the reference calls it) on the boards of
    eval_scores.npz."""
    pool = np.load(os.path.join(HERE, "eval_scores.npz"))["board"]
    env = Game2048Env()
    out = np.zeros(pool.shape[0], np.float64)
    for i in range(pool.shape[0]):
        set_env(env, tiles_of(pool[i]))
        out[i] = env._evaluate_pattern()
    np.savez_compressed(os.path.join(HERE, "pattern.npz"), board=pool, pattern=out)
    print("pattern", pool.shape[0])


def gen_eval_parts():
    """BeamSearchAgent._calculate_corner_bonus (:375-385) and _calculate_merge_potential (:387-403) on their own -- the two
    terms _evaluate_state weights -- on the boards of eval_scores.npz."""
    pool = np.load(os.path.join(HERE, "eval_scores.npz"))["board"]
    agent = BeamSearchAgent()
    corner = np.zeros(pool.shape[0], np.float64)
    merge = np.zeros(pool.shape[0], np.float64)
    for i in range(pool.shape[0]):
        g = tiles_of(pool[i]).reshape(4, 4)
        corner[i] = agent._calculate_corner_bonus(g.copy())
        merge[i] = agent._calculate_merge_potential(g.copy())
    np.savez_compressed(os.path.join(HERE, "eval_parts.npz"), board=pool, corner_bonus=corner, merge_potential=merge)
    print("eval_parts", pool.shape[0], "boards with a merge:", int((merge > 0).sum()), "without a corner tile:", int((corner == 0).sum()))


def gen_checkpoint():
    """The beam agent's checkpoint format (agents/beam_search_agent.py:413-478): the text of the reference-held
    checkpoints/BeamSearchAgent_*.pth files (JSON despite the name) and README files, and what the reference's own
    load() -> save() writes for each of them (JSON + README), saved under the same relative path from a scratch directory."""
    import contextlib
    import io
    import json
    import tempfile
    ck = os.path.join(REF, "checkpoints")
    out = {"files": {}, "resaved": {}}
    for name in sorted(os.listdir(ck)):
        if (name.startswith("BeamSearchAgent") and name.endswith(".pth")) or name.startswith("beam_search_config_readme"):
            out["files"][name] = open(os.path.join(ck, name)).read()
    here = os.getcwd()
    for name in [n for n in out["files"] if n.endswith(".pth")]:
        with tempfile.TemporaryDirectory() as tmp, contextlib.redirect_stdout(io.StringIO()):
            try:
                os.chdir(tmp)
                agent = BeamSearchAgent.load(os.path.join(ck, name))
                rel = os.path.join("checkpoints", name)
                agent.save(rel)
                readme = "beam_search_config_readme_%d_%d.txt" % (agent.beam_width, agent.search_depth)
                out["resaved"][name] = {"path": rel, "json": open(rel).read(), "readme_name": readme,
                                        "readme": open(os.path.join("checkpoints", readme)).read(),
                                        "beam_width": agent.beam_width, "search_depth": agent.search_depth,
                                        "early_game_threshold": agent.early_game_threshold,
                                        "mid_game_threshold": agent.mid_game_threshold}
            finally:
                os.chdir(here)
    with open(os.path.join(HERE, "beam_checkpoint.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("beam_checkpoint", sorted(out["files"]))


def main():
    if "--only" in sys.argv:
        what = sys.argv[sys.argv.index("--only") + 1]
        if what in ("pattern", "checkpoint", "games", "eval_parts"):
            {"pattern": gen_pattern, "checkpoint": gen_checkpoint, "games": gen_games, "eval_parts": gen_eval_parts}[what]()
            return
        if what == "step_noop":
            rng = np.random.default_rng(1)
            gen_step_noop(np.concatenate([random_code_boards(rng, 600, 0.30, 11), random_code_boards(rng, 300, 0.0, 3),
                                          random_code_boards(rng, 300, 0.6, 17)]).astype(np.uint8))
            return
        {"remember": gen_remember, "simulate_sampled": gen_simulate_sampled}[what]()
        return
    O.build()
    rng = np.random.default_rng(2048)
    t0 = time.time()
    print("self-play pools ...")
    sp_rand = selfplay_boards(24, np.random.default_rng(1), greedy=False, tag=1)
    sp_greedy = selfplay_boards(16, np.random.default_rng(2), greedy=True, tag=2)
    print("  pools", sp_rand.shape, sp_greedy.shape, "max code", sp_greedy.max(), "%.1fs" % (time.time() - t0))
    edges = edge_boards()

    def lifted(b, k):               # same structure, every tile 2^k times larger (mid / late game stand-ins)
        return np.where(b > 0, b + k, 0).astype(np.uint8)

    def pick(a, k):
        return a[rng.choice(a.shape[0], min(k, a.shape[0]), replace=False)]

    pool = np.concatenate([
        edges,
        pick(sp_rand, 2000),
        pick(sp_greedy, 2000),
        lifted(pick(sp_greedy, 500), 2), lifted(pick(sp_greedy, 500), 4), lifted(pick(sp_greedy, 300), 8),
        random_code_boards(rng, 1200, 0.30, 11),
        random_code_boards(rng, 800, 0.05, 5),      # dense, many merges / near-dead
        random_code_boards(rng, 600, 0.0, 3),       # full boards, lots of dead ones
        random_code_boards(rng, 500, 0.6, 17),      # sparse, huge tiles
        random_code_boards(rng, 400, 0.15, 17),
    ]).astype(np.uint8)
    print("pool", pool.shape)
    gen_rng_pin()
    gen_row_slide()
    gen_step_transitions(pool)
    small = np.concatenate([edges, pool[rng.choice(pool.shape[0], 1500, replace=False)]])
    gen_valid_and_agent_moves(small)
    gen_eval_scores(np.concatenate([edges, pool[rng.choice(pool.shape[0], 3000, replace=False)]]))
    gen_simulate(np.concatenate([edges, pool[np.random.default_rng(97).choice(pool.shape[0], 1200, replace=False)]]))
    # beam roots: early/mid/late, few/many empties, single-valid-move, no-valid-move
    big = sp_greedy[sp_greedy.max(axis=1) >= 6]
    lift_all = np.concatenate([big, lifted(big, 1), lifted(big, 2), lifted(big, 3), lifted(big, 4)])
    late = lift_all[lift_all.max(axis=1) >= 10]
    mid = lift_all[lift_all.max(axis=1) == 9]
    early = np.concatenate([sp_rand, sp_greedy])
    early = early[early.max(axis=1) < 9]
    roots = np.concatenate([
        edges,
        pick(early, 35),
        pick(mid, 30),
        pick(late, 40),
        random_code_boards(rng, 20, 0.30, 11),
        random_code_boards(rng, 10, 0.0, 4),
    ]).astype(np.uint8)
    print("beam roots", roots.shape)
    gen_beam(roots, with_mask=True)
    gen_beam_masks(np.concatenate([edges, random_code_boards(np.random.default_rng(321), 60, 0.05, 4),
                                   random_code_boards(np.random.default_rng(322), 25, 0.4, 8)]).astype(np.uint8))
    gen_episodes()
    gen_remember()
    gen_simulate_sampled()
    rng2 = np.random.default_rng(1)
    gen_step_noop(np.concatenate([random_code_boards(rng2, 600, 0.30, 11), random_code_boards(rng2, 300, 0.0, 3),
                                  random_code_boards(rng2, 300, 0.6, 17)]).astype(np.uint8))
    gen_pattern()
    gen_eval_parts()
    gen_checkpoint()
    gen_games()
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
