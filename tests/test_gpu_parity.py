"""GPU parity tests: the HIP path, called through the C-ABI (g2048.ops -> csrc/libg2048_hip.so), against
the committed golden vectors (captured from the reference) and against the CPU oracle on the same seeded
inputs. Bit-exact for boards / scores / flags / masks; f64 `==` for rewards and heuristic scores (the
north star allows 1e-6; the tolerance used here is 0). Full-size cases use the BASELINE configs."""
import numpy as np
import pytest
import torch

from conftest import load_golden, tiles_of

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
SEED = 0x2048


@pytest.fixture(scope="module")
def ops():
    import __graft_entry__ as ge
    ge.ensure_built()
    ge.import_package()
    from g2048 import ops as o, _lib
    _lib.lib()
    assert torch.cuda.is_available()
    return o


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=DEV)
    return t if dtype is None else t.to(dtype)


def host(t):
    return t.cpu().numpy()


def test_selftest_instruction_assumptions(ops):
    assert ops.selftest(DEV) == 0


def test_native_library_is_loaded(ops):
    import os
    from g2048 import _lib
    maps = open("/proc/self/maps").read()
    assert os.path.basename(_lib.library_path()) in maps


def test_synth_inputs_match_oracle(ops, oracle):
    n = 100003
    b = ops.synth_boards(n, seed=SEED, id_base=12345, device=DEV)
    a = ops.synth_actions(n, seed=SEED, step_index=9, id_base=12345, device=DEV)
    assert np.array_equal(host(b), oracle.synth_boards(n, seed=SEED, id_base=12345))
    assert np.array_equal(host(a), oracle.synth_actions(n, seed=SEED, step_index=9, id_base=12345))
    b2 = ops.synth_boards(4096, seed=3, id_base=2**40, p_empty=0.05, max_code=17, device=DEV)
    assert np.array_equal(host(b2), oracle.synth_boards(4096, seed=3, id_base=2**40, p_empty=0.05, max_code=17))


def test_step_golden_transitions(ops):
    """8.8k transitions recorded from the reference (incl. invalid moves, dead/full/empty boards)."""
    g = load_golden("step_transitions.npz")
    b, a = dev(g["board_in"]), dev(g["action"])
    sc = dev(g["score_in"].astype(np.int32))
    out, rw, fl = ops.step(b, a, sc, seed=SEED, step_index=5, id_base=0, reward_f64=True)
    fl = host(fl)
    assert np.array_equal(host(out), g["board_out"])
    assert np.array_equal(host(sc), g["score_out"])
    assert np.array_equal(host(rw), g["reward"], equal_nan=True)            # f64 ==
    assert np.array_equal(fl & 1, g["done"]) and np.array_equal((fl >> 1) & 1, g["valid"])
    assert np.array_equal(1 << (fl >> 3).astype(np.int64), np.maximum(g["highest_tile"], 1))
    # f32 reward mode = (float) of the f64 reward
    sc2 = dev(g["score_in"].astype(np.int32))
    _, rw32, _ = ops.step(b, a, sc2, seed=SEED, step_index=5, id_base=0)
    assert np.array_equal(host(rw32), g["reward"].astype(np.float32), equal_nan=True)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 1048576])
def test_step_vs_oracle_sizes(ops, oracle, n):
    """Config 2 (1,048,576 boards, all four actions in one launch) and ragged sizes."""
    b = ops.synth_boards(n, seed=SEED, device=DEV)
    a = ops.synth_actions(n, seed=SEED, step_index=1, device=DEV)
    sc = torch.zeros(n, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(b, a, sc, seed=SEED, step_index=1, reward_f64=True)
    bo, so, ro, fo = oracle.step_batch(host(b), host(a), np.zeros(n, np.uint32), seed=SEED, step_index=1)
    assert np.array_equal(host(out), bo)
    assert np.array_equal(host(sc).astype(np.uint32), so)
    assert np.array_equal(host(rw), ro, equal_nan=True)
    assert np.array_equal(host(fl), fo)
    if n >= 1000:
        assert set(np.unique(host(a))) == {0, 1, 2, 3}


def test_prepared_step_equals_step(ops):
    """ops.PreparedStep (arguments bound and checked once, then one ctypes call per launch) == ops.step."""
    n = 70000
    b = ops.synth_boards(n, seed=SEED + 5, device=DEV)
    a = ops.synth_actions(n, seed=SEED + 5, step_index=0, device=DEV)
    s1 = torch.zeros(n, dtype=torch.int32, device=DEV); s2 = torch.zeros_like(s1)
    o2 = torch.empty_like(b); r2 = torch.empty(n, dtype=torch.float32, device=DEV); f2 = torch.empty(n, dtype=torch.uint8, device=DEV)
    call = ops.PreparedStep(b, a, s2, SEED, 3, out=o2, reward=r2, flags=f2)
    s2.zero_()
    for t in (4, 5, 6):
        o1, r1, f1 = ops.step(b, a, s1, SEED, t, 3)
        call(t)
        assert bool((o1 == o2).all()) and bool((f1 == f2).all()) and bool((s1 == s2).all())
        assert np.array_equal(host(r1).view(np.uint32), host(r2).view(np.uint32))


@pytest.mark.parametrize("tune", [1, 2])
@pytest.mark.parametrize("n", [1, 1023, 4097, 300001])
def test_step_boards_per_lane_variants(ops, oracle, tune, n):
    """The G2048_STEP_TUNE variants (one / two boards per lane; both are what the library picks by itself at some size) are public
    opts: same results as the oracle,
    also when n is not a multiple of the block tile."""
    b = ops.synth_boards(n, seed=SEED + 9, device=DEV, p_empty=0.1, max_code=5)
    a = ops.synth_actions(n, seed=SEED + 9, step_index=3, device=DEV)
    sc = torch.full((n,), 7, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(b, a, sc, seed=SEED, step_index=3, id_base=11, reward_f64=True, auto_reset=True, tune=tune)
    bo, so, ro, fo = oracle.step_batch(host(b), host(a), np.full(n, 7, np.uint32), seed=SEED, step_index=3, id_base=11, opts=1)
    assert np.array_equal(host(out), bo) and np.array_equal(host(sc).astype(np.uint32), so)
    assert np.array_equal(host(rw), ro, equal_nan=True) and np.array_equal(host(fl), fo)


def test_step_in_place_and_auto_reset(ops, oracle):
    n = 200000
    hb = oracle.synth_boards(n, seed=5, p_empty=0.02, max_code=3)       # dense: many boards die
    ha = oracle.synth_actions(n, seed=5, step_index=0)
    b = dev(hb); sc = torch.full((n,), 100, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(b, dev(ha), sc, seed=5, step_index=0, out=b, reward_f64=True, auto_reset=True)
    assert out.data_ptr() == b.data_ptr()
    bo, so, ro, fo = oracle.step_batch(hb, ha, np.full(n, 100, np.uint32), seed=5, step_index=0, opts=1)
    assert (fo & 1).sum() > 50
    assert np.array_equal(host(b), bo) and np.array_equal(host(sc).astype(np.uint32), so)
    assert np.array_equal(host(rw), ro, equal_nan=True) and np.array_equal(host(fl), fo)


def test_reset_vs_oracle_and_golden(ops, oracle):
    b, sc = ops.reset(70001, seed=SEED, epoch=3, id_base=77, device=DEV)
    ob, osc = oracle.reset_batch(70001, seed=SEED, epoch=3, id_base=77)
    assert np.array_equal(host(b), ob) and int(sc.abs().sum()) == 0
    g = load_golden("episodes.npz")
    b1, _ = ops.reset(1, seed=int(g["seed"]), epoch=0, id_base=0, device=DEV)
    assert np.array_equal(host(b1)[0], g["c1_board0"])
    for e in range(4):
        be, _ = ops.reset(1, seed=int(g["seed"]), epoch=0, id_base=e, device=DEV)
        assert np.array_equal(host(be)[0], g["ep%d_board0" % e])


def test_valid_moves_both_semantics(ops, oracle):
    g = load_golden("moves.npz")
    b = dev(g["board"])
    assert np.array_equal(host(ops.valid_moves(b, False)), g["env_mask"])
    assert np.array_equal(host(ops.valid_moves(b, True)), g["agent_mask"])
    big = ops.synth_boards(300000, seed=11, p_empty=0.1, max_code=4, device=DEV)
    for agent in (False, True):
        assert np.array_equal(host(ops.valid_moves(big, agent)), oracle.valid_moves_batch(host(big), agent))


def test_eval_kernels(ops, oracle):
    from g2048 import _lib as L
    g = load_golden("eval_scores.npz")
    b = dev(g["board"]); n = b.shape[0]
    assert np.array_equal(host(ops.evaluate(b, L.EVAL_FAST)), g["fast"])
    for p in range(3):
        ph = torch.full((n,), p, dtype=torch.uint8, device=DEV)
        assert np.array_equal(host(ops.evaluate(b, L.EVAL_FULL, ph)), g["full"][:, p])       # f64 ==
    assert np.array_equal(host(ops.evaluate(b, L.EVAL_FULL)), g["full"][np.arange(n), g["phase"]])
    assert np.array_equal(host(ops.evaluate(b, L.EVAL_PPO_HEURISTIC)), g["ppo_heuristic"])
    for k in range(4):
        assert np.array_equal(host(ops.evaluate(b, L.EVAL_MONO_PP + k)), g["monotonicity"][:, k])
    assert np.array_equal(host(ops.evaluate(b, L.EVAL_PPO_SHAPING)), g["ppo_shaping"])
    assert np.array_equal(host(ops.obs(b)).view(np.uint32), g["normalize"].view(np.uint32))   # f32 bits
    big = ops.synth_boards(200000, seed=13, p_empty=0.2, max_code=15, device=DEV)
    hb = host(big)
    assert np.array_equal(host(ops.evaluate(big, L.EVAL_FAST)), oracle.eval_batch(hb, oracle.EVAL_FAST))
    ph = (torch.arange(200000, device=DEV) % 3).to(torch.uint8)
    assert np.array_equal(host(ops.evaluate(big, L.EVAL_FULL, ph)), oracle.eval_batch(hb, oracle.EVAL_FULL, host(ph)))
    assert np.array_equal(host(ops.evaluate(big, L.EVAL_PPO_HEURISTIC)), oracle.eval_batch(hb, oracle.EVAL_PPO))
    assert np.array_equal(host(ops.evaluate(big, L.EVAL_PPO_SHAPING)), oracle.eval_batch(hb, oracle.EVAL_PPO_SHAPING))
    assert np.array_equal(host(ops.obs(big)), oracle.obs_batch(hb))
    gp = load_golden("pattern.npz")                     # Game2048Env._evaluate_pattern (game_2048.py:313-339), f64 ==
    assert np.array_equal(host(ops.evaluate(dev(gp["board"]), L.EVAL_PATTERN)), gp["pattern"])
    assert np.array_equal(host(ops.evaluate(big, L.EVAL_PATTERN)), oracle.eval_batch(hb, oracle.EVAL_PATTERN))
    ge = load_golden("eval_parts.npz")                  # the two terms _evaluate_state weights, on their own (:375-403)
    assert np.array_equal(host(ops.evaluate(dev(ge["board"]), L.EVAL_CORNER_BONUS)), ge["corner_bonus"])
    assert np.array_equal(host(ops.evaluate(dev(ge["board"]), L.EVAL_MERGE_POTENTIAL)), ge["merge_potential"])
    assert np.array_equal(host(ops.evaluate(big, L.EVAL_CORNER_BONUS)), oracle.eval_batch(hb, oracle.EVAL_CORNER_BONUS))
    assert np.array_equal(host(ops.evaluate(big, L.EVAL_MERGE_POTENTIAL)), oracle.eval_batch(hb, oracle.EVAL_MERGE_POTENTIAL))
    from environment.game_2048 import Game2048Env       # ... and through the drop-in class
    env = Game2048Env(seed=3)
    for i in (0, 17, 400, 2999):
        env.board = tiles_of(gp["board"][i]).reshape(4, 4)
        assert env._evaluate_pattern() == gp["pattern"][i]
    from agents.beam_search_agent import BeamSearchAgent     # the agent's own per-board helpers (beam_search_agent.py:183-192, :271-373)
    agent, gm = BeamSearchAgent(20, 30, seed=1), load_golden("moves.npz")
    for i in (0, 5, 123, 1500, 3014):
        t = tiles_of(g["board"][i]).reshape(4, 4)
        assert agent._fast_evaluate(t) == g["fast"][i]
        for p, name in enumerate(("early", "mid", "late")):
            assert agent._evaluate_state(t, name) == g["full"][i, p]
        assert ("early", "mid", "late").index(agent._determine_game_phase(int(t.max()))) == g["phase"][i]
    for i in (0, 40, 700, 1514):
        want = [bool((int(gm["agent_mask"][i]) >> a) & 1) for a in range(4)]
        assert agent._check_valid_moves(tiles_of(gm["board"][i]).reshape(4, 4)) == want


def test_pack_unpack_roundtrip(ops, oracle):
    b = ops.synth_boards(65537, seed=17, p_empty=0.3, max_code=17, device=DEV)
    t = ops.unpack(b)
    assert np.array_equal(host(t), oracle.unpack(host(b)))
    assert np.array_equal(host(ops.pack(t)), host(b))


def test_config1_trace_through_vec_env(ops):
    """Config 1: one board, reset, 1000 hashed actions, auto-reset -- must equal what the reference did."""
    from g2048 import VecGame2048
    g = load_golden("episodes.npz")
    env = VecGame2048(1, device=DEV, seed=int(g["seed"]), auto_reset=True, reward_f64=True)
    assert np.array_equal(host(env.boards)[0], g["c1_board0"])
    boards, rewards, dones, scores = [], [], [], []
    for t in range(1000):
        a = env.random_actions()
        assert int(a.item()) == g["c1_action"][t]
        b, r, d, info = env.step(a)
        boards.append(b.clone()); rewards.append(r.clone()); dones.append(d.clone()); scores.append(info["score"].clone())
    assert np.array_equal(host(torch.cat(boards)), g["c1_board"])
    assert np.array_equal(host(torch.cat(rewards)), g["c1_reward"])
    assert np.array_equal(host(torch.cat(dones)).astype(np.uint8), g["c1_done"])
    exp_score = np.where(g["c1_done"] == 1, 0, g["c1_score"])
    assert np.array_equal(host(torch.cat(scores)), exp_score)


def test_full_episodes_replay(ops):
    g = load_golden("episodes.npz")
    for e in range(4):
        b, sc = ops.reset(1, seed=int(g["seed"]), epoch=0, id_base=e, device=DEV)
        for t in range(g["ep%d_action" % e].shape[0]):
            a = dev(g["ep%d_action" % e][t:t + 1])
            b, r, fl = ops.step(b, a, sc, seed=int(g["seed"]), step_index=t, id_base=e, reward_f64=True)
            assert np.array_equal(host(b)[0], g["ep%d_board" % e][t]), (e, t)
            assert float(r.item()) == g["ep%d_reward" % e][t], (e, t)
            assert (int(fl.item()) & 1) == g["ep%d_done" % e][t] and int(sc.item()) == g["ep%d_score" % e][t]


def test_step_properties_full_size(ops):
    """Size-independent properties on 1,048,576 boards: tile-sum conservation (+ the spawned tile),
    valid <=> board changed before the spawn, done <=> no valid move afterwards, invalid => no spawn."""
    n = 1 << 20
    b = ops.synth_boards(n, seed=21, device=DEV)
    a = ops.synth_actions(n, seed=21, step_index=0, device=DEV)
    sc = torch.zeros(n, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(b, a, sc, seed=21, step_index=0)
    tin, tout = ops.unpack(b).sum(dim=1), ops.unpack(out).sum(dim=1)
    valid = (fl & 2).bool()
    diff = tout - tin
    assert bool(((diff == 2) | (diff == 4))[valid].all()) and bool((diff == 0)[~valid].all())
    assert bool((out == b).all(dim=1)[~valid].all())
    mask_before = ops.valid_moves(b)
    assert bool((((mask_before >> a) & 1).bool() == valid).all())
    assert bool(((ops.valid_moves(out) == 0) == (fl & 1).bool()).all())
    frac4 = float((diff[valid] == 4).float().mean())
    assert 0.09 < frac4 < 0.11
    assert bool((sc >= 0).all()) and bool((sc % 4 == 0).all())


def test_sharding_invariance(ops):
    """Results keyed by GLOBAL board id: one launch over n boards == two launches over the halves."""
    n = 200000
    b = ops.synth_boards(n, seed=31, device=DEV)
    a = ops.synth_actions(n, seed=31, step_index=4, device=DEV)
    sc = torch.zeros(n, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(b, a, sc, seed=31, step_index=4, reward_f64=True)
    h = n // 2
    for lo, hi in ((0, h), (h, n)):
        bs = ops.synth_boards(hi - lo, seed=31, id_base=lo, device=DEV)
        as_ = ops.synth_actions(hi - lo, seed=31, step_index=4, id_base=lo, device=DEV)
        scs = torch.zeros(hi - lo, dtype=torch.int32, device=DEV)
        o2, r2, f2 = ops.step(bs, as_, scs, seed=31, step_index=4, id_base=lo, reward_f64=True)
        assert bool((o2 == out[lo:hi]).all()) and bool((r2 == rw[lo:hi]).all()) and bool((f2 == fl[lo:hi]).all())


def test_metrics_kernel(ops):
    n = 123457
    b = ops.synth_boards(n, seed=41, device=DEV)
    sc = (torch.arange(n, device=DEV) % 1000).to(torch.int32)
    fl = (torch.arange(n, device=DEV) % 7 == 0).to(torch.uint8)
    m = host(ops.metrics(b, sc, fl))
    assert m[0] == n and m[1] == int(sc.sum()) and m[2] == int(fl.sum())
    hist = np.bincount(host(b).max(axis=1), minlength=18)
    assert np.array_equal(m[4:22], hist)


def test_empty_batches_are_noops(ops):
    """n == 0 through every entry point: nothing launched, nothing raised."""
    from g2048 import _lib as L
    e = torch.empty((0, 16), dtype=torch.uint8, device=DEV)
    z8 = torch.empty(0, dtype=torch.uint8, device=DEV)
    z32 = torch.empty(0, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(e, z8, z32, seed=1, step_index=0)
    assert out.shape == (0, 16) and rw.numel() == 0 and fl.numel() == 0
    assert ops.valid_moves(e).numel() == 0 and ops.evaluate(e, L.EVAL_FAST).numel() == 0
    assert ops.obs(e).shape == (0, 16) and ops.unpack(e).shape == (0, 16)
    a, p = ops.beam_get_action(e, 20, 30)
    assert a.numel() == 0 and p.numel() == 0
    assert int(ops.metrics(e).sum()) == 0
    # round-2 entry points
    succ, r, d, c = ops.simulate_move_sampled(e, z8)
    assert succ.shape == (0, 8, 16) and c.numel() == 0
    o, act, pr, rw2, fl2 = ops.rollout_step(e, torch.empty((0, 4), dtype=torch.float32, device=DEV), z32, 1, 0)
    assert o.shape == (0, 16) and act.numel() == 0
    seen = ops.SeenStates(DEV, capacity_log2=4)
    assert ops.remember_shaping(seen, e, z8, z8, torch.empty(0, dtype=torch.float64, device=DEV)).numel() == 0 and seen.index == 0


def test_config5_total_on_one_gpu_properties(ops):
    """Config 5's global problem (8,388,608 boards) in one launch: size-independent properties, plus equality of
    eight 1,048,576-board shard launches with the single launch (what the 8-GPU run computes)."""
    n = 8 * (1 << 20)
    b = ops.synth_boards(n, seed=0x2048, device=DEV)
    a = ops.synth_actions(n, seed=0x2048, step_index=0, device=DEV)
    sc = torch.zeros(n, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(b, a, sc, seed=0x2048, step_index=0)
    valid = (fl & 2).bool()
    diff = ops.unpack(out).sum(dim=1) - ops.unpack(b).sum(dim=1)
    assert bool(((diff == 2) | (diff == 4))[valid].all()) and bool((diff == 0)[~valid].all())
    assert bool(((ops.valid_moves(out) == 0) == (fl & 1).bool()).all())
    m = ops.metrics(out, sc, fl)
    assert int(m[0]) == n and int(m[1]) == int(sc.sum(dtype=torch.int64)) and int(m[4:22].sum()) == n
    shard = 1 << 20
    for r in (0, 3, 7):
        bs = ops.synth_boards(shard, seed=0x2048, id_base=r * shard, device=DEV)
        as_ = ops.synth_actions(shard, seed=0x2048, step_index=0, id_base=r * shard, device=DEV)
        scs = torch.zeros(shard, dtype=torch.int32, device=DEV)
        o2, r2, f2 = ops.step(bs, as_, scs, seed=0x2048, step_index=0, id_base=r * shard)
        sl = slice(r * shard, (r + 1) * shard)
        assert bool((o2 == out[sl]).all()) and bool((r2 == rw[sl]).all()) and bool((f2 == fl[sl]).all())
        assert bool((scs == sc[sl]).all())


def test_drop_in_env_single_step_rate(ops):
    """Config 1 plumbing: the look-alike Game2048Env (one board, n = 1 launches). Prints its step rate next to the
    reference's ~2.3e3 steps/s; must at least be in that league."""
    import time
    from environment.game_2048 import Game2048Env
    env = Game2048Env(seed=5)
    env.reset()
    for _ in range(20):
        env.step(1)
    t0 = time.perf_counter(); steps = 0
    for t in range(400):
        s, r, d, info = env.step(t & 3)
        steps += 1
        if d:
            env.reset()
    dt = time.perf_counter() - t0
    print("drop-in Game2048Env: %.0f single-board steps/s (reference CPython env: ~2.3e3)" % (steps / dt))
    assert steps / dt > 1000


def test_drop_in_env_replays_a_reference_episode(ops):
    """Config 1 through the drop-in class against the REAL reference: episode 0 of tests/golden/episodes.npz (reset spawns,
    every state, f64 reward, done, score, valid flag of a complete seeded episode) with both record homes."""
    from environment.game_2048 import Game2048Env
    g = load_golden("episodes.npz")
    for home in ("host", "device"):
        env = Game2048Env(seed=int(g["seed"]), record=home)
        assert np.array_equal(env.get_state(), tiles_of(g["ep0_board0"]))
        for t in range(g["ep0_action"].shape[0]):
            s, r, d, info = env.step(int(g["ep0_action"][t]))
            assert np.array_equal(s, tiles_of(g["ep0_board"][t])) and r == g["ep0_reward"][t], (home, t)
            assert d == bool(g["ep0_done"][t]) and info["score"] == g["ep0_score"][t] and info["valid_move"] == bool(g["ep0_valid"][t])
            assert info["highest_tile"] == max(tiles_of(g["ep0_board0"]).max(), tiles_of(g["ep0_board"][:t + 1]).max())
        assert env.game_over and not any(env.get_valid_moves())


def test_drop_in_env_one_launch_per_iteration(ops, oracle):
    """Config 1 (train.py:55-75: get_valid_moves + step every iteration): one g2048_env_step launch and one synchronisation per
    iteration whether the 80-byte record lands in pinned host memory or in device memory; get_valid_moves() is a cache read
    until `board` is assigned; both record homes and the oracle agree on every transition. The loop's rate is printed next to the
    reference-style NumPy env's on this host (oracle/pyref.py, what bench.py reports as cpu_baseline_python), not compared."""
    import time
    from environment.game_2048 import Game2048Env
    from g2048 import _lib as L
    launches = {"n": 0}
    real = L.call

    def counting_call(dev, fn, *a):
        launches["n"] += 1
        return real(dev, fn, *a)
    envs = {k: Game2048Env(seed=4321, record=k) for k in ("host", "device")}
    state = {k: e.reset() for k, e in envs.items()}
    score = 0
    L.call = counting_call
    try:
        for t in range(300):
            masks = {k: e.get_valid_moves() for k, e in envs.items()}
            assert masks["host"] == masks["device"] == [bool((oracle.env_valid_mask(state["host"]) >> a) & 1) for a in range(4)]
            a = (t * 7 + 3) % 5                                    # 4 = an action outside 0..3: moves nothing
            out = {k: e.step(a) for k, e in envs.items()}
            k0, k1 = oracle.rng_keys(4321, oracle.DOM_STEP, t)
            b, score, r, d, v, hi = oracle.env_step(state["host"], score, a, oracle.rng_draw(k0, k1, 0, 0))
            for k, (s, rew, done, info) in out.items():
                assert np.array_equal(s, b) and rew == r and done == d and info["valid_move"] == v and info["score"] == score, (k, t)
            state = {k: out[k][0] for k in out}
            if d:
                break
        assert launches["n"] == 2 * (t + 1), "one launch per env.step, none for get_valid_moves"
        e = envs["host"]
        before = launches["n"]
        e.board = np.array([[2, 4, 2, 4], [4, 2, 4, 2], [2, 4, 2, 4], [4, 2, 4, 0]], dtype=np.int32)
        assert e.get_valid_moves() == [False, False, True, True] and e.get_valid_moves() == [False, False, True, True]
        assert launches["n"] == before + 2                         # pack + one PEEK; the second call is served from the cache
    finally:
        L.call = real
    from oracle import pyref
    rates = {}
    for k in ("host", "device"):
        e = Game2048Env(seed=5, record=k)
        for i in range(50):
            e.get_valid_moves(); e.step(i & 3)
        t0 = time.perf_counter(); steps = 0
        while steps < 2000:
            e.get_valid_moves()
            s, r, d, info = e.step(steps & 3)
            steps += 1
            if d:
                e.reset()
        rates[k] = steps / (time.perf_counter() - t0)
    ref = pyref.time_steps(2000, 5)
    print("drop-in Game2048Env train.py iteration: record in host memory %.0f /s, in device memory %.0f /s; reference-style NumPy env %.0f /s"
          % (rates["host"], rates["device"], ref))
    # the rates are printed, not compared with each other: two wall-clock loops on a shared host are not a correctness fact.
    # What is asserted is structural (above: one launch and one synchronisation per iteration) plus a generous floor.
    assert min(rates.values()) > 1000 and ref > 100


def test_simulate_move_f4(ops, oracle):
    """f4: Game2048Env.simulate_move -- all successors with the reference's quirks -- vs goldens taken from the
    reference, vs the oracle on a larger random set, and through the drop-in class."""
    g = load_golden("simulate_move.npz")
    succ, rw, dn, cnt = ops.simulate_move(dev(g["board"]), dev(g["action"]), dev(g["highest_code"]))
    assert np.array_equal(host(cnt), g["count"])
    assert np.array_equal(host(succ), g["succ"])
    assert np.array_equal(host(rw), g["reward"], equal_nan=True)          # f64 ==
    assert np.array_equal(host(dn).astype(np.uint8), g["done"])
    n = 20000
    b = ops.synth_boards(n, seed=71, p_empty=0.35, max_code=12, device=DEV)
    a = ops.synth_actions(n, seed=71, step_index=0, device=DEV)
    succ, rw, dn, cnt = ops.simulate_move(b, a)                         # highest_tile = the state's own max
    hb, ha = host(b), host(a); hs, hr, hd, hc = host(succ), host(rw), host(dn), host(cnt)
    for i in range(0, n, 37):
        t = tiles_of(hb[i])
        s, r, d = oracle.simulate_move(t, int(ha[i]), int(t.max()))
        assert hc[i] == s.shape[0], i
        assert np.array_equal(tiles_of(hs[i, :hc[i]]), s) and np.array_equal(hr[i, :hc[i]], r) and np.array_equal(hd[i, :hc[i]], d), i
    from environment.game_2048 import Game2048Env
    env = Game2048Env(seed=9)
    state = env.reset()
    before = (env.board.copy(), env.score, env.highest_tile)
    res = env.simulate_move(state, 2)
    s, r, d = oracle.simulate_move(state, 2, int(env.highest_tile))
    assert len(res) == s.shape[0]
    for k, (ns, rr, dd) in enumerate(res):
        assert ns.dtype == np.int32 and np.array_equal(ns, s[k]) and rr == r[k] and dd == bool(d[k])
        assert isinstance(rr, np.float64) and isinstance(dd, bool)
    assert np.array_equal(env.board, before[0]) and env.score == before[1] and env.highest_tile == before[2]


def test_step_out_of_range_actions(ops, oracle):
    """G2048_STEP_NOOP_ACTIONS: action bytes above 3 move nothing, as in the reference (golden), mixed with real moves in one
    launch (oracle); without the flag the low two bits count."""
    g = load_golden("step_noop.npz")
    n = g["board_in"].shape[0]
    sc = dev(g["score_in"].astype(np.int32))
    out, rw, fl = ops.step(dev(g["board_in"]), dev(g["action"]), sc, seed=1, step_index=0, reward_f64=True, noop_actions=True)
    assert np.array_equal(host(out), g["board_out"]) and np.array_equal(host(sc), g["score_out"])
    assert np.array_equal(host(rw), g["reward"], equal_nan=True)
    assert np.array_equal(host(fl) & 1, g["done"]) and not ((host(fl) >> 1) & 1).any()
    n = 100003
    hb = oracle.synth_boards(n, seed=8, p_empty=0.2, max_code=7)
    ha = np.random.default_rng(8).integers(0, 9, n).astype(np.uint8)
    for ar in (False, True):
        s1 = torch.zeros(n, dtype=torch.int32, device=DEV)
        o1, r1, f1 = ops.step(dev(hb), dev(ha), s1, seed=5, step_index=2, id_base=77, reward_f64=True, auto_reset=ar, noop_actions=True)
        bo, so, ro, fo = oracle.step_batch(hb, ha, np.zeros(n, np.uint32), seed=5, step_index=2, id_base=77, opts=2 | int(ar))
        assert np.array_equal(host(o1), bo) and np.array_equal(host(s1).astype(np.uint32), so)
        assert np.array_equal(host(r1), ro, equal_nan=True) and np.array_equal(host(f1), fo)
    s2 = torch.zeros(n, dtype=torch.int32, device=DEV)
    o2, r2, f2 = ops.step(dev(hb), dev(ha), s2, seed=5, step_index=2, id_base=77, reward_f64=True)
    bo, so, ro, fo = oracle.step_batch(hb, ha & 3, np.zeros(n, np.uint32), seed=5, step_index=2, id_base=77)
    assert np.array_equal(host(o2), bo) and np.array_equal(host(f2), fo)


def test_simulate_move_sampled(ops, oracle):
    """g2048_simulate_move_sampled (the hybrid agent's simulate_move, agents/hybrid.py:578-629): golden vectors taken from the
    reference, then 200k random (state, action) pairs against the oracle."""
    g = load_golden("simulate_sampled.npz")
    succ, rw, dn, cnt = ops.simulate_move_sampled(dev(g["board"]), dev(g["action"]), seed=int(g["seed"]), step_index=int(g["step_index"]))
    assert np.array_equal(host(cnt), g["count"]) and np.array_equal(host(succ), g["succ"])
    assert np.array_equal(host(rw), g["reward"]) and np.array_equal(host(dn), g["done"].astype(bool))
    n = 200000
    hb = np.concatenate([oracle.synth_boards(n // 2, seed=31, p_empty=0.3, max_code=12), oracle.synth_boards(n // 2, seed=32, p_empty=0.85, max_code=17)])
    ha = oracle.synth_actions(n, seed=31, step_index=0)
    succ, rw, dn, cnt = ops.simulate_move_sampled(dev(hb), dev(ha), seed=77, step_index=9, id_base=(1 << 40) + 3)
    os_, orw, odn, ocnt = oracle.hybrid_simulate_batch(hb, ha, seed=77, step_index=9, id_base=(1 << 40) + 3)
    assert np.array_equal(host(cnt), ocnt) and np.array_equal(host(succ), os_)
    assert np.array_equal(host(rw), orw) and np.array_equal(host(dn), odn)


@pytest.mark.parametrize("p_empty,max_code", [(0.0, 2), (0.02, 4), (0.3, 17), (0.7, 17), (0.95, 3)])
def test_step_fuzz_all_modes(ops, oracle, p_empty, max_code):
    """Differential fuzz: 2 Mi boards per distribution (dense low tiles ... sparse huge tiles), every template mode of
    the step kernel (f64 / f32 reward x auto-reset on / off), three consecutive steps each, against the oracle."""
    n = 1 << 21
    hb = oracle.synth_boards(n, seed=1000 + max_code, id_base=5 << 33, p_empty=p_empty, max_code=max_code)
    for f64 in (True, False):
        for ar in (False, True):
            b = dev(hb); hcur = hb
            sc = torch.zeros(n, dtype=torch.int32, device=DEV); hsc = np.zeros(n, np.uint32)
            for t in range(3):
                ha = oracle.synth_actions(n, seed=77, step_index=t, id_base=5 << 33)
                out, rw, fl = ops.step(b, dev(ha), sc, seed=77, step_index=t, id_base=5 << 33, reward_f64=f64, auto_reset=ar)
                hcur, hsc, hrw, hfl = oracle.step_batch(hcur, ha, hsc, seed=77, step_index=t, id_base=5 << 33, opts=int(ar))
                assert np.array_equal(host(out), hcur) and np.array_equal(host(fl), hfl)
                assert np.array_equal(host(sc).astype(np.uint32), hsc)
                want = hrw if f64 else hrw.astype(np.float32)
                assert np.array_equal(host(rw), want, equal_nan=True)
                b = out


def test_random_action_steps_equal_explicit_actions(ops):
    """G2048_STEP_RANDOM_ACTIONS: the kernel's own uniform actions == synth_actions for the same (seed, step, id)."""
    from g2048 import VecGame2048
    n = 300001
    e1 = VecGame2048(n, device=DEV, seed=5, id_base=1 << 35, auto_reset=True, reward_f64=True)
    e2 = VecGame2048(n, device=DEV, seed=5, id_base=1 << 35, auto_reset=True, reward_f64=True)
    for t in range(6):
        a = e1.random_actions()
        b1, r1, d1, _ = e1.step(a)
        b2, r2, d2, _ = e2.step()
        assert bool((b1 == b2).all()) and bool((d1 == d2).all()) and bool((e1.scores == e2.scores).all())
        assert np.array_equal(host(r1), host(r2), equal_nan=True)


def test_obs_f16_bf16(ops, oracle):
    """Reduced-precision observations = the reference's f32 normalize_state values rounded to nearest even."""
    b = ops.synth_boards(200003, seed=91, p_empty=0.3, max_code=17, device=DEV)
    f32 = torch.from_numpy(oracle.obs_batch(host(b)))
    for dt in (torch.float16, torch.bfloat16):
        got = ops.obs(b, dtype=dt).cpu()
        assert got.dtype == dt and bool((got.view(torch.int16) == f32.to(dt).view(torch.int16)).all())
    out = torch.empty((200003, 16), dtype=torch.bfloat16, device=DEV)
    assert ops.obs(b, out=out) is out


@pytest.mark.parametrize("steps", [1, 2, 128])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_step_many_equals_sequential_steps(ops, oracle, steps, auto_reset):
    """g2048_step_many (boards in registers for T steps, in-kernel random actions) == T sequential g2048_step launches with
    G2048_STEP_RANDOM_ACTIONS, bit for bit: final boards / scores / last flags, every step's reward (f64 ==) and flags, for a
    ragged n; and, at the first steps, == the oracle fed the actions synth_actions gives."""
    n, seed, base, t0 = 100003, 4242, 3 << 34, 7
    b0, s0 = ops.reset(n, seed, 1, base, device=DEV)
    if not auto_reset:                      # start near the end of the episodes so that finished boards get stepped too
        s0.zero_()
        b0 = ops.synth_boards(n, seed=seed, id_base=base, p_empty=0.05, max_code=6, device=DEV)
    for f64 in (True, False):
        b, sc = b0.clone(), s0.clone()
        rws, fls = [], []
        for t in range(steps):
            b, rw, fl = ops.step(b, None, sc, seed, t0 + t, base, reward_f64=f64, auto_reset=auto_reset)
            rws.append(rw); fls.append(fl)
        sm = s0.clone()
        out, flast, rstream, fstream, eps = ops.step_many(b0.clone(), sm, seed, t0, steps, base, reward_f64=f64, auto_reset=auto_reset,
                                                          want_rewards=True, want_flags=True, want_episodes=True)
        assert bool((out == b).all()) and bool((sm == sc).all()) and bool((flast == fls[-1]).all())
        assert np.array_equal(host(rstream), host(torch.stack(rws)), equal_nan=True)
        assert bool((fstream == torch.stack(fls)).all())
        done = (torch.stack(fls) & 1).to(torch.int32).sum(0)
        assert bool((eps == (done if auto_reset else torch.zeros_like(done))).all())
        # in place, no optional outputs
        sm2, bi = s0.clone(), b0.clone()
        out2, fl2, r2, f2, e2 = ops.step_many(bi, sm2, seed, t0, steps, base, out=bi, reward_f64=f64, auto_reset=auto_reset)
        assert out2 is bi and r2 is None and f2 is None and e2 is None
        assert bool((bi == b).all()) and bool((sm2 == sc).all()) and bool((fl2 == fls[-1]).all())
    # the oracle on the same draws (explicit actions), two steps
    hb, hs = host(b0), host(s0).astype(np.uint32)
    k = min(steps, 2)
    for t in range(k):
        ha = oracle.synth_actions(n, seed=seed, step_index=t0 + t, id_base=base)
        hb, hs, hr, hf = oracle.step_batch(hb, ha, hs, seed=seed, step_index=t0 + t, id_base=base, opts=1 if auto_reset else 0)
    sm = s0.clone()
    out, flast, rstream, _, _ = ops.step_many(b0.clone(), sm, seed, t0, k, base, reward_f64=True, auto_reset=auto_reset, want_rewards=True)
    assert np.array_equal(host(out), hb) and np.array_equal(host(sm).astype(np.uint32), hs) and np.array_equal(host(flast), hf)
    assert np.array_equal(host(rstream[k - 1]), hr, equal_nan=True)


def test_step_many_argument_checks(ops):
    b, s = ops.reset(64, 1, 0, 0, device=DEV)
    from g2048 import _lib as L
    fl = torch.empty(64, dtype=torch.uint8, device=DEV)
    rc = L.lib().g2048_step_many(b.data_ptr(), None, b.data_ptr(), s.data_ptr(), None, None, fl.data_ptr(), None, 1, 0, 4, 0, 64, 0, None)
    assert rc == -1 and b"RANDOM_ACTIONS" in L.lib().g2048_last_error()          # neither an actions stream nor the in-kernel policy
    rc = L.lib().g2048_step_many(b.data_ptr(), None, b.data_ptr(), s.data_ptr(), None, None, fl.data_ptr(), None, 1, 0, 0, 0, 64, 4, None)
    assert rc == -1
    with pytest.raises(ValueError):
        ops.step_many(b, s, 1, 0, 0)
    with pytest.raises(ValueError):
        ops.step_many(b, s, 1, 0, 3, actions=torch.zeros((2, 64), dtype=torch.uint8, device=DEV))
    assert L.lib().g2048_step_many(None, None, None, None, None, None, None, None, 1, 0, 4, 0, 0, 4, None) == 0       # n = 0: a no-op


@pytest.mark.parametrize("auto_reset", [False, True])
def test_step_many_with_explicit_actions(ops, oracle, auto_reset):
    """g2048_step_many fed a step-major stream of explicit actions == one g2048_step launch per step with those actions, and
    == the oracle (boards, scores, f64 rewards, flags) -- e.g. a recorded move sequence replayed on many boards at once."""
    n, T, seed, base = 50021, 40, 77, 1 << 36
    b0, s0 = ops.reset(n, seed, 0, base, device=DEV)
    g = torch.Generator().manual_seed(3)
    acts = torch.randint(0, 256, (T, n), generator=g, dtype=torch.uint8).to(DEV)        # high bits set: only the low two count
    b, sc = b0.clone(), s0.clone()
    hb, hs = host(b0), host(s0).astype(np.uint32)
    rws, fls = [], []
    for t in range(T):
        b, rw, fl = ops.step(b, acts[t].contiguous(), sc, seed, 5 + t, base, reward_f64=True, auto_reset=auto_reset)
        rws.append(rw); fls.append(fl)
        if t < 3:
            hb, hs, hr, hf = oracle.step_batch(hb, host(acts[t]) & 3, hs, seed=seed, step_index=5 + t, id_base=base, opts=int(auto_reset))
            assert np.array_equal(host(b), hb) and np.array_equal(host(rw), hr, equal_nan=True) and np.array_equal(host(fl), hf)
    sm = s0.clone()
    out, flast, rstream, fstream, _ = ops.step_many(b0.clone(), sm, seed, 5, T, base, reward_f64=True, auto_reset=auto_reset,
                                                    want_rewards=True, want_flags=True, actions=acts)
    assert bool((out == b).all()) and bool((sm == sc).all()) and bool((flast == fls[-1]).all())
    assert np.array_equal(host(rstream), host(torch.stack(rws)), equal_nan=True) and bool((fstream == torch.stack(fls)).all())


def test_vec_env_random_playout_equals_steps(ops):
    """VecGame2048.random_playout(T) == T calls of step() without actions (same boards, scores, flags, rewards, step counter)."""
    from g2048 import VecGame2048
    n, T = 70001, 37
    e1 = VecGame2048(n, device=DEV, seed=11, id_base=9 << 32, auto_reset=True)
    e2 = VecGame2048(n, device=DEV, seed=11, id_base=9 << 32, auto_reset=True)
    rs = []
    for t in range(T):
        b1, r1, d1, _ = e1.step()
        rs.append(r1.clone())
    b2, fl, rew, _, eps = e2.random_playout(T, want_rewards=True, want_episodes=True)
    assert bool((b1 == b2).all()) and bool((e1.scores == e2.scores).all()) and bool((e1.flags == fl).all()) and e1.t == e2.t == T
    assert np.array_equal(host(torch.stack(rs)), host(rew), equal_nan=True) and int(eps.sum()) > 0
    b1, r1, d1, _ = e1.step()                       # and the two envs stay in step afterwards
    b2, r2, d2, _ = e2.step()
    assert bool((b1 == b2).all()) and np.array_equal(host(r1), host(r2), equal_nan=True)



def test_replay_games_equals_the_step_path_on_arbitrary_action_streams(ops, oracle):
    """g2048_replay_games against the pinned step path on action streams nobody played: 20,000 boards x 160 uniform actions
    (many invalid moves, finished boards that stay put), ragged lengths, ids with a high word. Entry t of a history = the board
    g2048_step_many reaches after t steps; the score and flags streams agree; a byte above 3 ends a game's replay; and the
    first 40 games equal the oracle's env_step move by move."""
    n, T, seed, base = 20000, 160, 99, (3 << 33) + 17
    b0, s0 = ops.reset(n, seed, 0, base, device=DEV)
    acts = torch.stack([ops.synth_actions(n, seed=seed + 1, step_index=t, id_base=base, device=DEV) for t in range(T)])      # (T, n)
    lens = (torch.arange(n, device=DEV) % (T + 1)).to(torch.int32)                         # 0 .. T moves
    game_major = acts.t().contiguous()
    bh, sh, fh = ops.replay_games(b0, game_major, lens, seed, game_id_base=base)
    assert bh.shape == (n, T + 1, 16)
    cur, sc = b0.clone(), s0.clone()
    for t in range(T):
        at = lens > t                                                                      # games that play move t
        assert torch.equal(bh[at, t], cur[at]) and torch.equal(sh[at, t], sc[at]), t
        nxt, _, fl = ops.step(cur, acts[t], sc, seed, t, base, reward_f64=True)
        assert torch.equal(fh[at, t], fl[at]), t
        cur = nxt
        done_here = lens == t + 1
        assert torch.equal(bh[done_here, t + 1], cur[done_here]) and torch.equal(sh[done_here, t + 1], sc[done_here])
    # beyond a game's end nothing is written (the wrapper zero-fills)
    col = torch.arange(T + 1, device=DEV)[None, :]
    assert bool((bh[col.expand(n, -1) > lens[:, None]] == 0).all())
    # a byte above 3 ends the replay of that game: same histories as the shortened length
    cut = game_major.clone(); cut[:, 50] = 0xFF
    bh2, sh2, _ = ops.replay_games(b0, cut, torch.full((n,), T, dtype=torch.int32, device=DEV), seed, game_id_base=base)
    bh3, sh3, _ = ops.replay_games(b0, game_major, torch.full((n,), 50, dtype=torch.int32, device=DEV), seed, game_id_base=base, longest=T)
    assert torch.equal(bh2, bh3) and torch.equal(sh2, sh3)
    hb, ha = b0.cpu().numpy(), game_major.cpu().numpy()
    for g in range(40):
        b, score = oracle.unpack(hb[g:g + 1])[0], 0
        for t in range(int(lens[g])):
            k0, k1 = oracle.rng_keys(seed, oracle.DOM_STEP, t)
            b, score, r, d, v, hi = oracle.env_step(b, score, int(ha[g, t]), oracle.rng_draw(k0, k1, base + g, 0))
            assert np.array_equal(oracle.unpack(bh[g, t + 1].cpu().numpy()[None, :])[0], b) and int(sh[g, t + 1]) == score, (g, t)


def _env_state(e):
    return [host(x).copy() for x in (e.boards, e.scores, e.reward, e.flags)]


@pytest.mark.parametrize("n,chains", [(1048576, 2), (1048576, 4), (100003, 2), (100003, 3), (700, 2), (100, 4)])
def test_chains_equal_the_single_launch(ops, n, chains):
    """VecGame2048(chains=C): the boards stepped as C independent sub-batch launches on C streams (the launch form of the
    headline bench, environment/game_2048.py:170-210 per board) give the single launch's boards, scores, rewards (f64 ==) and
    flags bit for bit -- joined after every step, left open across steps, with explicit and with in-kernel random actions,
    auto-reset on, ragged sizes and sizes smaller than one chain."""
    from g2048 import VecGame2048
    kw = dict(device=DEV, seed=77, id_base=5 << 33, auto_reset=True, reward_f64=True)
    e1, e2, e3 = VecGame2048(n, **kw), VecGame2048(n, chains=chains, **kw), VecGame2048(n, chains=chains, **kw)
    start = ops.synth_boards(n, seed=5, id_base=1, device=DEV)
    for e in (e1, e2, e3):
        e.load(start)
    assert len(e2.chain_bounds) == min(chains, -(-n // 256)) and e2.chain_bounds[-1][1] == n
    acts = [ops.synth_actions(n, seed=9, step_index=t, device=DEV) if t % 3 else None for t in range(7)]
    torch.cuda.synchronize()
    for a in acts:
        e1.step(a)
        r = e2.step(a)                   # joined every step: results usable on the current stream right away
        assert r is not None and bool((r[0] == e1.boards).all())
        assert e3.step(a, join=False) is None or len(e3.chain_bounds) == 1
    e3.join()
    s1, s2, s3 = _env_state(e1), _env_state(e2), _env_state(e3)
    for x, y, z in zip(s1, s2, s3):
        assert np.array_equal(x, y, equal_nan=True) and np.array_equal(x, z, equal_nan=True)
    assert e1.t == e2.t == e3.t == len(acts)
    # the other methods close open chains themselves
    e3.step(acts[1], join=False)
    e1.step(acts[1])
    assert bool((e3.valid_moves() == e1.valid_moves()).all()) and not e3._chains.is_open


def test_chains_inside_one_hipgraph(ops):
    """The bench's form: K steps of two open chains captured as ONE hipGraph with two parallel branches (fork at the first
    step, join after the last); replays equal the same steps launched singly."""
    from g2048 import VecGame2048
    n, K = 1 << 20, 6
    kw = dict(device=DEV, seed=0x2048)
    e1, e2 = VecGame2048(n, **kw), VecGame2048(n, chains=2, **kw)
    start = ops.synth_boards(n, seed=SEED, device=DEV)
    actions = ops.synth_actions(n, seed=SEED, device=DEV)
    e1.load(start)
    e2.load(start)
    e2.step(actions)
    e1.step(actions)                     # (creates the side stream outside the capture)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=DEV)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            for t in range(K):
                e2.step(actions, join=False)
            e2.join()
    torch.cuda.synchronize()
    # K is even: a replay leaves `boards` / `_spare` where the capture found them, so the graph can be replayed as a loop body;
    # the step index of the captured launches is fixed, which the single-launch side repeats
    t_cap = e2.t - K
    for rep in range(2):
        g.replay()
        for t in range(K):
            e1.t = t_cap + t
            e1.step(actions)
        torch.cuda.synchronize()
        for x, y in zip(_env_state(e1), _env_state(e2)):
            assert np.array_equal(x, y, equal_nan=True)


def test_step_pieces_as_env_ops_and_drop_in_methods(ops, oracle):
    """The pieces of a step driven directly, as the reference's classes allow: Game2048Env._execute_move / _move_left /
    add_new_tile (game_2048.py:97-168, :59-67) and BeamSearchAgent._make_move / _add_random_tile / _calculate_corner_bonus /
    _calculate_merge_potential (beam_search_agent.py:194-269, :375-403) -- g2048_env_step ops MOVE / MOVE_AGENT / SPAWN and two
    g2048_eval kinds underneath -- against what the REAL reference returned for the same boards (moves.npz, eval_parts.npz) and
    against the oracle for the spawn."""
    from g2048 import _lib as L
    from environment.game_2048 import Game2048Env
    from agents.beam_search_agent import BeamSearchAgent
    g, ge = load_golden("moves.npz"), load_golden("eval_parts.npz")
    env, agent = Game2048Env(seed=77), BeamSearchAgent(20, 30, seed=78)
    rng = np.random.default_rng(4)
    rows = np.concatenate([np.arange(40), rng.choice(g["board"].shape[0], 160, replace=False)])
    for i in rows:
        t = tiles_of(g["board"][i]).reshape(4, 4)
        for a in range(4):
            env.board, env.score = t.copy(), 100
            env._execute_move(a)                                                   # :97-114
            assert np.array_equal(env.board.reshape(-1), tiles_of(g["env_board"][i, a])), (i, a)
            assert int(env.score) == 100 + int(g["env_gain"][i, a])
            assert env.get_valid_moves() == [bool((int(oracle.env_valid_mask(env.board.reshape(-1))) >> k) & 1) for k in range(4)]
            nb, sc, valid = agent._make_move(t.copy(), a)                         # :194-258, DOWN quirk included
            assert np.array_equal(nb.reshape(-1), tiles_of(g["agent_board"][i, a])), (i, a)
            assert sc == int(g["agent_score"][i, a]) and valid == bool(g["agent_valid"][i, a])
        env.board = t.copy()
        assert env._move_left() == (not np.array_equal(g["env_board"][i, 0], g["board"][i]))      # :116-168 returns `changed`
        assert np.array_equal(env.board.reshape(-1), tiles_of(g["env_board"][i, 0]))
    assert not np.array_equal(g["agent_board"][rows, 3], g["env_board"][rows, 3])                 # (the quirk is in the sample)
    # an action outside 0..3: nothing moves in the env (:97-114); the agent's _make_move slides LEFT (no transform applies)
    t = tiles_of(g["board"][5]).reshape(4, 4)
    env.board = t.copy(); env._execute_move(7)
    assert np.array_equal(env.board, t)
    assert np.array_equal(agent._make_move(t.copy(), 9)[0].reshape(-1), tiles_of(g["agent_board"][5, 0]))
    # add_new_tile / _add_random_tile: the k-th direct call takes the draw (seed, STEP, k, board 0, counter 1)
    for k, i in enumerate(rows[:60]):
        t = tiles_of(g["board"][i]).reshape(4, 4)
        k0, k1 = oracle.rng_keys(env.seed, 1, env._spawns)
        h = oracle.rng_draw(k0, k1, 0, 1)
        want = t.reshape(-1).copy()
        empty = np.flatnonzero(want == 0)
        if empty.size:
            want[empty[oracle.draw_index(h, empty.size)]] = 4 if oracle.draw_is4(h) else 2
        env.board = t.copy()
        env.add_new_tile()                                                         # :59-67
        assert np.array_equal(env.board.reshape(-1), want), i
        k0, k1 = oracle.rng_keys(agent.seed, 1, agent._spawns)
        h = oracle.rng_draw(k0, k1, 0, 1)
        want = t.reshape(-1).copy()
        if empty.size:
            want[empty[oracle.draw_index(h, empty.size)]] = 4 if oracle.draw_is4(h) else 2
        mine = t.copy()
        assert agent._add_random_tile(mine) is None and np.array_equal(mine.reshape(-1), want)   # :260-269, in place
    full = np.full((4, 4), 2, np.int32); full[::2, ::2] = 4; full[1::2, 1::2] = 4
    env.board = full.copy(); env.add_new_tile()
    assert np.array_equal(env.board, full)                                         # a full board takes no tile
    for i in (0, 9, 77, 1500, 3014):
        t = tiles_of(ge["board"][i]).reshape(4, 4)
        assert agent._calculate_corner_bonus(t) == ge["corner_bonus"][i]           # :375-385
        assert agent._calculate_merge_potential(t) == ge["merge_potential"][i]     # :387-403
    # the raw ops refuse what they do not know
    b = dev(g["board"][:1].copy()); sc = torch.zeros(1, dtype=torch.int32, device=DEV); rec = torch.zeros(80, dtype=torch.uint8, device=DEV)
    with pytest.raises(RuntimeError):
        ops.env_step(b, sc, rec, 1, 0, 0, 0, 6)


@pytest.mark.parametrize("id_base,n", [((1 << 32) - 1000, 5000), ((1 << 32) - 256, 70000), ((7 << 32) - 1, 3), ((1 << 33) - 513, 1025)])
def test_step_across_a_multiple_of_2_to_32_in_the_board_ids(ops, oracle, id_base, n):
    """The step kernel hashes the low word of a board id and takes the high word's term as a launch constant; g2048_step cuts a
    launch whose ids cross a multiple of 2^32 in two there. Same results as the oracle's 64-bit ids on both sides of the cut:
    explicit actions with the f64 reward, in-kernel random actions with auto-reset, and the no-op action form."""
    b = ops.synth_boards(n, seed=9, id_base=id_base, p_empty=0.4, max_code=6, device=DEV)
    a = ops.synth_actions(n, seed=9, step_index=4, id_base=id_base, device=DEV)
    hb, ha = host(b), host(a)
    sc = torch.full((n,), 12, dtype=torch.int32, device=DEV)
    out, rw, fl = ops.step(b, a, sc, 9, 4, id_base, reward_f64=True)
    bo, so, ro, fo = oracle.step_batch(hb, ha, np.full(n, 12, np.uint32), seed=9, step_index=4, id_base=id_base)
    assert np.array_equal(host(out), bo) and np.array_equal(host(sc).astype(np.uint32), so)
    assert np.array_equal(host(rw), ro, equal_nan=True) and np.array_equal(host(fl), fo)
    # in place, f32 reward: the bits of float32(f64 reward)
    b2, sc2 = b.clone(), torch.full((n,), 12, dtype=torch.int32, device=DEV)
    _, rw2, fl2 = ops.step(b2, a, sc2, 9, 4, id_base, out=b2)
    assert torch.equal(b2, out) and torch.equal(sc2, sc) and np.array_equal(host(rw2), ro.astype(np.float32), equal_nan=True)
    # random actions + auto-reset: what the shard [id_base, id_base + n) of a larger launch computes must not depend on where it was cut
    sc3, sc4 = torch.zeros(n, dtype=torch.int32, device=DEV), torch.zeros(n, dtype=torch.int32, device=DEV)
    o3, r3, f3 = ops.step(b, None, sc3, 9, 7, id_base, auto_reset=True)
    half = n // 2
    o4, r4, f4 = torch.empty_like(o3), torch.empty_like(r3), torch.empty_like(f3)
    for lo, hi in ((0, half), (half, n)):
        if hi > lo:
            ops.step(b[lo:hi].contiguous(), None, sc4[lo:hi], 9, 7, id_base + lo, out=o4[lo:hi], reward=r4[lo:hi], flags=f4[lo:hi], auto_reset=True)
    assert torch.equal(o3, o4) and torch.equal(sc3, sc4) and torch.equal(f3, f4) and torch.equal(r3.view(torch.int32), r4.view(torch.int32))
    # the keys from a device key block (g2048_step_dyn: graph-replayable loops) take the same cut and the same high-word term
    kb = ops.KeyBlock(9, start=4, device=DEV).advance()
    sc5 = torch.full((n,), 12, dtype=torch.int32, device=DEV)
    o5, r5, f5 = ops.step(b, a, sc5, 0, 0, id_base, reward_f64=True, keyblock=kb)
    assert torch.equal(o5, out) and torch.equal(sc5, sc) and torch.equal(f5, fl) and np.array_equal(host(r5), ro, equal_nan=True)
    acts = oracle.synth_actions(n, seed=9, step_index=7, id_base=id_base)
    bo3, so3, ro3, fo3 = oracle.step_batch(hb, acts, np.zeros(n, np.uint32), seed=9, step_index=7, id_base=id_base, opts=0)
    live = (fo3 & 1) == 0                                  # boards that did not end: no reset took their place
    assert np.array_equal(host(o3)[live], bo3[live]) and np.array_equal(host(f3), fo3)


@pytest.mark.parametrize("tune", [0, 2])
@pytest.mark.parametrize("n", [1, 65, 257, 1000, 70001])
def test_step_writes_nothing_past_the_last_board(ops, n, tune):
    """The lanes past the end of a ragged last block load a clamped index and compute like everybody else (no exec region around
    the body); they must store nothing: every output array carries a sentinel tail that has to survive, in place and out of place."""
    pad = 600
    boards = ops.synth_boards(n + pad, seed=3, device=DEV)
    acts = ops.synth_actions(n + pad, seed=3, device=DEV)
    for in_place in (False, True):
        src = boards.clone()
        out = src if in_place else torch.full_like(src, 0xAB)
        if in_place:
            out[n:] = 0xAB
        sc = torch.full((n + pad,), 0x55AA55, dtype=torch.int32, device=DEV)
        rw = torch.full((n + pad,), -12345.0, dtype=torch.float32, device=DEV)
        fl = torch.full((n + pad,), 0xEE, dtype=torch.uint8, device=DEV)
        ops.step(src[:n], acts[:n], sc[:n], 3, 1, 0, out=out[:n], reward=rw[:n], flags=fl[:n], tune=tune)
        torch.cuda.synchronize()
        assert bool((out[n:] == 0xAB).all()) and bool((sc[n:] == 0x55AA55).all()) and bool((rw[n:] == -12345.0).all()) and bool((fl[n:] == 0xEE).all())
        assert not bool((fl[:n] == 0xEE).all()) or n < 3          # (and the boards in range were written)


def test_vec_env_notices_a_replaced_state_tensor(ops, oracle):
    """VecGame2048.step launches from arguments prepared once; a caller who replaces `env.boards` / `env.scores` (rather than
    copying into them) must still get a step of what the env now holds."""
    from g2048 import VecGame2048
    n = 5000
    env = VecGame2048(n, device=DEV, seed=21, id_base=77, chains=2)
    a = env.random_actions()
    env.step(a)
    nb = ops.synth_boards(n, seed=4, device=DEV)
    env.boards = nb.clone()
    env.scores = torch.full((n,), 9, dtype=torch.int32, device=DEV)
    t = env.t
    b, r, d, info = env.step(a)
    bo, so, ro, fo = oracle.step_batch(host(nb), host(a), np.full(n, 9, np.uint32), seed=21, step_index=t, id_base=77)
    assert np.array_equal(host(b), bo) and np.array_equal(host(env.scores).astype(np.uint32), so)
    assert np.array_equal(host(r), ro.astype(np.float32), equal_nan=True) and np.array_equal(host(d), (fo & 1).astype(bool))
