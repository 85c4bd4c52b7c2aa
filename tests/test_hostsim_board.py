"""Unit tests of the SWAR board arithmetic the HIP kernels run (csrc/g2048_board.h), compiled for the
host by tests/hostsim (test harness only) and compared with the oracle and the golden vectors.
CPU only -- this is how the kernels' integer logic is checked without a GPU; the GPU parity tests
proper are in test_gpu_*.py and go through the C-ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO, load_golden

HS_DIR = os.path.join(REPO, "tests", "hostsim")


@pytest.fixture(scope="module")
def hs():
    subprocess.check_call(["make", "-C", HS_DIR, "-s"])
    return C.CDLL(os.path.join(HS_DIR, "libg2048_hostsim.so"))


def p(a, ty=C.c_uint8):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def all_rows_boards():
    codes = np.array(np.meshgrid(*[np.arange(18)] * 4, indexing="ij")).reshape(4, -1).T.astype(np.uint8)
    return np.ascontiguousarray(codes.reshape(-1, 16))          # 26244 boards, every 18^4 row once


def random_boards(n, seed=0):
    rng = np.random.default_rng(seed)
    parts = []
    for pe, mc in ((0.3, 11), (0.05, 5), (0.0, 3), (0.6, 17), (0.15, 17), (0.9, 4)):
        b = rng.integers(1, mc + 1, size=(n // 6, 16)).astype(np.uint8)
        b[rng.random(b.shape) < pe] = 0
        parts.append(b)
    return np.ascontiguousarray(np.concatenate(parts))


def hs_move(hs, boards, actions, agent):
    n = boards.shape[0]
    out = np.empty_like(boards); gain = np.empty(n, np.uint32); valid = np.empty(n, np.uint8)
    hs.hs_move(p(boards), p(actions), int(agent), p(out), p(gain, C.c_uint32), p(valid), C.c_size_t(n))
    return out, gain, valid


def test_rng_matches_oracle(hs, oracle):
    hs.hs_rng_draw.restype = C.c_uint32
    hs.hs_rng_draw.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    rows = load_golden("rng_pin.npz")["rows"]
    for seed, dom, idx, ident, ctr, k0, k1, h in rows[:: 3]:
        a, b = C.c_uint32(), C.c_uint32()
        hs.hs_rng_keys(C.c_uint64(int(seed)), C.c_uint32(int(dom)), C.c_uint64(int(idx)), C.byref(a), C.byref(b))
        assert (a.value, b.value) == (int(k0), int(k1))
        assert hs.hs_rng_draw(int(k0), int(k1), int(ident), int(ctr)) == int(h)
    # the split form (high word's term folded into the key, low word hashed: g2048_step's kernel) is the same function of the 64-bit id
    hs.hs_rng_draw_split.restype = C.c_uint32
    hs.hs_rng_draw_split.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    rng = np.random.default_rng(8)
    ids = np.concatenate([rng.integers(0, 2**63, 2000, dtype=np.uint64), np.array([0, 2**32 - 1, 2**32, 2**32 + 1, 2**63 + 5, 2**64 - 1], np.uint64)])
    for ident in ids:
        k0, k1, ctr = (int(x) for x in rng.integers(0, 2**32, 3, dtype=np.uint64))
        assert hs.hs_rng_draw_split(k0, k1, int(ident), ctr) == hs.hs_rng_draw(k0, k1, int(ident), ctr) == oracle.rng_draw(k0, k1, int(ident), ctr)


def test_transpose_rot180(hs):
    b = random_boards(600, 1)
    t = np.empty_like(b); r = np.empty_like(b)
    hs.hs_transpose(p(b), p(t), p(r), C.c_size_t(b.shape[0]))
    g = b.reshape(-1, 4, 4)
    assert np.array_equal(t.reshape(-1, 4, 4), g.transpose(0, 2, 1))
    assert np.array_equal(r.reshape(-1, 4, 4), g[:, ::-1, ::-1])


@pytest.mark.parametrize("action", [0, 1, 2, 3])
def test_slide_exhaustive_rows(hs, oracle, action):
    """Every 18^4 line, in every direction, against the golden row table (through the oracle's views)."""
    g = load_golden("row_slide.npz")
    boards = all_rows_boards()
    # orient the 4 lines of each board along the tested direction
    grid = boards.reshape(-1, 4, 4)
    want = g["out"].reshape(-1, 4, 4)
    gain = g["gain"].reshape(-1, 4).sum(axis=1)
    if action == 1:
        grid, want = grid.transpose(0, 2, 1), want.transpose(0, 2, 1)
    elif action == 2:
        grid, want = grid[:, :, ::-1], want[:, :, ::-1]
    elif action == 3:
        grid, want = grid.transpose(0, 2, 1)[:, ::-1, :], want.transpose(0, 2, 1)[:, ::-1, :]
    inp = np.ascontiguousarray(grid.reshape(-1, 16))
    out, gn, valid = hs_move(hs, inp, np.full(inp.shape[0], action, np.uint8), agent=False)
    assert np.array_equal(out.reshape(-1, 4, 4), want)
    assert np.array_equal(gn, gain.astype(np.uint32))
    assert np.array_equal(valid.astype(bool), (out != inp).any(axis=1))


def test_moves_vs_golden_and_oracle(hs, oracle):
    g = load_golden("moves.npz")
    b = np.ascontiguousarray(g["board"])
    for a in range(4):
        acts = np.full(b.shape[0], a, np.uint8)
        out, gn, valid = hs_move(hs, b, acts, agent=False)
        assert np.array_equal(out, g["env_board"][:, a]) and np.array_equal(gn, g["env_gain"][:, a].astype(np.uint32))
        out, gn, valid = hs_move(hs, b, acts, agent=True)
        assert np.array_equal(out, g["agent_board"][:, a]) and np.array_equal(gn, g["agent_score"][:, a].astype(np.uint32))
        assert np.array_equal(valid, g["agent_valid"][:, a])
    for agent, key in ((0, "env_mask"), (1, "agent_mask")):
        m = np.empty(b.shape[0], np.uint8)
        hs.hs_valid(p(b), agent, p(m), C.c_size_t(b.shape[0]))
        assert np.array_equal(m, g[key])
    rb = random_boards(60000, 2)
    for agent in (0, 1):
        m = np.empty(rb.shape[0], np.uint8)
        hs.hs_valid(p(rb), agent, p(m), C.c_size_t(rb.shape[0]))
        assert np.array_equal(m, oracle.valid_moves_batch(rb, bool(agent)))


def test_spawn_and_reset(hs, oracle):
    rb = random_boards(30000, 3)
    rng = np.random.default_rng(5)
    h = rng.integers(0, 2**32, size=rb.shape[0], dtype=np.uint64).astype(np.uint32)
    out = np.empty_like(rb)
    hs.hs_spawn(p(rb), p(h, C.c_uint32), p(out), C.c_size_t(rb.shape[0]))
    for i in range(0, rb.shape[0], 7):
        t = oracle.unpack(rb[i])[0]
        oracle.lib().g2048o_spawn(t.ctypes.data_as(C.POINTER(C.c_int32)), int(h[i]))
        assert np.array_equal(oracle.pack(t)[0], out[i]), i
    h1 = rng.integers(0, 2**32, size=5000, dtype=np.uint64).astype(np.uint32)
    h0 = h[:5000].copy()
    fb = np.empty((5000, 16), np.uint8)
    hs.hs_reset(p(h0, C.c_uint32), p(h1, C.c_uint32), p(fb), C.c_size_t(5000))
    for i in range(5000):
        assert np.array_equal(oracle.pack(oracle.env_reset(int(h0[i]), int(h1[i])))[0], fb[i])


def test_step_vs_golden(hs):
    g = load_golden("step_transitions.npz")
    b = np.ascontiguousarray(g["board_in"]); n = b.shape[0]
    out = np.empty_like(b); sc = g["score_in"].astype(np.uint32).copy()
    rw = np.empty(n, np.float64); fl = np.empty(n, np.uint8)
    hs.hs_step(p(b), p(np.ascontiguousarray(g["action"])), p(np.ascontiguousarray(g["h"]), C.c_uint32), p(out),
               p(sc, C.c_uint32), p(rw, C.c_double), p(fl), C.c_size_t(n))
    assert np.array_equal(out, g["board_out"])
    assert np.array_equal(sc, g["score_out"].astype(np.uint32))
    assert np.array_equal(rw, g["reward"], equal_nan=True)                 # f64 bit-exact
    assert np.array_equal(fl & 1, g["done"]) and np.array_equal((fl >> 1) & 1, g["valid"])
    assert np.array_equal(1 << (fl >> 3).astype(np.int64), np.maximum(g["highest_tile"], 1))


def test_step_vs_oracle_large(hs, oracle):
    b = random_boards(240000, 7); n = b.shape[0]
    rng = np.random.default_rng(9)
    acts = rng.integers(0, 4, size=n).astype(np.uint8)
    k0, k1 = oracle.rng_keys(99, oracle.DOM_STEP, 12)
    hs.hs_rng_draw.restype = C.c_uint32
    hs.hs_rng_draw.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    h = np.array([hs.hs_rng_draw(k0, k1, 1000 + i, 0) for i in range(n)], np.uint32)
    out = np.empty_like(b); sc = np.zeros(n, np.uint32); rw = np.empty(n, np.float64); fl = np.empty(n, np.uint8)
    hs.hs_step(p(b), p(acts), p(h, C.c_uint32), p(out), p(sc, C.c_uint32), p(rw, C.c_double), p(fl), C.c_size_t(n))
    bo, so, ro, fo = oracle.step_batch(b, acts, np.zeros(n, np.uint32), seed=99, step_index=12, id_base=1000)
    assert np.array_equal(out, bo) and np.array_equal(sc, so) and np.array_equal(fl, fo)
    assert np.array_equal(rw, ro, equal_nan=True)


def test_eval_vs_golden_and_oracle(hs, oracle):
    g = load_golden("eval_scores.npz")
    b = np.ascontiguousarray(g["board"]); n = b.shape[0]

    def ev(boards, kind, phase=None):
        out = np.empty(boards.shape[0], np.float64)
        hs.hs_eval(p(boards), kind, p(phase) if phase is not None else None, p(out, C.c_double), C.c_size_t(boards.shape[0]))
        return out
    assert np.array_equal(ev(b, 0), g["fast"])
    for ph in range(3):
        assert np.array_equal(ev(b, 1, np.full(n, ph, np.uint8)), g["full"][:, ph])
    assert np.array_equal(ev(b, 1), g["full"][np.arange(n), g["phase"]])        # phase derived on the fly
    assert np.array_equal(ev(b, 2), g["ppo_heuristic"])
    for k in range(4):
        assert np.array_equal(ev(b, 3 + k), g["monotonicity"][:, k])
    assert np.array_equal(ev(b, 7), g["ppo_shaping"])
    rb = random_boards(60000, 11)
    assert np.array_equal(ev(rb, 7), oracle.eval_batch(rb, oracle.EVAL_PPO_SHAPING))
    gp = load_golden("pattern.npz")                 # Game2048Env._evaluate_pattern (integer formulation vs the reference's floats)
    assert np.array_equal(ev(np.ascontiguousarray(gp["board"]), 8), gp["pattern"])
    assert np.array_equal(ev(rb, 8), oracle.eval_batch(rb, oracle.EVAL_PATTERN))
    ge = load_golden("eval_parts.npz")              # _calculate_corner_bonus / _calculate_merge_potential on their own
    gb = np.ascontiguousarray(ge["board"])
    assert np.array_equal(ev(gb, 9), ge["corner_bonus"]) and np.array_equal(ev(gb, 10), ge["merge_potential"])
    assert np.array_equal(ev(rb, 9), oracle.eval_batch(rb, oracle.EVAL_CORNER_BONUS))
    assert np.array_equal(ev(rb, 10), oracle.eval_batch(rb, oracle.EVAL_MERGE_POTENTIAL))
    assert np.array_equal(ev(rb, 0), oracle.eval_batch(rb, oracle.EVAL_FAST))
    for ph in range(3):
        pa = np.full(rb.shape[0], ph, np.uint8)
        assert np.array_equal(ev(rb, 1, pa), oracle.eval_batch(rb, oracle.EVAL_FULL, pa))
    assert np.array_equal(ev(rb, 2), oracle.eval_batch(rb, oracle.EVAL_PPO))


def test_simulate_move_vs_golden(hs):
    g = load_golden("simulate_move.npz")
    b = np.ascontiguousarray(g["board"]); n = b.shape[0]
    succ = np.zeros((n, 32, 16), np.uint8); rw = np.zeros((n, 32), np.float64)
    dn = np.zeros((n, 32), np.uint8); cnt = np.zeros(n, np.uint8)
    hs.hs_simulate(p(b), p(np.ascontiguousarray(g["action"])), p(np.ascontiguousarray(g["highest_code"])), p(succ),
                   p(rw, C.c_double), p(dn), p(cnt), C.c_size_t(n))
    assert np.array_equal(cnt, g["count"])
    assert np.array_equal(succ, g["succ"])
    assert np.array_equal(rw, g["reward"], equal_nan=True)
    assert np.array_equal(dn, g["done"])


def test_simulate_move_sampled_vs_golden(hs):
    """The hybrid agent's sampled simulate_move (agents/hybrid.py:578-629): the SWAR successor / reward arithmetic against
    what the reference returned for the same picks."""
    g = load_golden("simulate_sampled.npz")
    b = np.ascontiguousarray(g["board"]); n = b.shape[0]
    succ = np.zeros((n, 8, 16), np.uint8); rw = np.zeros((n, 8), np.float64)
    dn = np.zeros((n, 8), np.uint8); cnt = np.zeros(n, np.uint8)
    hs.hs_simulate_sampled(p(b), p(np.ascontiguousarray(g["action"])), p(np.ascontiguousarray(g["h"]), C.c_uint32), p(succ),
                           p(rw, C.c_double), p(dn), p(cnt), C.c_size_t(n))
    assert np.array_equal(cnt, g["count"])
    assert np.array_equal(succ, g["succ"])
    assert np.array_equal(rw, g["reward"])
    assert np.array_equal(dn, g["done"])
    assert set(np.unique(cnt)) == {1, 2, 4, 6}


def test_sample_action_vs_oracle_and_distribution(hs, oracle):
    rng = np.random.default_rng(3)
    n = 200000
    logits = rng.normal(size=(n, 4)).astype(np.float32) * 2
    probs = (np.exp(logits) / np.exp(logits).sum(axis=1, keepdims=True)).astype(np.float32)
    probs[:100] = 0.0                                   # degenerate rows
    mask = rng.integers(0, 16, size=n).astype(np.uint8)
    k0, k1 = oracle.rng_keys(5, 7, 9)
    h = np.array([oracle.rng_draw(k0, k1, i, 0) for i in range(n)], np.uint32)
    act = np.empty(n, np.uint8); pa = np.empty(n, np.float32)
    hs.hs_sample(p(np.ascontiguousarray(probs), C.c_float), p(mask), p(h, C.c_uint32), p(act), p(pa, C.c_float), C.c_size_t(n))
    oa, op = oracle.sample_batch(probs, mask, seed=5, step_index=9)
    assert np.array_equal(act, oa) and np.array_equal(pa.view(np.uint32), op.view(np.uint32))
    m = np.where(mask == 0, 15, mask)
    assert bool((((m >> act) & 1) == 1).all())          # never an invalid action
    # distribution: for one fixed row, empirical frequencies ~ masked renormalised probabilities
    row = np.array([[0.5, 0.2, 0.2, 0.1]], np.float32).repeat(n, 0)
    mk = np.full(n, 0b1011, np.uint8)
    hs.hs_sample(p(np.ascontiguousarray(row), C.c_float), p(mk), p(h, C.c_uint32), p(act), p(pa, C.c_float), C.c_size_t(n))
    freq = np.bincount(act, minlength=4) / n
    assert freq[2] == 0 and np.allclose(freq[[0, 1, 3]], np.array([0.5, 0.2, 0.1]) / 0.8, atol=0.005)
    assert np.allclose(pa[act == 0], 0.5 / 0.8, rtol=1e-6)
