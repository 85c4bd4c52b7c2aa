"""ctypes front-end of the CPU ORACLE (test infrastructure, NOT product code).

Loads ``oracle/libg2048_oracle.so`` (built by ``oracle/Makefile`` from
``g2048_oracle.c``, the plain-C restatement of the reference's
environment/game_2048.py, agents/beam_search_agent.py and the per-board parts
of agents/ppo_agent.py). Only tests/, ``__graft_entry__.smoke()`` and
bench.py's ``cpu_baseline`` leg may import this module; the product package
never does.

Parity status: PINNED by tests/golden/ (vectors captured from the reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libg2048_oracle.so")

DOM_STEP, DOM_RESET, DOM_BEAM, DOM_SYNTH_BOARD, DOM_SYNTH_ACTION, DOM_EPISODE, DOM_POLICY, DOM_SIMULATE = 1, 2, 3, 4, 5, 6, 7, 8
(EVAL_FAST, EVAL_FULL, EVAL_PPO, EVAL_MONO_PP, EVAL_MONO_PM, EVAL_MONO_MP, EVAL_MONO_MM, EVAL_PPO_SHAPING, EVAL_PATTERN,
 EVAL_CORNER_BONUS, EVAL_MERGE_POTENTIAL) = range(11)


def build(force=False):
    src = os.path.join(_HERE, "g2048_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libg2048_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        i32p, u8p, u32p, f64p, f32p = (C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_double), C.POINTER(C.c_float))
        sig = {
            "g2048o_rng_keys": (None, [C.c_uint64, C.c_uint32, C.c_uint64, u32p, u32p]),
            "g2048o_rng_draw": (C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]),
            "g2048o_draw_index": (C.c_uint32, [C.c_uint32, C.c_uint32]),
            "g2048o_draw_is4": (C.c_int, [C.c_uint32]),
            "g2048o_pack": (None, [i32p, u8p, C.c_size_t]),
            "g2048o_unpack": (None, [u8p, i32p, C.c_size_t]),
            "g2048o_env_move": (None, [i32p, C.c_int, i32p]),
            "g2048o_env_valid_mask": (C.c_int, [i32p]),
            "g2048o_spawn": (C.c_int, [i32p, C.c_uint32]),
            "g2048o_env_reward": (C.c_double, [i32p, i32p, C.c_int32, C.c_int]),
            "g2048o_env_step": (C.c_int, [i32p, i32p, C.c_int, C.c_uint32, f64p, C.POINTER(C.c_int), i32p]),
            "g2048o_env_reset": (None, [i32p, C.c_uint32, C.c_uint32]),
            "g2048o_simulate_move": (C.c_int, [i32p, C.c_int, C.c_int32, i32p, f64p, u8p]),
            "g2048o_agent_move": (None, [i32p, C.c_int, i32p, i32p, C.POINTER(C.c_int)]),
            "g2048o_agent_valid_mask": (C.c_int, [i32p]),
            "g2048o_phase": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
            "g2048o_fast_eval": (C.c_double, [i32p]),
            "g2048o_full_eval": (C.c_double, [i32p, C.c_int]),
            "g2048o_beam_get_action": (C.c_int, [i32p, C.c_int, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                                  u32p, C.c_size_t, C.c_uint64, C.c_uint64, C.c_uint64,
                                                  C.POINTER(C.c_int), f32p, u32p, u32p, f64p, i32p, C.c_int]),
            "g2048o_normalize_state": (None, [i32p, f32p]),
            "g2048o_monotonicity": (C.c_double, [i32p, C.c_int, C.c_int]),
            "g2048o_ppo_heuristic": (C.c_double, [i32p]),
            "g2048o_ppo_shaping": (C.c_double, [i32p, C.c_double]),
            "g2048o_synth_boards": (None, [u8p, C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint32, C.c_uint32]),
            "g2048o_synth_actions": (None, [u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t]),
            "g2048o_step_batch": (None, [u8p, u8p, u8p, u32p, f64p, u8p, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.c_size_t, C.c_uint32]),
            "g2048o_reset_batch": (None, [u8p, u32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t]),
            "g2048o_valid_moves_batch": (None, [u8p, u8p, C.c_size_t, C.c_int]),
            "g2048o_eval_batch": (None, [u8p, C.c_int, u8p, f64p, C.c_size_t]),
            "g2048o_obs_batch": (None, [u8p, f32p, C.c_size_t]),
            "g2048o_beam_batch": (None, [u8p, u8p, u8p, f32p, u32p, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                         C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t]),
            "g2048o_beam_batch_opt": (None, [u8p, u8p, u8p, f32p, u32p, C.c_int, C.c_int, C.c_int32, C.c_int32,
                                             C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int]),
            "g2048o_sample_action": (C.c_int, [f32p, C.c_int, C.c_uint32, f32p]),
            "g2048o_sample_batch": (None, [f32p, u8p, u8p, f32p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t]),
            "g2048o_hybrid_simulate_move": (C.c_int, [i32p, C.c_int, C.POINTER(C.c_int), i32p, f64p, u8p]),
            "g2048o_sample_picks": (None, [u32p, C.c_int, C.POINTER(C.c_int)]),
            "g2048o_hybrid_simulate_batch": (None, [u8p, u8p, u8p, f64p, u8p, u8p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_size_t]),
            "g2048o_remember_new": (C.c_void_p, []),
            "g2048o_remember_free": (None, [C.c_void_p]),
            "g2048o_remember_highest": (C.c_int32, [C.c_void_p]),
            "g2048o_remember_seen": (C.c_size_t, [C.c_void_p]),
            "g2048o_remember": (C.c_double, [C.c_void_p, i32p, i32p, C.c_double, C.POINTER(C.c_int)]),
            "g2048o_remember_batch": (None, [C.c_void_p, u8p, u8p, f64p, f64p, u8p, C.c_size_t]),
            "g2048o_num_threads": (C.c_int, []),
            "g2048o_set_num_threads": (None, [C.c_int]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty)) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int32).reshape(-1))


# ---------------------------------------------------------------- RNG -------
def rng_keys(seed, domain, index):
    k0, k1 = C.c_uint32(), C.c_uint32()
    lib().g2048o_rng_keys(seed & (2**64 - 1), domain, index & (2**64 - 1), C.byref(k0), C.byref(k1))
    return k0.value, k1.value


def rng_draw(k0, k1, ident, ctr=0):
    return lib().g2048o_rng_draw(k0, k1, ident & (2**64 - 1), ctr)


def draw_index(h, n):
    return lib().g2048o_draw_index(h, n)


def draw_is4(h):
    return bool(lib().g2048o_draw_is4(h))


# ------------------------------------------------------- single board -------
def pack(tiles):
    t = np.ascontiguousarray(np.asarray(tiles, dtype=np.int32).reshape(-1, 16))
    out = np.empty((t.shape[0], 16), dtype=np.uint8)
    lib().g2048o_pack(_p(t, C.c_int32), _p(out, C.c_uint8), t.shape[0])
    return out


def unpack(codes):
    c = np.ascontiguousarray(np.asarray(codes, dtype=np.uint8).reshape(-1, 16))
    out = np.empty((c.shape[0], 16), dtype=np.int32)
    lib().g2048o_unpack(_p(c, C.c_uint8), _p(out, C.c_int32), c.shape[0])
    return out


def env_move(board, action):
    b = _i32(board).copy()
    gain = C.c_int32()
    lib().g2048o_env_move(_p(b, C.c_int32), action, C.byref(gain))
    return b, gain.value


def env_valid_mask(board):
    return lib().g2048o_env_valid_mask(_p(_i32(board), C.c_int32))


def env_step(board, score, action, h, highest_tile=None):
    b = _i32(board).copy()
    sc, r, d = C.c_int32(score), C.c_double(), C.c_int()
    hi = C.c_int32(highest_tile if highest_tile is not None else 0)
    v = lib().g2048o_env_step(_p(b, C.c_int32), C.byref(sc), action, h, C.byref(r), C.byref(d),
                              C.byref(hi) if highest_tile is not None else None)
    return b, sc.value, r.value, bool(d.value), bool(v), (hi.value if highest_tile is not None else int(b.max()))


def env_reset(h0, h1):
    b = np.zeros(16, dtype=np.int32)
    lib().g2048o_env_reset(_p(b, C.c_int32), h0, h1)
    return b


def simulate_move(state, action, highest_tile):
    """Returns (succ int32 (k,16), reward f64 (k,), done bool (k,))."""
    b = _i32(state)
    succ = np.zeros((32, 16), dtype=np.int32)
    rw = np.zeros(32, dtype=np.float64)
    dn = np.zeros(32, dtype=np.uint8)
    k = lib().g2048o_simulate_move(_p(b, C.c_int32), action, int(highest_tile), _p(succ, C.c_int32), _p(rw, C.c_double),
                                   _p(dn, C.c_uint8))
    return succ[:k], rw[:k], dn[:k].astype(bool)


def agent_move(board, action):
    b = _i32(board)
    out = np.empty(16, dtype=np.int32)
    sc, v = C.c_int32(), C.c_int()
    lib().g2048o_agent_move(_p(b, C.c_int32), action, _p(out, C.c_int32), C.byref(sc), C.byref(v))
    return out, sc.value, bool(v.value)


def agent_valid_mask(board):
    return lib().g2048o_agent_valid_mask(_p(_i32(board), C.c_int32))


def fast_eval(board):
    return lib().g2048o_fast_eval(_p(_i32(board), C.c_int32))


def full_eval(board, phase):
    return lib().g2048o_full_eval(_p(_i32(board), C.c_int32), phase)


def ppo_heuristic(board):
    return lib().g2048o_ppo_heuristic(_p(_i32(board), C.c_int32))


def ppo_shaping(board, reward_in=0.0):
    return lib().g2048o_ppo_shaping(_p(_i32(board), C.c_int32), float(reward_in))


def monotonicity(board, row_dir, col_dir):
    return lib().g2048o_monotonicity(_p(_i32(board), C.c_int32), row_dir, col_dir)


def normalize_state(board):
    out = np.empty(16, dtype=np.float32)
    lib().g2048o_normalize_state(_p(_i32(board), C.c_int32), _p(out, C.c_float))
    return out


def beam_get_action(root, valid_mask4=-1, width=10, depth=15, early_thr=512, mid_thr=1024,
                    draws=None, seed=0, step_index=0, game_id=0, trace_levels=0):
    """Returns dict(action, prob, consumed, expanded[, trace_scores, trace_counts])."""
    b = _i32(root)
    a, p, nc, ne = C.c_int(), C.c_float(), C.c_uint32(), C.c_uint32()
    d = None if draws is None else np.ascontiguousarray(np.asarray(draws, dtype=np.uint32))
    ts = np.zeros((max(trace_levels, 1), width), dtype=np.float64)
    tc = np.zeros(max(trace_levels, 1), dtype=np.int32)
    rc = lib().g2048o_beam_get_action(_p(b, C.c_int32), valid_mask4, width, depth, early_thr, mid_thr,
                                      _p(d, C.c_uint32), 0 if d is None else d.size,
                                      seed, step_index, game_id, C.byref(a), C.byref(p), C.byref(nc), C.byref(ne),
                                      _p(ts, C.c_double) if trace_levels else None,
                                      _p(tc, C.c_int32) if trace_levels else None, trace_levels)
    if rc != 0:
        raise ValueError("g2048o_beam_get_action rc=%d" % rc)
    out = dict(action=a.value, prob=p.value, consumed=nc.value, expanded=ne.value)
    if trace_levels:
        out.update(trace_scores=ts, trace_counts=tc)
    return out


# ------------------------------------------------------------ batched -------
def synth_boards(n, seed=0x2048, id_base=0, p_empty=0.30, max_code=11):
    out = np.empty((n, 16), dtype=np.uint8)
    lib().g2048o_synth_boards(_p(out, C.c_uint8), seed, id_base, n, int(round(p_empty * 65536)), max_code)
    return out


def synth_actions(n, seed=0x2048, step_index=0, id_base=0):
    out = np.empty(n, dtype=np.uint8)
    lib().g2048o_synth_actions(_p(out, C.c_uint8), seed, step_index, id_base, n)
    return out


def step_batch(boards, actions, scores, seed=0x2048, step_index=0, id_base=0, opts=0):
    bi = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 16)
    n = bi.shape[0]
    ac = np.ascontiguousarray(actions, dtype=np.uint8)
    sc = np.ascontiguousarray(scores, dtype=np.uint32).copy()
    bo = np.empty_like(bi)
    rw = np.empty(n, dtype=np.float64)
    fl = np.empty(n, dtype=np.uint8)
    lib().g2048o_step_batch(_p(bi, C.c_uint8), _p(ac, C.c_uint8), _p(bo, C.c_uint8), _p(sc, C.c_uint32),
                            _p(rw, C.c_double), _p(fl, C.c_uint8), seed, step_index, id_base, n, opts)
    return bo, sc, rw, fl


def reset_batch(n, seed=0x2048, epoch=0, id_base=0):
    bo = np.empty((n, 16), dtype=np.uint8)
    sc = np.empty(n, dtype=np.uint32)
    lib().g2048o_reset_batch(_p(bo, C.c_uint8), _p(sc, C.c_uint32), seed, epoch, id_base, n)
    return bo, sc


def valid_moves_batch(boards, agent_semantics=False):
    bi = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 16)
    out = np.empty(bi.shape[0], dtype=np.uint8)
    lib().g2048o_valid_moves_batch(_p(bi, C.c_uint8), _p(out, C.c_uint8), bi.shape[0], int(agent_semantics))
    return out


def eval_batch(boards, kind, phase=None):
    bi = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 16)
    ph = None if phase is None else np.ascontiguousarray(phase, dtype=np.uint8)
    out = np.empty(bi.shape[0], dtype=np.float64)
    lib().g2048o_eval_batch(_p(bi, C.c_uint8), kind, _p(ph, C.c_uint8), _p(out, C.c_double), bi.shape[0])
    return out


def obs_batch(boards):
    bi = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 16)
    out = np.empty((bi.shape[0], 16), dtype=np.float32)
    lib().g2048o_obs_batch(_p(bi, C.c_uint8), _p(out, C.c_float), bi.shape[0])
    return out


def beam_batch(roots, width, depth, mask=None, early_thr=512, mid_thr=1024, seed=0x2048, step_index=0,
               game_id_base=0, fixed_down=False):
    bi = np.ascontiguousarray(roots, dtype=np.uint8).reshape(-1, 16)
    n = bi.shape[0]
    mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    act = np.empty(n, dtype=np.uint8)
    prob = np.empty(n, dtype=np.float32)
    exp = np.empty(n, dtype=np.uint32)
    lib().g2048o_beam_batch_opt(_p(bi, C.c_uint8), _p(mk, C.c_uint8), _p(act, C.c_uint8), _p(prob, C.c_float),
                                _p(exp, C.c_uint32), width, depth, early_thr, mid_thr, seed, step_index,
                                game_id_base, n, int(fixed_down))
    return act, prob, exp


def sample_batch(probs, mask=None, seed=0x2048, step_index=0, id_base=0):
    pr = np.ascontiguousarray(probs, dtype=np.float32).reshape(-1, 4)
    mk = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    act = np.empty(pr.shape[0], dtype=np.uint8)
    pa = np.empty(pr.shape[0], dtype=np.float32)
    lib().g2048o_sample_batch(_p(pr, C.c_float), _p(mk, C.c_uint8), _p(act, C.c_uint8), _p(pa, C.c_float),
                              seed, step_index, id_base, pr.shape[0])
    return act, pa


def num_threads():
    return lib().g2048o_num_threads()


def set_num_threads(n):
    lib().g2048o_set_num_threads(int(n))


# ------------------------------------------------------- PPOAgent.remember ---
class Remember:
    """PPOAgent.remember (agents/ppo_agent.py:234-269) with the agent's running state; strictly sequential."""

    def __init__(self):
        self._st = lib().g2048o_remember_new()

    def __del__(self):
        try:
            lib().g2048o_remember_free(self._st)
        except Exception:       # noqa: BLE001 -- interpreter shutdown
            pass

    @property
    def highest_tile_seen(self):
        return lib().g2048o_remember_highest(self._st)

    @property
    def n_seen(self):
        return lib().g2048o_remember_seen(self._st)

    def one(self, state, next_state, reward):
        nv = C.c_int()
        r = lib().g2048o_remember(self._st, _p(_i32(state), C.c_int32), _p(_i32(next_state), C.c_int32), float(reward),
                                  C.byref(nv))
        return r, bool(nv.value)

    def batch(self, state_codes, next_codes, reward_in):
        """uint8 (n,16) log2 codes, f64 (n,) -> (reward_out f64 (n,), novel bool (n,)), in order."""
        a = np.ascontiguousarray(state_codes, dtype=np.uint8).reshape(-1, 16)
        b = np.ascontiguousarray(next_codes, dtype=np.uint8).reshape(-1, 16)
        r = np.ascontiguousarray(reward_in, dtype=np.float64)
        out = np.empty(a.shape[0], dtype=np.float64)
        nov = np.empty(a.shape[0], dtype=np.uint8)
        lib().g2048o_remember_batch(self._st, _p(a, C.c_uint8), _p(b, C.c_uint8), _p(r, C.c_double), _p(out, C.c_double),
                                    _p(nov, C.c_uint8), a.shape[0])
        return out, nov.astype(bool)


# ------------------------------------------- hybrid agent's simulate_move ---
def hybrid_simulate_move(board, action, picks):
    """agents/hybrid.py:578-629. picks: indices into the row-major empty-cell list of the moved board, in random.sample's
    order. Returns (succ int32 (k,16), reward f64 (k,), done bool (k,))."""
    b = _i32(board)
    pk = (C.c_int * 3)(*([int(x) for x in picks] + [0, 0, 0])[:3])
    succ = np.zeros((6, 16), dtype=np.int32)
    rw = np.zeros(6, dtype=np.float64)
    dn = np.zeros(6, dtype=np.uint8)
    k = lib().g2048o_hybrid_simulate_move(_p(b, C.c_int32), int(action), pk, _p(succ, C.c_int32), _p(rw, C.c_double), _p(dn, C.c_uint8))
    return succ[:k], rw[:k], dn[:k].astype(bool)


def sample_picks(h3, n_empty):
    h = np.ascontiguousarray(h3, dtype=np.uint32)
    pk = (C.c_int * 3)()
    lib().g2048o_sample_picks(_p(h, C.c_uint32), int(n_empty), pk)
    return [pk[j] for j in range(min(3, int(n_empty)))]


def hybrid_simulate_batch(boards, actions, seed=0x2048, step_index=0, id_base=0):
    bi = np.ascontiguousarray(boards, dtype=np.uint8).reshape(-1, 16)
    n = bi.shape[0]
    ac = np.ascontiguousarray(actions, dtype=np.uint8)
    succ = np.zeros((n, 8, 16), dtype=np.uint8)
    rw = np.zeros((n, 8), dtype=np.float64)
    dn = np.zeros((n, 8), dtype=np.uint8)
    cnt = np.zeros(n, dtype=np.uint8)
    lib().g2048o_hybrid_simulate_batch(_p(bi, C.c_uint8), _p(ac, C.c_uint8), _p(succ, C.c_uint8), _p(rw, C.c_double), _p(dn, C.c_uint8),
                                       _p(cnt, C.c_uint8), seed, step_index, id_base, n)
    return succ, rw, dn.astype(bool), cnt
