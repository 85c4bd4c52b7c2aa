"""Property tests (hypothesis) of the SWAR board arithmetic against the oracle on ARBITRARY boards (codes 0..17 in
any arrangement, not just reachable ones), plus the game's invariants. CPU only, through tests/hostsim."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

from conftest import REPO

HS_DIR = os.path.join(REPO, "tests", "hostsim")
subprocess.check_call(["make", "-C", HS_DIR, "-s"])
HS = C.CDLL(os.path.join(HS_DIR, "libg2048_hostsim.so"))


def p(a, ty=C.c_uint8):
    return a.ctypes.data_as(C.POINTER(ty))


boards = st.lists(st.integers(0, 17), min_size=16, max_size=16).map(lambda v: np.array(v, np.uint8).reshape(1, 16))
sparse = st.lists(st.sampled_from([0, 0, 0, 1, 1, 2, 3]), min_size=16, max_size=16).map(lambda v: np.array(v, np.uint8).reshape(1, 16))


def hs_step(b, action, h):
    out = np.empty_like(b); sc = np.zeros(1, np.uint32); rw = np.empty(1, np.float64); fl = np.empty(1, np.uint8)
    HS.hs_step(p(b), p(np.array([action], np.uint8)), p(np.array([h], np.uint32), C.c_uint32), p(out), p(sc, C.c_uint32),
               p(rw, C.c_double), p(fl), C.c_size_t(1))
    return out, int(sc[0]), float(rw[0]), int(fl[0])


@settings(max_examples=400, deadline=None)
@given(b=st.one_of(boards, sparse), action=st.integers(0, 3), h=st.integers(0, 2**32 - 1))
def test_step_equals_oracle_and_invariants(oracle, b, action, h):
    b = np.ascontiguousarray(b)
    out, gain, rw, fl = hs_step(b, action, h)
    t = oracle.unpack(b)[0]
    ob, osc, orw, od, ov, ohi = oracle.env_step(t, 0, action, h)
    assert np.array_equal(oracle.unpack(out)[0], ob) and gain == osc
    assert (np.isnan(rw) and np.isnan(orw)) or rw == orw
    assert bool(fl & 1) == od and bool(fl & 2) == ov
    tin, tout = int(t.sum()), int(oracle.unpack(out)[0].sum())
    if ov:
        assert tout - tin in (2, 4)                     # a move conserves the tile sum; the spawn adds 2 or 4
        assert (b == 0).sum() + 0 >= 0 and (out == 0).sum() >= 0
    else:
        assert np.array_equal(out, b) and gain == 0     # invalid move: nothing changes, nothing is drawn
    assert gain % 4 == 0                                # every merge scores a multiple of 4
    m = np.empty(1, np.uint8)
    HS.hs_valid(p(np.ascontiguousarray(out)), 0, p(m), C.c_size_t(1))
    assert (m[0] == 0) == bool(fl & 1)                  # done <=> no valid move on the new board
    HS.hs_valid(p(b), 0, p(m), C.c_size_t(1))
    assert bool((m[0] >> action) & 1) == ov             # valid <=> the env's mask said so


@settings(max_examples=300, deadline=None)
@given(b=st.one_of(boards, sparse))
def test_masks_evals_and_moves_equal_oracle(oracle, b):
    b = np.ascontiguousarray(b)
    t = oracle.unpack(b)[0]
    for agent in (0, 1):
        m = np.empty(1, np.uint8)
        HS.hs_valid(p(b), agent, p(m), C.c_size_t(1))
        assert m[0] == (oracle.agent_valid_mask(t) if agent else oracle.env_valid_mask(t))
    for a in range(4):
        out = np.empty_like(b); g = np.empty(1, np.uint32); v = np.empty(1, np.uint8)
        HS.hs_move(p(b), p(np.array([a], np.uint8)), 1, p(out), p(g, C.c_uint32), p(v), C.c_size_t(1))
        ob, osc, ov = oracle.agent_move(t, a)
        assert np.array_equal(oracle.unpack(out)[0], ob) and g[0] == osc and bool(v[0]) == ov
    for kind, ref in ((0, oracle.fast_eval(t)), (2, oracle.ppo_heuristic(t)), (7, oracle.ppo_shaping(t))):
        o = np.empty(1, np.float64)
        HS.hs_eval(p(b), kind, None, p(o, C.c_double), C.c_size_t(1))
        assert o[0] == ref
    for ph in range(3):
        o = np.empty(1, np.float64)
        HS.hs_eval(p(b), 1, p(np.array([ph], np.uint8)), p(o, C.c_double), C.c_size_t(1))
        assert o[0] == oracle.full_eval(t, ph)


@settings(max_examples=60, deadline=None)
@given(n=st.one_of(st.integers(1, 70), st.integers(1, 5000), st.sampled_from([1 << 10, (1 << 10) + 1, 65535, 65536, 65537, 1 << 17, 300007])),
       k0=st.integers(0, 2**32 - 1), k1=st.integers(0, 2**32 - 1))
def test_minibatch_permutation_is_a_bijection_for_every_n(n, k0, k1):
    """PPOMemory.sample draws WITHOUT replacement (agents/ppo_agent.py:25): the Feistel network walked into range
    (csrc/g2048_rng.h, the definition the kernel compiles) maps 0 .. n-1 onto 0 .. n-1 one to one for any n and any key,
    so every prefix of it -- a minibatch -- has distinct indices."""
    out = np.empty(n, np.uint64)
    HS.hs_minibatch_indices(C.c_uint64(n), C.c_uint64(n), C.c_uint32(k0), C.c_uint32(k1), p(out, C.c_uint64))
    assert np.array_equal(np.sort(out), np.arange(n, dtype=np.uint64))


def test_minibatch_permutation_spreads_and_depends_on_the_key():
    n, B = 8388608, 65536                       # config 4's buffer, a 64 Ki-sample minibatch
    a, b = np.empty(B, np.uint64), np.empty(B, np.uint64)
    HS.hs_minibatch_indices(C.c_uint64(n), C.c_uint64(B), C.c_uint32(1), C.c_uint32(2), p(a, C.c_uint64))
    HS.hs_minibatch_indices(C.c_uint64(n), C.c_uint64(B), C.c_uint32(3), C.c_uint32(2), p(b, C.c_uint64))
    assert len(np.unique(a)) == B and len(np.unique(b)) == B and a.max() < n
    hist = np.bincount((a // (n // 64)).astype(np.int64), minlength=64)
    assert hist.min() > 0.8 * B / 64 and hist.max() < 1.2 * B / 64
    assert len(np.intersect1d(a, b)) < 4 * B * B / n + 200          # two keys: about B^2 / n common indices, as independent draws would have
