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
