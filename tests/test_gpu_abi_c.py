"""The C-ABI from plain C: tests/abi_c/abi_smoke.c (HIP runtime + libg2048_hip.so, no Python / torch in the
process) must produce the arrays the oracle produces for the same seeds -- compared by checksum."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def fnv1a(a):
    h = 1469598103934665603
    for b in np.ascontiguousarray(a).view(np.uint8).reshape(-1).tolist():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return "%016x" % h


def test_plain_c_client_matches_oracle(oracle):
    d = os.path.join(REPO, "tests", "abi_c")
    subprocess.check_call(["make", "-C", d, "-s"])
    n, steps, games, pg = 20000, 5, 128, 48
    out = subprocess.run([os.path.join(d, "abi_smoke"), str(n), str(steps), str(games), str(pg)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    got = dict(zip(out.stdout.split()[0::2], out.stdout.split()[1::2]))
    round4 = {k: got.pop(k) for k in ("replay_ok", "play_actions", "play_moves", "play_scores", "env_records")}
    b = oracle.synth_boards(n, seed=0x2048)
    sc = np.zeros(n, np.uint32)
    for t in range(steps):
        a = oracle.synth_actions(n, seed=0x2048, step_index=t)
        b, sc, rw, fl = oracle.step_batch(b, a, sc, seed=0x2048, step_index=t)
    act, prob, exp = oracle.beam_batch(b[:games], 20, 30, seed=0x2048, step_index=0)
    want = {"boards": fnv1a(b), "score": fnv1a(sc), "reward": fnv1a(rw), "flags": fnv1a(fl),
            "beam_action": fnv1a(act), "beam_expanded": fnv1a(exp)}
    assert got == want

    # ---- round 4 (ABI 3) from plain C: move-sets of complete games, their replay, the one-launch env step
    assert round4["replay_ok"] == "1"
    O = oracle
    seed, cap = 0x2048, 300
    acts = np.full((pg, cap), 0xFF, np.uint8); moves = np.zeros(pg, np.int32); scores = np.zeros(pg, np.uint32)
    k0, k1 = O.rng_keys(seed, O.DOM_RESET, 0)
    for g in range(pg):
        gid = 1000 + g
        b = O.env_reset(O.rng_draw(k0, k1, gid, 0), O.rng_draw(k0, k1, gid, 1))
        sc, t, done = 0, 0, False
        while not done and t < cap:
            a = O.beam_get_action(b, -1, 8, 6, seed=seed, step_index=t, game_id=gid)["action"]
            s0, s1 = O.rng_keys(seed, O.DOM_STEP, t)
            b, sc, r, done, v, hi = O.env_step(b, sc, a, O.rng_draw(s0, s1, gid, 0))
            acts[g, t] = a
            t += 1
        moves[g], scores[g] = t, sc
    assert round4["play_actions"] == fnv1a(acts) and round4["play_moves"] == fnv1a(moves) and round4["play_scores"] == fnv1a(scores)

    def record(board, score, flags, reward):
        rec = np.zeros(80, np.uint8)
        rec[0:64] = np.asarray(board, np.int32).reshape(16).view(np.uint8)
        rec[64:68] = np.asarray([score], np.int32).view(np.uint8)
        rec[68], rec[69] = flags, O.env_valid_mask(board)
        rec[72:80] = np.asarray([reward], np.float64).view(np.uint8)
        return rec
    b = O.env_reset(O.rng_draw(k0, k1, 7, 0), O.rng_draw(k0, k1, 7, 1))
    recs = [record(b, 0, int(np.log2(b.max())) << 3, 0.0)]
    sc = 0
    for t, a in ((0, 1), (1, 7)):
        s0, s1 = O.rng_keys(seed, O.DOM_STEP, t)
        b, sc, r, done, v, hi = O.env_step(b, sc, a, O.rng_draw(s0, s1, 7, 0))
        recs.append(record(b, sc, int(done) | (int(v) << 1) | (int(np.log2(b.max())) << 3), r))
    assert round4["env_records"] == fnv1a(np.concatenate(recs))
