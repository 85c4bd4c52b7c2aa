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
    n, steps, games = 20000, 5, 128
    out = subprocess.run([os.path.join(d, "abi_smoke"), str(n), str(steps), str(games)], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode == 0, out.stderr
    got = dict(zip(out.stdout.split()[0::2], out.stdout.split()[1::2]))
    b = oracle.synth_boards(n, seed=0x2048)
    sc = np.zeros(n, np.uint32)
    for t in range(steps):
        a = oracle.synth_actions(n, seed=0x2048, step_index=t)
        b, sc, rw, fl = oracle.step_batch(b, a, sc, seed=0x2048, step_index=t)
    act, prob, exp = oracle.beam_batch(b[:games], 20, 30, seed=0x2048, step_index=0)
    want = {"boards": fnv1a(b), "score": fnv1a(sc), "reward": fnv1a(rw), "flags": fnv1a(fl),
            "beam_action": fnv1a(act), "beam_expanded": fnv1a(exp)}
    assert got == want
