"""Shared test plumbing.

* registers the `gpu` marker (tests that need a real MI355X);
* puts the product package directory (``2048-using-reinforcement-learning_amd``)
  on sys.path, the way a user of the reference would (its top-level modules
  mirror the reference's: ``environment.game_2048``, ``agents.beam_search_agent``,
  plus the batched core ``g2048``);
* nothing here reads /root/reference: fixtures come from tests/golden/.
"""
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "2048-using-reinforcement-learning_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def tiles_of(codes):
    c = np.asarray(codes, dtype=np.int64)
    return np.where(c > 0, 1 << c, 0).astype(np.int32)
