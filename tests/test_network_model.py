"""The beam kernel's ranking network as an algorithm, on the CPU: the step sequence and the keep-the-larger lane masks of
csrc/g2048_beam.hip (cx_mask / sort_stages / top64_desc) restated with numpy -- partner of lane e at distance j is e ^ j --
and checked on random and adversarial inputs, including the claim the kernel relies on: with up to 16 extra keys in lanes
48..63 the first 48 places of the result are the 48 largest of all 80, in order. (The GPU tests run the real network:
tests/test_gpu_beam.py::test_ranking_network_*.)"""
import numpy as np
import pytest

E = np.arange(64)


def keep_max(k, j):                      # cx_mask(K, J)
    desc = np.ones(64, bool) if k >= 64 else (E & k) == 0
    return desc == ((E & j) == 0)


def cx(key, k, j):
    other = key[E ^ j]
    return np.where(keep_max(k, j), np.maximum(key, other), np.minimum(key, other))


def stages(key, kmax):
    k = 2
    while k <= kmax:
        j = k // 2
        while j >= 1:
            key = cx(key, k, j)
            j //= 2
        k *= 2
    return key


def merge64(key):
    for j in (32, 16, 8, 4, 2, 1):
        key = cx(key, 64, j)
    return key


def top64(a, b=None):
    a = stages(a, 64)
    if b is None:
        return a
    b = stages(b, 16)                    # row 3 (lanes 48..63) ends ascending
    return merge64(np.where(E >= 48, b, a))


def test_sort64_random_and_patterns():
    rng = np.random.default_rng(1)
    for _ in range(100):
        a = rng.permutation(1 << 20)[:64] + 1
        assert np.array_equal(top64(a), np.sort(a)[::-1])
    for a in (np.arange(1, 65), np.arange(64, 0, -1), np.r_[np.arange(1, 33), np.zeros(32, int)], np.zeros(64, int)):
        assert np.array_equal(top64(a.copy()), np.sort(a)[::-1])


def test_rows_after_sixteen_stages_alternate_direction():
    rng = np.random.default_rng(2)
    a = stages(rng.permutation(1000)[:64] + 1, 16)
    for r in range(4):
        row = a[16 * r:16 * r + 16]
        assert np.array_equal(row, np.sort(row)[::-1] if r % 2 == 0 else np.sort(row))


@pytest.mark.parametrize("n_extra", [1, 5, 16])
def test_first_48_exact_with_a_tail_row(n_extra):
    rng = np.random.default_rng(n_extra)
    for trial in range(90):
        vals = rng.permutation(1 << 20)[:64 + n_extra] + 1
        if trial % 3 == 1:
            vals = np.sort(vals)          # every extra key beats every base key
        elif trial % 3 == 2:
            vals = np.sort(vals)[::-1]    # every extra key is smaller
        a = vals[:64].copy()
        b = np.zeros(64, dtype=a.dtype)
        b[48:48 + n_extra] = vals[64:]
        got = top64(a, b)
        assert np.array_equal(got[:48], np.sort(vals)[::-1][:48])
