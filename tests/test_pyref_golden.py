"""The reference-style NumPy env (oracle/pyref.py, used only as a CPU baseline stand-in) against the goldens."""
import numpy as np

from conftest import load_golden, tiles_of


def test_pyref_step_transitions():
    from oracle.pyref import RefStyleEnv
    g = load_golden("step_transitions.npz")
    for i in range(0, g["board_in"].shape[0], 9):
        h = int(g["h"][i])
        env = RefStyleEnv(lambda: h)
        env.grid = tiles_of(g["board_in"][i]).reshape(4, 4).copy()
        env.score = int(g["score_in"][i])
        env.best = env.grid.max()
        s, r, d, info = env.step(int(g["action"][i]))
        assert np.array_equal(s, tiles_of(g["board_out"][i])), i
        assert (np.isnan(r) and np.isnan(g["reward"][i])) or r == g["reward"][i], i
        assert d == bool(g["done"][i]) and info["valid_move"] == bool(g["valid"][i]) and info["score"] == g["score_out"][i]


def test_pyref_episode_replay():
    from oracle.pyref import RefStyleEnv
    g = load_golden("episodes.npz")
    draws = []
    env = RefStyleEnv(lambda: draws.pop(0))
    draws.extend(int(x) for x in g["ep1_reset_h"])
    assert np.array_equal(env.reset(), tiles_of(g["ep1_board0"]))
    for t in range(g["ep1_action"].shape[0]):
        draws.append(int(g["ep1_h"][t]))
        s, r, d, info = env.step(int(g["ep1_action"][t]))
        draws.clear()
        assert np.array_equal(s, tiles_of(g["ep1_board"][t])) and r == g["ep1_reward"][t] and d == bool(g["ep1_done"][t])
