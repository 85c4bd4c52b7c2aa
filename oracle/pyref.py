"""Reference-STYLE CPU env: a per-board NumPy object with the reference's algorithmic structure and call pattern
(test infrastructure / cpu_baseline only -- never imported by the product).

Written from scratch from the behaviour documented in SURVEY.md section 8 and Appendix A, it does per step what
the reference's `Game2048Env.step` does (environment/game_2048.py:170-210): one trial move through array views,
a row-by-row compaction with Python lists and `np.pad` (:116-168), a spawn through an injected draw, the shaped
reward from NumPy reductions (:212-277) and the game-over test by four more trial moves with board copies
(:69-95, :279-288). Its purpose is to stand in for "the reference's Python env on this host's cores" where the
reference itself cannot be (the GPU box): same interpreter, same NumPy calls per step, hence the same order of
magnitude of cost; `bench.py` reports its rate with the calibration ratio measured against the real reference
in the build container. Parity: checked against tests/golden/ like the C oracle.
"""
import numpy as np


class RefStyleEnv:
    def __init__(self, draw):
        """draw() -> 32-bit int; the spawn decision mapping is the build's (index = ((h>>16)*n)>>16, four iff
        (h & 0xffff) >= 58982)."""
        self.draw = draw
        self.grid = np.zeros((4, 4), dtype=np.int32)
        self.score = 0
        self.best = 0
        self.over = False

    # -- pieces ---------------------------------------------------------------
    def _spawn(self):
        free = np.argwhere(self.grid == 0)
        if len(free):
            h = self.draw()
            r, c = free[((h >> 16) * len(free)) >> 16]
            self.grid[r, c] = 4 if (h & 0xFFFF) >= 58982 else 2

    def _squash_rows(self):
        for r in range(4):
            line = self.grid[r]
            tiles = line[line != 0]
            if tiles.size == 0:
                continue
            out, j = [], 0
            while j < tiles.size:
                if j + 1 < tiles.size and tiles[j] == tiles[j + 1]:
                    out.append(tiles[j] * 2)
                    self.score += out[-1]
                    j += 2
                else:
                    out.append(tiles[j])
                    j += 1
            self.grid[r] = np.pad(np.array(out, dtype=np.int32), (0, 4 - len(out)), "constant")

    def _shift(self, action):
        if action == 0:
            self._squash_rows()
        elif action == 1:
            self.grid = self.grid.T; self._squash_rows(); self.grid = self.grid.T
        elif action == 2:
            self.grid = np.fliplr(self.grid); self._squash_rows(); self.grid = np.fliplr(self.grid)
        else:
            self.grid = np.fliplr(self.grid.T); self._squash_rows(); self.grid = np.fliplr(self.grid).T

    def legal(self):
        keep, keep_score, res = self.grid.copy(), self.score, []
        for a in range(4):
            self._shift(a)
            res.append(not np.array_equal(keep, self.grid))
            self.grid, self.score = keep.copy(), keep_score
        return res

    def _shaped_reward(self, moved, before, score_before):
        rew = (self.score - score_before) / 4.0
        if not moved:
            rew -= 2.0
        free_after = np.sum(self.grid == 0)
        rew += (free_after - np.sum(before == 0)) * 0.5
        g = self.grid
        rim = np.sum(g[0, :]) + np.sum(g[-1, :]) + np.sum(g[:, 0]) + np.sum(g[:, -1])
        with np.errstate(invalid="ignore", divide="ignore"):
            rew += (rim / np.sum(g)) * 1.0
        if free_after <= 2:
            rew -= 2.0
        for i in range(4):
            row, col = g[i, :], g[:, i]
            up_r = sum(row[j] >= row[j - 1] for j in range(1, 4) if row[j] > 0 and row[j - 1] > 0)
            up_c = sum(col[j] >= col[j - 1] for j in range(1, 4) if col[j] > 0 and col[j - 1] > 0)
            rew += (up_r + up_c) * 0.1
        return rew

    # -- API ----------------------------------------------------------------------
    def reset(self):
        self.grid = np.zeros((4, 4), dtype=np.int32)
        self.score, self.over = 0, False
        self._spawn(); self._spawn()
        self.best = np.max(self.grid)
        return self.grid.flatten()

    def step(self, action):
        score_before, before = self.score, self.grid.copy()
        self._shift(action)
        moved = not np.array_equal(before, self.grid)
        if moved:
            self._spawn()
        rew = self._shaped_reward(moved, before, score_before)
        self.over = not any(self.legal())
        self.best = max(self.best, np.max(self.grid))
        return self.grid.flatten(), rew, self.over, {"score": self.score, "valid_move": moved, "highest_tile": self.best}


def time_steps(n_steps=3000, seed=0x2048):
    """Config-1 style loop: one board, hashed actions, auto-reset; returns board-steps per second (1 core)."""
    import time
    state = [seed & 0xFFFFFFFF]

    def draw():                      # cheap LCG; its cost is negligible next to a step
        state[0] = (state[0] * 1664525 + 1013904223) & 0xFFFFFFFF
        return state[0]
    env = RefStyleEnv(draw)
    env.reset()
    t0 = time.perf_counter()
    for t in range(n_steps):
        _, _, done, _ = env.step(draw() >> 30)
        if done:
            env.reset()
    return n_steps / (time.perf_counter() - t0)
