import sys, time, os
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
ge.import_package()
from environment.game_2048 import Game2048Env
import numpy as np
for rec in ("host", "device"):
    env = Game2048Env(seed=5, record=rec)
    env.reset()
    n = 20000
    t0 = time.perf_counter()
    for i in range(n):
        vm = env.get_valid_moves()
        s, r, d, info = env.step(i & 3)
        if d:
            env.reset()
    dt = time.perf_counter() - t0
    print("record in %s memory: %.0f train.py-shaped iterations/s (%.1f us each)" % (rec, n / dt, dt / n * 1e6))
