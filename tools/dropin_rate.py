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

# the drop-in BeamSearchAgent driven as run_game drives it (evaluate_beam_search.py:29-58): get_action(state) + env.step per move
from agents.beam_search_agent import BeamSearchAgent
for width, depth in ((10, 15), (20, 30)):
    env = Game2048Env(seed=9)
    agent = BeamSearchAgent(width, depth, seed=9)
    state = env.reset()
    n = 1500
    for i in range(50):
        a, p = agent.get_action(state)
    t0 = time.perf_counter(); tg = 0.0
    for i in range(n):
        g0 = time.perf_counter()
        a, p = agent.get_action(state)
        tg += time.perf_counter() - g0
        state, r, d, info = env.step(a)
        if d:
            state = env.reset()
    dt = time.perf_counter() - t0
    print("BeamSearchAgent(%d, %d).get_action + env.step: %.0f moves/s (%.1f us per move, of which get_action %.1f us)" % (width, depth, n / dt, dt / n * 1e6, tg / n * 1e6))
