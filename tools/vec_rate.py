"""VecGame2048.step from a Python loop: steps per second and host time per call, by batch size and number of chains -- what a user of the
batched front end gets (the bench times prepared launches).   python tools/vec_rate.py [n ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

ge.import_package()
from g2048 import VecGame2048  # noqa: E402

dev = torch.device("cuda", 0)
sizes = [int(x) for x in sys.argv[1:]] or [65536, 1 << 20]
K = 300
for n in sizes:
    for chains in (1, 2):
        for mode in ("random actions", "explicit actions", "explicit actions, join=False"):
            if mode.endswith("join=False") and chains == 1:
                continue
            env = VecGame2048(n, device=dev, seed=7, auto_reset=True, chains=chains)
            env.reset()
            acts = env.random_actions()
            for _ in range(20):
                env.step(None)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                if mode == "random actions":
                    env.step(None)
                elif mode == "explicit actions":
                    env.step(acts)
                else:
                    env.step(acts, join=False)
            t1 = time.perf_counter()
            if chains > 1:
                env.join()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("n = %8d chains = %d %-30s host %.1f us per step(), %.1f us per step with the final sync = %.3g board-steps/s" % (
                n, chains, mode, (t1 - t0) * 1e6 / K, (t2 - t0) * 1e6 / K, n * K / (t2 - t0)))
