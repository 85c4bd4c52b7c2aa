#!/usr/bin/env python3
"""Random playouts with the boards resident in registers (SURVEY 8d C2 "rollout" variant).

    python examples/random_playouts.py --boards 1048576 --steps 128

One g2048_step_many launch runs all `steps` env steps of every board (uniform actions drawn in the kernel, finished boards
restart); the per-step rewards come back as a (steps, boards) tensor. Prints the rate and a few statistics of the run."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "2048-using-reinforcement-learning_amd"))
from g2048 import VecGame2048

ap = argparse.ArgumentParser()
ap.add_argument("--boards", type=int, default=1 << 20)
ap.add_argument("--steps", type=int, default=128)
ap.add_argument("--seed", type=int, default=7)
a = ap.parse_args()
env = VecGame2048(a.boards, seed=a.seed, auto_reset=True)
env.random_playout(a.steps, want_rewards=True)              # warm-up (allocations, first touches)
torch.cuda.synchronize()
t0 = time.perf_counter()
boards, flags, rewards, _, episodes = env.random_playout(a.steps, want_rewards=True, want_episodes=True)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%d boards x %d steps in %.2f ms: %.3g board-steps/s" % (a.boards, a.steps, dt * 1e3, a.boards * a.steps / dt))
print("mean reward per step %.4f, episodes finished %d, mean score now %.1f, highest tile on a board %d" % (
    float(rewards.float().mean()), int(episodes.sum()), float(env.scores.float().mean()), 1 << int((flags >> 3).max())))
