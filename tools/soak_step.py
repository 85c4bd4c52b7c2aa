"""Randomised equality run of g2048_step against the oracle: sizes (ragged, tiny, up to a few hundred thousand boards), board-id bases
(anywhere in 64 bits, and just below multiples of 2^32 so that a launch crosses one), every opts combination (f64 / f32 reward,
auto-reset, in-kernel random actions, no-op actions, one / two boards per lane), in place or not, dense / sparse / huge-tile boards.
python tools/soak_step.py [seconds]"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

ge.ensure_built()
ge.import_package()
from g2048 import ops  # noqa: E402
from oracle import oracle as O  # noqa: E402

O.build()
DEV = "cuda:0"
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(20261005)
t_end = time.time() + budget
runs = boards_total = 0
last_note = time.time()
while time.time() < t_end:
    n = int(rng.choice([rng.integers(1, 70), rng.integers(1, 2000), rng.integers(1, 300000), 256 * rng.integers(1, 500)]))
    kind = rng.integers(0, 4)
    id_base = int([rng.integers(0, 2**63), (int(rng.integers(1, 2**20)) << 32) - int(rng.integers(0, n + 5)), rng.integers(0, 2**31), 0][kind])
    p_empty, max_code = [(0.3, 11), (0.02, 3), (0.6, 17), (0.1, 6)][rng.integers(0, 4)]
    seed, t = int(rng.integers(0, 2**62)), int(rng.integers(0, 2**40))
    f64, ar, rnd, tune = bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 4) == 0), int(rng.choice([0, 1, 2]))
    noop = (not rnd) and bool(rng.integers(0, 4) == 0)
    in_place = bool(rng.integers(0, 2))
    hb = O.synth_boards(n, seed=seed, id_base=id_base, p_empty=p_empty, max_code=max_code)
    if rnd:
        ha = O.synth_actions(n, seed=seed, step_index=t, id_base=id_base)
    elif noop:
        ha = rng.integers(0, 9, n).astype(np.uint8)
    else:
        ha = rng.integers(0, 4, n).astype(np.uint8)
    hs = rng.integers(0, 1 << 20, n).astype(np.uint32)
    b = torch.as_tensor(hb, device=DEV)
    sc = torch.as_tensor(hs.astype(np.int32), device=DEV)
    a = None if rnd else torch.as_tensor(ha, device=DEV)
    out, rw, fl = ops.step(b, a, sc, seed, t, id_base, out=b if in_place else None, reward_f64=f64, auto_reset=ar, tune=tune, noop_actions=noop)
    opts = (1 if ar else 0) | (2 if noop else 0)          # the oracle's bits: 1 auto-reset, 2 raw action bytes (no-op above 3)
    bo, so, ro, fo = O.step_batch(hb, ha, hs, seed=seed, step_index=t, id_base=id_base, opts=opts)
    ok = (np.array_equal(out.cpu().numpy(), bo) and np.array_equal(sc.cpu().numpy().astype(np.uint32), so) and np.array_equal(fl.cpu().numpy(), fo)
          and np.array_equal(rw.cpu().numpy(), ro if f64 else ro.astype(np.float32), equal_nan=True))
    if not ok:
        print("MISMATCH", dict(n=n, id_base=id_base, seed=seed, t=t, f64=f64, ar=ar, rnd=rnd, noop=noop, tune=tune, in_place=in_place, p_empty=p_empty, max_code=max_code))
        sys.exit(1)
    runs += 1
    boards_total += n
    if time.time() - last_note > 30.0:            # (a long silent run is taken for a hung one on the GPU box)
        print("  .. %d runs, %d board-steps so far" % (runs, boards_total), flush=True)
        last_note = time.time()
print("soak_step done: %d runs, %d board-steps, 0 mismatches" % (runs, boards_total))
