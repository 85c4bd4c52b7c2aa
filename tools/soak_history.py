"""Soak of the order-from-the-previous-call path (g2048_beam_get_action_hist through ops.beam_get_action): random sequences of
batches -- sizes 4096 .. 12000 that change now and then, fresh roots every call, widths 1 .. 32, depths 3 .. 30, caller masks,
both DOWN modes -- each compared with the caller-order run of the same batch.  usage: python tools/soak_history.py [seconds]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda")
gen = torch.Generator().manual_seed(2048)
def ri(lo, hi):
    return int(torch.randint(lo, hi + 1, (1,), generator=gen))
t0 = time.time(); calls = bad = 0; n = 4096; last = t0
while time.time() - t0 < secs:
    if ri(0, 5) == 0:
        n = ri(4096, 12000)
    w, d = ri(1, 32), ri(3, 30)
    pe = ri(5, 80) / 100.0
    roots = ops.synth_boards(n, seed=ri(0, 1 << 30), id_base=ri(0, 1 << 20), p_empty=pe, max_code=ri(3, 15), device=dev)
    mask = torch.randint(0, 16, (n,), generator=gen, dtype=torch.uint8).to(dev) if ri(0, 3) == 0 else None
    kw = dict(seed=ri(0, 1 << 30), step_index=ri(0, 1 << 20), game_id_base=ri(0, 1 << 30), fixed_down=bool(ri(0, 1)), want_expanded=True)
    a = ops.beam_get_action(roots, w, d, mask, **kw)
    b = ops.beam_get_action(roots, w, d, mask, balanced_order=False, **kw)
    ok = all(torch.equal(x, y) for x, y in zip(a, b))
    bad += not ok; calls += 1
    if time.time() - last > 20:
        print("%d calls, %d mismatches, %.0f s" % (calls, bad, time.time() - t0), flush=True); last = time.time()
print("soak done: %d calls, %d mismatches" % (calls, bad))
sys.exit(1 if bad else 0)
