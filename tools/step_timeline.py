"""When and where does each wavefront of a g2048_step launch run? (a -DG2048_INSTRUMENT=4 build writes start tick, end tick (10 ns)
and SIMD of every wavefront over lanes 0..2 of its reward output):  G2048_LIB=build_ab/libg2048_stiming.so python3 tools/step_timeline.py [n]"""
import os, sys
os.environ["G2048_ALLOW_INSTRUMENTED"] = "1"      # this tool reads the clock ticks a -DG2048_INSTRUMENT=4 build writes over real outputs
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops, _lib
assert _lib.lib().g2048_build_flags() & 4, "needs a -DG2048_INSTRUMENT=4 build (tools/build_ab.sh NAME -DG2048_INSTRUMENT=4; G2048_LIB=build_ab/libg2048_NAME.so): a product build's outputs are results, not clock ticks"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dev = torch.device("cuda", 0)
boards = ops.synth_boards(n, device=dev); actions = ops.synth_actions(n, device=dev)
out = torch.empty_like(boards); scores = torch.zeros(n, dtype=torch.int32, device=dev)
reward = torch.empty(n, dtype=torch.float32, device=dev); flags = torch.empty(n, dtype=torch.uint8, device=dev)
for t in range(5):
    ops.step(boards, actions, scores, 1, t, out=out, reward=reward, flags=flags)
torch.cuda.synchronize()
for rep in range(3):
    for t in range(4):      # back to back: the last one is looked at
        ops.step(boards, actions, scores, 1, 10 + t, out=out, reward=reward, flags=flags)
    torch.cuda.synchronize()
    w = reward.view(torch.int32).cpu().numpy().astype(np.int64).reshape(-1, 64) & 0xffffffff
    t0, t1, simd = w[:, 0], w[:, 1], w[:, 2]
    base = t0.min()
    s = (t0 - base) * 0.01; e = (t1 - base) * 0.01                         # us
    print("== launch %d: %d wavefronts on %d SIMDs; first start to last end %.2f us" % (rep, len(s), len(np.unique(simd)), e.max()))
    print("   starts: 50 %% by %.2f us, 90 %% by %.2f, 99 %% by %.2f, last %.2f; lifetime mean %.2f us (min %.2f, max %.2f)" % (
        np.percentile(s, 50), np.percentile(s, 90), np.percentile(s, 99), s.max(), (e - s).mean(), (e - s).min(), (e - s).max()))
    grid = np.arange(0, e.max() + 0.5, 0.5)
    print("   wavefronts resident per SIMD at t = 0, 0.5, ... us: " + " ".join("%.1f" % (np.sum((s <= t) & (e > t)) / 1024.0) for t in grid))
    print("   ends: first %.2f us, 10 %% by %.2f, 50 %% by %.2f, 90 %% by %.2f, 99 %% by %.2f" % (e.min(), np.percentile(e, 10), np.percentile(e, 50), np.percentile(e, 90), np.percentile(e, 99)))
    ids, cnt = np.unique(simd, return_counts=True)
    print("   wavefronts per SIMD: %s" % dict(zip(*np.unique(cnt, return_counts=True))))
