"""When and where does each search of a 4096-game beam launch run?

Pass 1 (shipped library):  python3 tools/beam_timeline.py expansions [n]   -> gpurun_out/beam_timeline_exp.npy
Pass 2 (tools/build_ab.sh timing -DG2048_INSTRUMENT=1):
    G2048_LIB=build_ab/libg2048_timing.so python3 tools/beam_timeline.py timeline [n]
The timing build writes, instead of prob / expanded, the wall-clock tick (100 MHz) at which a block started, how long it
ran and the SIMD (XCC, SE, SH, CU, SIMD of HW_ID) it ran on."""
import os, sys
os.environ["G2048_ALLOW_INSTRUMENTED"] = "1"      # this tool reads the clock ticks a -DG2048_INSTRUMENT=1 build writes over real outputs
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops, _lib
if len(sys.argv) > 1 and sys.argv[1] == "timeline":
    assert _lib.lib().g2048_build_flags() & 1, "the timeline pass needs a -DG2048_INSTRUMENT=1 build (tools/build_ab.sh timing -DG2048_INSTRUMENT=1; G2048_LIB=build_ab/libg2048_timing.so): a product build's outputs are results, not clock ticks"
import bench
bench.torch = torch
SEED = 0x2048
mode = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda")
roots = bench.beam_roots(ops, n, 0, dev)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
os.makedirs(out, exist_ok=True)
for w in range(3):
    r = ops.beam_get_action(roots, 20, 30, seed=SEED, step_index=10, want_expanded=True)
torch.cuda.synchronize()
if mode == "expansions":
    np.save(os.path.join(out, "beam_timeline_exp.npy"), r[2].cpu().numpy())
    np.save(os.path.join(out, "beam_timeline_roots.npy"), roots.cpu().numpy())
    print("saved", n, "expansion counts, mean %.1f max %d" % (r[2].float().mean().item(), r[2].max().item()))
    sys.exit(0)
exp = np.load(os.path.join(out, "beam_timeline_exp.npy")).astype(np.int64)
for rep in range(3):
    act, prob, ex = ops.beam_get_action(roots, 20, 30, seed=SEED, step_index=10, want_expanded=True)
    torch.cuda.synchronize()
    start = prob.view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff
    word = ex.cpu().numpy().astype(np.int64) & 0xffffffff
    dur = (word & 0x3ffff) * 0.01                                  # us
    simd = word >> 18
    t0 = (start - start.min()) * 0.01
    t1 = t0 + dur
    span = t1.max()
    print("== launch %d: first start to last end %.1f us; starts within %.1f us; mean lifetime %.1f us, max %.1f us" %
          (rep, span, t0.max(), dur.mean(), dur.max()))
    ids, cnt = np.unique(simd, return_counts=True)
    print("   SIMDs used %d; searches per SIMD: %s" % (len(ids), dict(zip(*np.unique(cnt, return_counts=True)))))
    work = np.zeros(len(ids)); end = np.zeros(len(ids))
    idx = {s: i for i, s in enumerate(ids)}
    for g in range(n):
        i = idx[simd[g]]; work[i] += exp[g]; end[i] = max(end[i], t1[g])
    print("   expansions per SIMD: mean %.0f max %.0f (max / mean %.3f); SIMD end time: mean %.1f us, min %.1f, max %.1f" %
          (work.mean(), work.max(), work.max() / work.mean(), end.mean(), end.min(), end.max()))
    # residency over time
    grid = np.arange(0, span, 5.0)
    res = [(np.sum((t0 <= t) & (t1 > t)) / len(ids)) for t in grid]
    print("   searches resident per SIMD at t = 0, 5, ... us: " + " ".join("%.2f" % r_ for r_ in res))
    # speed of a search against its company: expansions per us by lifetime class
    rate = exp / np.maximum(dur, 1e-3)
    for lo, hi in ((0, 800), (800, 1700), (1700, 2100), (2100, 10**6)):
        m = (exp >= lo) & (exp < hi)
        if m.any():
            print("   searches with %d..%d expansions: %d, lifetime mean %.1f us (min %.1f max %.1f), %.1f expansions/us each" %
                  (lo, hi, m.sum(), dur[m].mean(), dur[m].min(), dur[m].max(), rate[m].mean()))
    # correlation between the time a SIMD ends and its work
    c = np.corrcoef(work, end)[0, 1]
    print("   correlation SIMD work vs SIMD end time %.2f; the 5 latest SIMDs: %s" %
          (c, [(int(cnt[i]), int(work[i]), round(float(end[i]), 1)) for i in np.argsort(-end)[:5]]))
