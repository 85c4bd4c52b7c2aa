import sys, time, torch
sys.path.insert(0, '/root/repo'); import __graft_entry__ as ge
g = ge.import_package()
from g2048 import ops
for n in (1024, 4096, 16384):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = g.evaluate_beam_search(n, 20, 30, seed=1, max_moves=600, check_every=600)
    dt = time.perf_counter() - t0
    # kernel-only cost of the same number of beam+step launches on live boards
    env = g.VecGame2048(n, seed=1)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for t in range(200):
        a, p = ops.beam_get_action(env.boards, 20, 30, seed=1, step_index=t)
        env.step(a)
    torch.cuda.synchronize(); dk = (time.perf_counter() - t1) / 200
    print("n=%d: driver %.3f s for 600 moves = %.3f ms/move; beam+step alone %.3f ms/move" % (n, dt, dt / 600 * 1e3, dk * 1e3))
