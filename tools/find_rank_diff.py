import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
from g2048.vec import VecGame2048
dev = torch.device("cuda")
n, w, d, seed = 4096, 20, 30, 2025
def play(rbc):
    env = VecGame2048(n, device=dev, seed=seed)
    r = ops.play_games(env.boards, env.scores, w, d, 5000, 512, 1024, seed, 0, False, True, rank_by_counting=rbc)
    return env.boards.clone(), env.scores.clone(), r
b0, s0, r0 = play(False)
b1, s1, r1 = play(True)
diff = (r0["moves"] != r1["moves"]) | (s0 != s1)
idx = torch.nonzero(diff).flatten().tolist()
print("games that differ:", idx[:10], len(idx))
if idx:
    g = idx[0]
    # replay game g step by step in both modes until the decisions differ
    env = VecGame2048(1, device=dev, seed=seed, id_base=g)
    for t in range(5000):
        a0, p0, e0 = ops.beam_get_action(env.boards, w, d, seed=seed, step_index=t, game_id_base=g, want_expanded=True)
        a1, p1, e1 = ops.beam_get_action(env.boards, w, d, seed=seed, step_index=t, game_id_base=g, want_expanded=True, rank_by_counting=True)
        if int(a0) != int(a1) or int(e0) != int(e1):
            print("game", g, "move", t, "network", int(a0), int(e0), "counting", int(a1), int(e1))
            print("root", env.boards.cpu().numpy().tolist())
            np.save("/root/repo/gpurun_out/diff_root.npy", env.boards.cpu().numpy())
            open("/root/repo/gpurun_out/diff_info.txt", "w").write("%d %d %d %d\n" % (g, t, int(a0), int(a1)))
            break
        env.step(a1)
