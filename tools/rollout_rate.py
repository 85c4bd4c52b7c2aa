"""g2048_rollout_step alone (config 4's kernel): microseconds per launch at a given number of envs, from a hipGraph of T launches
with a fixed probability tensor (no policy network).  python3 tools/rollout_rate.py [envs ...]   (default: 65536 131072 262144 1048576)
Also the workload tools/profile_round4.sh collects the kernel's counters on (ROLLOUT_PLAIN=1: plain launches, no graph)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
dev = torch.device("cuda")
T = 128
sizes = [int(x) for x in sys.argv[1:]] or [65536, 131072, 262144, 1048576]
plain = os.environ.get("ROLLOUT_PLAIN") == "1"
for n in sizes:
    boards, scores = ops.reset(n, 7, 0, 0, device=dev)
    spare = torch.empty_like(boards)
    probs = torch.full((n, 4), 0.25, device=dev)
    obs = torch.empty((2, n, 16), dtype=torch.float32, device=dev)
    masks = torch.empty((2, n), dtype=torch.uint8, device=dev)
    ops.valid_moves(boards, out=masks[0])
    acts = torch.empty(n, dtype=torch.uint8, device=dev); pr = torch.empty(n, device=dev)
    rew = torch.empty(n, device=dev); fl = torch.empty(n, dtype=torch.uint8, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)

    def loop():
        b, s = boards, spare
        for t in range(T):
            ops.rollout_step(b, probs, scores, 7, t, 0, mask=masks[t & 1], out=s, actions=acts, prob=pr, reward=rew, flags=fl,
                             obs_next=obs[(t + 1) & 1], mask_next=masks[(t + 1) & 1], step_counter=counter)
            b, s = s, b
        counter.add_(T)
    loop(); torch.cuda.synchronize()
    if plain:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); loop(); e1.record(); torch.cuda.synchronize()
        print("%8d envs: %.2f us per launch (plain launches, host-bound below ~4 us)" % (n, e0.elapsed_time(e1) * 1e3 / T))
        continue
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            loop()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    best = None
    for rep in range(5):
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    us = best * 1e3 / T
    print("%8d envs: %.2f us per launch = %.3e env-steps/s, %.0f GB/s of the 132 B per env-step (%.3f of 8 TB/s), %.2f waves per SIMD"
          % (n, us, n / us * 1e6, n * 132 / us / 1e3, n * 132 / us / 1e3 / 8000, n / 64 / 1024), flush=True)
