"""g2048_step_many (T steps per launch, boards in registers) against T g2048_step launches (hipGraph): us per step at 1 Mi boards.
G2048_LIB=<other build> for A/B comparisons."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
SEED = 0x2048
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
T = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda")
rb, rs = ops.reset(n, SEED, 0, 0, device=dev)
flags = torch.empty(n, dtype=torch.uint8, device=dev)
reward = torch.empty((T, n), dtype=torch.float32, device=dev)
def run(stream_rewards, reps=5):
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.step_many(rb, rs, SEED, 0, T, 0, out=rb, flags=flags, auto_reset=True, reward_stream=reward if stream_rewards else None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    return best * 1e3 / T
for w in range(2):
    run(False, 1)
print("%s: step_many %d boards x %d steps: %.2f us/step without streams, %.2f us/step with the f32 reward stream" % (
    os.path.basename(os.environ.get("G2048_LIB", "libg2048_hip.so")), n, T, run(False), run(True)))
