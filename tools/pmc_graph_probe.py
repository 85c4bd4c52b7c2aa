"""Which call stalls when a hipGraph of g2048 launches is captured / replayed under rocprofv3 counter collection (--pmc)?
Writes one line per stage BEFORE and AFTER it runs to gpurun_out/<tag>_pmc_probe.txt (flushed), so that a run killed by its
timeout still shows the stage that never returned. Run once per question, under `timeout -k 10 SECONDS`:
    rocprofv3 --pmc SQ_WAVES --kernel-trace -d <dir> -- python3 tools/pmc_graph_probe.py <tag>"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops, RolloutCollector
tag = sys.argv[1] if len(sys.argv) > 1 else "probe"
log = open(os.path.join(ROOT, "gpurun_out", tag + "_pmc_probe.txt"), "w")
t00 = time.time()
def mark(s):
    log.write("%7.2f s  %s\n" % (time.time() - t00, s)); log.flush(); os.fsync(log.fileno())
dev = torch.device("cuda")
n = 65536
boards, scores = ops.reset(n, 1, 0, 0, device=dev)
flags = torch.empty(n, dtype=torch.uint8, device=dev); reward = torch.empty(n, dtype=torch.float32, device=dev)
def steps():
    for t in range(8):
        ops.step(boards, None, scores, 1, t, 0, out=boards, reward=reward, flags=flags, auto_reset=True)
mark("1 begin: eager launches on the current stream"); steps(); torch.cuda.synchronize(); mark("1 end")
side = torch.cuda.Stream(device=dev)
mark("2 begin: launches on a side stream after wait_stream, then side.synchronize()")
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    steps()
    side.synchronize()
mark("2 end")
g = torch.cuda.CUDAGraph()
mark("3 begin: capture of 8 g2048_step launches on the side stream (thread_local)")
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        steps()
torch.cuda.current_stream(dev).wait_stream(side)
mark("3 end")
mark("4 begin: first replay + synchronize"); g.replay(); torch.cuda.synchronize(); mark("4 end")
mark("5 begin: ten replays + synchronize")
for _ in range(10):
    g.replay()
torch.cuda.synchronize(); mark("5 end")
class Uniform(torch.nn.Module):
    def forward(self, x):
        return torch.full((x.shape[0], 4), 0.25, device=x.device)
T = int(os.environ.get("PROBE_T", "8"))
policy = Uniform()
if os.environ.get("PROBE_POLICY") == "transformer":      # the shape bench.py's config-4 leg uses (stock torch, random init)
    import torch.nn as nn
    class Policy(nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = nn.Linear(1, 64)
            self.enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(64, 4, 128, batch_first=True), 2)
            self.fc = nn.Sequential(nn.Linear(1024, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU())
            self.actor, self.critic = nn.Linear(64, 4), nn.Linear(64, 1)
        def forward(self, x):
            h = self.fc(self.enc(self.emb(x.view(x.shape[0], 16, 1))).reshape(x.shape[0], -1))
            return torch.softmax(self.actor(h), -1), self.critic(h)
    policy = Policy().to(dev).eval()
mark("policy %s, T = %d" % (type(policy).__name__, T))
rc = RolloutCollector(n, T, policy, device=dev, seed=3)
mark("6 begin: RolloutCollector.collect() #1 (warm-up pass on a side stream, capture, first replay)")
rc.collect(); torch.cuda.synchronize(); mark("6 end (graph captured: %s)" % (rc._graph is not None))
mark("7 begin: collect() #2 (replay)"); rc.collect(); torch.cuda.synchronize(); mark("7 end")
mark("done")
