"""RCCL on hardware, as far as one GPU allows: a world_size-1 "nccl" process group (backend "nccl" IS RCCL on ROCm) is
initialised exactly as bench.py initialises it, the product's dist helpers run their collectives on device tensors through
it, and a hipGraph of g2048_step launches is captured and replayed while the process group (and its watchdog thread) is
alive -- the combination the multi-GPU bench relies on. Runs in a child process so that the process group does not leak
into the test session. The N > 1 data path itself is covered by tests/test_dist_gloo.py (gloo, world size 2) and by
bench.py's own shard check."""
import os
import subprocess
import sys
import textwrap

import pytest

from conftest import PKG, REPO

pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent("""
    import os, sys
    sys.path[:0] = [%r, %r]
    import torch
    import __graft_entry__ as ge
    ge.import_package()
    from g2048 import ops, dist as gdist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    w, r, lr = gdist.init("nccl", dev)
    import torch.distributed as dist
    assert dist.is_initialized() and dist.get_backend() == "nccl" and (w, r) == (1, 0)
    n = 1 << 16
    boards = ops.synth_boards(n, seed=1, device=dev)
    actions = ops.synth_actions(n, seed=1, device=dev)
    out = torch.empty_like(boards)
    scores = torch.zeros(n, dtype=torch.int32, device=dev)
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    flags = torch.empty(n, dtype=torch.uint8, device=dev)
    def steps():
        for t in range(8):
            ops.step(boards, actions, scores, 7, t, out=out, reward=reward, flags=flags)
    steps()
    torch.cuda.synchronize()
    want = scores.clone()
    scores.zero_()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            steps()
    torch.cuda.current_stream(dev).wait_stream(side)
    g.replay()
    torch.cuda.synchronize()
    assert bool((scores == want).all()), "graph replay under a live process group differs"
    gdist.barrier()
    gathered = gdist.all_gather_scores(scores)                 # all_gather (sizes) + all_gather_into_tensor over RCCL
    assert gathered.data_ptr() != scores.data_ptr() and bool((gathered == scores).all())
    m = gdist.reduce_metrics(ops.metrics(out, scores, flags))  # all_reduce over RCCL
    assert int(m[0].item()) == n
    assert gdist.max_over_ranks(1.5, dev) == 1.5
    dist.destroy_process_group()
    print("rccl ok")
""") % (REPO, PKG)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_process_group_collectives_and_graph_capture(tmp_path):
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), G2048_DIST_FORCE="1")
    script = tmp_path / "rccl_child.py"
    script.write_text(CHILD)
    res = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0 and "rccl ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
