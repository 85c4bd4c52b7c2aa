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
    # control plane gloo, data plane RCCL (probed at init; a fallback to host staging would show up here)
    assert dist.is_initialized() and (w, r) == (1, 0)
    assert gdist.backends() == {"control": "gloo", "data": "nccl", "data_note": None}, gdist.backends()
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


SHARDED_CHILD = textwrap.dedent("""
    import os, sys, json
    sys.path[:0] = [%r, %r]
    import torch
    import __graft_entry__ as ge
    ge.import_package()
    import g2048
    from g2048 import dist as gdist
    dev = torch.device("cuda", 0)                      # both ranks share the one card of the test box
    torch.cuda.set_device(dev)
    w, r, lr = gdist.init("gloo", dev)
    res = g2048.evaluate_beam_search_sharded(45, 8, 6, seed=321, max_moves=400, device=dev, game_id_base=70)
    gdist.barrier()
    if r == 0:
        one = g2048.evaluate_beam_search(45, 8, 6, seed=321, max_moves=400, device=dev, game_id_base=70)
        keys = ("scores", "highest_tiles", "moves", "valid_moves", "invalid_moves", "milestones", "best_games",
                "total_expansions", "unfinished", "total_moves", "best_score", "best_game_idx")
        bad = [k for k in keys if res[k] != one[k]]
        assert not bad, bad
        assert (res["final_boards"] == one["final_boards"]).all() and res["parameters"]["world_size"] == 2
        print("sharded ok")
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()
""") % (REPO, PKG)


def test_two_rank_sharded_evaluation_equals_one_process(tmp_path):
    """evaluate_beam_search_sharded with two ranks (gloo, sharing the box's one GPU): 23 + 22 games, the gathered result
    equals the one-process evaluation of the 45 games -- games are keyed by global id, ranks never talk until the end."""
    script = tmp_path / "sharded_child.py"
    script.write_text(SHARDED_CHILD)
    port = str(_free_port())
    procs = []
    for rank in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                      text=True))
    try:
        outs = [p.communicate(timeout=280) for p in procs]
    finally:                                    # a stalled rank (rendezvous / barrier mismatch) must not keep the card
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                pass
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-2000:] + se[-4000:]
    assert "sharded ok" in outs[0][0]


def test_a_failing_rank_takes_the_job_down():
    """`python bench.py --gpus 2` with one rank raising right after the first barrier while the other goes on to wait at the
    next one: the job must end non-zero within seconds (the failing rank leaves through os._exit, torchrun stops the other,
    the launcher parent relays the code) -- not sit in the barrier until the driver's time limit. Two gloo ranks on one card."""
    import time
    env = dict(os.environ, G2048_DIST_BACKEND="gloo", G2048_BENCH_FAIL_RANK="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-beam",
                          "--no-rollout", "--no-extra", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert out.returncode != 0, "a rank failed but the job exited 0"
    assert "injected failure on rank 1" in out.stderr
    assert not any(l.startswith("{") for l in out.stdout.splitlines()), "no result line from a failed job"
    assert took < 240, "the surviving rank waited %.0f s" % took


def test_bench_two_ranks_on_one_card_fall_back_from_rccl_and_still_report():
    """The driver's command shape (`python bench.py --gpus 2 ...`, backend "nccl" asked for) on a box where RCCL cannot form the
    group -- here: both ranks share the one card, which RCCL refuses ("Duplicate GPU") --: the probe of the data plane fails, the
    ranks agree over gloo to stage the one post-timing exchange through the host, and the job still ends 0 with the N = 2 line:
    gathered scores equal to the one-GPU result, `backend` = gloo with a `backend_note` naming the RCCL error."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", G2048_RCCL_PROBE_TIMEOUT_S="60")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "G2048_DIST_BACKEND"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--no-beam",
                          "--no-rollout", "--no-extra", "--no-cpu-baseline", "--no-evaluation"], env=env, capture_output=True, text=True,
                         timeout=400)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["gathered_equals_single_gpu"] is True
    assert line["backend_control"] == "gloo"
    if line["backend"] != "nccl":           # (a box whose RCCL accepts two ranks on one device would simply use it)
        assert line["backend"] == "gloo" and "RCCL data plane unavailable" in line["backend_note"], line.get("backend_note")
    assert len(line["per_rank_ms_per_step"]) == 2 and line["scaling"] == "weak"
