#!/usr/bin/env python3
"""bench.py -- the hot path's benchmark (contract: python bench.py --gpus N --steps K --warmup W).

A "step" is one pass of Game2048Env.step (g2048_step, HIP) over this rank's batch of synthetic boards:
BASELINE.json configs[1] -- 1,048,576 boards per GPU (cells empty with p = 0.30, else codes uniform
1..11; uniform actions, all four in every launch), inputs resident in HBM, reading a fixed input buffer
and writing a separate output buffer so every step does identical work. With N GPUs every rank steps its
own contiguous shard of 1,048,576 boards (global ids rank*1,048,576 ..., weak scaling, config 5); the only
collective is the final all-gather of per-board scores over RCCL, issued after the timed region.

Timing: W warm-up steps, then EXACTLY K steps between barrier + synchronize pairs, max over ranks.
The K launches are replayed from one hipGraph (launch-bound otherwise: a step is ~10-20 us of GPU time);
pass --no-graph for eager launches. The roofline leg times the same kernel launch by launch with HIP
events on the launch stream; the cpu_baseline leg times the CPU oracle (the C port of the reference
algorithm, OpenMP over boards) on a bounded sample of the same workload, rank 0 at N = 1 only.

One JSON line on stdout (rank 0).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
import __graft_entry__ as ge  # noqa: E402

BOARDS_PER_GPU = 1 << 20
SEED = 0x2048
STEP_BYTES_F32 = 46          # R board 16 + action 1 + score 4; W board 16 + score 4 + reward 4 + flags 1
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
BEAM_GAMES, BEAM_WIDTH, BEAM_DEPTH = 4096, 20, 30
BEAM_BYTES = 29              # SURVEY 8(d) HBM-resident-beam accounting per expansion


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-beam", action="store_true")
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--no-evaluation", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--dry-launch", action="store_true", help="print the multi-GPU child command as JSON and exit")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_command(n_gpus, argv, port=None):
    """The child command `python bench.py --gpus N` (N > 1, no torchrun environment) starts: one rank per GPU through
    torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(int(n_gpus)),
            "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else _free_port()),
            os.path.abspath(__file__)] + [a for a in argv if a != "--dry-launch"]


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Parent of a multi-GPU run: it starts the ranks as a CHILD process and relays its exit code. Nothing in this
        # process has touched (or will touch) the GPU: no torch.cuda call, no HIP library load, no exec of a process
        # that initialised the device.
        cmd = launch_command(args.gpus, sys.argv[1:])
        if args.dry_launch:
            print(json.dumps({"launch": cmd}))
            return 0
        import subprocess
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        return subprocess.call(cmd, env=env)
    if args.dry_launch:
        print(json.dumps({"launch": None}))
        return 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world           # under torchrun the environment is authoritative
    global np, torch            # imported here, not at module level: the launcher parent above needs neither
    import numpy as np
    import torch
    ndev = max(torch.cuda.device_count(), 1)
    local_dev = local_rank % ndev            # normally local_rank; a 2-rank rehearsal on one GPU shares cuda:0
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    if rank == 0:
        ge.ensure_built()
    ge.import_package()
    from g2048 import ops, _lib, dist as gdist
    gdist.init("nccl", dev)
    gdist.barrier()             # rank 0 may just have built the library
    _lib.lib()
    assert ops.selftest(dev) == 0, "device self-test failed"

    n = BOARDS_PER_GPU
    id_base = rank * n
    boards = ops.synth_boards(n, seed=SEED, id_base=id_base, device=dev)
    actions = ops.synth_actions(n, seed=SEED, step_index=0, id_base=id_base, device=dev)
    out = torch.empty_like(boards)
    scores = torch.zeros(n, dtype=torch.int32, device=dev)
    reward = torch.empty(n, dtype=torch.float32, device=dev)
    flags = torch.empty(n, dtype=torch.uint8, device=dev)

    def one_step(t):
        ops.step(boards, actions, scores, SEED, t, id_base, out=out, reward=reward, flags=flags)

    barrier = gdist.barrier

    K, W = args.steps, args.warmup
    for t in range(W):
        one_step(t)
    torch.cuda.synchronize()

    graph = None
    if not args.no_graph:
        # thread_local capture mode: with a process group alive, the RCCL watchdog thread issues event queries
        # that would invalidate a global-mode capture. If capture fails for any reason, fall back to eager.
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                    for t in range(K):
                        one_step(W + t)
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize()
            for _ in range(max(3, -(-600 // K))):   # untimed: the first replay pays the graph's one-time upload, and
                graph.replay()                      # ~10 ms of load bring the clocks to their steady state
            torch.cuda.synchronize()
        except Exception as exc:            # noqa: BLE001
            print("bench.py: hipGraph capture failed (%s); timing eager launches" % exc, file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    # ---- timed region: exactly K steps -------------------------------------------------
    # wall clock between barrier + synchronize pairs (-> value), and a HIP event pair on the launch
    # stream around the same K launches (-> average launch duration for the roofline)
    scores.zero_()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    if graph is not None:
        graph.replay()
    else:
        for t in range(K):
            one_step(W + t)
    ev1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0      # this rank's K steps are complete here ...
    barrier()
    torch.cuda.synchronize()
    kernel_s = ev0.elapsed_time(ev1) * 1e-3 / K
    elapsed = gdist.max_over_ranks(elapsed, dev)    # ... and the job's time is the slowest rank's

    # ---- final metrics reduction: all-gather of per-board scores (config 5), timed separately
    gather_ms = None
    if world > 1:
        torch.cuda.synchronize()
        barrier()
        g0 = time.perf_counter()
        gathered = gdist.all_gather_scores(scores)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert gathered.numel() == world * n and bool((gathered[rank * n:(rank + 1) * n] == scores).all())
        m = gdist.reduce_metrics(ops.metrics(out, scores, flags))
        assert int(m[0].item()) == world * n
        if rank == 0:
            # the gathered vector must equal what ONE GPU computes for the same global ids: rank 0 re-runs the K timed
            # steps of every other shard (same seed, step indices and id_base = shard start) and compares
            vb, va = torch.empty_like(boards), torch.empty_like(actions)
            vo, vs = torch.empty_like(boards), torch.empty_like(scores)
            for r in range(1, world):
                ops.synth_boards(n, seed=SEED, id_base=r * n, device=dev, out=vb)
                ops.synth_actions(n, seed=SEED, step_index=0, id_base=r * n, device=dev, out=va)
                vs.zero_()
                for t in range(K):
                    ops.step(vb, va, vs, SEED, W + t, r * n, out=vo, reward=reward, flags=flags)
                assert bool((gathered[r * n:(r + 1) * n] == vs).all()), "shard %d differs from the 1-GPU result" % r

    # ---- roofline: algorithmic bytes per launch / average launch duration over the timed region ----
    achieved = n * STEP_BYTES_F32 / kernel_s / 1e9
    traffic = None
    pmc = os.path.join(REPO, "profiles", "pmc_step.json")     # HBM bytes per launch from the rocprofv3 PMC passes
    if os.path.exists(pmc):
        traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")

    result = {
        "metric": "board-steps/sec (batched env.step)",
        "value": world * n * K / elapsed,
        "unit": "board-steps/s",
        "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8 (packed log2 tiles; f64 reward arithmetic, f32 reward out)",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: batched env.step, 1,048,576 random boards per GPU "
                               "(p_empty=0.30, codes 1..11), 4 actions per launch, input->output buffers",
                   "boards_per_gpu": n, "launch": "hipGraph of K launches" if graph is not None else "eager",
                   "parallelism": "%d shard(s) of 1,048,576 boards, no data-path collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                     "kernel": "step_kernel<false,false,1>", "kernel_us": kernel_s * 1e6,
                     "algorithmic_bytes_per_launch": n * STEP_BYTES_F32,
                     "timing": "HIP event pair on the launch stream around the K timed launches / K"},
    }
    if gather_ms is not None:
        import torch.distributed as tdist
        result["allgather_scores_ms"] = gather_ms
        result["n_ranks_seen"] = tdist.get_world_size()
        result["backend"] = tdist.get_backend()
        result["gathered_equals_single_gpu"] = True        # asserted above on rank 0

    # ---- beam search leg (config 3): 4096 concurrent games, width 20, depth 30 -----------
    if not args.no_beam:
        roots = torch.cat([ops.synth_boards(BEAM_GAMES // 2, seed=SEED + 1, id_base=rank * BEAM_GAMES, device=dev),
                           ops.synth_boards(BEAM_GAMES // 2, seed=SEED + 2, id_base=rank * BEAM_GAMES, p_empty=0.45,
                                            max_code=9, device=dev)])
        for w in range(2):
            a, p, e = ops.beam_get_action(roots, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=w,
                                          game_id_base=rank * BEAM_GAMES, want_expanded=True)
        torch.cuda.synchronize()
        breps = 20
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        exps = []
        b0.record()
        for w in range(breps):          # back to back on one stream: the GPU never waits for the host
            a, p, e = ops.beam_get_action(roots, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=10 + w,
                                          game_id_base=rank * BEAM_GAMES, want_expanded=True)
            exps.append(e)
        b1.record()
        torch.cuda.synchronize()
        bsec = b0.elapsed_time(b1) * 1e-3
        total_exp = int(torch.stack(exps).sum().item())
        result["beam"] = {"metric": "beam node-expansions/s (width=20, depth=30, 4096 concurrent games)",
                          "value": total_exp / bsec, "unit": "expansions/s",
                          "decisions_per_s": BEAM_GAMES * breps / bsec, "ms_per_batch_decision": bsec / breps * 1e3,
                          "expansions_per_decision": total_exp / (BEAM_GAMES * breps),
                          "hbm_equivalent_GBs_at_29B": total_exp / bsec * BEAM_BYTES / 1e9}

    # ---- evaluation leg: 4096 beam-search games (w=20, d=30) played to completion, fused per game
    if not args.no_evaluation and not args.no_beam and world == 1:
        from g2048 import evaluate_beam_search
        evaluate_beam_search(256, BEAM_WIDTH, BEAM_DEPTH, seed=1, max_moves=50, device=dev)      # warm
        ev = evaluate_beam_search(BEAM_GAMES, BEAM_WIDTH, BEAM_DEPTH, seed=2025, max_moves=5000, device=dev)
        sm = ev["summary"]
        result["evaluation"] = {"metric": "4096 complete beam-search games (width 20, depth 30, 5000-move cap), one launch",
                                "seconds": ev["elapsed_s"], "moves": ev["total_moves"], "moves_per_s": sm["moves_per_s"],
                                "expansions_per_s": sm["expansions_per_s"], "rate_2048_or_more": sm["rate_2048_or_more"],
                                "average_score": sm["average_score"], "reference_report_md": {"rate_2048_or_more": 0.35,
                                                                                                "average_score": 18945.6}}

    # ---- C2 "rollout" variant: 128 consecutive in-place steps from reset states, on-device random actions
    # drawn inside the step kernel (G2048_STEP_RANDOM_ACTIONS; realistic tile distribution instead of the synthetic
    # one), auto-reset on; one hipGraph of 128 step launches
    if not args.no_rollout:
        rb, rs = ops.reset(n, SEED, 0, id_base, device=dev)

        def rollout_steps(t0):
            for t in range(t0, t0 + 128):
                ops.step(rb, None, rs, SEED, t, id_base, out=rb, reward=reward, flags=flags, auto_reset=True)
        rollout_steps(0)
        torch.cuda.synchronize()
        rg = None
        try:
            side2 = torch.cuda.Stream(device=dev)
            side2.wait_stream(torch.cuda.current_stream(dev))
            rg = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side2):
                with torch.cuda.graph(rg, stream=side2, capture_error_mode="thread_local"):
                    rollout_steps(128)
            torch.cuda.current_stream(dev).wait_stream(side2)
            torch.cuda.synchronize()
            rg.replay()
            torch.cuda.synchronize()
        except Exception:           # noqa: BLE001
            rg = None
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        q0.record()
        if rg is not None:
            rg.replay()
        else:
            rollout_steps(128)
        q1.record()
        torch.cuda.synchronize()
        rsec = q0.elapsed_time(q1) * 1e-3
        result["rollout_random"] = {"metric": "board-steps/s, 1,048,576 boards x 128 consecutive in-place steps from reset, "
                                              "uniform actions drawn in the kernel, auto-reset (realistic tile distribution)",
                                    "value": n * 128 / rsec, "unit": "board-steps/s", "us_per_step": rsec / 128 * 1e6}

    # ---- PPO rollout leg (config 4): 65,536 envs x 128 steps, transformer policy on PyTorch-ROCm -----
    if not args.no_rollout and world == 1:
        import torch.nn as nn
        from g2048 import RolloutCollector

        class Policy(nn.Module):            # the reference's models/transformer.py shape, stock torch, random init
            def __init__(self):
                super().__init__()
                self.emb = nn.Linear(1, 64)
                self.enc = nn.TransformerEncoder(nn.TransformerEncoderLayer(64, 4, 128, batch_first=True), 2)
                self.fc = nn.Sequential(nn.Linear(1024, 128), nn.ReLU(), nn.Linear(128, 64), nn.ReLU())
                self.actor, self.critic = nn.Linear(64, 4), nn.Linear(64, 1)

            def forward(self, x):
                h = self.fc(self.enc(self.emb(x.view(x.shape[0], 16, 1))).reshape(x.shape[0], -1))
                return torch.softmax(self.actor(h), -1), self.critic(h)

        class Uniform(nn.Module):           # no network: isolates the env side of the rollout
            def forward(self, x):
                return torch.full((x.shape[0], 4), 0.25, device=x.device)

        torch.manual_seed(0)
        rres = {}
        for name, pol in (("transformer_policy", Policy().to(dev).eval()), ("uniform_policy_env_only", Uniform())):
            rc = RolloutCollector(65536, 128, pol, device=dev, seed=SEED)
            rc.collect()
            torch.cuda.synchronize()
            r0 = time.perf_counter()
            rc.collect()
            torch.cuda.synchronize()
            rres[name] = 65536 * 128 / (time.perf_counter() - r0)
        result["rollout"] = {"metric": "env-steps/s, 65,536 envs x 128 steps (obs -> mask -> policy -> sample -> step, auto-reset)",
                             "unit": "env-steps/s", **rres}

    # ---- cpu_baseline leg: the oracle (C port of the reference algorithm) on the host cores --
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        # libgomp is already loaded by torch, so OMP_NUM_THREADS is moot: size the pool explicitly to the
        # box's CPU share (16 for one GPU) or to the cores this process may run on, whichever is smaller
        O.set_num_threads(min(16, len(os.sched_getaffinity(0))))
        hb, ha = boards.cpu().numpy(), actions.cpu().numpy()
        hs = np.zeros(n, np.uint32)
        O.step_batch(hb[:4096], ha[:4096], hs[:4096], seed=SEED, step_index=0)       # load + warm
        c0 = time.perf_counter(); passes = 0
        while time.perf_counter() - c0 < args.cpu_seconds and passes < 1000:
            bo, so, ro, fo = O.step_batch(hb, ha, hs, seed=SEED, step_index=W + passes, id_base=id_base)
            passes += 1
        csec = time.perf_counter() - c0
        result["cpu_baseline"] = {"value": n * passes / csec, "unit": "board-steps/s", "cores": O.num_threads(),
                                  "kind": "port",
                                  "sample": "%d passes of g2048o_step_batch over the same 1,048,576 boards "
                                            "(%.1f s, OpenMP static over boards)" % (passes, csec)}
        from oracle import pyref
        prate = pyref.time_steps(4000, SEED)
        result["cpu_baseline_python"] = {"value": prate, "unit": "board-steps/s", "cores": 1, "kind": "port",
                                         "sample": "4000 steps of one board, reference-style per-board NumPy env "
                                                   "(oracle/pyref.py), auto-reset",
                                         "calibration": "the reference's own Game2048Env ran at 0.88x this env's rate on "
                                                        "the same core in the build container (2.55e3 vs 2.92e3 steps/s)"}
        # the last CPU pass doubles as a full-size parity check of what the GPU just computed
        one_step(W + passes - 1)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), bo) and np.array_equal(flags.cpu().numpy(), fo), "GPU != oracle"
        if not args.no_beam:
            hr = roots.cpu().numpy()
            c0 = time.perf_counter(); cexp = 0; cdec = 0
            while time.perf_counter() - c0 < args.cpu_seconds and cdec < 100:
                oa, op, oe = O.beam_batch(hr, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=10 + cdec, game_id_base=0)
                cexp += int(oe.sum()); cdec += 1
            csec = time.perf_counter() - c0
            result["beam"]["cpu_baseline"] = {"value": cexp / csec, "unit": "expansions/s",
                                              "cores": O.num_threads(), "kind": "port",
                                              "sample": "%d batch decisions over the same 4096 roots (%.1f s, OpenMP "
                                                        "dynamic over games)" % (cdec, csec)}
            a, p, e = ops.beam_get_action(roots, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=10 + cdec - 1,
                                          game_id_base=0, want_expanded=True)
            assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(e.cpu().numpy().astype(np.uint32), oe), \
                "GPU beam != oracle"

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
