#!/usr/bin/env python3
"""bench.py -- the hot path's benchmark (contract: python bench.py --gpus N --steps K --warmup W).

A "step" is one pass of Game2048Env.step (g2048_step, HIP) over this rank's batch of synthetic boards:
BASELINE.json configs[1] -- 1,048,576 boards per GPU (cells empty with p = 0.30, else codes uniform
1..11; uniform actions, all four in every launch), inputs resident in HBM, reading a fixed input buffer
and writing a separate output buffer so every step does identical work. With N GPUs every rank steps its
own contiguous shard of 1,048,576 boards (global ids rank*1,048,576 ..., weak scaling, config 5); the only
collective is the final all-gather of per-board scores over RCCL, issued after the timed region.

`python bench.py --gpus N` with N > 1 and no torchrun environment starts its own N ranks: the parent builds a
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>` command, runs it as a CHILD process
and relays its exit code; the parent itself never touches the GPU (no torch import, no HIP call).

Timing: W warm-up steps, then EXACTLY K steps between barrier + synchronize pairs, max over ranks.
The K launches of the one-chain form are replayed from one hipGraph (launch-bound otherwise: a step is ~10-15 us of GPU time);
pass --no-graph for eager launches. roofline.frac has ONE definition, kept across rounds: algorithmic bytes / (HIP event
pair on the launch stream around the timed region's own K launches / K). Two more readings of the same launches are
reported beside it and never replace it: frac_wall (bytes / ms_per_step, the driver-visible clock) and
frac_plain_launches (an event pair around K plain launches of the same call queued behind K untimed ones: no graph-replay
fixed cost, no host launch latency). The cpu_baseline leg times the CPU oracle (the C port of
the reference algorithm) on a bounded sample of the same workload, rank 0 at N = 1 only.

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
BIG_BOARDS = 1 << 24         # beyond the 256 MiB Infinity Cache: 16 Mi boards = 738 MB of streams per launch
SEED = 0x2048
STEP_BYTES_F32 = 46          # R board 16 + action 1 + score 4; W board 16 + score 4 + reward 4 + flags 1
STEP_BYTES_F64 = 50          # the same with the f64 reward (parity mode)
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
BEAM_GAMES, BEAM_WIDTH, BEAM_DEPTH = 4096, 20, 30
BEAM_BYTES_PER_EXPANSION = 29    # "HBM-resident-beam" accounting (parent 16 B / 4 children + child 16 + score 8 + root action 1)
SIMDS_PER_CU = 4                 # MI355X_MICROARCH.md; the CU count and the clock come from the device properties


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph replay")
    ap.add_argument("--chains", type=int, default=2,
                    help="a step = this many independent sub-batch launches on parallel hipGraph branches (1 = one launch per step)")
    ap.add_argument("--chain-tune", type=int, default=-1, help="boards per lane inside a chain: 1 or 2 (default 2); 0 = the library's choice by size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-beam", action="store_true")
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--no-ppo-rollout", action="store_true", help="skip only the config-4 PPO rollout leg (stock-torch policies)")
    ap.add_argument("--no-evaluation", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the HBM-resident and f64-reward step legs")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
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


def cpu_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    quota = None            # the container's CPU share, if a cgroup quota says so (an affinity mask of 256 can sit on a 16-core share)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except (OSError, ValueError, IndexError):
            continue
    return {"cpu_model": model, "nproc": os.cpu_count(), "affinity": len(os.sched_getaffinity(0)), "cgroup_cpu_quota_cores": quota}


def recorded_single_gpu_line():
    """The N = 1 line recorded on an MI355X in this round's (else the latest round's) profiles/rNN_bench.json -- the figure a
    weak-scaling efficiency divides by: per-GPU value at N ranks / this value. Not a measurement of this run."""
    import glob
    for path in sorted(glob.glob(os.path.join(REPO, "profiles", "r[0-9][0-9]_bench.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        if d.get("n_gpus") == 1 and d.get("value"):
            return {"value": d["value"], "ms_per_step": d.get("ms_per_step"), "steps": d.get("steps"), "chains": (d.get("config") or {}).get("chains", 1),
                    "source": os.path.relpath(path, REPO) + " (recorded N = 1 run of this bench on one MI355X; compare like with like: same --steps)"}
    return None


_IN_GRAPH_EVENTS = [True]        # cleared the first time a capture with timing events in it is refused


def graph_of(fn, dev, with_events=True):
    """Capture fn() into a hipGraph on a side stream (thread_local capture mode: with a process group alive the RCCL
    watchdog thread issues event queries that would invalidate a global-mode capture). Returns (graph, ev0, ev1) where the
    two events are recorded as the first and the last node of the graph (None if this torch / HIP cannot record timing
    events inside a capture); (None, None, None) if capture is unavailable."""
    ev0 = ev1 = None
    if getattr(torch.version, "hip", None):
        _IN_GRAPH_EVENTS[0] = False     # known: "External events are disallowed in rocm" -- do not even start such a capture
    with_events = with_events and _IN_GRAPH_EVENTS[0]
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        graph = torch.cuda.CUDAGraph()
        if with_events:
            ev0 = torch.cuda.Event(enable_timing=True, external=True)
            ev1 = torch.cuda.Event(enable_timing=True, external=True)
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                if ev0 is not None:
                    ev0.record(side)
                fn()
                if ev1 is not None:
                    ev1.record(side)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        return graph, ev0, ev1
    except Exception as exc:            # noqa: BLE001
        print("bench.py: hipGraph capture failed%s (%s: %s)" % (" with in-graph events" if with_events else "",
                                                                type(exc).__name__, exc), file=sys.stderr)
        torch.cuda.synchronize()
        if with_events:                 # retry without in-graph events before giving the graph up
            _IN_GRAPH_EVENTS[0] = False
            return graph_of(fn, dev, with_events=False)
        return None, None, None


def timed_replay(graph, ev0, ev1, eager, reps=1):
    """GPU time of one replay (ms), best of `reps`: in-graph events if there are any; else an event pair around a replay
    that is queued directly behind another one (so that the host's graph-launch latency is not inside the pair)."""
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(reps):
        if graph is not None:
            graph.replay()
        else:
            eager()
        a.record()
        if graph is not None:
            graph.replay()
        else:
            eager()
        b.record()
        torch.cuda.synchronize()
        ms = None
        if graph is not None and ev0 is not None:
            try:
                ms = ev0.elapsed_time(ev1)
            except Exception:           # noqa: BLE001
                ms = None
        if ms is None:
            ms = a.elapsed_time(b)
        best = ms if best is None else min(best, ms)
    return best


def beam_roots(ops, n, id_base, dev):
    """half the roots reached by 200 moves of seeded self-play under a greedy fast-evaluation policy (a
    width-1, depth-1 beam decision is exactly that: every valid move, its spawn, _fast_evaluate, best one), auto-reset on;
    half independent cells as in C2 (empty with p = 0.30, else codes uniform 1..11)."""
    half = n // 2
    boards, scores = ops.reset(half, SEED + 1, 0, id_base, device=dev)
    for t in range(200):
        a, _ = ops.beam_get_action(boards, 1, 1, seed=SEED + 1, step_index=t, game_id_base=id_base)
        ops.step(boards, a, scores, SEED + 1, t, id_base, out=boards, auto_reset=True)
    return torch.cat([boards, ops.synth_boards(n - half, seed=SEED + 2, id_base=id_base + half, device=dev)])


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
    if os.environ.get("G2048_BENCH_FAIL_RANK") == str(rank):       # tests/test_bench_launch.py: a rank that dies while the others wait
        raise RuntimeError("injected failure on rank %d (G2048_BENCH_FAIL_RANK)" % rank)
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

    # A step = `--chains` independent sub-batch launches (contiguous slices of the shard, ops.StepChains -- the launch form of
    # VecGame2048(chains=C)): chain c's step t+1 is ordered behind chain c's step t only, so one chain's launch head and drain
    # overlap the other's arithmetic. Same kernel, same bytes, same results as one launch per step (checked below).
    sc = ops.StepChains(n, max(1, args.chains), dev)
    # (inside a chain a lane takes two boards, G2048_STEP_TUNE 2: a 512 Ki-board launch is then 4 wavefronts per SIMD, so both chains'
    # launches are resident together instead of taking turns at the 8 slots -- 9.3 against 10.6 us per step in steady state, 1-2 % at
    # K = 20; same kernel arithmetic, same results: profiles/r05_chains_launch_forms.txt, section 10)
    chain_tune = (args.chain_tune if args.chain_tune >= 0 else 2) if max(1, args.chains) > 1 else 0
    chain_calls = [ops.PreparedStep(boards[lo:hi], actions[lo:hi], scores[lo:hi], SEED, id_base + lo, out=out[lo:hi],
                                    reward=reward[lo:hi], flags=flags[lo:hi], tune=chain_tune) for lo, hi in sc.bounds]
    if len(sc) > 1:
        sc.keep_alive(boards, actions, out, scores, reward, flags)

    def steps_in_chains(t0, count):
        sc.fork()                   # (no-op if the chains are open already: the timed region opens them before its clock starts)
        lanes = [(call, sc.stream(c).cuda_stream) for c, call in enumerate(chain_calls)]
        for t in range(t0, t0 + count):
            for call, sp_ in lanes:
                call(t, sp_)
        sc.join()

    barrier = gdist.barrier
    K, W = args.steps, args.warmup
    steps_in_chains(0, W)
    torch.cuda.synchronize()

    def k_steps():
        steps_in_chains(W, K)

    # Launch form of the timed region. One chain: the K launches are replayed from one hipGraph (launch-bound otherwise). Several
    # chains: PLAIN launches on the chains' streams, paced by the host (~4.5 us per launch against ~5 us of GPU time per sub-batch
    # launch). A hipGraph with parallel branches is not used for them: this ROCm's hipGraphLaunch takes its slow path for a
    # multi-branch graph and, from an idle stream, gets the second branch's first launch to the GPU ~55 us late (K = 20: 12.5 us
    # per step against 10.4 us for the same launches made directly; tools/chains_wave_timeline.py, profiles/r05_chains_*.txt).
    graph = ev0 = ev1 = None
    if not args.no_graph and len(sc) == 1:
        graph, ev0, ev1 = graph_of(k_steps, dev)
    # untimed: the first replay pays the graph's one-time upload, and ~20 ms of load bring the clocks to steady state
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.02:
        if graph is not None:
            graph.replay()
        else:
            k_steps()
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps -------------------------------------------------
    # wall clock between barrier + synchronize pairs (-> value); the HIP events bracket the same K launches on the launch
    # stream (-> average launch duration for the roofline)
    scores.zero_()
    o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    barrier()
    if len(sc) > 1:
        sc.fork()       # stream plumbing only (the side streams wait for the -- idle -- launch stream): no GPU work, like a graph's capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    o0.record()
    if graph is not None:
        graph.replay()
    else:
        k_steps()
    o1.record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0      # this rank's K steps are complete here ...
    barrier()
    torch.cuda.synchronize()
    outer_ms = o0.elapsed_time(o1)
    per_rank_s = gdist.gather_floats(elapsed, dev)  # every rank's own K-step time, in rank order
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
            # (into buffers of its own: this rank's out / reward / flags still have to pass the chains check below)
            vb, va = torch.empty_like(boards), torch.empty_like(actions)
            vo, vs = torch.empty_like(boards), torch.empty_like(scores)
            vr, vf = torch.empty_like(reward), torch.empty_like(flags)
            for r in range(1, world):
                ops.synth_boards(n, seed=SEED, id_base=r * n, device=dev, out=vb)
                ops.synth_actions(n, seed=SEED, step_index=0, id_base=r * n, device=dev, out=va)
                vs.zero_()
                for t in range(K):
                    ops.step(vb, va, vs, SEED, W + t, r * n, out=vo, reward=vr, flags=vf)
                assert bool((gathered[r * n:(r + 1) * n] == vs).all()), "shard %d differs from the 1-GPU result" % r
            del vb, va, vo, vs, vr, vf

    # ---- the chain form must BE the single launch: the same K steps, one launch each, into buffers of their own
    chains_equal = None
    if len(sc) > 1:
        vo, vs, vr, vf = torch.empty_like(out), torch.zeros_like(scores), torch.empty_like(reward), torch.empty_like(flags)
        for t in range(K):
            ops.step(boards, actions, vs, SEED, W + t, id_base, out=vo, reward=vr, flags=vf)
        chains_equal = bool(torch.equal(vo, out) and torch.equal(vs, scores) and torch.equal(vf, flags)
                            and torch.equal(vr.view(torch.int32), reward.view(torch.int32)))
        assert chains_equal, "the %d-chain step differs from the single launch" % len(sc)
        del vo, vs, vr, vf

    # ---- the single-launch form beside it (rounds 1-4's headline): one hipGraph of K launches of all 1,048,576 boards, the same
    # event-pair definition (a replay from an idle stream), mean of three
    def k_single():
        for t in range(K):
            one_step(W + t)
    single_s = None
    if len(sc) > 1:
        gs, _, _ = (None, None, None) if args.no_graph else graph_of(k_single, dev)
        reads = []
        for _ in range(4):
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            s0.record()
            if gs is not None:
                gs.replay()
            else:
                k_single()
            s1.record()
            torch.cuda.synchronize()
            reads.append(s0.elapsed_time(s1) * 1e-3 / K)
        single_s = sum(reads[1:]) / 3.0
        del gs

    # ---- kernel time of the timed region's launches (after the score exchange above: the extra launches below add to `scores`)
    # THE definition of roofline.frac (frozen in round 3): the HIP event pair around the timed region's own K launches / K.
    # With a hipGraph that pair also holds the replay's fixed cost (~8 us per replay: +0.4 us per launch at K = 20) and, from an
    # idle stream, the host's launch latency -- so it under-reads the kernel; the two other readings below bracket it.
    region_s = outer_ms * 1e-3 / K
    kernel_s = region_s
    timing = ("HIP event pair on the launch stream around the timed region's own K steps (one chain: one hipGraph replay unless "
              "--no-graph; several chains: plain launches on the chains' streams, the pair spans fork and join), / K -- the definition "
              "roofline.frac keeps across rounds")
    # plain launches: the same g2048_step call of the timed region (same buffers, same arguments), K times behind K untimed
    # ones through a prepared call (ops.PreparedStep, ~4 us of host time per launch, so the GPU queue never runs dry)
    prepared = ops.PreparedStep(boards, actions, scores, SEED, id_base, out=out, reward=reward, flags=flags)
    sp = torch.cuda.current_stream(dev).cuda_stream
    plain = []
    for _ in range(3):
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for t in range(K):
            prepared(W + t, sp)
        q0.record()
        for t in range(K):
            prepared(W + t, sp)
        q1.record()
        torch.cuda.synchronize()
        plain.append(q0.elapsed_time(q1) * 1e-3 / K)
    plain_s, plain_mean_s = min(plain), sum(plain) / len(plain)
    # ---- roofline: algorithmic bytes per launch / average launch duration over the timed region ----
    achieved = n * STEP_BYTES_F32 / kernel_s / 1e9
    traffic = traffic_src = None
    pmc_extra = {}
    pmc = os.path.join(REPO, "profiles", "pmc_step.json")     # HBM bytes per launch from the rocprofv3 PMC passes
    if os.path.exists(pmc):
        pj = json.load(open(pmc))
        traffic = pj.get("hbm_bytes_per_launch")
        pmc_extra = pj
        traffic_src = "recorded rocprofv3 PMC passes (%s), not an observation of this run" % pj.get("source", "profiles/pmc_step.json")

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
                   "boards_per_gpu": n,
                   "launch": ("%s; a step = %d independent sub-batch launches of %s boards on %s, chain c's step t+1 ordered only behind "
                              "chain c's step t (ops.StepChains, the launch form of VecGame2048(chains=%d)); roofline.single_launch is "
                              "the one-launch-per-step form of rounds 1-4"
                              % ("one hipGraph of K steps" if graph is not None else "plain launches paced by the host", len(sc),
                                 "/".join(str(hi - lo) for lo, hi in sc.bounds),
                                 "parallel branches of the graph" if graph is not None else "%d HIP streams" % len(sc), len(sc)))
                             if len(sc) > 1 else ("hipGraph of K launches" if graph is not None else "eager"),
                   "chains": len(sc),
                   "working_set": "48 MB per launch, re-used by every launch: resident in the 256 MiB Infinity Cache (LLC), "
                                  "not streamed from HBM -- see roofline_hbm_resident for the beyond-LLC size",
                   "parallelism": "%d shard(s) of 1,048,576 boards, no data-path collective" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": ("step_kernel<false,false,2,256> x %d sub-batch launches per step (two boards per lane inside a chain)" % len(sc)) if len(sc) > 1 else "step_kernel<false,false,1,256>",
                     "kernel_us": kernel_s * 1e6,
                     "kernel_us_is": ("one whole-batch STEP (all %d sub-batch launches, which overlap): event pair / K. A single sub-batch "
                                      "launch lasts longer than its share of this (tools/chains_wave_timeline.py, profiles/r05_chains_*), "
                                      "and a kernel tracer serialises the two queues, so a rocprofv3 per-launch average prices the lone "
                                      "sub-launch, not the step; the single-launch figure below is the one a trace reproduces" % len(sc))
                                     if len(sc) > 1 else "one launch",
                     "frac_wall": n * STEP_BYTES_F32 / (elapsed / K) / 1e9 / HBM_PEAK_GBS,
                     "algorithmic_bytes_per_launch": n * STEP_BYTES_F32, "algorithmic_bytes_per_step": n * STEP_BYTES_F32, "timing": timing,
                     "chains_equal_single_launch": chains_equal,
                     "single_launch": {"kernel_us": (single_s if single_s is not None else kernel_s) * 1e6,
                                       "frac": n * STEP_BYTES_F32 / (single_s if single_s is not None else kernel_s) / 1e9 / HBM_PEAK_GBS,
                                       "frac_plain_launches": n * STEP_BYTES_F32 / plain_s / 1e9 / HBM_PEAK_GBS,
                                       "kernel_us_plain_launches": plain_s * 1e6, "kernel_us_plain_launches_mean_of_3": plain_mean_s * 1e6,
                                       "timing": "one hipGraph of K launches of all 1,048,576 boards, event pair around a replay from an idle "
                                                 "stream / K, mean of three (rounds 3-4's roofline.frac: 0.447-0.456); plain launches: K "
                                                 "launches queued behind K untimed ones, best / mean of three"},
                     "note": "LLC-resident working set at this size; what binds is VALU issue (valu_issue below), not memory"},
    }
    # what really binds the step kernel: VALU issue. Instructions per step from the recorded SQ_INSTS_VALU pass, the issue-cycle
    # model of tools/isa_cost.py, and the counter's own busy time: SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (SIMDs x clock)
    if pmc_extra.get("valu_wave_instructions_per_launch"):
        props = torch.cuda.get_device_properties(dev)
        n_simd = props.multi_processor_count * SIMDS_PER_CU
        clock_ghz = (getattr(props, "clock_rate", 0) or 2400000) / 1e6
        insts = float(pmc_extra["valu_wave_instructions_per_launch"])
        cyc = float(pmc_extra.get("issue_cycles_per_instruction_model", 3.45))
        busy_s = float(pmc_extra.get("active_inst_valu_quad_cycles_per_launch", 0)) * 4.0 / (n_simd * clock_ghz * 1e9)
        peak = n_simd * clock_ghz / cyc
        result["roofline"]["valu_issue"] = {
            "bound": "valu_issue", "achieved": insts / kernel_s / 1e9, "peak": peak, "unit": "G wave-instr/s", "frac": insts / kernel_s / 1e9 / peak,
            "valu_wave_instructions_per_step": insts, "valu_instructions_per_board": insts / (n / 64.0),
            "issue_cycles_per_instruction_model": cyc, "simds": n_simd, "clock_ghz": clock_ghz,
            "valu_busy_us_per_step_by_counter": busy_s * 1e6, "valu_busy_frac_by_counter": busy_s / kernel_s,
            "valu_busy_frac_single_launch": busy_s / (single_s if single_s is not None else kernel_s),
            "source": "instruction and busy-cycle counts from recorded rocprofv3 passes (%s); times measured in this run; "
                      "valu_busy_frac_by_counter = SQ_ACTIVE_INST_VALU x 4 / (SIMDs x clock x kernel_us): the share of the step during "
                      "which every SIMD would be issuing VALU work if the work were spread evenly" % pmc_extra.get("sq_source", "profiles/pmc_step.json")}
        # `bound` names whichever limit is nearer: the HBM-algorithmic frac (SURVEY 8d's definition, kept) or VALU issue
        if result["roofline"]["valu_issue"]["valu_busy_frac_by_counter"] > result["roofline"]["frac"]:
            result["roofline"]["bound"] = "valu_issue"
            result["roofline"]["bound_note"] = ("frac / achieved / peak stay the HBM-algorithmic figures SURVEY 8(d) defines (46 B per board-step "
                                                "against 8 TB/s); the kernel is nearer its VALU-issue limit (valu_issue.valu_busy_frac_by_counter) "
                                                "than the memory one, so that is the bound named")
    if gather_ms is not None:
        import torch.distributed as tdist
        planes = gdist.backends()
        result["allgather_scores_ms"] = gather_ms
        result["n_ranks_seen"] = tdist.get_world_size()
        # the data plane (all-gather of the scores, all-reduce of the metrics) is RCCL unless its probe failed on this node, in
        # which case the same exchange was staged through the host over gloo and data_note says why; barriers and clock
        # readings always travel over gloo (g2048/dist.py)
        result["backend"] = planes["data"]
        result["backend_control"] = planes["control"]
        if planes.get("data_note"):
            result["backend_note"] = "RCCL data plane unavailable, staged through the host over gloo: " + planes["data_note"]
        result["gathered_equals_single_gpu"] = True        # asserted above on rank 0
        # what the scaling curve is read against: every rank's own K-step time, their spread, and the recorded N = 1 figure
        result["per_rank_ms_per_step"] = [t / K * 1e3 for t in per_rank_s]
        result["slowest_over_fastest_rank"] = max(per_rank_s) / min(per_rank_s)
        result["per_gpu_value"] = result["value"] / world
        ref1 = recorded_single_gpu_line()
        if ref1 is not None:
            result["single_gpu_reference"] = ref1
            if ref1.get("value"):
                result["single_gpu_reference"]["this_run_per_gpu_over_it"] = result["per_gpu_value"] / ref1["value"]

    # ---- the other legs. With one rank a failure in any of them fails the run, as it should. With several ranks the headline
    # figures above are what the scaling curve needs: a leg that raises (or a collective in it that times out: g2048/dist.py
    # bounds every wait) is recorded in the line as optional_legs_error and the line is still printed.
    def _optional_legs():
        # ---- extra step legs: f64-reward parity mode at configs[1]; a working set beyond the Infinity Cache -----------
        if not args.no_extra:
            reward64 = torch.empty(n, dtype=torch.float64, device=dev)

            def k_steps_f64():
                for t in range(K):
                    ops.step(boards, actions, scores, SEED, W + t, id_base, out=out, reward=reward64, flags=flags, reward_f64=True)
            k_steps_f64()
            g64, a64, b64 = (None, None, None) if args.no_graph else graph_of(k_steps_f64, dev)
            ms = timed_replay(g64, a64, b64, k_steps_f64, reps=3)
            us = ms * 1e3 / K
            result["roofline_f64_reward"] = {"bound": "hbm", "achieved": n * STEP_BYTES_F64 / us / 1e3, "peak": HBM_PEAK_GBS,
                                             "unit": "GB/s", "frac": n * STEP_BYTES_F64 / us / 1e3 / HBM_PEAK_GBS, "kernel_us": us,
                                             "kernel": "step_kernel<true,false,1,256>", "bytes_per_board": STEP_BYTES_F64,
                                             "board_steps_per_s": n / us * 1e6,
                                             "traffic": (pmc_extra.get("f64_reward_mode") or {}).get("hbm_bytes_per_launch"),
                                             "traffic_source": "recorded rocprofv3 PMC passes (%s), not an observation of this run"
                                                               % pmc_extra.get("extra_legs_source", "profiles/pmc_step.json"),
                                             "note": "bit-exact parity mode: reward written as f64 (50 B per board-step)"}
            del reward64, g64
            nb = BIG_BOARDS
            bb = ops.synth_boards(nb, seed=SEED + 7, id_base=id_base, device=dev)
            ba = ops.synth_actions(nb, seed=SEED + 7, step_index=0, id_base=id_base, device=dev)
            bo = torch.empty_like(bb)
            bs = torch.zeros(nb, dtype=torch.int32, device=dev)
            br = torch.empty(nb, dtype=torch.float32, device=dev)
            bf = torch.empty(nb, dtype=torch.uint8, device=dev)
            KB = 40          # 40 launches per replay (6.5 ms): long enough for steady-state clocks, the replay's fixed cost amortised

            def big_steps():
                for t in range(KB):
                    ops.step(bb, ba, bs, SEED, t, id_base, out=bo, reward=br, flags=bf)
            big_steps()
            gb, ab, bb2 = (None, None, None) if args.no_graph else graph_of(big_steps, dev)
            for _ in range(3):                  # untimed: fresh 738 MB of buffers (first touches, TLB), clocks to steady state
                if gb is not None:
                    gb.replay()
                else:
                    big_steps()
            torch.cuda.synchronize()
            all_ms = [timed_replay(gb, ab, bb2, big_steps, reps=1) for _ in range(4)]
            us = min(all_ms) * 1e3 / KB
            us_mean = sum(all_ms) / len(all_ms) * 1e3 / KB
            result["roofline_hbm_resident"] = {"bound": "hbm", "achieved": nb * STEP_BYTES_F32 / us / 1e3, "peak": HBM_PEAK_GBS,
                                               "unit": "GB/s", "frac": nb * STEP_BYTES_F32 / us / 1e3 / HBM_PEAK_GBS,
                                               "frac_mean_of_4": nb * STEP_BYTES_F32 / us_mean / 1e3 / HBM_PEAK_GBS,
                                               "kernel_us": us, "kernel_us_mean_of_4": us_mean, "boards_per_launch": nb,
                                               "traffic": (pmc_extra.get("hbm_resident_leg") or {}).get("hbm_bytes_per_launch"),
                                               "traffic_source": "recorded rocprofv3 PMC passes (%s), not an observation of this run"
                                                                 % pmc_extra.get("extra_legs_source", "profiles/pmc_step.json"),
                                               "kernel": "step_kernel<false,false,2,256> (two boards per lane from 4 Mi boards per launch on)",
                                               "algorithmic_bytes_per_launch": nb * STEP_BYTES_F32,
                                               "board_steps_per_s": nb / us * 1e6,
                                               "note": "16,777,216 boards per launch: 738 MB of streams, beyond the 256 MiB "
                                                       "Infinity Cache, so reads come from and writes go to HBM3E"}
            # rounds 1-2 timed 10 launches per replay (1.6 ms): kept beside the 40-launch reading for like-for-like comparison
            def big_steps10():
                for t in range(10):
                    ops.step(bb, ba, bs, SEED, t, id_base, out=bo, reward=br, flags=bf)
            g10, a10, b10 = (None, None, None) if args.no_graph else graph_of(big_steps10, dev)
            ms10 = [timed_replay(g10, a10, b10, big_steps10, reps=1) for _ in range(4)]
            result["roofline_hbm_resident"]["frac_10_launch_replays"] = nb * STEP_BYTES_F32 / (min(ms10) * 1e3 / 10) / 1e3 / HBM_PEAK_GBS
            result["roofline_hbm_resident"]["frac_10_launch_replays_mean_of_4"] = (nb * STEP_BYTES_F32 / (sum(ms10) / len(ms10) * 1e3 / 10) / 1e3
                                                                                  / HBM_PEAK_GBS)
            result["roofline_hbm_resident"]["timing"] = ("event pair around one hipGraph replay of 40 launches queued behind another one "
                                                         "(6 ms: steady-state clocks), best and mean of four; rounds 1-2 replayed 10 launches "
                                                         "(frac_10_launch_replays: 0.586 in round 2)")
            del bb, ba, bo, bs, br, bf, gb, g10
            torch.cuda.empty_cache()

        # ---- beam search leg (config 3): 4096 concurrent games, width 20, depth 30 -----------
        if not args.no_beam:
            roots = beam_roots(ops, BEAM_GAMES, rank * BEAM_GAMES, dev)
            for w in range(2):
                a, p, e = ops.beam_get_action(roots, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=w,
                                              game_id_base=rank * BEAM_GAMES, want_expanded=True)
            torch.cuda.synchronize()
            breps = 20
            batches = []
            for rep in range(3):                # three batches of 20 calls; `value` is the best batch, the mean is reported beside it
                b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                exps = []
                b0.record()
                for w in range(breps):          # back to back on one stream: the GPU never waits for the host
                    a, p, e = ops.beam_get_action(roots, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=10 + w,
                                                  game_id_base=rank * BEAM_GAMES, want_expanded=True)
                    exps.append(e)
                b1.record()
                torch.cuda.synchronize()
                batches.append((b0.elapsed_time(b1) * 1e-3, int(torch.stack(exps).sum().item())))
            # a fourth, separate batch with an event between every two calls: the per-call durations (kernel + one launch boundary), whose
            # median is what a rocprofv3 kernel trace of this command shows for beam_kernel<2> (profiles/r04_bench_kernel_trace_stats.txt)
            marks = [torch.cuda.Event(enable_timing=True) for _ in range(breps + 1)]
            marks[0].record()
            for w in range(breps):
                ops.beam_get_action(roots, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=10 + w, game_id_base=rank * BEAM_GAMES, want_expanded=True)
                marks[w + 1].record()
            torch.cuda.synchronize()
            per_call_us = sorted(marks[w].elapsed_time(marks[w + 1]) * 1e3 for w in range(breps))
            bsec, total_exp = min(batches)
            beam_mean = sum(x / t for t, x in batches) / len(batches)
            beam_best = total_exp / bsec
            bsec_mean = sum(t for t, _ in batches) / len(batches)
            # N > 1: every rank searched its own 4096 roots (games shard like boards); the job's figure is all ranks' expansions over
            # the slowest rank's mean batch time
            job_exp = world * total_exp if world == 1 else int(gdist.reduce_metrics(torch.tensor([total_exp], dtype=torch.int64, device=dev))[0].item())
            job_sec = gdist.max_over_ranks(bsec_mean, dev)
            result["beam"] = {"metric": "beam node-expansions/s (width=20, depth=30, 4096 concurrent games per GPU)",
                              "value": beam_mean if world == 1 else job_exp / job_sec, "unit": "expansions/s",
                              "value_best_of_3_batches": beam_best, "n_gpus": world,
                              "timing": "HIP event pair around 20 calls queued back to back (one beam kernel each; the blocks take the games in the "
                                        "depth-balanced order the previous call left behind, g2048_beam_get_action_hist); `value` = mean of three "
                                        "such batches (round 3 reported the best of three: 6.94e10 best / 6.80e10 mean then; rounds 1-2 one batch)",
                              "decisions_per_s": BEAM_GAMES * breps / bsec_mean, "ms_per_batch_decision": bsec_mean / breps * 1e3,
                              "parents_expanded_per_s_approx": beam_mean / 4.0,     # "parents-expanded/s (= expansions / ~4)"
                              "ms_per_batch_decision_best": bsec / breps * 1e3,
                              "us_per_call_median_of_20_event_pairs": per_call_us[len(per_call_us) // 2], "us_per_call_min": per_call_us[0],
                              "gbs_equivalent_29B": beam_mean * BEAM_BYTES_PER_EXPANSION / 1e9,
                              "gbs_equivalent_29B_frac_of_hbm_peak": beam_mean * BEAM_BYTES_PER_EXPANSION / 1e9 / HBM_PEAK_GBS,
                              "gbs_equivalent_note": "SURVEY 8(d): 29 B per expansion if the beam lived in HBM; the search keeps it in LDS "
                                                     "(16 B in, 5 B out per decision), so this is a comparability figure, not traffic",
                              "expansions_per_decision": total_exp / (BEAM_GAMES * breps),
                              "kernel": "beam_kernel<2> (one wavefront per game; spawn + score in up to two 64-child passes per level, ranking by "
                                        "a bitonic network over the lanes; issue priority by remaining levels while the whole launch is resident; "
                                        "games dealt to the SIMDs by depth class from the lists the previous call's blocks filed them in "
                                        "-- no order kernel)"}
            # the beam lives in LDS (HBM traffic per decision: 16 B in, 5 B out), so its bound is VALU issue, not memory:
            # wave-instructions per launch (SQ_INSTS_VALU, recorded rocprofv3 pass) / measured launch time, against what the
            # chip's 1024 SIMDs can issue at the kernel's average cost per instruction (tools/isa_cost.py)
            pb = os.path.join(REPO, "profiles", "pmc_beam.json")
            if os.path.exists(pb):
                pj = json.load(open(pb))
                insts = float(pj["valu_wave_instructions_per_launch"])
                cyc = float(pj["issue_cycles_per_instruction"])
                props = torch.cuda.get_device_properties(dev)
                n_simd = props.multi_processor_count * SIMDS_PER_CU      # 256 CUs x 4 on MI355X; taken from the device
                clock_ghz = (getattr(props, "clock_rate", 0) or 2400000) / 1e6
                peak = n_simd * clock_ghz / cyc             # G wave-instructions / s
                ach = insts / bsec_mean * breps / 1e9            # (priced on the mean batch, like `value`)
                result["beam"]["roofline"] = {"bound": "valu_issue", "achieved": ach, "peak": peak, "unit": "G wave-instr/s",
                                              "frac": ach / peak, "valu_wave_instructions_per_launch": insts,
                                              "issue_cycles_per_instruction": cyc, "simds": n_simd, "clock_ghz": clock_ghz,
                                              "source": "instruction count from a recorded rocprofv3 SQ_INSTS_VALU pass (%s); "
                                                        "time measured in this run" % pj.get("source", "profiles/pmc_beam.json")}

            # BASELINE's metric has two halves; the driver's record keeps `roofline` and `cpu_baseline` whole, so a compact copy of the
            # beam half lives there too (nested, and as flat scalars in case nested objects are dropped)
            br = result["beam"].get("roofline", {})
            compact = {"value": result["beam"]["value"], "unit": "expansions/s", "value_best_of_3_batches": beam_best,
                       "ms_per_batch_decision": bsec_mean / breps * 1e3, "kernel": "beam_kernel<2>",
                       "valu_issue_frac": br.get("frac"), "gbs_equivalent_29B_frac": result["beam"]["gbs_equivalent_29B_frac_of_hbm_peak"],
                       "expansions_per_decision": total_exp / (BEAM_GAMES * breps), "games": BEAM_GAMES, "width": BEAM_WIDTH, "depth": BEAM_DEPTH}
            result["roofline"]["beam"] = compact
            for k in ("value", "ms_per_batch_decision", "valu_issue_frac", "gbs_equivalent_29B_frac", "expansions_per_decision"):
                result["roofline"]["beam_" + k] = compact[k]

        # ---- evaluation leg: 4096 beam-search games (w=20, d=30) played to completion, fused per game
        if not args.no_evaluation and not args.no_beam and world == 1:
            from g2048 import evaluate_beam_search
            evaluate_beam_search(256, BEAM_WIDTH, BEAM_DEPTH, seed=1, max_moves=50, device=dev)      # warm
            def play(games, **kw):
                return evaluate_beam_search(games, BEAM_WIDTH, BEAM_DEPTH, seed=2025, max_moves=5000, device=dev, **kw)
            runs = [(play(BEAM_GAMES, one_phase=True), play(BEAM_GAMES), play(100)) for _ in range(2)]     # interleaved, best of two
            ev1, ev, ev100 = (min((r[k] for r in runs), key=lambda x: x["elapsed_s"]) for k in range(3))
            sm = ev["summary"]
            result["evaluation"] = {"metric": "4096 complete beam-search games (width 20, depth 30, 5000-move cap), one launch",
                                    "seconds": ev["elapsed_s"], "seconds_without_helper_wavefronts": ev1["elapsed_s"],
                                    "same_games_without_helpers": ev["scores"] == ev1["scores"] and ev["moves"] == ev1["moves"],
                                    "seconds_100_games": ev100["elapsed_s"],
                                    "moves": ev["total_moves"], "moves_per_s": sm["moves_per_s"],
                                    "expansions_per_s": sm["expansions_per_s"], "rate_2048_or_more": sm["rate_2048_or_more"],
                                    "average_score": sm["average_score"], "reference_report_md": {"rate_2048_or_more": 0.35,
                                                                                                    "average_score": 18945.6}}
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            evh = play(BEAM_GAMES, histories="best5")           # the same evaluation with the action stream recorded + five games replayed
            torch.cuda.synchronize()
            # evaluate_beam_search stops its own clock before the replay of the asked games: `elapsed_s` holds the action stream only,
            # the whole call (replay launch, unpack, the host-side history lists) is timed here
            result["evaluation"]["seconds_with_action_stream"] = evh["elapsed_s"]
            result["evaluation"]["seconds_with_action_stream_and_best5_histories"] = time.perf_counter() - h0
            result["evaluation"]["same_games_with_action_stream"] = evh["scores"] == ev["scores"] and evh["moves"] == ev["moves"]
            ec = {"seconds": ev["elapsed_s"], "moves": ev["total_moves"], "same_games_without_helpers": result["evaluation"]["same_games_without_helpers"],
                  "seconds_without_helper_wavefronts": ev1["elapsed_s"], "expansions_per_s": sm["expansions_per_s"], "games": BEAM_GAMES}
            result["roofline"]["evaluation"] = ec
            for k in ("seconds", "moves", "same_games_without_helpers"):
                result["roofline"]["evaluation_" + k] = ec[k]

        # ---- sharded evaluation (N > 1): every rank plays 512 games of one evaluation, one all-gather of the per-game table.
        # A failure here must not cost the headline line: it is reported, not raised.
        if not args.no_evaluation and not args.no_beam and world > 1:
            try:
                from g2048 import evaluate_beam_search_sharded
                evs = evaluate_beam_search_sharded(512 * world, BEAM_WIDTH, BEAM_DEPTH, seed=2025, max_moves=5000, device=dev)
                result["evaluation_sharded"] = {"metric": "one beam-search evaluation (width 20, depth 30, 5000-move cap) of 512 games "
                                                          "per rank, games sharded by contiguous id range, one all-gather at the end",
                                                "games": len(evs["scores"]), "n_ranks": evs["parameters"]["world_size"],
                                                "seconds_slowest_rank": evs["elapsed_s"], "moves": evs["total_moves"],
                                                "rate_2048_or_more": evs["summary"]["rate_2048_or_more"],
                                                "average_score": evs["summary"]["average_score"]}
            except Exception as exc:                # noqa: BLE001
                result["evaluation_sharded"] = {"error": "%s: %s" % (type(exc).__name__, exc)}

        # ---- C2 "rollout" variant: 128 consecutive in-place steps from reset states, on-device random actions,
        # auto-reset on (realistic tile distribution instead of the synthetic one). g2048_step_many runs all 128 steps of a board
        # in ONE launch with the board in registers (per-step f32 rewards streamed out, final boards / scores / flags); the
        # 128-launch form (one hipGraph of g2048_step launches, each re-reading and re-writing 46 B per board) is timed beside it.
        if not args.no_rollout:
            T_ROLL = 128
            rb, rs = ops.reset(n, SEED, 0, id_base, device=dev)
            rstream = torch.empty((T_ROLL, n), dtype=torch.float32, device=dev)

            def many(t0, stream):
                ops.step_many(rb, rs, SEED, t0, T_ROLL, id_base, out=rb, flags=flags, auto_reset=True, reward_stream=stream)

            def time_many(stream):
                best = None
                for rep in range(5):
                    m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    m0.record()
                    many(T_ROLL * (3 + rep), stream)
                    m1.record()
                    torch.cuda.synchronize()
                    ms = m0.elapsed_time(m1)
                    best = ms if best is None else min(best, ms)
                return best
            for w in range(3):                      # untimed: first-touch of the 512 MB reward stream, clocks to steady state
                many(w * T_ROLL, rstream)
            torch.cuda.synchronize()
            ms_stream, ms_bare = time_many(rstream), time_many(None)

            def rollout_steps(t0=128):
                for t in range(t0, t0 + 128):
                    ops.step(rb, None, rs, SEED, t, id_base, out=rb, reward=reward, flags=flags, auto_reset=True)
            rollout_steps(0)
            torch.cuda.synchronize()
            rg, ra, rbv = (None, None, None) if args.no_graph else graph_of(rollout_steps, dev)
            ms = timed_replay(rg, ra, rbv, rollout_steps, reps=2)
            result["rollout_random"] = {"metric": "board-steps/s, 1,048,576 boards x 128 consecutive in-place steps from reset, "
                                                  "uniform actions drawn in the kernel, auto-reset (realistic tile distribution); "
                                                  "ONE g2048_step_many launch, boards in registers, per-step f32 rewards streamed out",
                                        "value": n * T_ROLL / (ms_stream * 1e-3), "unit": "board-steps/s",
                                        "us_per_step": ms_stream * 1e3 / T_ROLL,
                                        "kernel": "step_many_kernel<false,true,256>",
                                        "without_reward_stream": {"value": n * T_ROLL / (ms_bare * 1e-3), "us_per_step": ms_bare * 1e3 / T_ROLL,
                                                                  "note": "no per-step output requested: the shaped reward is not computed"},
                                        "as_128_step_launches": {"value": n * 128 / (ms * 1e-3), "us_per_step": ms * 1e3 / 128,
                                                                 "note": "round 2's form: one hipGraph of 128 g2048_step launches"}}
            del rstream

        # ---- PPO rollout leg (config 4): 65,536 envs x 128 steps, transformer policy on PyTorch-ROCm -----
        if not args.no_rollout and not args.no_ppo_rollout and world == 1:
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

            class ActorCritic(nn.Module):       # the MLP shapes agents/ppo_agent.py:61-136 actually uses (16-256-128-64-4 / -1,
                def __init__(self):             # BatchNorm + ReLU, eval mode), stock torch, random init
                    super().__init__()
                    def trunk():
                        return nn.Sequential(nn.Linear(16, 256), nn.BatchNorm1d(256), nn.ReLU(), nn.Linear(256, 128), nn.BatchNorm1d(128),
                                             nn.ReLU(), nn.Linear(128, 64), nn.BatchNorm1d(64), nn.ReLU())
                    self.actor, self.critic = nn.Sequential(trunk(), nn.Linear(64, 4)), nn.Sequential(trunk(), nn.Linear(64, 1))

                def forward(self, x):
                    return torch.softmax(self.actor(x), -1), self.critic(x)

            class Uniform(nn.Module):           # no network: isolates the env side of the rollout
                def forward(self, x):
                    return torch.full((x.shape[0], 4), 0.25, device=x.device)

            torch.manual_seed(0)
            rres = {}
            for name, pol in (("transformer_policy", Policy().to(dev).eval()), ("mlp_actor_critic_policy", ActorCritic().to(dev).eval()),
                              ("uniform_policy_env_only", Uniform())):
                rc = RolloutCollector(65536, 128, pol, device=dev, seed=SEED)
                rc.collect()                    # includes the one-time graph capture
                rc.collect()
                torch.cuda.synchronize()
                best = None
                for _ in range(3 if name != "transformer_policy" else 1):
                    r0 = time.perf_counter()
                    rc.collect()
                    torch.cuda.synchronize()
                    dt = time.perf_counter() - r0
                    best = dt if best is None else min(best, dt)
                rres[name] = 65536 * 128 / best
                rres[name + "_graph"] = rc._graph is not None
            # the env half of config 4 on its own: g2048_rollout_step at 65,536 envs with a fixed probability tensor, 128 launches from one
            # hipGraph (tools/rollout_rate.py) -> the kernel's launch time for its roofline block
            RN, RT, ROLLOUT_BYTES = 65536, 128, 132          # 132 B per env-step: R board 16 + probs 16 + mask 1 + score 4; W board 16 +
            rbd, rsc = ops.reset(RN, SEED, 0, 0, device=dev)  # score 4 + action 1 + prob 4 + reward 4 + flags 1 + next obs 64 + next mask 1
            rsp, rpr = torch.empty_like(rbd), torch.full((RN, 4), 0.25, device=dev)
            robs = torch.empty((2, RN, 16), dtype=torch.float32, device=dev)
            rmk = torch.empty((2, RN), dtype=torch.uint8, device=dev)
            ops.valid_moves(rbd, out=rmk[0])
            rac, rpb = torch.empty(RN, dtype=torch.uint8, device=dev), torch.empty(RN, device=dev)
            rrw, rfl = torch.empty(RN, device=dev), torch.empty(RN, dtype=torch.uint8, device=dev)
            rcnt = torch.zeros(1, dtype=torch.int64, device=dev)

            def rollout_launches():
                b, sp_ = rbd, rsp
                for t in range(RT):
                    ops.rollout_step(b, rpr, rsc, SEED, t, 0, mask=rmk[t & 1], out=sp_, actions=rac, prob=rpb, reward=rrw, flags=rfl,
                                     obs_next=robs[(t + 1) & 1], mask_next=rmk[(t + 1) & 1], step_counter=rcnt)
                    b, sp_ = sp_, b
            rollout_launches()
            torch.cuda.synchronize()
            rg_, ra_, rb_ = (None, None, None) if args.no_graph else graph_of(rollout_launches, dev)
            r_us = timed_replay(rg_, ra_, rb_, rollout_launches, reps=4) * 1e3 / RT
            pr_path = os.path.join(REPO, "profiles", "pmc_rollout.json")
            prj = json.load(open(pr_path)) if os.path.exists(pr_path) else {}
            props = torch.cuda.get_device_properties(dev)
            n_simd_r = props.multi_processor_count * SIMDS_PER_CU
            clock_r = (getattr(props, "clock_rate", 0) or 2400000) / 1e6
            rollout_roofline = {"bound": "latency (one wavefront per SIMD)", "achieved": RN * ROLLOUT_BYTES / r_us / 1e3, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": RN * ROLLOUT_BYTES / r_us / 1e3 / HBM_PEAK_GBS, "kernel": "rollout_step_kernel",
                                "kernel_us": r_us, "envs_per_launch": RN, "algorithmic_bytes_per_launch": RN * ROLLOUT_BYTES,
                                "traffic": prj.get("hbm_bytes_per_launch"), "env_steps_per_s_kernel_only": RN / r_us * 1e6,
                                "valu_busy_frac_by_counter": (float(prj["active_inst_valu_quad_cycles_per_launch"]) * 4.0
                                                              / (n_simd_r * clock_r * 1e9) / (r_us * 1e-6)) if prj.get("active_inst_valu_quad_cycles_per_launch") else None,
                                "wait_frac_of_wave_cycles_by_counter": prj.get("wait_any_frac"),
                                "timing": "event pair around one hipGraph replay of 128 launches queued behind another, best of four, / 128",
                                "traffic_source": "recorded rocprofv3 PMC passes (%s), not an observation of this run" % prj.get("source", "profiles/pmc_rollout.json"),
                                "note": "65,536 envs = 1,024 wavefronts = ONE per SIMD: nothing hides a wavefront's load -> 868 VALU -> store "
                                        "chain, so the launch is latency-bound (61 % of wave-cycles parked on s_waitcnt, 22 % VALU-active), far "
                                        "from both the HBM and the VALU-issue limit; at 1,048,576 envs the same kernel reaches 0.58 of HBM "
                                        "(tools/rollout_rate.py). The policy network dominates a real rollout either way."}
            del rg_, rbd, rsp, robs
            result["rollout"] = {"metric": "env-steps/s, 65,536 envs x 128 steps: policy -> g2048_rollout_step (sample + step + next "
                                           "obs + next mask in one launch, auto-reset), the T-step loop replayed from one hipGraph",
                                 "unit": "env-steps/s", **rres, "roofline": rollout_roofline,
                                 "note": "the policies are stock PyTorch-ROCm modules (the consumers of the rollout, not part of the hot "
                                         "path): the reference's unused 16-token transformer (its time is torch's layer-norm / attention "
                                         "kernels), the MLP actor / critic its PPO agent really uses, and no network at all"}

        # ---- cpu_baseline leg: the oracle (C port of the reference algorithm) on the host cores --
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            info = cpu_info()
            hb, ha = boards.cpu().numpy(), actions.cpu().numpy()
            hs = np.zeros(n, np.uint32)
            O.step_batch(hb[:4096], ha[:4096], hs[:4096], seed=SEED, step_index=0)       # load + warm

            def time_steps(threads, seconds):
                O.set_num_threads(threads)
                c0 = time.perf_counter(); passes = 0; last = None
                while time.perf_counter() - c0 < seconds and passes < 1000:
                    last = O.step_batch(hb, ha, hs, seed=SEED, step_index=W + passes, id_base=id_base)
                    passes += 1
                return passes, time.perf_counter() - c0, last
            # libgomp is already loaded by torch, so OMP_NUM_THREADS is moot: size the pool explicitly to the box's CPU
            # share (16 for one GPU) or to the cores this process may run on, whichever is smaller
            many = min(16, info["affinity"])
            every = min(info["affinity"], 512)      # all the host cores this process may run on ("1 core and all host cores")
            p1, s1, _ = time_steps(1, args.cpu_seconds * 0.35)
            pa, sa, _ = time_steps(every, args.cpu_seconds * 0.25) if every > many else (0, 1.0, None)
            passes, csec, (bo_, so_, ro_, fo_) = time_steps(many, args.cpu_seconds * 0.4)
            result["cpu_baseline"] = {"value": n * passes / csec, "unit": "board-steps/s", "cores": O.num_threads(),
                                      "kind": "port",
                                      "sample": "%d passes of g2048o_step_batch over the same 1,048,576 boards "
                                                "(%.1f s, OpenMP static over boards)" % (passes, csec),
                                      "one_thread": {"value": n * p1 / s1, "cores": 1,
                                                     "sample": "%d passes, %.1f s" % (p1, s1)},
                                      "all_affinity_cores": ({"value": n * pa / sa, "cores": every,
                                                              "sample": "%d passes, %.1f s, one OpenMP thread per core in the affinity mask "
                                                                        "(slower than `value` where the box's CPU share is smaller than the mask: "
                                                                        "see cgroup_cpu_quota_cores)" % (pa, sa)} if every > many else None),
                                      **info}
            from oracle import pyref
            prate = pyref.time_steps(4000, SEED)
            result["cpu_baseline_python"] = {"value": prate, "unit": "board-steps/s", "cores": 1, "kind": "port",
                                             "sample": "4000 steps of one board, reference-style per-board NumPy env "
                                                       "(oracle/pyref.py), auto-reset",
                                             "calibration": "the reference's own Game2048Env ran at 0.88x this env's rate on "
                                                            "the same core in the build container (2.55e3 vs 2.92e3 steps/s)"}
            # config 1 (the reference's own CPU-runnable case) through the drop-in class: a train.py-shaped iteration
            # (get_valid_moves + step, train.py:55-75) = one g2048_env_step launch + one synchronisation
            from environment.game_2048 import Game2048Env
            denv = Game2048Env(seed=SEED)
            for i in range(200):
                denv.get_valid_moves(); denv.step(i & 3)
            d0 = time.perf_counter(); dsteps = 0
            while dsteps < 4000:
                denv.get_valid_moves()
                if denv.step(dsteps & 3)[2]:
                    denv.reset()
                dsteps += 1
            drate = dsteps / (time.perf_counter() - d0)
            result["cpu_baseline_python"]["drop_in_steps_per_s"] = drate
            result["cpu_baseline_python"]["drop_in_note"] = ("environment.game_2048.Game2048Env on the GPU, one board, get_valid_moves() + step() "
                                                             "per iteration: one g2048_env_step launch + one synchronisation")
            result["cpu_baseline"]["config1_drop_in_steps_per_s"] = drate
            result["cpu_baseline"]["config1_reference_style_python_steps_per_s"] = prate
            # the last CPU pass doubles as a full-size parity check of what the GPU just computed
            one_step(W + passes - 1)
            torch.cuda.synchronize()
            assert np.array_equal(out.cpu().numpy(), bo_) and np.array_equal(flags.cpu().numpy(), fo_), "GPU != oracle"
            if not args.no_beam:
                hr = roots.cpu().numpy()

                def time_beam(threads, seconds, cap):
                    O.set_num_threads(threads)
                    c0 = time.perf_counter(); cexp = 0; cdec = 0; last = None
                    while time.perf_counter() - c0 < seconds and cdec < cap:
                        last = O.beam_batch(hr if threads > 1 else hr[:256], BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=10 + cdec,
                                            game_id_base=0)
                        cexp += int(last[2].sum()); cdec += 1
                    return cexp, cdec, time.perf_counter() - c0, last
                e1, d1, sec1, _ = time_beam(1, args.cpu_seconds * 0.3, 100)
                ea, da, seca, _ = time_beam(every, args.cpu_seconds * 0.3, 100) if every > many else (0, 0, 1.0, None)
                cexp, cdec, csec, (oa, op, oe) = time_beam(many, args.cpu_seconds * 0.5, 100)
                result["beam"]["cpu_baseline"] = {"value": cexp / csec, "unit": "expansions/s",
                                                  "cores": O.num_threads(), "kind": "port",
                                                  "sample": "%d batch decisions over the same 4096 roots (%.1f s, OpenMP "
                                                            "dynamic over games)" % (cdec, csec),
                                                  "one_thread": {"value": e1 / sec1, "cores": 1,
                                                                 "sample": "%d batch decisions over the first 256 roots, %.1f s" % (d1, sec1)},
                                                  "all_affinity_cores": ({"value": ea / seca, "cores": every,
                                                                          "sample": "%d batch decisions over the same 4096 roots, %.1f s"
                                                                                    % (da, seca)} if every > many else None),
                                                  **info}
                cb = result["beam"]["cpu_baseline"]
                result["cpu_baseline"]["beam"] = {"value": cb["value"], "unit": "expansions/s", "cores": cb["cores"], "kind": "port",
                                                  "one_thread": cb["one_thread"]["value"], "sample": cb["sample"]}
                result["cpu_baseline"]["beam_value"], result["cpu_baseline"]["beam_cores"] = cb["value"], cb["cores"]
                result["cpu_baseline"]["beam_one_thread_value"] = cb["one_thread"]["value"]
                result["cpu_baseline"]["one_thread_value"] = result["cpu_baseline"]["one_thread"]["value"]
                a, p, e = ops.beam_get_action(roots, BEAM_WIDTH, BEAM_DEPTH, seed=SEED, step_index=10 + cdec - 1,
                                              game_id_base=0, want_expanded=True)
                assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(e.cpu().numpy().astype(np.uint32), oe), \
                    "GPU beam != oracle"

    if world == 1:
        _optional_legs()
    else:
        try:
            _optional_legs()
        except Exception as exc:            # noqa: BLE001
            import traceback
            traceback.print_exc()
            text = str(exc).strip().splitlines()
            result["optional_legs_error"] = "%s: %s" % (type(exc).__name__, text[0][:300] if text else "")

    if rank == 0:
        # the numbers of BASELINE's two-part metric once more at the END of the line (a log that keeps only a tail keeps these)
        result["headline"] = {"board_steps_per_s": result["value"], "ms_per_step": result["ms_per_step"], "n_gpus": world,
                              "roofline_frac": result["roofline"]["frac"], "step_kernel_us": result["roofline"]["kernel_us"],
                              "beam_expansions_per_s": (result.get("beam") or {}).get("value"),
                              "beam_ms_per_batch_decision": (result.get("beam") or {}).get("ms_per_batch_decision"),
                              "beam_valu_issue_frac": ((result.get("beam") or {}).get("roofline") or {}).get("frac"),
                              "evaluation_seconds": (result.get("evaluation") or {}).get("seconds"),
                              "cpu_board_steps_per_s": (result.get("cpu_baseline") or {}).get("value"),
                              "cpu_beam_expansions_per_s": ((result.get("beam") or {}).get("cpu_baseline") or {}).get("value")}
        print(json.dumps(result))
    if world > 1:
        gdist.shutdown()                # rank 0 was still verifying / printing: leave together
    return 0


def _main_guarded():
    """A rank that raises must take the whole job down with a non-zero exit, at once: under torchrun the other ranks may be
    sitting in a barrier or a collective, and an interpreter shutdown that first tries to tear down the process group can wait
    for them. So: print the traceback, flush, and leave through os._exit -- torchrun sees the dead worker, stops the others
    and exits non-zero; the launcher parent above relays that code."""
    try:
        return main() or 0
    except SystemExit:
        raise
    except BaseException:           # noqa: BLE001
        import traceback
        traceback.print_exc()
        sys.stderr.write("bench.py: rank %s failed\n" % os.environ.get("RANK", "0"))
        sys.stderr.flush()
        sys.stdout.flush()
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:
            os._exit(1)
        return 1


if __name__ == "__main__":
    sys.exit(_main_guarded())
