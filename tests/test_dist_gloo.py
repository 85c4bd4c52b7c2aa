"""N > 1 path on CPU: world_size-2 gloo. Each rank steps its own contiguous shard of the global board range
(here through the oracle, standing in for the GPU) with id_base = shard start, then the product's
g2048.dist helpers gather the scores and reduce the metrics. The union must equal a single-process run over
all boards -- the multi-GPU correctness property (results keyed by global board id)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import PKG, REPO

N_GLOBAL = 20001      # odd on purpose: ragged shards


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from g2048 import dist as gdist
    from oracle import oracle as O
    O.set_num_threads(1)
    w, r, _ = gdist.init("gloo")
    assert (w, r) == (world, rank)
    lo, hi = gdist.shard(N_GLOBAL, rank, world)
    b = O.synth_boards(hi - lo, seed=3, id_base=lo)
    a = O.synth_actions(hi - lo, seed=3, step_index=2, id_base=lo)
    bo, sc, rw, fl = O.step_batch(b, a, np.zeros(hi - lo, np.uint32), seed=3, step_index=2, id_base=lo)
    scores = torch.from_numpy(sc.astype(np.int32))
    gathered = gdist.all_gather_scores(scores)
    m = torch.zeros(24, dtype=torch.int64)
    m[0], m[1], m[2] = hi - lo, int(sc.sum()), int((fl & 1).sum())
    gdist.reduce_metrics(m)
    t = gdist.max_over_ranks(1.0 + rank, torch.device("cpu"))
    gdist.barrier()
    if rank == 0:
        q.put((gathered.numpy(), m.numpy(), t))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_union_equals_single_run(oracle):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, m, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    b = oracle.synth_boards(N_GLOBAL, seed=3)
    a = oracle.synth_actions(N_GLOBAL, seed=3, step_index=2)
    bo, sc, rw, fl = oracle.step_batch(b, a, np.zeros(N_GLOBAL, np.uint32), seed=3, step_index=2)
    assert np.array_equal(gathered.astype(np.uint32), sc)
    assert m[0] == N_GLOBAL and m[1] == int(sc.sum()) and m[2] == int((fl & 1).sum())
    assert t == 2.0


def test_shard_ranges_cover_exactly():
    sys.path.insert(0, PKG)
    from g2048 import dist as gdist
    for n in (0, 1, 7, 8, 1 << 20, 8388608, 8388609):
        for w in (1, 2, 3, 8):
            r = [gdist.shard(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1


def _rows_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from g2048 import dist as gdist
    from g2048.evaluate import TABLE_COLUMNS
    gdist.init("gloo")
    lo, hi = gdist.shard(37, rank, world)                       # ragged: 19 + 18 games
    full = np.random.default_rng(5).integers(0, 1 << 40, size=(37, TABLE_COLUMNS), dtype=np.int64)
    got = gdist.all_gather_rows(torch.from_numpy(full[lo:hi].copy()))
    gdist.barrier()
    if rank == 0:
        q.put((got.numpy(), full))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_evaluation_table_gather():
    """The evaluation's end-of-run exchange (evaluate_beam_search_sharded): ragged per-rank tables come back in global
    game order on every rank."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rows_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, full = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got, full)


def _empty_shard_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    sys.path.insert(0, PKG)
    from g2048 import dist as gdist
    from g2048.evaluate import TABLE_COLUMNS
    gdist.init("gloo", torch.device("cpu"))
    lo, hi = gdist.shard(1, rank, world)                        # ONE game for two ranks: a rank's shard is empty
    full = np.arange(TABLE_COLUMNS, dtype=np.int64)[None, :] + 7
    got = gdist.all_gather_rows(torch.from_numpy(full[lo:hi].copy()))
    gdist.barrier()
    q.put((rank, lo, hi, got.numpy()))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_gather_with_an_empty_shard():
    """Fewer games than ranks (evaluate_beam_search_sharded with num_games < world size): the rank whose shard is empty still
    takes part in the gather, and every rank ends up with the whole table."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_empty_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=120) for _ in range(2)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    sizes = sorted(hi - lo for _, lo, hi, _ in res)
    assert sizes == [0, 1]
    for _, _, _, got in res:
        assert got.shape[0] == 1 and int(got[0, 0]) == 7
    for p in procs:
        assert p.exitcode == 0


def test_results_from_table_is_the_single_process_result():
    """results_from_table over concatenated shard tables == over the whole table (the merge is a concatenation)."""
    sys.path.insert(0, PKG)
    from g2048.evaluate import results_from_table, TABLE_COLUMNS, MILESTONES
    rng = np.random.default_rng(11)
    n = 23
    t = np.zeros((n, TABLE_COLUMNS), dtype=np.int64)
    t[:, 0] = rng.integers(0, 60000, n)
    t[:, 1] = rng.integers(100, 5000, n); t[:, 2] = t[:, 1] - 7; t[:, 3] = 7
    t[:, 4] = rng.integers(0, 2, n); t[:, 5] = rng.integers(0, 1 << 33, n)
    t[:, 6:14] = rng.integers(-1, 3000, (n, 8))
    t[:, 14:30] = 2 ** rng.integers(1, 12, (n, 16))
    r = results_from_table(t, 1.0, 20, 30, 7, 5000)
    assert r["scores"] == [int(x) for x in t[:, 0]] and r["total_moves"] == int(t[:, 1].sum())
    assert r["unfinished"] == int(t[:, 4].sum()) and r["total_expansions"] == int(t[:, 5].sum())
    assert r["highest_tiles"] == [int(x) for x in t[:, 14:30].max(axis=1)]
    assert r["best_games"] == sorted(range(n), key=lambda i: t[i, 0], reverse=True)[:5]
    for k, m in enumerate(MILESTONES):
        assert r["milestones"][m] == [int(v) for v in t[:, 6 + k] if v >= 0]
    assert r["final_boards"].shape == (n, 4, 4) and r["summary"]["games"] == n


def _eight_worker(rank, world, port, q, n_global, n_games):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    torch.set_num_threads(1)
    from g2048 import dist as gdist
    from g2048.evaluate import TABLE_COLUMNS
    from oracle import oracle as O
    O.set_num_threads(1)
    w, r, _ = gdist.init("gloo", torch.device("cpu"))
    assert (w, r) == (world, rank)
    lo, hi = gdist.shard(n_global, rank, world)
    b = O.synth_boards(hi - lo, seed=3, id_base=lo) if hi > lo else np.zeros((0, 16), np.uint8)
    a = O.synth_actions(hi - lo, seed=3, step_index=2, id_base=lo) if hi > lo else np.zeros(0, np.uint8)
    if hi > lo:
        bo, sc, rw, fl = O.step_batch(b, a, np.zeros(hi - lo, np.uint32), seed=3, step_index=2, id_base=lo)
    else:
        sc, fl = np.zeros(0, np.uint32), np.zeros(0, np.uint8)
    gathered = gdist.all_gather_scores(torch.from_numpy(sc.astype(np.int32)))
    m = torch.zeros(24, dtype=torch.int64)
    m[0], m[1], m[2] = hi - lo, int(sc.sum()), int((fl & 1).sum())
    gdist.reduce_metrics(m)
    glo, ghi = gdist.shard(n_games, rank, world)
    full = np.random.default_rng(5).integers(0, 1 << 40, size=(n_games, TABLE_COLUMNS), dtype=np.int64)
    rows = gdist.all_gather_rows(torch.from_numpy(full[glo:ghi].copy()))
    times = gdist.gather_floats(10.0 + rank, torch.device("cpu"))
    slowest = gdist.max_over_ranks(10.0 + rank, torch.device("cpu"))
    gdist.barrier()
    q.put((rank, hi - lo, ghi - glo, gathered.numpy(), m.numpy(), rows.numpy(), times, slowest))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_global,n_games", [(20003, 37), (5, 3)])
def test_eight_rank_gloo_ragged_and_empty_shards(oracle, n_global, n_games):
    """The driver's rank count on CPU: world size 8 over gloo. dist.shard + all_gather_scores + reduce_metrics +
    all_gather_rows + gather_floats + max_over_ranks with ragged shards (20,003 boards, 37 games) and with more ranks than units
    (5 boards, 3 games: three / five ranks own nothing and still take part in every collective). EVERY rank must end up with the
    single-process result in global id order."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eight_worker, args=(r, world, port, q, n_global, n_games)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=240) for _ in range(world)]
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    b = oracle.synth_boards(n_global, seed=3)
    a = oracle.synth_actions(n_global, seed=3, step_index=2)
    bo, sc, rw, fl = oracle.step_batch(b, a, np.zeros(n_global, np.uint32), seed=3, step_index=2)
    from g2048.evaluate import TABLE_COLUMNS
    full = np.random.default_rng(5).integers(0, 1 << 40, size=(n_games, TABLE_COLUMNS), dtype=np.int64)
    assert sorted(r[0] for r in res) == list(range(world))
    assert sum(r[1] for r in res) == n_global and sum(r[2] for r in res) == n_games
    if n_global < world:
        assert sum(1 for r in res if r[1] == 0) == world - n_global
    for rank, nb, ng, gathered, m, rows, times, slowest in res:
        assert np.array_equal(gathered.astype(np.uint32), sc), rank
        assert m[0] == n_global and m[1] == int(sc.sum()) and m[2] == int((fl & 1).sum())
        assert np.array_equal(rows, full), rank
        assert times == [10.0 + k for k in range(world)] and slowest == 17.0


def _fallback_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), G2048_RCCL_PROBE_TIMEOUT_S="30")
    os.environ.pop("G2048_DIST_BACKEND", None)
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    from g2048 import dist as gdist
    gdist.init("nccl", None)            # what bench.py asks for; there is no GPU here, so the RCCL probe cannot succeed
    planes = gdist.backends()
    lo, hi = gdist.shard(11, rank, world)
    gathered = gdist.all_gather_scores(torch.arange(lo, hi, dtype=torch.int32))
    m = torch.tensor([hi - lo, rank + 1], dtype=torch.int64)
    gdist.reduce_metrics(m)
    slowest = gdist.max_over_ranks(5.0 + rank)
    q.put((rank, planes, gathered.tolist(), m.tolist(), slowest))
    gdist.shutdown()


def test_rccl_probe_failure_falls_back_to_gloo_on_every_rank():
    """init("nccl") on a host where RCCL cannot run (this container has no GPU): the data plane's probe fails, the ranks agree
    over the control plane, and every collective of the bench still works over gloo -- the multi-GPU bench line is produced
    (with backend_note) instead of the job dying in its one post-timing collective."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fallback_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = sorted(q.get(timeout=180) for _ in range(world))
    finally:
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    for rank, planes, gathered, m, slowest in res:
        assert planes["control"] == "gloo" and planes["data"] == "gloo" and planes["data_note"], planes
        assert gathered == list(range(11)) and m == [11, 3] and slowest == 6.0
