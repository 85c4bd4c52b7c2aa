"""Multi-GPU plumbing: independent shards, one process per GPU, one collective at the end.

Boards (and beam-search games) are independent and every random draw is keyed by the GLOBAL board id,
so the path shards by contiguous id ranges with no data-path collective. What is exchanged is
only the final per-board scores (all-gather, 4 MiB per rank at 1,048,576 boards) and the small metrics
vector (all-reduce).

Two planes (`init("nccl", device)`, what bench.py and the sharded evaluation use on GPUs):
  * control -- the default process group, backend "gloo": barriers, the max / gather of per-rank clock readings. Host
    tensors over local TCP; it does not depend on the GPUs' IPC / xGMI state, so a rank can always tell the others that
    something went wrong;
  * data    -- a second group, backend "nccl" (= RCCL over xGMI on ROCm), for the device-tensor collectives the north star
    names: the all-gather of the per-board scores and the all-reduce of the metrics vector. It is PROBED once at init (a
    one-element all-reduce with a short timeout, its outcome agreed over the control plane); if RCCL cannot run on this node
    (no IPC, one card shared by several ranks in a rehearsal, ...) every rank falls back -- together -- to staging the same
    exchange through the host over gloo, and `backends()` says so. The timed region of the bench contains no collective
    either way.
`init("gloo")` (the CPU tests) is one plane: everything over gloo.
"""
import datetime
import os

import torch
import torch.distributed as dist


# G2048_DIST_FORCE=1: initialise the process group and run the collectives even at world size 1 (tests/test_gpu_rccl.py uses
# it to exercise RCCL on a one-GPU box); normally a single rank takes the shortcuts below
_FORCE = os.environ.get("G2048_DIST_FORCE") == "1"
_DATA = None            # the RCCL group of the data plane (None: device tensors are staged through the host over gloo)
_DATA_NOTE = None       # why the data plane is not RCCL, if it was asked for
_PROBE_TIMEOUT_S = float(os.environ.get("G2048_RCCL_PROBE_TIMEOUT_S", "90"))
# every wait on the control plane is bounded too (torch's default is 30 minutes): a rank that never arrives costs the others
# five minutes and an exception, not the job's time limit
_CONTROL_TIMEOUT_S = float(os.environ.get("G2048_CONTROL_TIMEOUT_S", "300"))


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def _first_line(e):
    text = str(e).strip().splitlines()
    return "%s: %s" % (type(e).__name__, text[0][:300] if text else "")


def _open_data_plane(device):
    """RCCL group for the device collectives, probed; every rank takes the same decision (agreed over gloo)."""
    global _DATA, _DATA_NOTE
    group, err = None, None
    try:
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        group = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=_PROBE_TIMEOUT_S))
        probe = torch.ones(1, dtype=torch.int32, device=device)
        dist.all_reduce(probe, group=group)
        torch.cuda.synchronize(device)
        if int(probe.item()) != dist.get_world_size():
            err = "probe all-reduce returned %d for world size %d" % (int(probe.item()), dist.get_world_size())
    except Exception as e:      # noqa: BLE001 -- whatever RCCL raises here (ncclInvalidUsage, IPC failure, timeout): fall back
        err = _first_line(e)
    ok = torch.tensor([0 if err else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)           # control plane
    if int(ok.item()) == 1:
        _DATA, _DATA_NOTE = group, None
    else:
        _DATA, _DATA_NOTE = None, err or "the RCCL probe failed on another rank"


def init(backend=None, device=None):
    """Initialise the process group(s) from the torchrun environment (no-op for world size 1)."""
    w, r, lr = world()
    if (w > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("G2048_DIST_BACKEND") or backend      # rehearsal override (e.g. gloo on one GPU)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            # a collective that cannot complete raises in the caller after its timeout (-> the fallback above) instead of
            # the watchdog thread aborting the process
            os.environ.setdefault("TORCH_NCCL_BLOCKING_WAIT", "1")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=_CONTROL_TIMEOUT_S))
            _open_data_plane(device)
        else:
            dist.init_process_group(backend)
    return w, r, lr


def backends():
    """{"control": ..., "data": ..., "data_note": ...} of the initialised planes (None before init / at world size 1)."""
    if not dist.is_initialized():
        return None
    ctl = dist.get_backend()
    return {"control": ctl, "data": "nccl" if _DATA is not None else ctl, "data_note": _DATA_NOTE}


def _staged(t):
    """Device tensors without an RCCL data plane (gloo rehearsal on a GPU, or the fallback): the exchange is staged through
    the host."""
    return _DATA is None and t.is_cuda


def _group(t):
    return _DATA if (_DATA is not None and t.is_cuda) else None


def shard(n_global, rank, world_size):
    """Contiguous range [lo, hi) of the global board ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_global), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    """Control plane. (Callers bracket timed regions with barrier() + torch.cuda.synchronize(): the barrier orders the hosts,
    the synchronize drains this rank's GPU.)"""
    if dist.is_initialized():
        dist.barrier()


def _sizes(n_local):
    w = dist.get_world_size()
    n = torch.tensor([int(n_local)], dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)                           # control plane (host)
    return [int(s.item()) for s in sizes]


def all_gather_scores(scores):
    """Per-board scores of every shard, in global id order. Equal shard sizes (the benchmark's case) use one
    all_gather_into_tensor; ragged shards fall back to all_gather of padded tensors."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return scores.clone()
    if _staged(scores):
        return all_gather_scores(scores.cpu()).to(scores.device)
    g = _group(scores)
    w = dist.get_world_size()
    sizes = _sizes(scores.numel())
    if len(set(sizes)) == 1:
        out = torch.empty(w * sizes[0], dtype=scores.dtype, device=scores.device)
        dist.all_gather_into_tensor(out, scores.contiguous(), group=g)
        return out
    m = max(sizes)
    padded = torch.zeros(m, dtype=scores.dtype, device=scores.device)
    padded[:scores.numel()] = scores
    parts = [torch.empty_like(padded) for _ in range(w)]
    dist.all_gather(parts, padded, group=g)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)])


def all_gather_rows(rows):
    """Row-wise concatenation of every rank's (n_r, C) int64 table in rank order (ragged n_r allowed)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return rows.clone()
    flat = all_gather_scores(rows.reshape(-1))
    return flat.reshape(-1, rows.shape[1])


def reduce_metrics(metrics):
    """Sum of the per-shard metric vectors (g2048_metrics layout) over all ranks, in place."""
    if dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE):
        if _staged(metrics):
            h = metrics.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            metrics.copy_(h)
        else:
            dist.all_reduce(metrics, op=dist.ReduceOp.SUM, group=_group(metrics))
    return metrics


def max_over_ranks(value, device=None):
    """Control plane: the slowest rank's clock reading."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device=None):
    """Control plane: every rank's `value` (one float per rank) in rank order, on every rank: the per-rank step times of the
    bench line."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]


def shutdown():
    """Leave together and drop the groups (a failed RCCL probe leaves nothing to tear down on the data plane)."""
    global _DATA
    if dist.is_initialized():
        try:
            dist.barrier()
        finally:
            _DATA = None
            dist.destroy_process_group()
