"""Multi-GPU plumbing: independent shards, one process per GPU, one collective at the end.

Boards (and beam-search games) are independent and every random draw is keyed by the GLOBAL board id,
so the path shards by contiguous id ranges with no data-path collective. What is exchanged is
only the final per-board scores (all-gather, 4 MiB per rank at 1,048,576 boards) and the small metrics
vector (all-reduce). Backend: "nccl" (= RCCL over xGMI) on GPUs; the same code runs on "gloo" for the
CPU tests.
"""
import os

import torch
import torch.distributed as dist


# G2048_DIST_FORCE=1: initialise the process group and run the collectives even at world size 1 (tests/test_gpu_rccl.py uses
# it to exercise RCCL on a one-GPU box); normally a single rank takes the shortcuts below
_FORCE = os.environ.get("G2048_DIST_FORCE") == "1"


def world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None, device=None):
    """Initialise the default process group from the torchrun environment (no-op for world size 1)."""
    w, r, lr = world()
    if (w > 1 or _FORCE) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("G2048_DIST_BACKEND") or backend      # rehearsal override (e.g. gloo on one GPU)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return w, r, lr


def _staged(t):
    """gloo has no device collectives for every op used here: when rehearsing on gloo with device tensors the
    exchange is staged through the host (never taken on RCCL)."""
    return dist.get_backend() == "gloo" and t.is_cuda


def shard(n_global, rank, world_size):
    """Contiguous range [lo, hi) of the global board ids owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(n_global), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def all_gather_scores(scores):
    """Per-board scores of every shard, in global id order. Equal shard sizes (the benchmark's case) use one
    all_gather_into_tensor; ragged shards fall back to all_gather of padded tensors."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return scores.clone()
    if _staged(scores):
        return all_gather_scores(scores.cpu()).to(scores.device)
    w = dist.get_world_size()
    n = torch.tensor([scores.numel()], dtype=torch.int64, device=scores.device)
    sizes = [torch.zeros_like(n) for _ in range(w)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if len(set(sizes)) == 1:
        out = torch.empty(w * sizes[0], dtype=scores.dtype, device=scores.device)
        dist.all_gather_into_tensor(out, scores.contiguous())
        return out
    m = max(sizes)
    padded = torch.zeros(m, dtype=scores.dtype, device=scores.device)
    padded[:scores.numel()] = scores
    parts = [torch.empty_like(padded) for _ in range(w)]
    dist.all_gather(parts, padded)
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
            dist.all_reduce(metrics, op=dist.ReduceOp.SUM)
    return metrics


def max_over_ranks(value, device):
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value, device):
    """Every rank's `value` (one float per rank) in rank order, on every rank: the per-rank step times of the bench line."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not _FORCE):
        return [float(value)]
    t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    parts = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    return [float(p.item()) for p in parts]
