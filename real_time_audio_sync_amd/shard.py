"""Multi-GPU sharding of independent live streams.

Streams never interact: each rank (one process per GPU) takes a contiguous slice of the streams
and runs the single-GPU kernel on it with its own replica of the (<= 1 MB) reference chroma.  There is
NO collective on the data path; torch.distributed is used only for (a) the timing barrier / max-clock
of the benchmark and (b) the final host-side gather of paths.  Works unchanged on backend "nccl"
(= RCCL, GPU ranks) and "gloo" (CPU ranks, used by the tests)."""
import torch
import torch.distributed as dist


def partition(n_streams, world, rank):
    """Contiguous, balanced slice [lo, hi) of ``n_streams`` for ``rank`` (first ranks take the remainder)."""
    base, rem = divmod(int(n_streams), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def stream_seed(base_seed, global_stream):
    """Seed of a synthetic stream depends on its global index only, so any sharding aligns the same data."""
    return int(base_seed) + 1 + int(global_stream)


def reduce_clock_and_count(elapsed_s, count, device=None):
    """(max over ranks of elapsed, sum over ranks of count); identity when not initialised."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), int(count)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    n = torch.tensor([int(count)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return float(t.item()), int(n.item())


def gather_paths(local_paths, dst=0):
    """Final host gather: list of per-stream (P_b, 2) int32 arrays from every rank -> on ``dst`` the
    concatenation in global stream order (ranks hold contiguous slices), elsewhere None."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return list(local_paths)
    world, rank = dist.get_world_size(), dist.get_rank()
    out = [None] * world if rank == dst else None
    dist.gather_object(list(local_paths), out, dst=dst)
    if rank != dst:
        return None
    return [p for chunk in out for p in chunk]
