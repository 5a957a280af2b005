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


class ShardedOTW:
    """One host process driving several devices ("one Python process driving 8 devices with per-device
    streams is enough"): the streams are cut into contiguous slices (``partition``), every device gets its own replica
    of the reference and its own ``BatchedOTW``, launches are asynchronous per device, and nothing is exchanged between
    devices.  ``devices``: list of device indices (default: all visible ones); stream b lives on the device whose
    slice contains it.  The interface follows BatchedOTW's, with global stream indices."""

    def __init__(self, ref, c, max_run_count, batch, devices=None, **kwargs):
        from .otw_batch import BatchedOTW
        if devices is None:
            devices = list(range(torch.cuda.device_count()))
        if not devices:
            raise RuntimeError("ShardedOTW needs at least one ROCm GPU (no CPU fallback)")
        self.B = int(batch)
        before = torch.cuda.current_device()
        self.slices, self.engines = [], []
        for r, d in enumerate(devices):
            lo, hi = partition(self.B, len(devices), r)
            if hi > lo:
                self.slices.append((lo, hi))
                self.engines.append(BatchedOTW(ref, c, max_run_count, batch=hi - lo, device="cuda:%d" % int(d), **kwargs))
        torch.cuda.set_device(before)

    def _locate(self, b):
        for (lo, hi), eng in zip(self.slices, self.engines):
            if lo <= b < hi:
                return eng, b - lo
        raise IndexError("stream %d of %d" % (b, self.B))

    def pack(self, lives, dtype=None):
        """List of B (12, T_b) arrays -> one (frames, lengths) pair per device, resident there."""
        assert len(lives) == self.B
        return [eng.pack(lives[lo:hi], dtype) for (lo, hi), eng in zip(self.slices, self.engines)]

    def run(self, packed, mode="insert"):
        """Launch every device's slice (asynchronous: returns once all launches are queued)."""
        for eng, (lv, ln) in zip(self.engines, packed):
            eng.run(lv, ln, mode=mode)

    def reset(self):
        for eng in self.engines:
            eng.reset()

    def synchronize(self):
        for eng in self.engines:
            torch.cuda.synchronize(eng.device)

    def path(self, b):
        eng, i = self._locate(b)
        return eng.path(i)

    def paths(self):
        return [p for eng in self.engines for p in eng.paths()]

    def state(self, b):
        eng, i = self._locate(b)
        return eng.state(i)

    def bands(self, b):
        eng, i = self._locate(b)
        return eng.bands(i)

    def close(self):
        for eng in self.engines:
            eng.close()
        self.engines = []
