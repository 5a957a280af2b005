"""The N>1 path: stream partition, seeds, clock/count reduction and the final host
gather, exercised with two real ranks over the gloo backend on CPU.  The per-rank compute here is the
CPU oracle standing in for the HIP kernel (tests may use it as the checker): what is under test is that
sharding changes neither which streams are aligned nor their paths."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from real_time_audio_sync_amd import shard, synth


def test_partition_is_contiguous_and_balanced():
    for n, w in ((512, 8), (64, 1), (130, 4), (3, 8), (0, 2)):
        parts = [shard.partition(n, w, r) for r in range(w)]
        assert parts[0][0] == 0 and parts[-1][1] == n
        assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in parts]
        assert max(sizes) - min(sizes) <= 1
    assert shard.partition(512, 8, 3) == (192, 256)   # BASELINE configs[3]: 64 streams per GPU


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_streams, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    ref = synth.synth_ref(150, seed=7)
    lo, hi = shard.partition(n_streams, world, rank)
    paths, frames = [], 0
    for g in range(lo, hi):
        live = synth.synth_live(ref, seed=shard.stream_seed(7, g))
        o = oracle.OtwOracle(ref, 20, 3)
        frames += o.run(live)
        paths.append(o.path)
    elapsed, total = shard.reduce_clock_and_count(0.5 + rank, frames)
    allp = shard.gather_paths(paths, dst=0)
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), elapsed=elapsed, total=total, n=len(allp),
                 **{"p%d" % i: p for i, p in enumerate(allp)})
    else:
        assert allp is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    n_streams, world = 5, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_streams, str(tmp_path)), nprocs=world, join=True)
    z = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    assert int(z["n"]) == n_streams and float(z["elapsed"]) == 1.5      # max over ranks
    import oracle
    ref = synth.synth_ref(150, seed=7)
    total = 0
    for g in range(n_streams):                                           # single-process result
        live = synth.synth_live(ref, seed=shard.stream_seed(7, g))
        o = oracle.OtwOracle(ref, 20, 3)
        total += o.run(live)
        assert np.array_equal(z["p%d" % g], o.path), g                  # same streams, same order, same paths
    assert int(z["total"]) == total


def test_identity_without_process_group():
    assert shard.reduce_clock_and_count(1.25, 7) == (1.25, 7)
    assert shard.gather_paths([np.zeros((1, 2))])[0].shape == (1, 2)
