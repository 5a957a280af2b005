"""bench.py's N>1 branch over RCCL (backend "nccl"), one process per GPU, on however many gfx950 devices the box
has -- the path the driver's scaling run takes.  Skipped on a one-GPU box (RCCL refuses two ranks on one
device; the 2-rank rehearsal over gloo lives in tests/test_shard_cpu.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_over_rccl_on_all_devices():
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from real_time_audio_sync_amd import _native as nat
    n = nat.lib.rts_device_count()
    if n < 2:
        pytest.skip("one gfx950 device: the RCCL leg needs at least two")
    n = min(n, 6)  # the GPU box allows at most 6 processes on its cards at once
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", str(n), "--steps", "3", "--warmup", "1", "--batch", "8", "--n-ref", "500", "--c", "100"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["n_gpus"] == n and r["scaling"] == "weak" and r["config"]["streams_total"] == 8 * n
    assert r["parity"]["path_mismatches"] == 0 and r["value"] > 0


def test_one_process_many_devices():
    """shard.ShardedOTW: one host process, one BatchedOTW per device, contiguous stream slices, no exchange.  Runs on
    every visible device, and always also as two slices on device 0 (which exercises the slicing, the per-handle
    device switch and two handles side by side on a one-GPU box); results must equal the oracle's."""
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np
    import oracle
    from real_time_audio_sync_amd import shard, synth
    ref, lives = synth.synth_batch(400, 7, seed=77)
    lives[4] = lives[4][:, :150]
    want = []
    for live in lives:
        o = oracle.OtwOracle(ref, 100, 3)
        o.run(live)
        want.append(np.asarray(o.path))
    layouts = [[0, 0], [0, 0, 0]]
    if torch.cuda.device_count() > 1:
        layouts.append(list(range(min(torch.cuda.device_count(), 8))))
    for devices in layouts:
        eng = shard.ShardedOTW(ref, 100, 3, batch=7, devices=devices, dtype=torch.float32)
        assert [hi - lo for lo, hi in eng.slices] == [hi - lo for lo, hi in
                                                      (shard.partition(7, len(devices), r) for r in range(len(devices)))]
        packed = eng.pack(lives)
        eng.run(packed)
        eng.synchronize()
        got = eng.paths()
        assert len(got) == 7
        for b in range(7):
            assert np.array_equal(got[b], want[b]), (devices, b)
            assert np.array_equal(eng.path(b), want[b])
            assert eng.state(b)["n_path"] == len(want[b])
        eng.close()
