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
