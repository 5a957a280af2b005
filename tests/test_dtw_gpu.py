"""HIP DTW (csrc/dtw.hip through the C-ABI) against the reference-generated golden vectors and the
CPU oracle.  Bar: cost, acc_cost and back-pointers bit-exact (float64 ==), path bit-exact."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_dtw_golden(dtw_golden, otw_golden):
    import oracle
    from real_time_audio_sync_amd.dtw import DTW
    g = dtw_golden
    for cid in g["cases"]:
        cid = str(cid)
        if cid == "dtw_chopin":
            a, b = otw_golden["G/live"], otw_golden["G/ref"]
        else:
            a, b = g[cid + "/a"].astype(np.float64), g[cid + "/b"].astype(np.float64)
        cost, acc, path = DTW(a, b)
        assert np.array_equal(path, g[cid + "/path"]), cid
        ocost, oacc, opath, oback = oracle.dtw(a, b)
        assert np.array_equal(cost, ocost) and np.array_equal(acc, oacc), cid   # bit-exact vs oracle
        assert acc[-1, -1] == oacc[-1, -1]
        if max(acc.shape) <= 200:  # sizes where the reference's dgemm is canonical (see test_oracle_golden)
            assert sha(cost) == str(g[cid + "/cost_sha"]) and sha(acc) == str(g[cid + "/acc_sha"]), cid


def test_dtw_batch_shapes_and_backpointers():
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.dtw import dtw_batch
    from real_time_audio_sync_amd.otw_batch import frames_tensor
    dev = torch.device("cuda:0")
    ref = synth.synth_ref(300, seed=3)
    lives = [synth.synth_live(ref, seed=10 + b, max_frames=257)[:, :257] for b in range(5)]
    a = torch.stack([frames_tensor(l, dev, torch.float32) for l in lives])     # [5][257][12]
    b = frames_tensor(ref, dev, torch.float32)                                 # shared [300][12]
    cost, acc, back, path, plen = dtw_batch(a, b)
    torch.cuda.synchronize()
    for k, l in enumerate(lives):
        ocost, oacc, opath, oback = oracle.dtw(l, ref)
        n = int(plen[k])
        assert np.array_equal(path[k, :n].cpu().numpy(), opath), k
        assert np.array_equal(back[k].cpu().numpy(), oback), k
        assert np.array_equal(acc[k].cpu().numpy(), oacc), k


@pytest.mark.parametrize("M,N", [(1, 1), (1, 7), (9, 1), (2, 2), (513, 40), (1100, 90)])
def test_dtw_edge_shapes(M, N):
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.dtw import DTW
    a = synth.synth_ref(M, seed=M)
    b = synth.synth_ref(N, seed=N + 1)
    cost, acc, path = DTW(a, b)
    ocost, oacc, opath, _ = oracle.dtw(a, b)
    assert np.array_equal(path, opath) and np.array_equal(acc, oacc) and np.array_equal(cost, ocost)


def test_dtw_config1_sizes():
    """BASELINE configs[0] shapes: two 30 s clips at hop 2048 (322 frames) and hop 512 (1289)."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.dtw import DTW
    for n in (322, 1289):
        r = synth.synth_ref(n, seed=n)
        l = synth.synth_live(r, seed=n + 1, max_frames=n)
        cost, acc, path = DTW(l, r)
        ocost, oacc, opath, _ = oracle.dtw(l, r)
        assert np.array_equal(path, opath) and np.array_equal(acc, oacc)
        # size-independent properties of a DTW path
        assert tuple(path[0]) == (0, 0) and tuple(path[-1]) == (l.shape[1] - 1, n - 1)
        step = np.diff(path, axis=0)
        assert ((step >= 0).all() and (step <= 1).all() and (step.sum(axis=1) >= 1).all())
