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


# 768 / 769 rows: 12 strips (whole backtrack in one launch) / 13 strips (a workgroup per strip)
@pytest.mark.parametrize("M,N", [(1, 1), (1, 7), (9, 1), (2, 2), (513, 40), (1100, 90), (768, 130), (769, 130), (64, 700)])
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


def test_dtw_long_sequences_hbm_diagonals():
    """Long pairs: the strip pipeline spans hundreds of workgroups (csrc/sdp.h).  7000 x 900 against the oracle, then a
    30-minute-sized pair (19 380 x 19 380: 3 GB each for cost and acc_cost) through size-independent properties."""
    import oracle
    from real_time_audio_sync_amd import _native as nat, synth
    from real_time_audio_sync_amd.dtw import DTW, dtw_batch
    from real_time_audio_sync_amd.otw_batch import frames_tensor
    a = synth.synth_ref(7000, seed=70)
    b = synth.synth_live(synth.synth_ref(900, seed=71), seed=72, max_frames=900)
    cost, acc, path = DTW(a, b)
    ocost, oacc, opath, _ = oracle.dtw(a, b)
    assert np.array_equal(path, opath) and np.array_equal(acc, oacc) and np.array_equal(cost, ocost)
    dev = torch.device("cuda:0")
    n = 19380
    ref = synth.synth_ref(n, seed=80)
    live = synth.synth_live(ref, seed=81)
    ad, bd = frames_tensor(live, dev, torch.float32), frames_tensor(ref, dev, torch.float32)
    cost, acc, back, pth, plen = dtw_batch(ad, bd, check=True)
    m = ad.shape[0]
    p = pth[0, : int(plen[0])].cpu().numpy()
    assert tuple(p[0]) == (0, 0) and tuple(p[-1]) == (m - 1, n - 1)
    step = np.diff(p, axis=0)
    assert (step >= 0).all() and (step <= 1).all() and (step.sum(axis=1) >= 1).all()
    # the accumulated cost at the end equals the sum of weighted costs along the returned path (dtw.py:35-37)
    c = cost[0][torch.from_numpy(p[:, 0]).to(dev).long(), torch.from_numpy(p[:, 1]).to(dev).long()].cpu().numpy()
    w = np.where(step.sum(axis=1) == 2, 2.0, 1.0)
    total = c[0] + float((c[1:] * w).sum())
    assert abs(total - float(acc[0, -1, -1])) <= 1e-9 * max(1.0, abs(total))
    # synthetic warps stay within 0.8..1.25 of the diagonal
    assert np.abs(p[:, 1] - p[:, 0] * (n / float(m))).max() < 0.2 * n


def test_dtw_batched_pipelines_side_by_side():
    """Several pairs whose strip pipelines span many workgroups each, in one launch (every pair's row groups hand
    their bottom rows to the next through HBM while the other pairs' pipelines run beside them), with float64 and
    float32 inputs and a shared b: every pair's cost, acc_cost, back-pointers and path against the oracle."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.dtw import dtw_batch
    from real_time_audio_sync_amd.otw_batch import frames_tensor
    dev = torch.device("cuda:0")
    ref = synth.synth_ref(333, seed=901)
    lives = [synth.synth_live(synth.synth_ref(900, seed=910 + k), seed=920 + k, max_frames=700)[:, :700] for k in range(6)]
    assert all(l.shape[1] == 700 for l in lives)
    for tdt in (torch.float64, torch.float32):
        a = torch.stack([frames_tensor(l, dev, tdt) for l in lives])       # [6][700][12]: 11 strips per pair
        b = frames_tensor(ref, dev, tdt)
        cost, acc, back, path, plen = dtw_batch(a, b, check=True)
        for k, l in enumerate(lives):
            lk = l.astype(np.float32).astype(np.float64) if tdt == torch.float32 else l
            rk = ref.astype(np.float32).astype(np.float64) if tdt == torch.float32 else ref
            ocost, oacc, opath, oback = oracle.dtw(lk, rk)
            n = int(plen[k])
            assert np.array_equal(path[k, :n].cpu().numpy(), opath), (str(tdt), k)
            assert np.array_equal(acc[k].cpu().numpy(), oacc), (str(tdt), k)
            assert np.array_equal(cost[k].cpu().numpy(), ocost), (str(tdt), k)
            assert np.array_equal(back[k].cpu().numpy(), oback), (str(tdt), k)


def test_strip_dp_soak_short():
    """tests/sdp_soak.py, shortened: seeded random DTW shapes / batches / dtypes and WTW windows on the strip-DP path,
    everything bit-exact against the oracle (cost, acc_cost, back-pointers, paths, pointers).  (The long form, run by
    hand when sdp.h changes: 600 trials = 1 038 problems at the end of round 2.)"""
    import sdp_soak
    assert sdp_soak.run(45, seed=17, verbose=False) >= 45
