"""The strip DP (csrc/sdp.h) does not depend on how many of its workgroups the device holds at once.

The row groups of one DTW / WTW problem wait for one another inside a single launch.  They are handed out by ticket
(sdp::for_each_rowgroup), so a row group is only ever waited for once a *running* workgroup has taken it: progress needs
no particular residency.  These tests force the situations the round-2 review named: fewer resident workgroups than the
grid (LDS padding: one workgroup per CU; forced grid larger than that), and another kernel occupying the CUs on a second
stream while the pipeline runs.  Results must be bit-exact and no fault may be reported."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


class _Env(object):
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update({k: str(v) for k, v in self.kw.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _dtw_pairs_vs_oracle(a_list, b, tdt=torch.float32):
    import oracle
    from real_time_audio_sync_amd.dtw import dtw_batch
    from real_time_audio_sync_amd.otw_batch import frames_tensor
    dev = torch.device("cuda:0")
    a = torch.stack([frames_tensor(x, dev, tdt) for x in a_list])
    bd = frames_tensor(b, dev, tdt)
    cost, acc, _, path, plen = dtw_batch(a, bd, want_back=False, check=True)
    for k, x in enumerate(a_list):
        _, oacc, opath, _ = oracle.dtw(x, b)
        n = int(plen[k])
        assert n > 0, "pair %d: device pipeline fault" % k
        assert np.array_equal(path[k, :n].cpu().numpy(), opath), k
        assert np.array_equal(acc[k].cpu().numpy(), oacc), k


def test_more_workgroups_than_the_device_holds():
    """6 pairs x 79 strips, one strip per workgroup, 79 workgroups per pair forced = 474 workgroups, with the LDS padded
    so that only one fits a CU (256 resident): the surplus workgroups start only when earlier ones exit.  With row groups
    bound to workgroups (round 2) the wrapped-around ones waited for workgroups that could never start."""
    from real_time_audio_sync_amd import synth
    b = synth.synth_ref(260, seed=31)
    a_list = [synth.synth_ref(5050, seed=40 + k) for k in range(6)]
    with _Env(RTS_SDP_LDS_PAD=80000, RTS_SDP_CONFIG=1, RTS_SDP_GRID=79):
        _dtw_pairs_vs_oracle(a_list, b)
    # and the opposite corner: far fewer workgroups than row groups (each takes many tickets in turn)
    with _Env(RTS_SDP_CONFIG=1, RTS_SDP_GRID=3):
        _dtw_pairs_vs_oracle(a_list[:2], b)
    with _Env(RTS_SDP_CONFIG=2, RTS_SDP_GRID=1):
        _dtw_pairs_vs_oracle(a_list[:1], b)


def test_wtw_windows_with_padded_lds_and_forced_grid():
    """The WTW side of the same machinery: W = 700 windows (11 strips) for 30 streams = 330 workgroups of the one-strip
    kernel with one workgroup per CU resident."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.wtw import BatchedWTW
    dev = torch.device("cuda:0")
    B, W, hopf = 30, 700, 350
    ref, lives = synth.synth_batch(1800, B, seed=55)
    with _Env(RTS_SDP_LDS_PAD=80000, RTS_SDP_CONFIG=1, RTS_SDP_GRID=11):
        eng = BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), W, hopf, B)
        tmax = max(l.shape[1] for l in lives)
        cols = np.zeros((B, tmax, 12))
        for i, l in enumerate(lives):
            cols[i, :l.shape[1]] = l.T
        n_new = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)
        eng.push(torch.from_numpy(cols).to(dev), n_new, precheck=True)
        st = eng.states()
    assert (st[:, 3] != 3).all(), "device fault reported"
    for b in (0, 7, 29):
        o = oracle.WtwOracle(ref, W, hopf)
        o.insert_precheck()
        for q in range(lives[b].shape[1]):
            if o.push_col(lives[b][:, q]) != oracle.RUNNING:
                break
        assert o.counters["windows"] >= 2
        assert np.array_equal(eng.path(b), o.path), b
    eng.close()


def test_strip_dp_beside_another_kernel():
    """A 4 000 x 4 100 DTW (63 row groups in flight at once) on stream A while a 256-stream OTW batch (one workgroup per
    CU for ~4 ms) runs on stream B, launched first so that it holds LDS on every CU when the pipeline starts; then the
    other way round.  Both results bit-exact."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.dtw import dtw_batch
    from real_time_audio_sync_amd.otw_batch import BatchedOTW, frames_tensor
    dev = torch.device("cuda:0")
    ref_o, lives_o = synth.synth_batch(2200, 256, seed=1000)
    eng = BatchedOTW(ref_o, 500, 3, batch=256, dtype=torch.float32, device=dev)
    lv, ln = eng.pack(lives_o)
    b = synth.synth_ref(4100, seed=61)
    a = synth.synth_live(b, seed=62)[:, :4000]
    ad, bd = frames_tensor(a, dev, torch.float32), frames_tensor(b, dev, torch.float32)
    _, oacc, opath, _ = oracle.dtw(a, b)
    want_otw = {}
    for k in (0, 100, 255):
        o = oracle.OtwOracle(ref_o, 500, 3)
        o.run(lives_o[k])
        want_otw[k] = o.path
    sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    torch.cuda.synchronize()
    for order in ("otw_first", "dtw_first"):
        res = {}

        def launch_otw():
            with torch.cuda.stream(sb):
                for _ in range(3):
                    eng.run(lv, ln)

        def launch_dtw():
            with torch.cuda.stream(sa):
                res["dtw"] = dtw_batch(ad, bd, want_back=False)
        if order == "otw_first":
            launch_otw()
            launch_dtw()
        else:
            launch_dtw()
            launch_otw()
        torch.cuda.synchronize()
        cost, acc, _, path, plen = res["dtw"]
        n = int(plen[0])
        assert n > 0, "%s: device pipeline fault" % order
        assert np.array_equal(path[0, :n].cpu().numpy(), opath), order
        assert np.array_equal(acc[0].cpu().numpy(), oacc), order
        with torch.cuda.stream(sb):
            for k, want in want_otw.items():
                assert np.array_equal(eng.path(k), want), (order, k)
    eng.close()
