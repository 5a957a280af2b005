"""rts_live_* (csrc/live.hip): raw audio of many microphones -> device ring buffers -> HIP chroma -> OTW / WTW state,
one host-to-device copy per feed and no read-back.  Checked against the same computation done offline, against the
CPU oracle, and -- for the WTW form -- against the reference's own known-answer file."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

PARAMS = {'fft_len': 4096, 'hop_size': 2048, 'dtw_win_size': 4096 * 10, 'dtw_hop_size': 2048 * 10}  # tests.py:174


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _offline_paths(ref_chroma, lives, c):
    """Un-padded hop framing of each stream's samples (numpy oracle chroma), then the C oracle's OTW insert loop."""
    import oracle
    from oracle import chroma_oracle
    out = []
    for x in lives:
        n = (len(x) - 4096) // 2048 + 1 if len(x) >= 4096 else 0
        cols = np.stack([chroma_oracle.wav_to_chroma_col(x[m * 2048:m * 2048 + 4096]) for m in range(n)], axis=1) if n else np.zeros((12, 0))
        o = oracle.OtwOracle(ref_chroma, c, 3)
        if n:
            o.run(cols)
        out.append((o.path, o.state))
    return out


def test_int16_and_float_feeds_agree_with_the_oracle(chopin_audio, otw_golden):
    """Four microphones with different buffer sizes, two of them delivering PCM16 through feed_block-style int16 feeds;
    the device path must equal the oracle's on the same samples (the chroma differs from numpy's by ~1e-15, far below
    what could move an OTW decision on real audio; the same comparison test_live_gpu.py makes)."""
    from real_time_audio_sync_amd.live import LiveSession
    ref_chroma = otw_golden["G/ref"]
    pcm = np.round(chopin_audio["live"] * 32768.0).astype(np.int16)     # the recording as a mono PCM16 microphone would deliver it
    live = pcm.astype(np.float32) / np.float32(32768.0)                # ... and as librosa.load would return that (exact)
    lens = (len(live), len(live) // 3, len(live) - 12345, 5000)
    want = _offline_paths(ref_chroma, [live[:n] for n in lens], 50)
    for dt in (np.float32, np.int16):
        sess = LiveSession(ref_chroma, batch=4, c=50, max_run_count=3)
        src = live if dt == np.float32 else pcm
        sizes = (1500, 4096, 9000, 700)
        pos = [0, 0, 0, 0]
        while any(pos[b] < lens[b] for b in range(4)):
            bufs = []
            for b in range(4):
                n = min(sizes[b], lens[b] - pos[b])
                bufs.append(src[pos[b]:pos[b] + n] if n > 0 else None)
                pos[b] += max(n, 0)
            sess.feed(bufs)
        sess.sync()
        info = sess.poll()
        assert info["feeds_done"] == info["feeds_submitted"] > 0
        for b in range(4):
            path, st = want[b]
            assert np.array_equal(sess.path(b), path), (str(dt), b)
            assert tuple(info["positions"][b]) == (st["t"], st["j"]), (str(dt), b)
            assert info["status"][b] == st["status"]
        # the host mirror of the pending counts is what the device holds
        want_pending = [n - ((n - 4096) // 2048 + 1) * 2048 if n >= 4096 else n for n in lens]
        assert list(sess.pending()) == want_pending
        sess.close()


def test_wtw_live_session_reproduces_the_known_answer(chopin_audio, wtw_known_answer):
    """The reference's file-driven WTW run (tests.py:180-190: np.array_split(live, 4096) buffers into WTW.insert) through
    the device-side ingestion, for three streams at once -- two with the reference's buffering, one with 1-second
    buffers: all three must give Songs/chopin/tests/wtw_test_20b.txt."""
    from real_time_audio_sync_amd import chroma
    from real_time_audio_sync_amd.live import LiveSession
    plan = chroma._plan()
    ref_dev = torch.from_numpy(chopin_audio["ref"]).to(plan.device)
    ref_chroma = plan.frames(ref_dev, pad_left=2048)[0].t().contiguous().cpu().numpy()      # wtw.py:37-41
    sess = LiveSession(ref_chroma, batch=3, wtw_params=PARAMS)
    live = chopin_audio["live"]
    parts = np.array_split(live, 4096)
    big = [live[i:i + 22050] for i in range(0, len(live), 22050)]
    k = 0
    for i, buf in enumerate(parts):
        third = None
        if i % 40 == 0 and k < len(big):
            third = big[k]
            k += 1
        sess.feed([buf, buf, third])
    while k < len(big):
        sess.feed([None, None, big[k]])
        k += 1
    sess.sync()
    info = sess.poll()
    for b in range(3):
        assert np.array_equal(sess.path(b), wtw_known_answer), b
    assert [tuple(p) for p in info["positions"]] == [(380, 360)] * 3
    assert sess.stopped() == [0, 1, 2]                                        # insert() returned "stop" (wtw.py:96-97)
    sess.close()


def test_feed_block_stop_and_overflow_reporting():
    from real_time_audio_sync_amd import _native as nat, synth
    from real_time_audio_sync_amd.live import LiveSession
    ref = synth.synth_ref(30, seed=3)
    sess = LiveSession(ref, batch=5, c=10, max_run_count=3, max_pending=3 * 4096)
    rs = np.random.RandomState(1)
    sess.feed_block((rs.rand(5, 4096) - 0.5).astype(np.float32))             # leaves 2048 samples pending per stream
    with pytest.raises(nat.RtsyncError):
        sess.feed_block((rs.rand(5, 3 * 4096) - 0.5).astype(np.float32))     # 2048 + 12288 > max_pending: refused, nothing changes
    assert (sess.pending() == 2048).all()
    for _ in range(39):                                                       # 40 x 4096 samples = 80 hops > 2N = 60 frames
        sess.feed_block((rs.rand(5, 4096) - 0.5).astype(np.float32))
    sess.sync()
    info = sess.poll()
    assert info["feeds_done"] == 40
    assert set(info["status"]) <= {nat.STOP_REF_END, nat.LIVE_OVERFLOW}         # every stream ran out of one or the other
    assert sess.stopped() == [b for b in range(5) if info["status"][b] == nat.STOP_REF_END]
    assert [int(s) for s in info["status"]] == [sess.otw.state(b)["status"] for b in range(5)]
    assert (sess.pending() == 2048).all()
    sess.reset()
    assert (sess.pending() == 0).all() and sess.poll()["feeds_submitted"] == 0
    sess.feed_block((rs.rand(5, 4096) - 0.5).astype(np.float32))
    sess.sync()
    assert sess.otw.state(0)["consumed"] == 1
    sess.close()
