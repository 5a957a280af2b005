"""HIP WTW (csrc/wtw.hip + csrc/chroma.hip through the C-ABI) against the reference's known-answer
file, reference-generated window vectors and the CPU oracle.  Bar: path indices, pointers and the
window DP matrix D bit-exact given identical chroma columns."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

PARAMS = {'fft_len': 4096, 'hop_size': 2048, 'dtw_win_size': 4096 * 10, 'dtw_hop_size': 2048 * 10}  # tests.py:174


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_wtw_known_answer_on_gpu(chopin_audio, wtw_known_answer):
    """Songs/chopin/tests/wtw_test_20b.txt reproduced end to end on the GPU: audio -> HIP chroma ->
    HIP window DP -> 509 (live, ref) pairs, driven exactly like tests.py:180-190 drives the reference."""
    from real_time_audio_sync_amd.wtw import WTW
    wtw = WTW.from_samples(chopin_audio["ref"], PARAMS, {'chroma': False})
    assert wtw.chroma_ref.shape == (12, 380)
    for buf in np.array_split(chopin_audio["live"], 4096):
        cont = wtw.insert(buf.tolist())
        if cont == "stop":
            break
    wtw_path = np.array(wtw.path)
    assert wtw_path.shape == (509, 2) and np.array_equal(wtw_path, wtw_known_answer)
    assert (wtw.live_ptr, wtw.ref_ptr) == (380, 360)
    assert wtw.insert([0.0] * 10) == "stop"  # sticky


def test_windows_against_reference_vectors(wtw_window_golden, wtw_path):
    """One window = get_cost_matrix + run_dtw + find_path of the reference (called directly in
    make_golden.py).  The window is forced by giving the handle a reference a few frames longer
    than the window and hop = W-1 so the whole sub-path is handed over."""
    from real_time_audio_sync_amd.wtw import BatchedWTW
    g = wtw_window_golden
    dev = torch.device("cuda:0")
    for cid in ("win20", "win16_silent", "win100"):
        x, y = g[cid + "/x"], g[cid + "/y"]
        W = x.shape[1]
        assert y.shape[1] == W
        ref = np.concatenate([y, np.ones((12, 4))], axis=1)        # frames beyond the window are never read
        eng = BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), W, W - 1, 1, keep_last_d=True)
        eng.push(torch.from_numpy(np.ascontiguousarray(x.T))[None].to(dev), precheck=True)
        st = eng.state()
        assert st["windows"] == 1 and st["cells"] == W * W
        assert np.array_equal(eng.last_d(), g[cid + "/D"], equal_nan=True), cid
        sub = g[cid + "/sub"]
        assert np.array_equal(eng.path(), sub), cid  # every point has l <= W-1
        eng.close()


@pytest.fixture(params=["win", "win_two_waves", "older"])
def wtw_path(request, monkeypatch):
    """Windows of at most 128 frames can run on wtw_win_kernel (every window of a push in one launch; the default up to
    104 frames, forced up to 128 here); RTS_WTW_WIN=0 selects the older kernels (anti-diagonal sweep up to 64 frames,
    strip DP above), which stay covered this way.  "win_two_waves" sends windows of 65 frames or fewer through the
    two-wave form of the kernel as well (RTS_WIN_FORCE_R2: its second DP wave then has no rows)."""
    monkeypatch.setenv("RTS_WTW_WIN", "0" if request.param == "older" else "1")
    if request.param == "win_two_waves":
        monkeypatch.setenv("RTS_WIN_FORCE_R2", "1")
    return request.param


def test_batched_streams_vs_oracle(wtw_path):
    """Synthetic chroma, several streams, W=20/hop=10 and W=100/hop=50 (wtw_live.py's setting)."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.wtw import BatchedWTW
    dev = torch.device("cuda:0")
    ref, lives = synth.synth_batch(600, 4, seed=41)
    lives[2] = lives[2][:, :250]
    # un-normalise columns a little: the cosine cost must renormalise them
    lives = [l * (0.5 + np.random.RandomState(b).rand(1, l.shape[1])) for b, l in enumerate(lives)]
    for W, hopf in ((20, 10), (100, 50), (130, 7)):
        eng = BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), W, hopf, 4)
        tmax = max(l.shape[1] for l in lives)
        cols = np.zeros((4, tmax, 12))
        for b, l in enumerate(lives):
            cols[b, : l.shape[1]] = l.T
        n_new = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)
        # push in two uneven chunks to exercise the persisted state
        cut = 123
        eng.push(torch.from_numpy(cols[:, :cut].copy()).to(dev), torch.clamp(n_new, max=cut), precheck=True)
        eng.push(torch.from_numpy(cols[:, cut:].copy()).to(dev), torch.clamp(n_new - cut, min=0), precheck=True)
        for b, l in enumerate(lives):
            o = oracle.WtwOracle(ref, W, hopf)
            for q in range(l.shape[1]):
                if o.push_col(l[:, q]) != oracle.RUNNING:
                    break
            st, so = eng.state(b), o.state
            assert np.array_equal(eng.path(b), o.path), (W, b)
            assert (st["live_ptr"], st["ref_ptr"], st["status"]) == (so["live_ptr"], so["ref_ptr"], so["status"]), (W, b)
            assert (st["windows"], st["cells"]) == (o.counters["windows"], o.counters["cells"])
        eng.close()


# 768 / 800 frames: 12 strips (backtrack and control step in one launch) / 13 strips (separate kernels)
@pytest.mark.parametrize("W,hopf,n_ref", [(700, 350, 2500), (2000, 1000, 5000), (768, 300, 2600), (800, 410, 2700)])
def test_large_windows_hbm_resident(W, hopf, n_ref):
    """Windows beyond the 512 frames that fit LDS run from an HBM workspace; results stay bit-exact."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.wtw import BatchedWTW
    dev = torch.device("cuda:0")
    ref, lives = synth.synth_batch(n_ref, 2, seed=90 + W)
    eng = BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), W, hopf, 2)
    tmax = max(l.shape[1] for l in lives)
    cols = np.zeros((2, tmax, 12))
    for b, l in enumerate(lives):
        cols[b, : l.shape[1]] = l.T
    n_new = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)
    eng.push(torch.from_numpy(cols).to(dev), n_new, precheck=True)
    for b, l in enumerate(lives):
        o = oracle.WtwOracle(ref, W, hopf)
        for q in range(l.shape[1]):
            if o.push_col(l[:, q]) != oracle.RUNNING:
                break
        st, so = eng.state(b), o.state
        assert o.counters["windows"] >= 2
        assert np.array_equal(eng.path(b), o.path), (W, b)
        assert (st["live_ptr"], st["ref_ptr"], st["status"], st["windows"]) == (
            so["live_ptr"], so["ref_ptr"], so["status"], o.counters["windows"]), (W, b)
    eng.close()


def test_config5_window_10000():
    """BASELINE configs[4] shape: 30-minute reference (19 380 frames), window 10 000 frames, window hop
    5 000 (wtw.py:242's hop = W/2), one stream.  float64 throughout (the config's fp16 band would not
    be bit-exact); checked against the CPU oracle on the same chroma."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.wtw import BatchedWTW
    dev = torch.device("cuda:0")
    ref = synth.synth_ref(19380, seed=500)
    live = synth.synth_live(ref, seed=501)
    eng = BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), 10000, 5000, 1)
    eng.push(torch.from_numpy(np.ascontiguousarray(live.T))[None].to(dev), precheck=True)
    st = eng.state()
    p = eng.path()
    assert st["windows"] >= 1 and st["cells"] == st["windows"] * 10000 * 10000
    # size-independent properties of a WTW path: monotone steps of at most one frame inside a window
    step = np.diff(p, axis=0)
    assert (step >= 0).all() and (step <= 1).all()
    o = oracle.WtwOracle(ref, 10000, 5000)
    for q in range(live.shape[1]):
        if o.push_col(live[:, q]) != oracle.RUNNING:
            break
    assert np.array_equal(p, o.path)
    assert (st["live_ptr"], st["ref_ptr"], st["windows"]) == (o.state["live_ptr"], o.state["ref_ptr"], o.counters["windows"])
    eng.close()


def test_config5_window_against_the_reference_fixture():
    """configs[4] pinned by the REFERENCE: tests/golden/wtw10k_golden.json holds what wtw.py's own get_cost_matrix /
    run_dtw / find_path gave for the first W = 10 000 window of this workload (tests/golden/make_wtw10k_golden.py, 15
    minutes of the reference's Python; the C port was checked equal there): sha256 of D's last row and last column,
    D[-1, -1], and of the path WTW.insert hands over with dtw_hop = 5 000 frames (wtw.py:107-117)."""
    import hashlib
    import json
    import os
    from conftest import GOLDEN
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.wtw import BatchedWTW
    gold = json.load(open(os.path.join(GOLDEN, "wtw10k_golden.json")))
    assert "reference" in gold and gold["reference"] == gold["oracle"]
    dev = torch.device("cuda:0")
    ref = synth.synth_ref(19380, seed=500)
    live = synth.synth_live(ref, seed=501)[:, :10000]
    eng = BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), 10000, 5000, 1, keep_last_d=True)
    eng.push(torch.from_numpy(np.ascontiguousarray(live.T))[None].to(dev), precheck=True)
    st = eng.state()
    assert st["windows"] == 1 and st["cells"] == 10000 * 10000 and st["status"] == 0
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
    p = eng.path().astype(np.int32)
    assert len(p) == gold["path_len"] and sha(p) == gold["path_sha256"]
    D = eng.last_d()
    assert float(D[-1, -1]) == gold["reference"]["d_end"]
    assert sha(D[-1]) == gold["reference"]["d_last_row_sha"]
    assert sha(np.ascontiguousarray(D[:, -1])) == gold["reference"]["d_last_col_sha"]
    eng.close()


def test_randomized_small_windows(wtw_path):
    """Seeded sweep: window sizes from 1 frame up, hops from 1 to W, references barely longer than a
    window, live streams shorter / longer than the reference, silent (all-zero) frames -> NaN costs."""
    import oracle
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.wtw import BatchedWTW
    dev = torch.device("cuda:0")
    rs = np.random.RandomState(77)
    for trial in range(40):
        W = int(rs.choice([1, 2, 3, 5, 8, 16, 20, 33, 63, 64, 65, 100, 127, 128, 129, 200]))
        hopf = int(rs.randint(1, W + 1))
        M = int(W + rs.choice([1, 2, 3, 10, 60, 150]))
        ref = synth.synth_ref(M, seed=trial)
        live = synth.synth_live(ref, seed=500 + trial, lo=float(rs.uniform(0.5, 1.0)), hi=float(rs.uniform(1.0, 2.0)))
        if live.shape[1] == 0:
            live = ref[:, :1].copy()
        extra = int(rs.choice([0, 0, W, 2 * M]))
        if extra:
            live = np.concatenate([live, np.repeat(live[:, -1:], extra, axis=1)], axis=1)
        live = live * (0.5 + rs.rand(1, live.shape[1]))       # un-normalised columns
        if rs.rand() < 0.3 and live.shape[1] > 3:
            live[:, int(rs.randint(0, live.shape[1]))] = 0.0   # a silent frame (wtw.py:169 divides by zero)
        eng = BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), W, hopf, 1)
        cut = int(rs.randint(0, live.shape[1] + 1))
        cols = torch.from_numpy(np.ascontiguousarray(live.T))[None].to(dev)
        if cut > 0:
            eng.push(cols[:, :cut].contiguous(), precheck=True)
        if cut < live.shape[1]:
            eng.push(cols[:, cut:].contiguous(), precheck=True)
        o = oracle.WtwOracle(ref, W, hopf)
        with np.errstate(all="ignore"):
            for q in range(live.shape[1]):
                # the drop-in applies insert()'s entry check once per push; the oracle mirrors that
                if q == 0 or q == cut:
                    if o.insert_precheck() != oracle.RUNNING:
                        break
                if o.push_col(live[:, q]) != oracle.RUNNING:
                    break
        st, so = eng.state(), o.state
        tag = (trial, W, hopf, M, live.shape[1], cut)
        assert np.array_equal(eng.path(), o.path), tag
        assert (st["live_ptr"], st["ref_ptr"], st["windows"]) == (so["live_ptr"], so["ref_ptr"], o.counters["windows"]), tag
        assert (st["status"] != 0) == (so["status"] != 0), tag
        eng.close()


def test_wtw_argument_errors():
    from real_time_audio_sync_amd import _native as nat
    from real_time_audio_sync_amd.wtw import BatchedWTW
    ref = torch.zeros((50, 12), dtype=torch.float64, device="cuda:0")
    with pytest.raises(nat.RtsyncError):
        BatchedWTW(ref, 20, 0)      # dtw_hop_size < hop_size: the reference would loop forever
    with pytest.raises(nat.RtsyncError):
        BatchedWTW(ref, 16385, 10)  # beyond the supported window


def test_known_answer_through_path_file_and_metric(chopin_audio, tmp_path):
    """SURVEY 8(f1)+(f2) on device output: the WTW path the HIP kernels produce for the chopin pair, written with
    pathfile.write_path_file exactly like test_simple.py:183-185 writes it, is byte-identical to the reference's
    Songs/chopin/tests/wtw_test_20b.txt; read back (tests.py:20-27) and scored with the accuracy metric
    (tests.py:29-137) it gives the numbers the reference's own test_simple class printed for that file
    (tests/golden/eval_golden.json, made by executing that class)."""
    import json
    import os
    from conftest import GOLDEN
    from real_time_audio_sync_amd import evaluate, pathfile
    from real_time_audio_sync_amd.wtw import WTW
    wtw = WTW.from_samples(chopin_audio["ref"], PARAMS, {'chroma': False})
    for buf in np.array_split(chopin_audio["live"], 4096):
        if wtw.insert(buf.tolist()) == "stop":
            break
    f = tmp_path / "wtw_test.txt"
    pathfile.write_path_file(str(f), wtw.path)
    assert f.read_bytes() == open(os.path.join(GOLDEN, "wtw_test_20b.txt"), "rb").read()
    path = pathfile.read_path_file(str(f), header_lines=0)
    ev = evaluate.AlignmentError(os.path.join(GOLDEN, "chopin_rubinstein_20b.csv"),
                                 os.path.join(GOLDEN, "chopin_rachmaninoff_20b.csv"), path)
    gold = json.load(open(os.path.join(GOLDEN, "eval_golden.json")))["wtw_known_answer"]
    st = ev.stats()
    lines = [st["pct_off_beats"][t] for t in (1, 3, 5, 10)] + [st["pct_off_seconds"][t] for t in (1, 3, 5, 10)]
    assert lines == gold["percent_lines"]          # exact float equality
    assert ev.get_error(verbose=False) == gold["returned"]
