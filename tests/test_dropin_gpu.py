"""The drop-in classes, used exactly like the reference's harnesses use theirs
(test_simple.py:101-162, tests.py:143-172), against the golden vectors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_otw_insert_loop_like_test_simple(otw_golden, capsys):
    from real_time_audio_sync_amd.otw_eran import OnlineTimeWarping
    g = otw_golden
    ref_seq, live_seq = g["G/ref"], g["G/live"]
    otw_params = {'c': 50, 'max_run_count': 3}
    otw2 = OnlineTimeWarping(ref_seq, otw_params)
    for i in range(live_seq.shape[1]):
        cont = otw2.insert(live_seq[:, i])
        if cont == "stop":
            break
    path = otw2.path
    assert isinstance(path, list) and isinstance(path[0], tuple)
    assert np.array_equal(np.array(path), g["G_otw_c50_insert/path"])
    assert (otw2.t, otw2.j) == (int(g["G_otw_c50_insert/t"]), int(g["G_otw_c50_insert/j"]))
    assert otw2.direction in ("Both", "Row", "Column")


def test_otw_stop_and_message(otw_golden, capsys):
    from real_time_audio_sync_amd.otw_eran import OnlineTimeWarping
    g = otw_golden
    o = OnlineTimeWarping(g["F/ref"].astype(np.float64), {'c': 20, 'max_run_count': 3})
    live = g["F/live"].astype(np.float64)
    n = 0
    for i in range(live.shape[1]):
        n += 1
        if o.insert(live[:, i]) == "stop":
            break
    assert n == int(g["F_otw_stop_c20_insert/consumed"])
    assert "Ran out of ref-sequence" in capsys.readouterr().out
    assert o.insert(live[:, 0]) == "stop"  # sticky
    assert np.array_equal(np.array(o.path), g["F_otw_stop_c20_insert/path"])


def test_otw_set_live_returns_array(otw_golden):
    from real_time_audio_sync_amd.otw_eran import OnlineTimeWarping
    g = otw_golden
    o = OnlineTimeWarping(g["A/ref"].astype(np.float64), {'c': 10, 'max_run_count': 3})
    o.set_live(g["A/live"].astype(np.float64))
    assert isinstance(o.path, np.ndarray) and np.array_equal(o.path, g["A_otw_c10_set_live/path"])


def test_livenote_and_v2(otw_golden):
    from real_time_audio_sync_amd.livenote import LiveNote
    from real_time_audio_sync_amd.livenote_v2 import LiveNoteV2
    g = otw_golden
    params = {'search_band_width': 50, 'max_run_count': 3}
    debug_params = {'seq': False, 'all': False}
    ref, live = g["A/ref"].astype(np.float64), g["A/live"].astype(np.float64)
    ln = LiveNote(ref, params, debug_params)
    for i in range(live.shape[1]):
        if ln.insert(live[:, i]) == "stop":
            break
    assert np.array_equal(np.array(ln.path), g["A_livenote_c50_insert/path"])
    assert (ln.live_ptr, ln.ref_ptr) == (int(g["A_livenote_c50_insert/t"]), int(g["A_livenote_c50_insert/j"]))
    assert ln.direction in ("both", "row", "column")
    v2 = LiveNoteV2(g["C/ref"].astype(np.float64), params, debug_params, chroma_diff=True)  # tests.py:156
    lc = g["C/live"].astype(np.float64)
    for i in range(lc.shape[1]):
        if v2.insert(lc[:, i]) == "stop":
            break
    assert np.array_equal(np.array(v2.path), g["C_livenote_v2_euclid_c50_insert/path"])


def test_dense_matrices_hash_equal_to_reference(otw_golden):
    """.acc_cost / .cost, the dense (2N x N) float64 matrices the reference keeps (otw_eran.py:23,27),
    as written by the HIP kernel: sha256 of the whole matrices equals the sha256 of the matrices the
    reference's own code produced (tests/golden/make_golden.py)."""
    import hashlib
    from conftest import parse_case
    from real_time_audio_sync_amd.otw_eran import OnlineTimeWarping
    from real_time_audio_sync_amd.livenote import LiveNote
    from real_time_audio_sync_amd.livenote_v2 import LiveNoteV2
    g = otw_golden

    def sha(a):
        return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()

    for cid in ("A_otw_c50_insert", "B_otw_c500_insert", "D_otw_tie_c10_insert", "F_otw_stop_c20_insert",
                "E_livenote_overflow_c10_insert", "C_livenote_v2_euclid_c50_insert", "G_otw_c500_insert",
                "A_livenote_v2_c10_set_live"):
        case = parse_case([m for m in g["cases"] if str(m).split("|")[0] == cid][0])
        ref = g[case["group"] + "/ref"].astype(np.float64)
        live = g[case["group"] + "/live"].astype(np.float64)
        if case["variant"] == "otw":
            o = OnlineTimeWarping(ref, {'c': case["c"], 'max_run_count': case["mrc"]})
        elif case["variant"] == "livenote":
            o = LiveNote(ref, {'search_band_width': case["c"], 'max_run_count': case["mrc"]}, {})
        else:
            o = LiveNoteV2(ref, {'search_band_width': case["c"], 'max_run_count': case["mrc"]}, {},
                           chroma_diff=case["euclid"])
        if case["mode"] == "set_live":
            o.set_live(live)
        else:
            lv, ln = o._eng.pack([live], dtype=torch.float64)
            o._eng.run(lv, ln)           # same as the insert loop, one launch
        acc, cost = o.acc_cost, o.cost
        assert acc.shape == (2 * ref.shape[1], ref.shape[1])
        assert sha(acc) == str(g[cid + "/acc_sha"]), cid
        assert sha(cost) == str(g[cid + "/cost_sha"]), cid
        assert int((cost != -1).sum()) == int(g[cid + "/cells"]), cid


def test_dense_matrices_refused_when_huge(monkeypatch):
    from real_time_audio_sync_amd import _dropin
    from real_time_audio_sync_amd.otw_eran import OnlineTimeWarping
    from real_time_audio_sync_amd import synth
    monkeypatch.setattr(_dropin.OtwDropIn, "DENSE_LIMIT_BYTES", 1000)
    o = OnlineTimeWarping(synth.synth_ref(64, seed=2), {'c': 10, 'max_run_count': 3})
    with pytest.raises(NotImplementedError):
        o.acc_cost
