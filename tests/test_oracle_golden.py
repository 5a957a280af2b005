"""The CPU oracle (oracle/) against golden vectors produced by the reference's own code
(tests/golden/make_golden.py) and against the reference's known-answer file.  CPU only.

Bit-exactness claimed here: alignment-path indices, end state, the two live accumulated-cost bands
and the sha256 of the whole dense float64 acc_cost / cost matrices."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from oracle import chroma_oracle
from conftest import parse_case

VARIANT = {"otw": oracle.OTW, "livenote": oracle.LIVENOTE, "livenote_v2": oracle.LIVENOTE_V2}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_oracle(case, ref, live, keep_cost=True):
    o = oracle.OtwOracle(ref, case["c"], case["mrc"], VARIANT[case["variant"]],
                         oracle.COST_EUCLID if case["euclid"] else oracle.COST_DOT, keep_cost=keep_cost)
    consumed = 0
    if case["mode"] == "set_live":
        o.set_live(live)
    else:
        consumed = o.run(live)
    return o, consumed


def test_dot_orders_match_numpy():
    """The two BLAS summation orders the restatement assumes, against this numpy build."""
    rs = np.random.RandomState(3)
    A = rs.rand(12, 500)
    B = rs.rand(12, 500)
    G = np.dot(A.T[:50], B[:, :50])
    for i in range(500):
        assert np.dot(A[:, i], B[:, i]) == oracle.dot_strided(A[:, i], B[:, i])
    for i in range(50):
        for j in range(50):
            assert G[i, j] == oracle.dot_chain(A[:, i], B[:, j])
    for i in range(200):
        assert np.sqrt(np.sum((A[:, i] - B[:, i]) ** 2)) == oracle.euclid(A[:, i], B[:, i])


def test_otw_family_bit_exact(otw_golden):
    g = otw_golden
    for meta in g["cases"]:
        case = parse_case(meta)
        cid = case["cid"]
        ref = g[case["group"] + "/ref"].astype(np.float64)
        live = g[case["group"] + "/live"].astype(np.float64)
        o, consumed = run_oracle(case, ref, live)
        st = o.state
        assert np.array_equal(o.path, g[cid + "/path"]), cid
        assert (st["t"], st["j"]) == (int(g[cid + "/t"]), int(g[cid + "/j"])), cid
        assert st["direction"] == int(g[cid + "/direction"]), cid
        assert st["previous"] == int(g[cid + "/previous"]), cid
        assert st["run_count"] == int(g[cid + "/run_count"]), cid
        if case["mode"] == "insert":
            assert consumed == int(g[cid + "/consumed"]), cid
            assert (st["status"] == oracle.STOP_REF_END) == bool(g[cid + "/stopped"]), cid
        rb, cb = o.bands()
        assert np.array_equal(rb, g[cid + "/row_band"], equal_nan=True), cid
        assert np.array_equal(cb, g[cid + "/col_band"], equal_nan=True), cid
        assert sha(o.acc_cost()) == str(g[cid + "/acc_sha"]), cid
        assert sha(o.cost()) == str(g[cid + "/cost_sha"]), cid
        assert o.counters["cells"] >= int(g[cid + "/cells"]), cid  # re-evaluated cells count twice


def test_set_live_is_insert_loop_plus_origin(otw_golden):
    """set_live's path = (0,0)-decision + the insert-loop path (when neither mode is cut
    short by the other's end condition)."""
    g = otw_golden
    for v in ("otw", "livenote", "livenote_v2"):
        for c in (10, 50):
            a = g["A_%s_c%d_insert/path" % (v, c)]
            b = g["A_%s_c%d_set_live/path" % (v, c)]
            assert np.array_equal(b[1:], a) and tuple(b[0]) == (0, 0)


def test_overflow_and_stop_status(otw_golden):
    g = otw_golden
    case = parse_case([m for m in g["cases"] if str(m).startswith("E_otw")][0])
    o, _ = run_oracle(case, g["E/ref"].astype(np.float64), g["E/live"].astype(np.float64), keep_cost=False)
    assert o.state["status"] == oracle.LIVE_OVERFLOW
    case = parse_case([m for m in g["cases"] if str(m).startswith("F_otw_stop_c20_insert")][0])
    o, consumed = run_oracle(case, g["F/ref"].astype(np.float64), g["F/live"].astype(np.float64), keep_cost=False)
    assert o.state["status"] == oracle.STOP_REF_END and consumed < g["F/live"].shape[1]


def test_dtw_bit_exact(dtw_golden, otw_golden):
    g = dtw_golden
    for cid in g["cases"]:
        cid = str(cid)
        if cid == "dtw_chopin":
            a, b = otw_golden["G/live"], otw_golden["G/ref"]
        else:
            a, b = g[cid + "/a"].astype(np.float64), g[cid + "/b"].astype(np.float64)
        cost, acc, path, back = oracle.dtw(a, b)
        assert np.array_equal(path, g[cid + "/path"]), cid
        assert acc[-1, -1] == float(g[cid + "/acc_end"]), cid
        # dtw.py:11 is a dgemm; OpenBLAS rounds the remainder column of each thread's partition
        # differently (1 ulp, 46 of 99 820 elements at 310x322), so the reference's own cost matrix
        # is canonical only up to 1 ulp at sizes where dgemm threads.  Values: <= 1 ulp; hashes:
        # only where dgemm is single-block.
        for key, full in (("acc_sub", acc[::9, ::7]), ("cost_sub", cost[::9, ::7]),
                          ("acc_last_row", acc[-1]), ("acc_last_col", acc[:, -1]), ("acc_diag", np.diag(acc))):
            want = g[cid + "/" + key]
            # 1 ulp of the O(1) dot product is 2.2e-16 absolute in cost = 1 - dot
            tol = 2.3e-16 if key == "cost_sub" else 1e-14 * np.maximum(np.abs(want), 1.0)
            assert np.all(np.abs(full - want) <= tol), (cid, key)
        if max(acc.shape) <= 200:
            assert sha(cost) == str(g[cid + "/cost_sha"]), cid
            assert sha(acc) == str(g[cid + "/acc_sha"]), cid
    assert len(g["dtw_chopin/path"]) == 657 and abs(float(g["dtw_chopin/acc_end"]) - 63.029) < 1e-3


def test_wtw_window_functions_bit_exact(wtw_window_golden):
    g = wtw_window_golden
    for cid in g["cases"]:
        cid = str(cid)
        x, y = g[cid + "/x"], g[cid + "/y"]
        with np.errstate(all="ignore"):
            C = oracle.wtw_cost_matrix(x, y)
            D, B = oracle.wtw_run_dtw(C)
        sub = oracle.wtw_find_path(B)
        assert np.array_equal(C, g[cid + "/C"], equal_nan=True), cid
        assert np.array_equal(D, g[cid + "/D"], equal_nan=True), cid
        assert np.array_equal(B, g[cid + "/B"]), cid
        assert np.array_equal(sub, g[cid + "/sub"]), cid


def test_wtw_known_answer(chopin_audio, wtw_known_answer):
    """The reference's only reproducible known-answer test: Songs/chopin/tests/wtw_test_20b.txt,
    produced by test_simple.py:166-185 / tests.py:174-190 (W = 20 frames, hop = 10 frames,
    np.array_split(live, 4096) buffers).  Pins chroma + WTW end to end at path level."""
    params = {"fft_len": 4096, "hop_size": 2048, "dtw_win_size": 4096 * 10, "dtw_hop_size": 2048 * 10}
    w = chroma_oracle.WtwAudioOracle(chopin_audio["ref"], params)
    for buf in np.array_split(chopin_audio["live"], 4096):
        if w.insert(buf.tolist()) == "stop":
            break
    assert w.path.shape == wtw_known_answer.shape == (509, 2)
    assert np.array_equal(w.path, wtw_known_answer)


def test_chroma_oracle_shapes_and_norm(chopin_audio, otw_golden):
    ch = chroma_oracle.wav_to_chroma(chopin_audio["ref"])
    assert ch.shape == (12, 380)
    assert np.array_equal(ch, otw_golden["G/ref"])
    assert np.allclose(np.sqrt((ch ** 2).sum(axis=0)), 1.0, atol=1e-12) and (ch > 0).all()
    fb = chroma_oracle.chroma_filterbank()
    assert fb.shape == (12, 2049)
    col = chroma_oracle.wav_to_chroma_col(chopin_audio["ref"][10000:10000 + 4096])
    assert col.shape == (12, 1) or col.shape == (12,)
    d = chroma_oracle.wav_to_chroma_diff(chopin_audio["ref"][:60000])
    assert d.shape[0] == 12 and (d >= 0).all()


def test_create_stft_pinned_by_the_reference(chopin_audio):
    """chroma.py:44-65 was executed from its own text on the two chopin WAVs (tests/golden/make_golden.py::
    make_stft_golden); the numpy restatement in oracle/chroma_oracle.py reproduces its complex STFT bit for bit:
    sha256 of the whole (2049, M) matrix, the stored columns, the per-frame power sums."""
    import hashlib
    from conftest import GOLDEN
    from oracle import chroma_oracle as co
    g = np.load(os.path.join(GOLDEN, "stft_golden.npz"))
    for key in ("ref", "live"):
        ft = co.create_stft(chopin_audio[key])
        assert tuple(g[key + "/shape"]) == ft.shape and ft.dtype == np.complex128
        assert hashlib.sha256(np.ascontiguousarray(ft).tobytes()).hexdigest() == str(g[key + "/sha"]), key
        assert np.array_equal(ft[:, g[key + "/cols"]], g[key + "/stft_cols"])
        assert np.array_equal((np.abs(ft) ** 2).sum(axis=0), g[key + "/power_sum"])


def test_numpy_restatement_matches_reference_goldens(otw_golden):
    """oracle/otw_numpy.py (the "numpy otw_eran.py path" bench.py times as a cpu_baseline leg) against every
    OnlineTimeWarping insert-loop case the reference's own code produced: path, end state, both live bands and the
    sha256 of the dense float64 acc_cost / cost matrices, bit for bit."""
    from oracle import otw_numpy
    g = otw_golden
    done = 0
    for meta in g["cases"]:
        case = parse_case(meta)
        if case["variant"] != "otw" or case["mode"] != "insert" or case["c"] > 64:
            continue  # c = 500 cases take minutes in pure Python; the c <= 64 ones cover every branch
        cid = case["cid"]
        ref = g[case["group"] + "/ref"].astype(np.float64)
        live = g[case["group"] + "/live"].astype(np.float64)
        o = otw_numpy.NumpyOTW(ref, case["c"], case["mrc"])
        consumed = o.run(live)
        assert np.array_equal(np.array(o.path, dtype=np.int32).reshape(-1, 2), g[cid + "/path"]), cid
        assert (o.t, o.j, consumed) == (int(g[cid + "/t"]), int(g[cid + "/j"]), int(g[cid + "/consumed"])), cid
        assert (o.status == otw_numpy.STOP_REF_END) == bool(g[cid + "/stopped"]), cid
        assert o.run_count == int(g[cid + "/run_count"]), cid
        rb, cb = o.bands()
        assert np.array_equal(rb, g[cid + "/row_band"], equal_nan=True), cid
        assert np.array_equal(cb, g[cid + "/col_band"], equal_nan=True), cid
        assert sha(o.acc) == str(g[cid + "/acc_sha"]) and sha(o.cost) == str(g[cid + "/cost_sha"]), cid
        done += 1
    assert done >= 6
