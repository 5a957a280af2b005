#!/usr/bin/env python3
"""Generate tests/golden/*.npz by executing the REFERENCE's own code (build container only).

/root/reference does not exist on the GPU box and never travels; this script is run once here and
its small outputs are committed.  What it does:

  * reads otw_eran.py / livenote.py / livenote_v2.py / dtw.py / wtw.py as text, applies lib2to3's
    ``fix_print`` in memory (the files are Python 2: ``print`` statements), sets ``np.int = int``
    (dtw.py:17 uses the alias numpy removed) and exec()s the result into throw-away modules.
    For wtw.py only the class body's numpy-only methods are used (get_cost_matrix / run_dtw /
    find_path); its constructor and insert() need librosa, which is absent from this image and is
    NOT stood in for -- the chroma -> WTW chain is pinned by the reference's own known-answer file
    instead (wtw_test_20b.txt, copied next to this script as data).
  * runs those classes/functions on seeded synthetic chroma (real_time_audio_sync_amd.synth) and
    on the chroma of the two WAVs the checkout still holds (computed by oracle/chroma_oracle.py,
    since chroma.py itself needs librosa; recorded as *inputs* of the fixture),
  * stores inputs + paths + end state + the two live accumulated-cost bands + sha256 of the dense
    float64 acc_cost (and cost) matrices,
  * cuts ``create_stft`` out of chroma.py as text (chroma.py:44-65 is pure numpy; the module itself
    cannot be imported: IPython / librosa / pyaudio are absent), rewrites its three ``/`` -- all with
    int operands, Python 2 floor division -- to ``//``, and runs it on the two chopin WAVs:
    stft_golden.npz holds a handful of STFT columns and the sha256 of the whole complex matrix.

Usage:  python tests/golden/make_golden.py      (takes ~5 minutes; reference is pure Python)
"""
import contextlib
import hashlib
import io
import os
import shutil
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from real_time_audio_sync_amd import synth  # noqa: E402
from oracle import chroma_oracle  # noqa: E402


def load_reference_module(name, drop_imports=()):
    from lib2to3 import refactor
    src = open(os.path.join(REF, name + ".py")).read()
    if not src.endswith("\n"):
        src += "\n"
    if drop_imports:  # wtw.py imports matplotlib/IPython/librosa/pyaudio at module level
        keep = []
        for line in src.splitlines():
            s = line.strip()
            if (s.startswith("import ") or s.startswith("from ") or s.startswith("plt.")) and any(
                    d in s for d in drop_imports):
                continue
            keep.append(line)
        src = "\n".join(keep) + "\n"
    tool = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
    src3 = str(tool.refactor_string(src, name))
    mod = types.ModuleType("reference_" + name)
    exec(compile(src3, os.path.join(REF, name + ".py"), "exec"), mod.__dict__)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


DIRS = {None: -1, "Both": 0, "both": 0, "Row": 1, "row": 1, "Column": 2, "column": 2}


def bands(acc, t, j, c):
    rb = np.full(c + 1, np.nan)
    cb = np.full(c + 1, np.nan)
    t = min(t, acc.shape[0] - 1)
    j = min(j, acc.shape[1] - 1)
    for i in range(c + 1):
        y, x = j - c + i, t - c + i
        if y >= 0:
            rb[i] = acc[t, y]
        if x >= 0:
            cb[i] = acc[x, j]
    return rb, cb


def run_otw_like(mods, variant, ref, live, c, mrc, mode, euclid=False):
    """variant: 'otw' | 'livenote' | 'livenote_v2'.  Returns dict of outputs."""
    with contextlib.redirect_stdout(io.StringIO()):
        if variant == "otw":
            o = mods["otw_eran"].OnlineTimeWarping(ref, {"c": c, "max_run_count": mrc})
        elif variant == "livenote":
            o = mods["livenote"].LiveNote(ref, {"search_band_width": c, "max_run_count": mrc}, {})
        else:
            o = mods["livenote_v2"].LiveNoteV2(ref, {"search_band_width": c, "max_run_count": mrc}, {},
                                               chroma_diff=euclid)
        consumed = 0
        rets = []
        if mode == "set_live":
            o.set_live(live)
        else:
            for i in range(live.shape[1]):
                r = o.insert(live[:, i])
                consumed += 1
                if r == "stop":
                    rets.append(i)
                    break
    if variant == "otw":
        t, j = o.t, o.j
    else:
        t, j = o.live_ptr, o.ref_ptr
    path = np.array(o.path, dtype=np.int64).reshape(-1, 2)
    rb, cb = bands(o.acc_cost, t, j, c)
    return dict(path=path.astype(np.int32), t=t, j=j, direction=DIRS[o.direction],
                previous=DIRS[o.previous], run_count=o.run_count, consumed=consumed,
                stopped=int(bool(rets)), row_band=rb, col_band=cb, acc_sha=sha(o.acc_cost),
                cost_sha=sha(o.cost), cells=int((o.cost != -1).sum()))


def make_stft_golden():
    """chroma.py:44-65 executed from its own text -> stft_golden.npz (direct pin of create_stft)."""
    src = open(os.path.join(REF, "chroma.py")).read()
    a = src.index("def create_stft(wav):")
    b = src.index("def create_chroma(", a)
    fn_src = src[a:b]
    assert fn_src.count("/") == 3, "chroma.py:49,53,54 are the only divisions expected"
    fn_src = fn_src.replace("/", "//")
    ns = {"np": np, "fft_len": 4096, "hop_size": 2048}  # chroma.py:20-21
    exec(compile(fn_src, os.path.join(REF, "chroma.py"), "exec"), ns)
    st = {}
    for key, fn in (("ref", "chopin_rubinstein_20b.wav"), ("live", "chopin_rachmaninoff_20b.wav")):
        wav, fs = chroma_oracle.load_wav_mono(os.path.join(REF, "Songs/chopin", fn))  # what librosa.load returns
        assert fs == 22050 and wav.dtype == np.float32
        ft = ns["create_stft"](wav)
        cols = np.array(sorted({0, 1, 2, 47, 100, 200, ft.shape[1] - 2, ft.shape[1] - 1}))
        st[key + "/shape"] = np.array(ft.shape)
        st[key + "/cols"] = cols
        st[key + "/stft_cols"] = np.ascontiguousarray(ft[:, cols])
        st[key + "/sha"] = np.asarray(sha(ft))
        st[key + "/power_sum"] = (np.abs(ft) ** 2).sum(axis=0)  # one float64 per frame: a whole-matrix value check
        print("%-28s create_stft -> %s sha %s" % ("stft_" + key, ft.shape, sha(ft)[:16]))
    np.savez_compressed(os.path.join(HERE, "stft_golden.npz"), **st)


def main():
    np.int = int  # dtw.py:17
    if "--only-stft" in sys.argv:
        make_stft_golden()
        return
    make_stft_golden()
    mods = {n: load_reference_module(n) for n in ("otw_eran", "livenote", "livenote_v2", "dtw")}
    mods["wtw"] = load_reference_module(
        "wtw", drop_imports=("matplotlib", "IPython", "librosa", "pyaudio", "plt.rcParams"))

    out = {}
    meta = []

    def add_case(cid, variant, ref, live, c, mrc, mode, euclid=False):
        r = run_otw_like(mods, variant, ref, live, c, mrc, mode, euclid)
        meta.append("|".join([cid, variant, str(c), str(mrc), mode, str(int(euclid))]))
        for k, v in r.items():
            out[cid + "/" + k] = np.asarray(v)
        print("%-28s path %5d  t=%d j=%d cells=%d stop=%d" % (cid, len(r["path"]), r["t"], r["j"],
                                                             r["cells"], r["stopped"]))

    # ---- A: small synthetic, all variants / modes ------------------------------------------
    refA = synth.synth_ref(300, seed=11)
    liveA = synth.synth_live(refA, seed=12)
    out["A/ref"], out["A/live"] = refA.astype(np.float32), liveA.astype(np.float32)
    for variant in ("otw", "livenote", "livenote_v2"):
        for c in (10, 50):
            for mode in ("insert", "set_live"):
                add_case("A_%s_c%d_%s" % (variant, c, mode), variant, refA, liveA, c, 3, mode)
    add_case("A_otw_c50_mrc2_insert", "otw", refA, liveA, 50, 2, "insert")
    add_case("A_otw_c64_mrc5_insert", "otw", refA, liveA, 64, 5, "insert")

    # ---- B: c = 500 past the warm-up (t > c) -----------------------------------------------
    refB = synth.synth_ref(900, seed=21)
    liveB = synth.synth_live(refB, seed=22)
    out["B/ref"], out["B/live"] = refB.astype(np.float32), liveB.astype(np.float32)
    add_case("B_otw_c500_insert", "otw", refB, liveB, 500, 3, "insert")
    add_case("B_livenote_v2_c500_insert", "livenote_v2", refB, liveB, 500, 3, "insert")

    # ---- C: LiveNoteV2 with the Euclidean chroma-diff cost (tests.py:146-148,156) ----------
    refC = np.clip(np.diff(refA), 0, np.inf)
    liveC = np.clip(np.diff(liveA), 0, np.inf)
    refC = refC.astype(np.float32).astype(np.float64)
    liveC = liveC.astype(np.float32).astype(np.float64)
    out["C/ref"], out["C/live"] = refC.astype(np.float32), liveC.astype(np.float32)
    add_case("C_livenote_v2_euclid_c50_insert", "livenote_v2", refC, liveC, 50, 3, "insert", euclid=True)
    add_case("C_livenote_v2_euclid_c10_set_live", "livenote_v2", refC, liveC, 10, 3, "set_live", euclid=True)

    # ---- D: ties (exactly repeated frames) --------------------------------------------------
    refD, liveD = synth.synth_tie(120, seed=5)
    out["D/ref"], out["D/live"] = refD.astype(np.float32), liveD.astype(np.float32)
    for variant in ("otw", "livenote", "livenote_v2"):
        add_case("D_%s_tie_c10_insert" % variant, variant, refD, liveD, 10, 3, "insert")

    # ---- E: live overflow (2N rows exhausted before the reference ends) ---------------------
    refE = synth.synth_ref(40, seed=31)
    liveE = synth.synth_live(refE, seed=32, lo=0.25, hi=0.4)
    liveE = np.concatenate([liveE, liveE[:, -1:].repeat(40, axis=1)], axis=1)
    liveE = synth._as_f32_values(liveE + 0.01 * np.random.RandomState(33).rand(*liveE.shape))
    out["E/ref"], out["E/live"] = refE.astype(np.float32), liveE.astype(np.float32)
    add_case("E_otw_overflow_c10_insert", "otw", refE, liveE, 10, 3, "insert")
    add_case("E_livenote_overflow_c10_insert", "livenote", refE, liveE, 10, 3, "insert")

    # ---- F: reference exhausted early ("stop") ----------------------------------------------
    refF = synth.synth_ref(80, seed=41)
    liveF = synth.synth_live(refF, seed=42, lo=1.5, hi=2.0)
    liveF = np.concatenate([liveF, synth.synth_ref(60, seed=43)], axis=1)
    out["F/ref"], out["F/live"] = refF.astype(np.float32), liveF.astype(np.float32)
    add_case("F_otw_stop_c20_insert", "otw", refF, liveF, 20, 3, "insert")
    add_case("F_otw_stop_c20_set_live", "otw", refF, liveF, 20, 3, "set_live")

    # ---- H: band wider than 500 cells (the 1024-cell window of the kernel) ------------------------
    refH = synth.synth_ref(1300, seed=81)
    liveH = synth.synth_live(refH, seed=82)
    out["H/ref"], out["H/live"] = refH.astype(np.float32), liveH.astype(np.float32)
    add_case("H_otw_c800_insert", "otw", refH, liveH, 800, 3, "insert")
    add_case("H_livenote_v2_c1000_insert", "livenote_v2", refH, liveH, 1000, 3, "insert")

    # ---- I: bands wider than 1012 cells (the 2048-cell window of the kernel; round 3) -------------------------
    refI = synth.synth_ref(2300, seed=91)
    liveI = synth.synth_live(refI, seed=92)
    out["I/ref"], out["I/live"] = refI.astype(np.float32), liveI.astype(np.float32)
    add_case("I_otw_c1500_insert", "otw", refI, liveI, 1500, 3, "insert")
    add_case("I_livenote_v2_c2000_set_live", "livenote_v2", refI, liveI, 2000, 3, "set_live")

    # ---- chopin pair (real audio; chroma from the numpy oracle -- chroma.py needs librosa) ---
    wav_r, _ = chroma_oracle.load_wav_mono(os.path.join(REF, "Songs/chopin/chopin_rubinstein_20b.wav"))
    wav_l, _ = chroma_oracle.load_wav_mono(os.path.join(REF, "Songs/chopin/chopin_rachmaninoff_20b.wav"))
    refG = chroma_oracle.wav_to_chroma(wav_r)
    liveG = chroma_oracle.wav_to_chroma(wav_l)
    out["G/ref"], out["G/live"] = refG, liveG  # float64: real chroma is not float32-exact
    add_case("G_otw_c50_insert", "otw", refG, liveG, 50, 3, "insert")
    add_case("G_otw_c500_insert", "otw", refG, liveG, 500, 3, "insert")
    add_case("G_livenote_v2_c50_insert", "livenote_v2", refG, liveG, 50, 3, "insert")
    out["cases"] = np.array(meta)
    np.savez_compressed(os.path.join(HERE, "otw_golden.npz"), **out)

    # ---- DTW --------------------------------------------------------------------------------
    dt = {}
    dmeta = []

    def add_dtw(cid, a, b, store_inputs=True):
        cost, acc, path = mods["dtw"].DTW(a, b)
        if store_inputs:
            dt[cid + "/a"], dt[cid + "/b"] = a, b
        dt[cid + "/path"] = path.astype(np.int32)
        dt[cid + "/acc_end"] = np.float64(acc[-1, -1])
        dt[cid + "/acc_sha"] = np.asarray(sha(acc))
        dt[cid + "/cost_sha"] = np.asarray(sha(cost))
        dt[cid + "/acc_diag"] = np.diag(acc).copy()
        dt[cid + "/acc_sub"] = acc[::9, ::7].copy()    # coarse samples + borders for a value check
        dt[cid + "/cost_sub"] = cost[::9, ::7].copy()
        dt[cid + "/acc_last_row"] = acc[-1].copy()
        dt[cid + "/acc_last_col"] = acc[:, -1].copy()
        dmeta.append(cid)
        print("%-28s path %5d acc_end %.6f" % (cid, len(path), acc[-1, -1]))

    r64 = synth.synth_ref(48, seed=51)
    l64 = synth.synth_live(r64, seed=52, max_frames=64)
    add_dtw("dtw_small", l64, r64, store_inputs=False)
    dt["dtw_small/a"], dt["dtw_small/b"] = l64.astype(np.float32), r64.astype(np.float32)
    r322 = synth.synth_ref(322, seed=61)
    l322 = synth.synth_live(r322, seed=62, max_frames=322)
    add_dtw("dtw_322", l322, r322, store_inputs=False)
    dt["dtw_322/a"], dt["dtw_322/b"] = l322.astype(np.float32), r322.astype(np.float32)
    add_dtw("dtw_chopin", liveG, refG, store_inputs=False)  # inputs live in otw_golden.npz (G/)
    td_r, td_l = synth.synth_tie(40, seed=7)
    add_dtw("dtw_tie", td_l, td_r, store_inputs=False)
    dt["dtw_tie/a"], dt["dtw_tie/b"] = td_l.astype(np.float32), td_r.astype(np.float32)
    dt["cases"] = np.array(dmeta)
    np.savez_compressed(os.path.join(HERE, "dtw_golden.npz"), **dt)

    # ---- WTW window functions (wtw.py:162-240), called unbound: they never touch self --------
    W = mods["wtw"].WTW
    wt = {}
    wmeta = []
    rs = np.random.RandomState(71)
    for cid, (n, m, silent) in {"win20": (20, 20, False), "win20x13": (20, 13, False),
                                "win16_silent": (16, 16, True), "win100": (100, 100, False)}.items():
        r = synth.synth_ref(max(n, m) + 5, seed=rs.randint(1000))
        x = synth.synth_live(r, seed=rs.randint(1000), max_frames=n)[:, :n]
        # WTW windows hold un-rounded float64 chroma scaled arbitrarily (cosine cost renormalises)
        x = x * rs.uniform(0.5, 2.0, size=(1, x.shape[1]))
        y = r[:, :m].copy()
        if silent:
            x[:, 5] = 0.0  # silent frame -> 0/0 = NaN in wtw.py:169
        with np.errstate(all="ignore"), contextlib.redirect_stdout(io.StringIO()):
            C = W.get_cost_matrix(None, x, y)
            D, B = W.run_dtw(None, C)
            sub = np.array(W.find_path(None, B), dtype=np.int32)
        wt[cid + "/x"], wt[cid + "/y"] = x, y
        wt[cid + "/C"], wt[cid + "/D"], wt[cid + "/B"], wt[cid + "/sub"] = C, D, B.astype(np.int8), sub
        wmeta.append(cid)
        print("%-28s sub-path %d  D_end %r" % (cid, len(sub), D[-1, -1]))
    wt["cases"] = np.array(wmeta)
    np.savez_compressed(os.path.join(HERE, "wtw_window_golden.npz"), **wt)

    # ---- the reference's own known-answer file + the audio it was made from ------------------
    shutil.copyfile(os.path.join(REF, "Songs/chopin/tests/wtw_test_20b.txt"),
                    os.path.join(HERE, "wtw_test_20b.txt"))
    aud = {}
    for key, fn in (("ref", "chopin_rubinstein_20b.wav"), ("live", "chopin_rachmaninoff_20b.wav")):
        import wave
        with wave.open(os.path.join(REF, "Songs/chopin", fn), "rb") as w:
            raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, w.getnchannels())
        # mono float32 sample = (L + R) / 65536 exactly (librosa.load semantics); store L + R
        aud[key + "_lr_sum"] = raw.astype(np.int32).sum(axis=1).astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "chopin_20b_audio.npz"), **aud)

    # ---- accuracy metric (tests.py:29-137): the reference's test_simple class, cut out of tests.py as
    # text (the module itself runs a whole evaluation at import, tests.py:278-283), on the chopin
    # pair's ground-truth CSVs -- which are copied here as data -- for the WTW known-answer path and
    # the DTW golden path.
    import json
    src = open(os.path.join(REF, "tests.py")).read()
    a = src.index("class test_simple():")
    b = src.index("params = {'search_band_width': 50", a)
    from lib2to3 import refactor
    tool = refactor.RefactoringTool(["lib2to3.fixes.fix_print"])
    cls_src = str(tool.refactor_string("import csv\n" + src[a:b] + "\n", "tests_class"))
    ns = {}
    exec(compile(cls_src, "tests.py:test_simple", "exec"), ns)
    for fn in ("chopin_rubinstein_20b.csv", "chopin_rachmaninoff_20b.csv"):
        shutil.copyfile(os.path.join(REF, "Songs/chopin", fn), os.path.join(HERE, fn))
    kat = np.loadtxt(os.path.join(HERE, "wtw_test_20b.txt"), dtype=np.int64)
    ev = {}
    shifted = kat + np.array([[0, 25]])   # a deliberately bad alignment so the counters are exercised
    drift = np.stack([kat[:, 0], (kat[:, 1] * 0.8).astype(np.int64)], axis=1)
    ev_paths = {"shifted": shifted.tolist(), "drift": drift.tolist()}
    for name, path in (("wtw_known_answer", kat), ("dtw_chopin", dt["dtw_chopin/path"]), ("shifted", shifted),
                       ("drift", drift)):
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            t = ns["test_simple"](os.path.join(REF, "Songs/chopin/chopin_rubinstein_20b.wav"),
                                  os.path.join(REF, "Songs/chopin/chopin_rachmaninoff_20b.wav"),
                                  [(int(l), int(r)) for l, r in path])
            ret = t.get_error()
        pcts = [float(line.split(":")[1].strip().rstrip("%")) for line in buf.getvalue().splitlines()
                if line.startswith("Percent incorrect")]
        ev[name] = dict(returned=ret, percent_lines=pcts)
        print("%-28s get_error() -> %r  %s" % (name, ret, pcts))
    ev["_paths"] = ev_paths
    json.dump(ev, open(os.path.join(HERE, "eval_golden.json"), "w"))
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
