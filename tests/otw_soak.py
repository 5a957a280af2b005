#!/usr/bin/env python3
"""Soak test (GPU box; the pytest suite runs a shortened form, tests/test_otw_gpu.py): many seeded OTW / LiveNote / LiveNoteV2 configurations with the
band widths that use the 128-, 256-, 512- and 1024-cell windows, run through the library's default (pipelined) kernel and
compared bit for bit with the dense CPU oracle -- path, end state, both bands.

    python3 tests/otw_soak.py [n_trials] [seed]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n_trials=120, seed=7, verbose=True):
    import torch
    import oracle
    from real_time_audio_sync_amd import otw_batch as ob, synth
    rs = np.random.RandomState(seed)
    vmap = {"otw": oracle.OTW, "livenote": oracle.LIVENOTE, "livenote_v2": oracle.LIVENOTE_V2}
    t0 = time.time()
    checked = 0
    for trial in range(n_trials):
        c = int(rs.choice([53, 60, 64, 100, 116, 117, 130, 200, 244, 245, 300, 400, 500, 500, 501, 640, 1012, 1013, 1500, 2036]))
        n_ref = int(rs.choice([c // 2 + 3, c + 1, c + 40, 2 * c, 3 * c]))
        n_ref = min(n_ref, 1400 if c <= 1012 else 2600)
        mrc = int(rs.choice([1, 2, 3, 5]))
        variant = str(rs.choice(["otw", "otw", "livenote", "livenote_v2"]))
        euclid = bool(variant == "livenote_v2" and rs.rand() < 0.4)
        mode = "set_live" if rs.rand() < 0.25 else "insert"
        f32 = bool(rs.rand() < 0.5) and not euclid
        batch = int(rs.choice([1, 2, 4]))
        if rs.rand() < 0.15:
            ref, base_live = synth.synth_tie(n_ref, seed=seed * 1000 + trial)
            n_ref = ref.shape[1]
        else:
            ref = synth.synth_ref(n_ref, seed=seed * 1000 + trial)
            base_live = None
        lives = []
        for b in range(batch):
            if base_live is not None:
                lv = base_live
            else:
                lv = synth.synth_live(ref, seed=seed * 100000 + 100 * trial + b, lo=float(rs.uniform(0.4, 1.0)),
                                      hi=float(rs.uniform(1.0, 2.2)))
                if lv.shape[1] == 0:
                    lv = ref[:, :1].copy()
            extra = int(rs.choice([0, 0, 0, 7, n_ref]))  # run past the reference end now and then
            if extra:
                lv = np.concatenate([lv, np.repeat(lv[:, -1:], extra, axis=1)], axis=1)
                lv = synth._as_f32_values(lv + 1e-3 * rs.rand(*lv.shape))
            if euclid:
                lv = synth._as_f32_values(np.abs(lv - 0.2))
            lives.append(lv)
        refx = synth._as_f32_values(np.abs(ref - 0.2)) if euclid else ref
        tdt = torch.float32 if f32 else torch.float64
        eng = ob.BatchedOTW(refx, c, mrc, batch=batch, variant=variant, euclid=euclid, dtype=tdt)
        lvd, lnd = eng.pack(lives, dtype=tdt)
        eng.run(lvd, lnd, mode=mode)
        for b, lv in enumerate(lives):
            o = oracle.OtwOracle(refx, c, mrc, vmap[variant], oracle.COST_EUCLID if euclid else oracle.COST_DOT)
            if mode == "set_live":
                o.set_live(lv)
            else:
                o.run(lv)
            tag = (trial, b, n_ref, c, mrc, variant, euclid, mode, f32, lv.shape[1])
            st, so = eng.state(b), o.state
            assert np.array_equal(eng.path(b), o.path), ("path", tag)
            for key in ("t", "j", "previous", "run_count", "status"):
                assert st[key] == so[key], (key, tag)
            if mode == "insert":
                assert st["direction"] == so["direction"], ("direction", tag)
                cnt = o.counters
                assert (st["cells"], st["row_strips"], st["col_strips"]) == (cnt["cells"], cnt["row_strips"], cnt["col_strips"]), ("counters", tag)
                rb, cb = eng.bands(b)
                orb, ocb = o.bands()
                assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), ("bands", tag)
            checked += 1
        eng.close()
        if verbose and trial % 100 == 99:
            print("trial %d, %d streams checked, %.0f s" % (trial + 1, checked, time.time() - t0), flush=True)
    if verbose:
        print("soak ok: %d configurations, %d streams bit-exact vs oracle (%.0f s)" % (n_trials, checked, time.time() - t0))
    return checked


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
