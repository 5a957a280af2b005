"""HIP OTW / LiveNote / LiveNoteV2 kernels (through librtsync.so's C-ABI) against the golden vectors
the reference's own code produced and against the CPU oracle on seeded inputs.

Bar: alignment-path indices, end state and the two live accumulated-cost bands are BIT-EXACT
(float64 compared with ==), not merely within north_star's 1e-4."""
import numpy as np
import pytest

from conftest import parse_case

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from real_time_audio_sync_amd import _native, otw_batch, synth
    import oracle
    return dict(nat=_native, ob=otw_batch, synth=synth, oracle=oracle)


def _check_against_golden(g, case, eng, b=0, check_bands=True):
    cid = case["cid"]
    st = eng.state(b)
    assert np.array_equal(eng.path(b), g[cid + "/path"]), cid
    assert st["n_path"] == len(g[cid + "/path"]), cid
    assert (st["t"], st["j"]) == (int(g[cid + "/t"]), int(g[cid + "/j"])), cid
    assert st["direction"] == int(g[cid + "/direction"]), cid
    assert st["previous"] == int(g[cid + "/previous"]), cid
    assert st["run_count"] == int(g[cid + "/run_count"]), cid
    assert st["path_truncated"] == 0
    if case["mode"] == "insert":
        assert st["consumed"] == int(g[cid + "/consumed"]), cid
        assert (st["status"] == 1) == bool(g[cid + "/stopped"]), cid
        if check_bands:
            rb, cb = eng.bands(b)
            assert np.array_equal(rb, g[cid + "/row_band"], equal_nan=True), cid
            assert np.array_equal(cb, g[cid + "/col_band"], equal_nan=True), cid


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_golden_cases(gpu, otw_golden, dtype):
    g = otw_golden
    ob = gpu["ob"]
    recomputes = 0
    for meta in g["cases"]:
        case = parse_case(meta)
        if dtype == "f32" and case["group"] == "G":
            continue  # real chroma is not float32-exact
        ref = g[case["group"] + "/ref"].astype(np.float64)
        live = g[case["group"] + "/live"].astype(np.float64)
        tdt = torch.float64 if dtype == "f64" else torch.float32
        eng = ob.BatchedOTW(ref, case["c"], case["mrc"], batch=1, variant=case["variant"],
                            euclid=case["euclid"], dtype=tdt)
        lv, ln = eng.pack([live], dtype=tdt)
        eng.run(lv, ln, mode=case["mode"])
        _check_against_golden(g, case, eng)
        recomputes += eng.state(0)["band_recomputes"]
        eng.close()
    # the rare branch of the incremental band-minimum bookkeeping (minimum slid out of the window ->
    # full wave reduction) must actually have been taken by some of these cases
    assert recomputes > 0


@pytest.mark.parametrize("waves", [1, 2, 4, 8])
def test_wave_count_does_not_change_results(gpu, otw_golden, waves):
    g = otw_golden
    ob = gpu["ob"]
    for cid in ("A_otw_c50_insert", "B_otw_c500_insert", "D_otw_tie_c10_insert", "C_livenote_v2_euclid_c50_insert",
                "F_otw_stop_c20_insert", "E_otw_overflow_c10_insert", "A_livenote_v2_c10_set_live"):
        case = parse_case([m for m in g["cases"] if str(m).split("|")[0] == cid][0])
        ref = g[case["group"] + "/ref"].astype(np.float64)
        live = g[case["group"] + "/live"].astype(np.float64)
        eng = ob.BatchedOTW(ref, case["c"], case["mrc"], batch=1, variant=case["variant"], euclid=case["euclid"],
                            dtype=torch.float64, waves=waves)
        lv, ln = eng.pack([live])
        eng.run(lv, ln, mode=case["mode"])
        _check_against_golden(g, case, eng)
        eng.close()


@pytest.mark.parametrize("spec", ["0", "1"])
def test_plain_and_pipelined_kernels_agree(gpu, otw_golden, monkeypatch, spec):
    """The pipelined kernel (speculative strips, hit steps) is the default; RTS_OTW_SPEC=0 selects the plain one
    (chain phase, then control phase).  Both must reproduce the goldens and the oracle."""
    monkeypatch.setenv("RTS_OTW_SPEC", spec)
    g = otw_golden
    ob, synth, oracle = gpu["ob"], gpu["synth"], gpu["oracle"]
    for cid in ("A_otw_c50_insert", "B_otw_c500_insert", "D_otw_tie_c10_insert", "C_livenote_v2_euclid_c50_insert",
                "F_otw_stop_c20_insert", "E_otw_overflow_c10_insert", "A_livenote_v2_c10_set_live"):
        case = parse_case([m for m in g["cases"] if str(m).split("|")[0] == cid][0])
        ref = g[case["group"] + "/ref"].astype(np.float64)
        live = g[case["group"] + "/live"].astype(np.float64)
        eng = ob.BatchedOTW(ref, case["c"], case["mrc"], batch=1, variant=case["variant"], euclid=case["euclid"],
                            dtype=torch.float64)
        lv, ln = eng.pack([live])
        eng.run(lv, ln, mode=case["mode"])
        _check_against_golden(g, case, eng)
        eng.close()
    # band widths that use the 256- and 512-cell windows, past the warm-up, ragged batch
    for c, n_ref in ((200, 500), (500, 700)):
        ref, lives = synth.synth_batch(n_ref, 5, seed=900 + c)
        lives[2] = lives[2][:, :c + 37]
        eng = ob.BatchedOTW(ref, c, 3, batch=5, dtype=torch.float32)
        lv, ln = eng.pack(lives)
        eng.run(lv, ln)
        for b, live in enumerate(lives):
            o = oracle.OtwOracle(ref, c, 3)
            n = o.run(live)
            st, so = eng.state(b), o.state
            assert np.array_equal(eng.path(b), o.path), (spec, c, b)
            for k in ("t", "j", "direction", "previous", "run_count", "status"):
                assert st[k] == so[k], (spec, c, b, k)
            assert st["consumed"] == n
            cnt = o.counters
            assert (st["cells"], st["row_strips"], st["col_strips"]) == (cnt["cells"], cnt["row_strips"], cnt["col_strips"])
            rb, cb = eng.bands(b)
            orb, ocb = o.bands()
            assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), (spec, c, b)
        eng.close()


def test_residency_flavour_agrees(gpu, otw_golden, monkeypatch):
    """Batches of more than two streams per CU run a third flavour of the pipelined kernel (no live ring, 73 VGPRs,
    one cost cell in flight per helper thread, the chains of steps that are not hits on wave 2).  RTS_OTW_TP_FROM=0
    selects it at any batch size: goldens (every variant, both modes, tie and stop cases) and seeded batches against
    the oracle, float32 and float64 features, including the 1024-cell window's neighbour c = 500."""
    monkeypatch.setenv("RTS_OTW_TP_FROM", "0")
    g = otw_golden
    ob, synth, oracle = gpu["ob"], gpu["synth"], gpu["oracle"]
    n = 0
    for meta in g["cases"]:
        case = parse_case(meta)
        if case["c"] > 500:
            continue  # the 1024-cell window has its own no-ring flavour
        ref = g[case["group"] + "/ref"].astype(np.float64)
        live = g[case["group"] + "/live"].astype(np.float64)
        eng = ob.BatchedOTW(ref, case["c"], case["mrc"], batch=1, variant=case["variant"], euclid=case["euclid"],
                            dtype=torch.float64)
        lv, ln = eng.pack([live])
        eng.run(lv, ln, mode=case["mode"])
        _check_against_golden(g, case, eng)
        eng.close()
        n += 1
    assert n >= 20
    for c, n_ref, dt in ((200, 500, torch.float32), (500, 900, torch.float32), (500, 700, torch.float64)):
        ref, lives = synth.synth_batch(n_ref, 6, seed=1200 + c)
        lives[2] = lives[2][:, :c + 37]
        eng = ob.BatchedOTW(ref, c, 3, batch=6, dtype=dt)
        lv, ln = eng.pack(lives)
        eng.run(lv, ln)
        for b, live in enumerate(lives):
            o = oracle.OtwOracle(ref, c, 3)
            o.run(live)
            assert np.array_equal(eng.path(b), o.path), (c, b)
            rb, cb = eng.bands(b)
            orb, ocb = o.bands()
            assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), (c, b)
        eng.close()


def test_batch_vs_oracle_c500(gpu):
    """Config-3 shaped, scaled down so the dense oracle stays small: 16 different warps of one
    reference, c=500, past the warm-up."""
    ob, synth, oracle = gpu["ob"], gpu["synth"], gpu["oracle"]
    ref, lives = synth.synth_batch(800, 16, seed=300)
    lives[3] = lives[3][:, :417]   # ragged lengths
    lives[7] = lives[7][:, :1]     # a single frame: first insert only
    eng = ob.BatchedOTW(ref, 500, 3, batch=16, dtype=torch.float32)
    lv, ln = eng.pack(lives)
    eng.run(lv, ln)
    for b, live in enumerate(lives):
        o = oracle.OtwOracle(ref, 500, 3)
        n = o.run(live)
        st, so = eng.state(b), o.state
        assert np.array_equal(eng.path(b), o.path), b
        for k in ("t", "j", "direction", "previous", "run_count", "status"):
            assert st[k] == so[k], (b, k)
        assert st["consumed"] == n
        cnt = o.counters
        assert (st["cells"], st["row_strips"], st["col_strips"]) == (cnt["cells"], cnt["row_strips"], cnt["col_strips"])
        rb, cb = eng.bands(b)
        orb, ocb = o.bands()
        assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), b
    eng.close()


def test_empty_stream(gpu):
    ob, synth = gpu["ob"], gpu["synth"]
    ref, lives = synth.synth_batch(100, 2, seed=5)
    eng = ob.BatchedOTW(ref, 20, 3, batch=2, dtype=torch.float32)
    lv, ln = eng.pack(lives)
    ln[1] = 0
    eng.run(lv, ln)
    st = eng.state(1)
    assert st["first_insert"] == 1 and st["n_path"] == 0 and st["consumed"] == 0 and st["t"] == 0
    assert eng.state(0)["n_path"] > 0
    eng.close()


@pytest.mark.parametrize("variant,c", [("otw", 20), ("livenote_v2", 50), ("otw", 500)])
def test_insert_mode_matches_run(gpu, variant, c):
    """rts_otw_insert (one frame per launch, state persisted in HBM between launches) must equal
    the whole-sequence run, including after "stop" (sticky) and with streams of unequal length."""
    ob, synth, oracle = gpu["ob"], gpu["synth"], gpu["oracle"]
    n_ref = 120 if c < 500 else 560
    ref, lives = synth.synth_batch(n_ref, 3, seed=77)
    lives[1] = lives[1][:, : lives[1].shape[1] // 2]
    lives[2] = np.concatenate([lives[2], lives[2][:, -20:]], axis=1)  # runs past the reference end
    eng = ob.BatchedOTW(ref, c, 3, batch=3, variant=variant, dtype=torch.float64)
    tmax = max(l.shape[1] for l in lives)
    dev = eng.device
    for i in range(tmax):
        frames = torch.zeros((3, 12), dtype=torch.float64)
        active = torch.zeros(3, dtype=torch.uint8)
        for b, l in enumerate(lives):
            if i < l.shape[1]:
                frames[b] = torch.from_numpy(l[:, i].copy())
                active[b] = 1
        eng.insert(frames.to(dev), active.to(dev))
    vmap = {"otw": oracle.OTW, "livenote_v2": oracle.LIVENOTE_V2}
    for b, live in enumerate(lives):
        o = oracle.OtwOracle(ref, c, 3, vmap[variant])
        o.run(live)
        st, so = eng.state(b), o.state
        assert np.array_equal(eng.path(b), o.path), b
        for k in ("t", "j", "direction", "previous", "run_count", "status"):
            assert st[k] == so[k], (b, k)
        rb, cb = eng.bands(b)
        orb, ocb = o.bands()
        assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), b
    eng.close()


def test_full_size_properties(gpu):
    """BASELINE config 3 at full size (B=64, N=2200, c=500): the dense oracle is checked on 4 of
    the 64 streams; all streams are checked through size-independent properties of an OTW path."""
    ob, synth, oracle = gpu["ob"], gpu["synth"], gpu["oracle"]
    ref, lives = synth.synth_batch(2200, 64, seed=1000)
    eng = ob.BatchedOTW(ref, 500, 3, batch=64, dtype=torch.float32)
    lv, ln = eng.pack(lives)
    eng.run(lv, ln)
    states = eng.states()
    nat = gpu["nat"]
    for b in range(64):
        p = eng.path(b)
        st = eng.state(b)
        assert st["path_truncated"] == 0 and len(p) == st["n_path"]
        # every path point lies on the current row or current column of its decide() and inside the band
        assert (p[:, 0] >= 0).all() and (p[:, 1] >= 0).all() and (p[:, 0] <= st["t"]).all() and (p[:, 1] < 2200).all()
        # the frontier never moves backwards: max(live), max(ref) over prefixes are what decide() saw
        assert st["consumed"] == st["t"] + 1
        assert st["row_strips"] == st["t"]
        assert st["col_strips"] == st["j"] - (1 if st["status"] == nat.STOP_REF_END else 0)
        # synthetic warps stay within 0.8..1.25: the alignment must end near the diagonal's end
        assert abs(int(p[-1, 1]) - min(2199, int(round(p[-1, 0] * 2200.0 / lives[b].shape[1])))) < 120
    for b in (0, 21, 42, 63):
        o = oracle.OtwOracle(ref, 500, 3)
        o.run(lives[b])
        assert np.array_equal(eng.path(b), o.path), b
        rb, cb = eng.bands(b)
        orb, ocb = o.bands()
        assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), b
    eng.close()


def test_argument_errors(gpu):
    nat, ob, synth = gpu["nat"], gpu["ob"], gpu["synth"]
    ref = synth.synth_ref(50, seed=1)
    with pytest.raises(nat.RtsyncError):
        ob.BatchedOTW(ref, 2037, 3)   # band too wide for the LDS-resident kernel
    with pytest.raises(nat.RtsyncError):
        ob.BatchedOTW(ref, 0, 3)
    with pytest.raises(nat.RtsyncError):
        ob.BatchedOTW(ref[:11], 10, 3)  # not 12 chroma bins


def test_randomized_small_configs(gpu):
    """Seeded sweep over the corners the fixed cases do not reach: band widths down to c=1, references
    and live sequences of a few frames, every variant / cost / mode / wave count, exact ties.  Each
    configuration must match the dense CPU oracle bit for bit."""
    ob, synth, oracle = gpu["ob"], gpu["synth"], gpu["oracle"]
    rs = np.random.RandomState(20261004)
    vmap = {"otw": oracle.OTW, "livenote": oracle.LIVENOTE, "livenote_v2": oracle.LIVENOTE_V2}
    checked = 0
    for trial in range(90):
        n_ref = int(rs.choice([1, 2, 3, 5, 9, 17, 40, 90, 160]))
        c = int(rs.choice([1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 31, 33, 52, 63, 64, 65, 120]))
        mrc = int(rs.choice([1, 2, 3, 5]))
        variant = str(rs.choice(["otw", "livenote", "livenote_v2"]))
        euclid = bool(variant == "livenote_v2" and rs.rand() < 0.4)
        mode = "set_live" if rs.rand() < 0.3 else "insert"
        waves = int(rs.choice([1, 2, 4, 8]))
        batch = int(rs.choice([1, 3]))
        if rs.rand() < 0.25:
            ref, base_live = synth.synth_tie(max(n_ref, 2), seed=trial)
            n_ref = ref.shape[1]
        else:
            ref = synth.synth_ref(n_ref, seed=trial)
            base_live = None
        lives = []
        for b in range(batch):
            if base_live is not None:
                lv = base_live
            else:
                lv = synth.synth_live(ref, seed=1000 * trial + b, lo=float(rs.uniform(0.3, 1.0)), hi=float(rs.uniform(1.0, 2.5)))
                if lv.shape[1] == 0:
                    lv = ref[:, :1].copy()
            extra = int(rs.choice([0, 0, 1, 5, 3 * n_ref]))        # run past the reference end / into overflow
            if extra:
                lv = np.concatenate([lv, np.repeat(lv[:, -1:], extra, axis=1)], axis=1)
                lv = synth._as_f32_values(lv + 1e-3 * rs.rand(*lv.shape))
            if euclid:
                lv = synth._as_f32_values(np.abs(lv - 0.2))
            lives.append(lv)
        refx = synth._as_f32_values(np.abs(ref - 0.2)) if euclid else ref
        eng = ob.BatchedOTW(refx, c, mrc, batch=batch, variant=variant, euclid=euclid, dtype=torch.float64, waves=waves)
        lvd, lnd = eng.pack(lives, dtype=torch.float64)
        eng.run(lvd, lnd, mode=mode)
        for b, lv in enumerate(lives):
            o = oracle.OtwOracle(refx, c, mrc, vmap[variant], oracle.COST_EUCLID if euclid else oracle.COST_DOT)
            if mode == "set_live":
                o.set_live(lv)
            else:
                o.run(lv)
            tag = (trial, b, n_ref, c, mrc, variant, euclid, mode, waves, lv.shape[1])
            st, so = eng.state(b), o.state
            assert np.array_equal(eng.path(b), o.path), tag
            for key in ("t", "j", "previous", "run_count", "status"):
                assert st[key] == so[key], (tag, key)
            if mode == "insert":
                assert st["direction"] == so["direction"], tag
                rb, cb = eng.bands(b)
                orb, ocb = o.bands()
                assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), tag
            checked += 1
        eng.close()
    assert checked >= 90


def test_soak_medium_and_wide_bands(gpu):
    """tests/otw_soak.py, shortened: seeded configurations with band widths 53..1012 (the 128-, 256-, 512- and
    1024-cell windows), every variant / cost / mode, float32 and float64 features, runs past the reference end -- all
    bit-exact against the dense oracle.  (The long form is run by hand on the GPU box when the kernel changes: round 2's
    last runs were 1 500 configurations (3 475 streams) with the default kernel flavours and 1 000 (2 388 streams) with the
    residency flavour forced.)"""
    import otw_soak
    assert otw_soak.run(120, seed=31, verbose=False) >= 120


def test_bands_wider_than_500_cells(gpu):
    """500 < c <= 1012 run on the 1024-cell window, 1012 < c <= 2036 on the 2048-cell one: every variant against the
    dense oracle -- path, end state, both live bands -- plus the refusals the header documents.  (The reference-made
    goldens H_otw_c800 / H_livenote_v2_c1000 are covered by test_golden_cases.)"""
    import oracle
    from real_time_audio_sync_amd import _native as nat, synth
    ob = gpu["ob"]
    vmap = {"otw": oracle.OTW, "livenote": oracle.LIVENOTE, "livenote_v2": oracle.LIVENOTE_V2}
    checked = 0
    for c, n_ref, variant, mode in ((501, 700, "otw", "insert"), (640, 1500, "livenote", "set_live"),
                                    (900, 1000, "livenote_v2", "insert"), (1012, 1600, "otw", "insert"),
                                    (1012, 1100, "otw", "set_live"), (1013, 1500, "otw", "insert"),
                                    (1700, 2100, "livenote", "insert"), (2036, 2300, "livenote_v2", "set_live")):
        ref, lives = synth.synth_batch(n_ref, 3, seed=700 + c)
        lives[1] = lives[1][:, : max(5, lives[1].shape[1] // 2)]
        eng = ob.BatchedOTW(ref, c, 3, batch=3, variant=variant, dtype=torch.float32)
        lv, ln = eng.pack(lives)
        eng.run(lv, ln, mode=mode)
        for b, live in enumerate(lives):
            o = oracle.OtwOracle(ref, c, 3, vmap[variant])
            if mode == "set_live":
                o.set_live(live)
            else:
                o.run(live)
            st, so = eng.state(b), o.state
            tag = (c, variant, mode, b)
            assert np.array_equal(eng.path(b), o.path), tag
            assert (st["t"], st["j"], st["run_count"], st["direction"]) == (so["t"], so["j"], so["run_count"], so["direction"]), tag
            rb, cb = eng.bands(b)
            orb, ocb = o.bands()
            assert np.array_equal(rb, orb, equal_nan=True) and np.array_equal(cb, ocb, equal_nan=True), tag
            checked += 1
        eng.close()
    assert checked == 24
    # float64 features and the per-call ingestion paths take the same window (no live ring in LDS: its helper waves
    # read live frames from global memory), so the drop-in classes work at these widths too
    ref, lives = synth.synth_batch(900, 2, seed=5)
    eng = ob.BatchedOTW(ref, 600, 3, batch=2, dtype=torch.float64)
    lv, ln = eng.pack(lives, dtype=torch.float64)
    eng.run(lv, ln)
    for b, live in enumerate(lives):
        o = oracle.OtwOracle(ref, 600, 3, oracle.OTW)
        o.run(live)
        assert np.array_equal(eng.path(b), o.path), ("f64", b)
    eng.reset()
    n_push = 150
    for f in range(n_push):  # one frame per stream per call (OnlineTimeWarping.insert)
        eng.insert(torch.from_numpy(np.ascontiguousarray(np.stack([l[:, f] for l in lives]))).to("cuda:0"))
    for b, live in enumerate(lives):
        o = oracle.OtwOracle(ref, 600, 3, oracle.OTW)
        for f in range(n_push):
            o.insert(live[:, f])
        assert np.array_equal(eng.path(b), o.path), ("insert", b)
    eng.close()
    from real_time_audio_sync_amd import otw_eran as otw_mod
    drop = otw_mod.OnlineTimeWarping(ref, {"c": 700, "max_run_count": 3})
    o = oracle.OtwOracle(ref, 700, 3, oracle.OTW, keep_cost=True)
    for f in range(120):
        drop.insert(lives[0][:, f])
        o.insert(lives[0][:, f])
    assert np.array_equal(np.asarray(drop.path), o.path)
    # the dense (2N x N) matrices at these widths (otw_eran.py:23,27): the plain kernel without rings, against the oracle's
    assert np.array_equal(drop.acc_cost, o.acc_cost()) and np.array_equal(drop.cost, o.cost())
    for c, n_ref, variant, mode in ((1100, 1250, "livenote", "set_live"), (2036, 2100, "otw", "insert")):
        ref2, lives2 = synth.synth_batch(n_ref, 1, seed=900 + c)
        eng = ob.BatchedOTW(ref2, c, 3, batch=1, variant=variant, dtype=torch.float32)
        lv, ln = eng.pack(lives2)
        eng.run(lv, ln, mode=mode)
        acc, cost = eng.replay_dense()
        o = oracle.OtwOracle(ref2, c, 3, vmap[variant], keep_cost=True)
        (o.set_live if mode == "set_live" else o.run)(lives2[0])
        assert np.array_equal(eng.path(0), o.path), (c, "path")
        assert np.array_equal(acc[0].cpu().numpy(), o.acc_cost()) and np.array_equal(cost[0].cpu().numpy(), o.cost()), (c, "dense")
        eng.close()
        del acc, cost, o
    with pytest.raises(nat.RtsyncError):
        ob.BatchedOTW(ref, 2037, 3, batch=1, dtype=torch.float32)
