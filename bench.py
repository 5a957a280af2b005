#!/usr/bin/env python3
"""Headline benchmark: aligned chroma frames/sec, batch=64 concurrent live streams against one
reference, OTW (otw_eran semantics) c=500, max_run_count=3 -- BASELINE.json configs[2] on one GPU,
configs[3] (64 streams per GPU, independent shards, no collective on the data path) on N GPUs.

A "step" = one whole-sequence pass of the hot path over the batch: rts_otw_run on B streams of
~2150 synthetic chroma frames each (inputs already resident in HBM).  value = live frames consumed
by all ranks per second of wall time (barrier + synchronize on both sides, max over ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W]          one process per GPU (torch.distributed.run for N > 1)
    python bench.py --gpus N --single-process                    one host process driving N devices (shard.ShardedOTW)

Extra JSON objects (see the task contract):
  roofline      dominant kernel (otw_advance_kernel): algorithmic bytes per launch / mean launch
                duration from HIP events recorded on the launch stream, against HBM 8 TB/s.
  cpu_baseline  N = 1 only: timed on this box's host cores on the same streams, before the GPU is touched (the
                worker processes are forked while the process is still GPU-free): the C port of otw_eran.py's insert
                loop (oracle/) on 1 core and on all usable cores (one stream per task), and the numpy
                restatement (oracle/otw_numpy.py -- the reference's own per-cell numpy calls) on 1 core for one stream.
  parity        EVERY stream of EVERY rank against the C port's path (each rank checks its own shard with workers it
                forks before touching the GPU; the mismatch and stream counts are all-reduced); a mismatch voids `value`.
  secondary     N = 1 only, after the timed headline: the other BASELINE configs and kernels on the same box, each
                {workload, ms, algorithmic_bytes, frac, cpu_port_ms, parity} (configs[0] DTW, configs[1] B = 1,
                configs[4] one W = 10 000 window, chroma, float64 features, WTW at the reference's two window settings).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
F_CLK_HZ = 2.4e9         # MI355X_MICROARCH.md "Max clock"; an assumed figure, the run does not read the clock


def usable_cpus():
    """Cores this process may actually use: scheduler affinity, further limited by a cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_worker(task):
    """One stream through the C oracle in a worker process: (frames, seconds in the insert loop, path)."""
    ref, live, c, mrc = task
    import oracle
    o = oracle.OtwOracle(ref, c, mrc)
    lv = np.ascontiguousarray(live.T)
    t0 = time.perf_counter()
    n = o._L.orc_otw_run(o._h, lv.ctypes.data, lv.shape[0])  # insert loop only, like SURVEY 6
    dt = time.perf_counter() - t0
    return n, dt, o.path, os.getpid()


def oracle_paths(ref, lives, c, mrc, procs):
    """The C port's path for every stream (the parity checker of a rank's shard), over `procs` forked workers.
    Must run before anything touches the GPU (it forks)."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    tasks = [(ref, l, c, mrc) for l in lives]
    if procs <= 1 or len(tasks) <= 1:
        return [_cpu_worker(t)[2] for t in tasks]
    with mp.get_context("fork").Pool(min(procs, len(tasks))) as pool:
        return [r[2] for r in pool.map(_cpu_worker, tasks, chunksize=1)]


def cpu_legs(ref, lives, c, mrc, n_cpu, numpy_leg=True, allow_fork=True):
    """The CPU baseline legs.  Must run before anything touches the GPU (it forks)."""
    import multiprocessing as mp
    import oracle
    oracle.build()
    tasks = [(ref, lives[b], c, mrc) for b in range(n_cpu)]
    # 1 core, streams one after another
    seq = [_cpu_worker(t) for t in tasks]
    frames1 = sum(r[0] for r in seq)
    secs1 = sum(r[1] for r in seq)
    paths = [r[2] for r in seq]
    legs = {"c_port_1core": {"value": frames1 / secs1, "unit": "frames/s", "cores": 1,
                             "sample": "%d streams, %d frames, one after another" % (n_cpu, frames1)}}
    # all usable cores, one stream per task; the rate is frames / the busiest worker's time in the insert loop
    procs = min(n_cpu, usable_cpus()) if allow_fork else 1
    if procs > 1:
        with mp.get_context("fork").Pool(procs) as pool:
            pool.map(_cpu_worker, tasks[:procs], chunksize=1)  # warm every worker (library load, first touch)
            w0 = time.perf_counter()
            par = pool.map(_cpu_worker, tasks, chunksize=1)
            wall = time.perf_counter() - w0
        per_proc = {}
        for n, dt, _, pid in par:
            per_proc[pid] = per_proc.get(pid, 0.0) + dt
        framesp = sum(r[0] for r in par)
        legs["c_port_allcores"] = {"value": framesp / max(per_proc.values()), "unit": "frames/s", "cores": procs,
                                   "wall_value": framesp / wall,
                                   "sample": "%d streams over %d forked processes; value = frames / busiest process's "
                                             "insert-loop time, wall_value includes the 2N x N matrix allocation per stream"
                                             % (n_cpu, procs)}
    if numpy_leg:
        from oracle import otw_numpy
        o = otw_numpy.NumpyOTW(ref, c, mrc)
        t0 = time.perf_counter()
        n = o.run(lives[0])
        dt = time.perf_counter() - t0
        ok = np.array_equal(np.array(o.path, dtype=np.int32).reshape(-1, 2), paths[0])
        legs["numpy_1core"] = {"value": n / dt, "unit": "frames/s", "cores": 1, "path_equals_c_port": bool(ok),
                               "sample": "stream 0 only (%d frames): the reference's per-cell numpy calls, pure Python "
                                         "loop; the reference itself: 262 frames/s (BASELINE.md 3a, survey container)" % n}
    return legs, paths


# ------------------------------------------------------------------------------------------------------------------
# secondary workloads (N = 1): inputs, CPU side (C port timings + expected results), GPU side
# ------------------------------------------------------------------------------------------------------------------
def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def secondary_inputs():
    from real_time_audio_sync_amd import synth
    inp = {}
    for n in (322, 1289):   # configs[0]: two 30 s clips at hop 2048 / hop 512
        r = synth.synth_ref(n, seed=n)
        inp["dtw%d" % n] = (synth.synth_live(r, seed=n + 1, max_frames=n), r)
    inp["wtw64"] = synth.synth_batch(2200, 64, seed=3)          # 64 live streams for the two reference window settings
    ref5 = synth.synth_ref(19380, seed=500)                     # configs[4]: 30-minute reference, W = 10 000
    inp["wtw10k"] = (ref5, synth.synth_live(ref5, seed=501)[:, :10000])
    # 30 minutes at 22 050 Hz, PCM16-valued like a WAV through librosa.load (x / 32768)
    inp["audio"] = np.random.RandomState(0).randint(-32768, 32768, 30 * 60 * 22050, dtype=np.int16).astype(np.float32) / np.float32(32768.0)
    return inp


def _wtw_oracle_paths(ref, lives, W, hopf):
    import oracle
    paths = []
    t0 = time.perf_counter()
    for l in lives:
        o = oracle.WtwOracle(ref, W, hopf)
        o.insert_precheck()
        for q in range(l.shape[1]):
            if o.push_col(l[:, q]) != oracle.RUNNING:
                break
        paths.append(o.path)
    return paths, (time.perf_counter() - t0) * 1e3


CHROMA_CHECK_SEGMENTS = 12   # segments of 24 frames spread over the 30 minutes, checked against the numpy oracle


def secondary_cpu(inp):
    """Expected results and C-port timings (one host core) of the secondary workloads."""
    import oracle
    from oracle import chroma_oracle
    cpu = {}
    for key in ("dtw322", "dtw1289"):
        a, b = inp[key]
        oracle.dtw(a[:, :8], b[:, :8])   # library load
        t0 = time.perf_counter()
        _, acc, path, _ = oracle.dtw(a, b)
        cpu[key] = {"ms": (time.perf_counter() - t0) * 1e3, "path": path, "acc_sha": _sha(acc)}
    ref, lives = inp["wtw64"]
    for W, hopf in ((20, 10), (100, 50)):
        paths, ms = _wtw_oracle_paths(ref, lives, W, hopf)
        cpu["wtw%d" % W] = {"ms": ms, "paths": paths}
    # W = 10 000: one window is 1e8 cells = about a minute of the C port; the expected path is a committed fixture
    # (tests/golden/wtw10k_golden.json, made by tests/golden/make_golden.py from the same seeded input)
    gpath = os.path.join(ROOT, "tests", "golden", "wtw10k_golden.json")
    cpu["wtw10k"] = json.load(open(gpath)) if os.path.exists(gpath) else None
    # chroma: the numpy oracle on CHROMA_CHECK_SEGMENTS segments of the 30 minutes
    wav = inp["audio"]
    hop, L, nseg, per = 2048, 4096, CHROMA_CHECK_SEGMENTS, 24
    m_total = (len(wav) + L // 2 - L) // hop + 1
    firsts = sorted(set([0] + [int(x) for x in np.linspace(1, m_total - per, nseg - 1)]))
    want, secs = {}, 0.0
    for m0 in firsts:
        # global frame m covers samples [m*hop - L/2, m*hop + L/2).  A segment cut at the first sample of frame m0 > 0
        # goes through the oracle's own zero padding: its frame 0 is a throw-away, its frames 1.. are frames m0..
        lo = m0 * hop - L // 2
        seg = wav[lo: lo + (per - 1) * hop + L] if m0 > 0 else wav[: (per - 1) * hop + L // 2]
        t0 = time.perf_counter()
        ch = chroma_oracle.wav_to_chroma(seg)
        secs += time.perf_counter() - t0
        ch = ch[:, 1:1 + per] if m0 > 0 else ch[:, :per]
        assert ch.shape[1] == per
        want[m0] = ch
    frames_done = sum(v.shape[1] for v in want.values())
    cpu["chroma"] = {"want": want, "ms_per_frame": secs * 1e3 / frames_done, "frames": frames_done}
    return cpu


def _timed(fn, reps=5, warm=2, before=None):
    """Median device time of fn() in ms (HIP events on the current stream); `before()` runs ahead of every call, outside
    the timed interval (re-arming a handle: the constructor's work, not the path's)."""
    import torch
    for _ in range(warm):
        if before:
            before()
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        if before:
            before()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def _entry(workload, ms, alg_bytes, parity, **kw):
    e = {"workload": workload, "ms": ms, "algorithmic_bytes": int(alg_bytes),
         "frac": alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "parity": parity}
    e.update(kw)
    return e


def _otw_alg_bytes(states, nat):
    frames = int(states[:, nat.ST_CONSUMED].sum())
    cells = int(sum((int(np.uint32(s[nat.ST_CELLS_HI])) << 32) | int(np.uint32(s[nat.ST_CELLS_LO])) for s in states))
    n_col = int(states[:, nat.ST_COL_STRIPS].sum())
    n_path = int(states[:, nat.ST_N_PATH].sum())
    # A = 4*cells + 48 + 48*n_col + 8*n_path bytes per live frame, summed exactly
    return 4 * cells + 48 * frames + 48 * n_col + 8 * n_path, frames, cells, n_col, n_path


def secondary_gpu(inp, cpu, dev, head_ref, head_lives, head_paths, c, mrc, only=None):
    """The other BASELINE configs on the same box, device time from HIP events on the launch stream (median of 5).
    `only`: set of entry keys to run (profiling: tools/collect_profile.py traces one entry per rocprofv3 run)."""
    import torch
    from real_time_audio_sync_amd import _native as nat
    from real_time_audio_sync_amd import chroma, dtw, otw_batch, wtw
    from real_time_audio_sync_amd.otw_batch import frames_tensor
    out = []

    def wanted(key):
        return only is None or key in only

    # ---- configs[0]: offline DTW on two 30 s clips (322 frames at hop 2048, 1289 at hop 512), one pair
    for key in ("dtw322", "dtw1289"):
        if not wanted(key):
            continue
        a, b = inp[key]
        ad, bd = frames_tensor(a, dev, torch.float32), frames_tensor(b, dev, torch.float32)
        ms = _timed(lambda: dtw.dtw_batch(ad, bd, want_back=False))
        _, acc, _, path, plen = dtw.dtw_batch(ad, bd, want_back=False)
        n = int(plen[0].item())
        ok = n > 0 and np.array_equal(path[0, :n].cpu().numpy(), cpu[key]["path"]) and _sha(acc[0].cpu().numpy()) == cpu[key]["acc_sha"]
        cells = a.shape[1] * b.shape[1]
        out.append(_entry("configs[0]: dtw.DTW %d x %d, one pair (cost + strip DP + backtrack)" % (a.shape[1], b.shape[1]),
                          ms, cells * 16.25, {"path_and_acc_equal_c_port": bool(ok)}, key=key, cpu_port_ms=cpu[key]["ms"], cpu_cores=1,
                          kernels=["dtw_cost_kernel", "dtw_prep_kernel", "dtw_sdp_kernel", "dtw_tail_kernel"],
                          bytes_per_unit="16.25 B per cell: 8 cost + 8 acc + 2 bits of step code"))

    # ---- configs[1]: single-stream OTW c=500 (stream 0 of the headline batch), and the batch with float64 features
    for key, name, B, tdt in (("otw_b1", "configs[1]: single-stream OTW c=%d (stream 0 of the batch), f32 features" % c, 1, torch.float32),
                              ("otw_b64_f64", "configs[2] with float64 features in HBM (what wav_to_chroma produces)", len(head_lives), torch.float64)):
        if not wanted(key):
            continue
        eng = otw_batch.BatchedOTW(head_ref, c, mrc, batch=B, dtype=tdt, device=dev)
        lv, ln = eng.pack(head_lives[:B])
        ms = _timed(lambda: eng.run(lv, ln))
        st = eng.states()
        alg, frames, _, _, _ = _otw_alg_bytes(st, nat)
        bad = sum(0 if np.array_equal(eng.path(b), head_paths[b]) else 1 for b in range(B)) if head_paths else None
        out.append(_entry(name, ms, alg, {"streams_checked": B if head_paths else 0, "path_mismatches": bad}, key=key,
                          kernels=["otw_reset_kernel", "otw_advance_kernel"], frames=frames, frames_per_s=frames / (ms * 1e-3)))
        eng.close()

    # ---- chroma front end: 30 minutes of audio -> 19 379 frames (configs[4]'s reference length)
    if wanted("chroma"):
        plan = chroma.ChromaPlan(4096, 2048, 22050, dev)
        wav = torch.from_numpy(inp["audio"]).to(dev)
        m = plan.num_frames(wav.numel(), 2048)
        ms = _timed(lambda: plan.frames(wav, pad_left=2048))
        got = plan.frames(wav, pad_left=2048)[0].cpu().numpy()
        worst = 0.0
        for m0, expect in cpu["chroma"]["want"].items():
            worst = max(worst, float(np.abs(got[m0:m0 + expect.shape[1]].T - expect).max()))
        out.append(_entry("chroma.wav_to_chroma: 30 min of audio, fft 4096 / hop 2048, %d frames" % m, ms, m * (2048 * 4 + 96),
                          {"frames_checked": cpu["chroma"]["frames"], "max_abs_diff_vs_numpy_oracle": worst, "tolerance": 1e-11,
                           "ok": bool(worst <= 1e-11)}, key="chroma", kernels=["chroma_frames4096_kernel"],
                          cpu_port_ms=cpu["chroma"]["ms_per_frame"] * m, cpu_cores=1,
                          cpu_sample="numpy oracle on %d frames, scaled to %d" % (cpu["chroma"]["frames"], m),
                          frames_per_s=m / (ms * 1e-3), bytes_per_unit="8 KB of new float32 samples + 96 B chroma per frame"))
        plan.close()
        del wav

    # ---- WTW, 64 streams, at the reference's two window settings: W=20/hop=10 (tests.py:174), W=100/hop=50 (wtw_live.py)
    if wanted("wtw20") or wanted("wtw100"):
        ref, lives = inp["wtw64"]
        refd = torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev)
        tmax = max(l.shape[1] for l in lives)
        cols = np.zeros((len(lives), tmax, 12))
        for i, l in enumerate(lives):
            cols[i, :l.shape[1]] = l.T
        cols_d = torch.from_numpy(cols).to(dev)
        n_new = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)
        for W, hopf in ((20, 10), (100, 50)):
            if not wanted("wtw%d" % W):
                continue
            eng = wtw.BatchedWTW(refd, W, hopf, len(lives))

            ms = _timed(lambda: eng.push(cols_d, n_new, precheck=True), before=eng.reset)
            st = eng.states()
            windows = int(st[:, 5].sum())
            frames = int(st[:, 0].sum())
            bad = sum(0 if np.array_equal(eng.path(b), cpu["wtw%d" % W]["paths"][b]) else 1 for b in range(len(lives)))
            out.append(_entry("wtw.WTW: %d streams, W=%d frames / hop=%d, ref 2200 frames" % (len(lives), W, hopf), ms,
                              windows * (2 * W * W + 96 * W), {"streams_checked": len(lives), "path_mismatches": bad},
                              key="wtw%d" % W, kernels=["wtw_append_kernel", "wtw_win_kernel"],
                              cpu_port_ms=cpu["wtw%d" % W]["ms"], cpu_cores=1, windows=windows, frames=frames,
                              frames_per_s=frames / (ms * 1e-3), bytes_per_unit="2 W^2 + 96 W per window (SURVEY 8(d))",
                              includes="insert()'s entry check + append + every window of every stream in one launch"))
            eng.close()

    # ---- configs[4]: one W = 10 000 window against the 30-minute reference
    if wanted("wtw10k"):
        ref5, live5 = inp["wtw10k"]
        eng = wtw.BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref5.T)).to(dev), 10000, 5000, 1)
        c5 = torch.from_numpy(np.ascontiguousarray(live5.T))[None].to(dev)

        ms = _timed(lambda: eng.push(c5, precheck=True), reps=5, warm=1, before=eng.reset)
        s5 = eng.state()
        p5 = eng.path()
        gold = cpu["wtw10k"]
        par = {"windows": s5["windows"], "path_len": int(len(p5)), "path_sha256": _sha(p5.astype(np.int32))}
        if gold:
            par["equal_to_fixture"] = bool(par["path_sha256"] == gold["path_sha256"] and par["path_len"] == gold["path_len"])
            par["fixture"] = "tests/golden/wtw10k_golden.json (%s)" % gold.get("made_by", "?")
        out.append(_entry("configs[4]: wtw.WTW long-form, ref 19 380 frames, one window of W=10 000 frames (float64, not the "
                          "config's fp16 band)", ms / max(s5["windows"], 1), 2 * 10000 * 10000 + 96 * 10000, par, key="wtw10k",
                          kernels=["wtw_big_ctl_kernel", "wtw_big_fill_kernel", "wtw_big_dp_kernel", "wtw_big_hops_kernel",
                                   "wtw_big_segment_kernel"],
                          cpu_port_ms=gold.get("c_port_ms") if gold else None, cpu_cores=1,
                          cpu_sample="one window through the C port when the fixture was made (build container); the "
                                     "reference's own Python took %.0f s there" % (gold.get("reference_s", float("nan")) if gold else float("nan")),
                          bytes_per_unit="2 W^2 + 96 W per window (SURVEY 8(d))"))
        eng.close()
    return out


# ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="streams per GPU")
    ap.add_argument("--n-ref", type=int, default=2200)
    ap.add_argument("--c", type=int, default=500)
    ap.add_argument("--max-run-count", type=int, default=3)
    ap.add_argument("--waves", type=int, default=0, help="waves per stream workgroup (0 = library default)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"], help="feature dtype in HBM")
    ap.add_argument("--cpu-streams", type=int, default=-1, help="streams timed on the CPU oracle (-1 = all on rank 0)")
    ap.add_argument("--no-cpu", action="store_true", help="skip cpu_baseline / parity gate / secondary (profiling runs)")
    ap.add_argument("--no-numpy", action="store_true", help="skip the (slow) numpy leg of cpu_baseline")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary workloads")
    ap.add_argument("--secondary-only", default="", help="comma-separated secondary entry keys (dtw322, dtw1289, otw_b1, "
                    "otw_b64_f64, chroma, wtw20, wtw100, wtw10k): run only those (profiling, one entry per rocprofv3 run)")
    ap.add_argument("--no-fork", action="store_true", help="CPU legs without worker processes (1-core leg only): safe "
                    "under rocprofv3, whose preloaded runtime must not be forked")
    ap.add_argument("--single-process", action="store_true",
                    help="one host process drives all --gpus devices through shard.ShardedOTW (no torch.distributed)")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "otw_traffic.json"),
                    help="PMC-derived HBM bytes per launch (from the committed rocprofv3 passes) to report as roofline.traffic")
    args = ap.parse_args()
    verbose = bool(os.environ.get("BENCH_VERBOSE"))

    def note(msg):
        if verbose:
            print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)

    if verbose:
        import faulthandler
        faulthandler.dump_traceback_later(90, repeat=True, file=sys.stderr)
    note("start rank=%s world=%s" % (os.environ.get("RANK"), os.environ.get("WORLD_SIZE")))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    single = args.single_process
    if single and world != 1:
        raise SystemExit("--single-process is one host process: do not launch it under torch.distributed.run")
    if not single and world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d (or pass --single-process)" % args.gpus)
    n_gpus = args.gpus if single else world

    from real_time_audio_sync_amd import shard, synth

    # ---- workload: one reference, `batch` different time-warped noisy renditions per GPU
    B = args.batch
    ref = synth.synth_ref(args.n_ref, seed=1000)
    if single:
        lo, hi = 0, B * n_gpus        # this process holds every shard
    else:
        lo, hi = shard.partition(B * world, world, rank)  # contiguous slice of the global stream list
    lives = [synth.synth_live(ref, seed=shard.stream_seed(1000, g)) for g in range(lo, hi)]

    # ---- CPU side, before the GPU is initialised (the workers are forked): at N = 1 the cpu_baseline legs, whose paths
    # are also the parity checker; at N > 1 every rank computes the C port's paths of its own shard
    cpu = opaths = sec_inp = sec_cpu = None
    want_secondary = (n_gpus == 1 and rank == 0 and not args.no_cpu and not args.no_secondary and B == 64
                      and args.c == 500 and args.n_ref == 2200 and args.dtype == "f32")
    if not args.no_cpu:
        if n_gpus == 1:
            n_cpu = B if args.cpu_streams < 0 else min(B, args.cpu_streams)
            cpu = cpu_legs(ref, lives, args.c, args.max_run_count, n_cpu, numpy_leg=not args.no_numpy, allow_fork=not args.no_fork)
            opaths = cpu[1]
        else:
            opaths = oracle_paths(ref, lives, args.c, args.max_run_count, max(1, usable_cpus() // max(local_world, 1)))
        note("cpu legs done")
        if want_secondary:
            sec_inp = secondary_inputs()
            sec_cpu = secondary_cpu(sec_inp)
            note("secondary cpu side done")

    import torch
    import torch.distributed as dist

    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # Rehearsal switches for a one-GPU box (never set by the driver): all ranks / shards on device 0 and the two
    # tiny report reductions over gloo, since RCCL refuses two ranks on one device.
    single_device = bool(os.environ.get("BENCH_SINGLE_DEVICE"))
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = 0 if single_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from real_time_audio_sync_amd import _native as nat
    from real_time_audio_sync_amd import otw_batch

    tdt = torch.float32 if args.dtype == "f32" else torch.float64
    if single:
        devices = [0] * n_gpus if single_device else list(range(n_gpus))
        if not single_device and torch.cuda.device_count() < n_gpus:
            raise SystemExit("--gpus %d but only %d devices are visible" % (n_gpus, torch.cuda.device_count()))
        eng = shard.ShardedOTW(ref, args.c, args.max_run_count, batch=B * n_gpus, devices=devices, variant="otw",
                               dtype=tdt, waves=(args.waves or None))
        packed = eng.pack(lives)
        eng0 = eng.engines[0]

        def run_step():
            eng.run(packed)

        def barrier():
            eng.synchronize()
    else:
        eng = otw_batch.BatchedOTW(ref, args.c, args.max_run_count, batch=B, variant="otw", dtype=tdt, device=dev,
                                   waves=(args.waves or None))
        live_dev, len_dev = eng.pack(lives)
        eng0 = eng

        def run_step():
            eng.run(live_dev, len_dev)

        def barrier():
            torch.cuda.synchronize(dev)
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize(dev)
    note("inputs resident")

    for _ in range(args.warmup):
        run_step()
    barrier()
    # HIP events on the launch stream of the first engine (device 0's slice in single-process mode)
    with torch.cuda.device(eng0.device):
        ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
        ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    if single:
        for i in range(args.steps):
            for k, (e, (lv, ln)) in enumerate(zip(eng.engines, packed)):
                if k == 0:
                    with torch.cuda.device(e.device):
                        ev0[i].record()
                        e.run(lv, ln)
                        ev1[i].record()
                else:
                    e.run(lv, ln)
    else:
        for i in range(args.steps):
            ev0[i].record()
            run_step()
            ev1[i].record()
    barrier()
    t1 = time.perf_counter()
    note("timed region done")
    elapsed = t1 - t0
    launch_ms = [a.elapsed_time(b) for a, b in zip(ev0, ev1)]

    states = np.concatenate([e.states() for e in eng.engines]) if single else eng.states()
    alg_all, frames, cells, n_col, n_path = _otw_alg_bytes(states, nat)
    st0 = eng0.states()                       # the slice the HIP events bracket (one device's launch)
    alg_bytes = _otw_alg_bytes(st0, nat)[0]

    # ---- parity (outside the timed region): every stream of this rank's shard against the C port's path
    mism = checked = 0
    if opaths is not None:
        got = eng.paths()
        checked = len(opaths)
        mism = sum(0 if np.array_equal(got[b], opaths[b]) else 1 for b in range(checked))

    # the only collectives in the run: max clock, frame count and the parity tallies for the report (never on the data path)
    rdev = dev if backend == "nccl" else None
    elapsed, total_frames = shard.reduce_clock_and_count(elapsed, frames, device=rdev)
    _, mism_total = shard.reduce_clock_and_count(0.0, mism, device=rdev)
    _, checked_total = shard.reduce_clock_and_count(0.0, checked, device=rdev)

    result = None
    if rank == 0:
        value = total_frames * args.steps / elapsed
        mean_launch_s = float(np.mean(launch_ms)) * 1e-3
        achieved = alg_bytes / mean_launch_s / 1e9
        traffic, traffic_src = None, None
        default_workload = (B == 64 and args.c == 500 and args.n_ref == 2200 and args.dtype == "f32")
        if default_workload and args.traffic_json and os.path.exists(args.traffic_json):
            tj = json.load(open(args.traffic_json))
            src_now = hashlib.sha256(open(os.path.join(ROOT, "real_time_audio_sync_amd", "csrc", "otw.hip"), "rb").read()).hexdigest()[:16]
            if tj.get("otw_hip_sha16") in (None, src_now):
                traffic = tj.get("hbm_bytes_per_launch")
            traffic_src = {"file": os.path.relpath(args.traffic_json, ROOT), "measured_in_this_run": False,
                           "pass": tj.get("source"), "kernel_source_sha16": tj.get("otw_hip_sha16"),
                           "kernel_source_sha16_now": src_now,
                           "note": "PMC counters need their own rocprofv3 --pmc passes (tools/collect_profile.py); the figure "
                                   "is dropped (null) when csrc/otw.hip has changed since that pass"}
        # latency model: every stream is one workgroup walking its own dependent chain of steps
        # (one decide() = one path point per step), so a launch lasts as long as its longest stream:
        #   frames/s <= B * f_clk * frames_per_step / cycles_per_step
        steps_max = int(st0[:, nat.ST_N_PATH].max())
        cycles_per_step = mean_launch_s * F_CLK_HZ / max(steps_max, 1)
        result = {
            "metric": "aligned chroma frames/sec, batch=64 OTW c=500; path-index match vs CPU ref",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64" if args.dtype == "f64" else "f64 arithmetic, f32 features in HBM",
            "data": "synthetic",
            "config": {"workload": "configs[%d]: batch=%d concurrent live streams per GPU vs one reference, OTW c=%d, "
                                   "max_run_count=%d, ref %d frames, live ~%d frames/stream, %s chroma in HBM"
                                   % (2 if n_gpus == 1 else 3, B, args.c, args.max_run_count, args.n_ref,
                                      total_frames // (B * n_gpus), args.dtype),
                       "streams_total": B * n_gpus, "streams_per_gpu": B, "frames_per_step": total_frames,
                       "cells_per_frame": cells / max(frames, 1), "waves_per_stream": args.waves or 8,
                       "launcher": ("one host process, %d devices (shard.ShardedOTW)" % n_gpus) if single
                                   else ("one process per GPU, %s" % (backend if world > 1 else "no process group")),
                       "partition": "contiguous: rank r holds streams [r*%d, (r+1)*%d)" % (B, B)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": eng0.kernel_name, "launch_ms": mean_launch_s * 1e3,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "latency_model": {
                             "f_clk_hz": F_CLK_HZ, "f_clk_source": "assumed: the 2.4 GHz maximum clock of MI355X_MICROARCH.md, not read in this run",
                             "steps_longest_stream": steps_max,
                             "cycles_per_step": cycles_per_step,
                             "steps_per_frame": n_path / max(frames, 1),
                             "cells_per_step": cells / max(n_path, 1),
                             "bound_frames_per_s": B * F_CLK_HZ * (frames / max(n_path, 1)) / cycles_per_step,
                             "formula": "frames/s <= B * f_clk / (steps_per_frame * cycles_per_step); a step is one "
                                        "decide(): ~c cells of a dependent float64 (min,+) chain plus the control "
                                        "decision, executed by one workgroup per stream"},
                         "note": "B=64 occupies 64 of 256 CUs and each strip is a dependent float64 "
                                 "(min,+) chain: latency/occupancy-bound by construction, not HBM-bound"},
        }
        if single_device and n_gpus > 1:
            result["config"]["rehearsal"] = "BENCH_SINGLE_DEVICE: all %d shards on device 0 (not a scaling measurement)" % n_gpus
        if cpu is not None:
            legs = cpu[0]
            main_leg = legs.get("c_port_allcores", legs["c_port_1core"])
            result["cpu_baseline"] = {
                "value": main_leg["value"], "unit": "frames/s", "cores": main_leg["cores"], "kind": "port",
                "sample": "all %d streams of the batch (%d frames each on average) through the C port of otw_eran.py's "
                          "insert loop (oracle/rtsync_oracle.c, dense 2N x N float64 matrices, constructor excluded) on "
                          "%d host cores; legs: the same on 1 core, and the numpy restatement on 1 core"
                          % (len(opaths), frames // B, main_leg["cores"]),
                "legs": legs, "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus()}
        if not args.no_cpu:
            result["parity"] = {"streams_checked": checked_total, "streams_total": B * n_gpus,
                                "path_mismatches": mism_total,
                                "how": "every rank checks every stream of its own shard against the C port's path"}
            if mism_total or checked_total != B * n_gpus:
                result["value"] = 0.0  # a fast kernel whose results differ is not done
                result["parity"]["note"] = "PATH MISMATCH vs CPU oracle or unchecked streams: value voided"

    # ---- secondary workloads (N = 1): after the timed headline, on the same device
    if result is not None and want_secondary:
        t_sec = time.perf_counter()
        only = set(k for k in args.secondary_only.split(",") if k) or None
        result["secondary"] = secondary_gpu(sec_inp, sec_cpu, dev, ref, lives, opaths, args.c, args.max_run_count, only)
        result["secondary_wall_s"] = time.perf_counter() - t_sec
    if rank == 0:
        print(json.dumps(result))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
