#!/usr/bin/env python3
"""Secondary measurements (not the headline metric): chroma, DTW and WTW kernels on synthetic input,
device time from HIP events on the launch stream.  Prints one JSON object per line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=5, warm=2):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    return float(np.median(ts))


def main():
    import torch
    from real_time_audio_sync_amd import chroma, dtw, synth, wtw
    from real_time_audio_sync_amd.otw_batch import frames_tensor
    dev = torch.device("cuda:0")
    out = []

    # ---- chroma: 30 minutes of audio at 22 050 Hz -> 19 380 frames (configs[4]'s reference length)
    plan = chroma.ChromaPlan(4096, 2048, 22050)
    rs = np.random.RandomState(0)
    wav = torch.from_numpy((rs.rand(30 * 60 * 22050) - 0.5).astype(np.float32)).to(dev)
    m = plan.num_frames(wav.numel(), 2048)
    t = timed(lambda: plan.frames(wav, pad_left=2048))
    out.append(dict(kernel="chroma_frames_kernel", frames=m, seconds=t, frames_per_s=m / t,
                    algorithmic_bytes=m * (2048 * 4 + 96), hbm_GBps=m * (2048 * 4 + 96) / t / 1e9,
                    note="8 KB of new float32 samples + 96 B chroma per frame; fp64 FFT; filterbank (197 KB) re-read from L2 per frame"))
    plan512 = chroma.ChromaPlan(4096, 512, 22050)
    m5 = plan512.num_frames(wav.numel(), 2048)
    t = timed(lambda: plan512.frames(wav, pad_left=2048))
    out.append(dict(kernel="chroma_frames_kernel(hop=512)", frames=m5, seconds=t, frames_per_s=m5 / t))

    # ---- DTW: configs[0] shapes, one pair and a batch of 256 pairs
    for n in (322, 1289):
        r = synth.synth_ref(n, seed=n)
        l = synth.synth_live(r, seed=n + 1, max_frames=n)
        a, b = frames_tensor(l, dev, torch.float32), frames_tensor(r, dev, torch.float32)
        t = timed(lambda: dtw.dtw_batch(a, b, want_back=False))
        assert int(dtw.dtw_batch(a, b, want_back=False, check=True)[4].min()) > 0
        cells = a.shape[0] * b.shape[0]
        out.append(dict(kernel="rts_dtw (cost + strip DP + backtrack)", M=int(a.shape[0]), N=int(b.shape[0]), pairs=1,
                        seconds=t, cells_per_s=cells / t, algorithmic_bytes=cells * 16.25,
                        note="8 B cost + 8 B acc + 2 bits of step code per cell; strips of 64 rows pipelined over waves / workgroups"))
        if n == 322:
            ab = a.unsqueeze(0).repeat(256, 1, 1).contiguous()
            t = timed(lambda: dtw.dtw_batch(ab, b, want_back=False))
            out.append(dict(kernel="rts_dtw (cost + strip DP + backtrack)", M=int(a.shape[0]), N=int(b.shape[0]), pairs=256,
                            seconds=t, cells_per_s=256 * cells / t, hbm_GBps=256 * cells * 16.25 / t / 1e9))

    # ---- WTW: 64 streams, W=100/hop=50 (wtw_live.py's setting) on a 2200-frame reference
    ref, lives = synth.synth_batch(2200, 64, seed=3)
    refd = torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev)
    tmax = max(l.shape[1] for l in lives)
    cols = np.zeros((64, tmax, 12))
    for i, l in enumerate(lives):
        cols[i, :l.shape[1]] = l.T
    cols_d = torch.from_numpy(cols).to(dev)
    n_new = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)
    eng = wtw.BatchedWTW(refd, 100, 50, 64)

    def run_wtw():
        eng.reset()
        eng.push(cols_d, n_new, precheck=True)
    t = timed(run_wtw)
    st = eng.states()
    frames = int(st[:, 0].sum())
    cells = int(sum((int(np.uint32(s[7])) << 32) | int(np.uint32(s[6])) for s in st))
    out.append(dict(kernel="wtw_win_kernel (every window of the push in one launch)", streams=64, W=100, hop=50, frames=frames, windows=int(st[:, 5].sum()),
                    seconds=t, frames_per_s=frames / t, cells_per_s=cells / t))
    # one long-form stream, W = 10 000 (configs[4] shape)
    ref5 = synth.synth_ref(19380, seed=500)
    live5 = synth.synth_live(ref5, seed=501)
    eng5 = wtw.BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref5.T)).to(dev), 10000, 5000, 1)
    c5 = torch.from_numpy(np.ascontiguousarray(live5.T))[None].to(dev)

    def run5():
        eng5.reset()
        eng5.push(c5, precheck=True)
    t = timed(run5, reps=3, warm=1)
    s5 = eng5.state()
    out.append(dict(kernel="wtw_big_dp_kernel + wtw_big_ctl_kernel (strip DP over many workgroups)", streams=1, W=10000,
                    hop=5000, windows=s5["windows"], seconds=t, seconds_per_window=t / max(1, s5["windows"]),
                    cells_per_s=s5["cells"] / t, frames_per_s=s5["chroma_ptr"] / t,
                    algorithmic_bytes_per_window=2 * 10000 * 10000 + 2 * 12 * 4 * 10000,
                    note="SURVEY 8(d) figure 2 W^2 + 96 W bytes per window; the kernel itself writes 2 bits per cell"))
    # ---- BASELINE configs[1]: single-stream OTW c=500, N=2200 -- batched engine with B=1, and the drop-in class
    from real_time_audio_sync_amd.otw_batch import BatchedOTW
    from real_time_audio_sync_amd.otw_eran import OnlineTimeWarping
    ref1 = synth.synth_ref(2200, seed=1000)
    live1 = synth.synth_live(ref1, seed=1001)
    for dt_name, tdt in (("f32", torch.float32), ("f64", torch.float64)):
        e1 = BatchedOTW(ref1, 500, 3, batch=1, dtype=tdt)
        lv1, ln1 = e1.pack([live1])
        t = timed(lambda: e1.run(lv1, ln1))
        fr = int(e1.states()[0, 8])
        out.append(dict(kernel="rts_otw_run B=1 (configs[1])", features=dt_name, frames=fr, seconds=t, frames_per_s=fr / t))
        e1.close()
    for dt_name, tdt in (("f64", torch.float64),):
        refb, livesb = synth.synth_batch(2200, 64, seed=1000)
        e64 = BatchedOTW(refb, 500, 3, batch=64, dtype=tdt)
        lvb, lnb = e64.pack(livesb)
        t = timed(lambda: e64.run(lvb, lnb))
        fr = int(e64.states()[:, 8].sum())
        out.append(dict(kernel="rts_otw_run B=64", features=dt_name, frames=fr, seconds=t, frames_per_s=fr / t,
                        note="float64 features in HBM (what wav_to_chroma produces): the <double> ring instantiation"))
        e64.close()
    o1 = OnlineTimeWarping(ref1, {'c': 500, 'max_run_count': 3})
    t0 = time.perf_counter()
    n1 = 0
    for i in range(live1.shape[1]):
        n1 += 1
        if o1.insert(live1[:, i]) == "stop":
            break
    wall = time.perf_counter() - t0
    out.append(dict(kernel="OnlineTimeWarping.insert loop (drop-in class, one launch + state read-back per frame)",
                    frames=n1, seconds=wall, frames_per_s=n1 / wall, us_per_insert=wall / n1 * 1e6,
                    note="host wall time; the reference's own class does 262 frames/s at this setting (BASELINE.md 3a)"))
    o2 = OnlineTimeWarping(ref1, {'c': 500, 'max_run_count': 3})
    t = timed(lambda: o2.set_live(live1), reps=3, warm=1)
    out.append(dict(kernel="OnlineTimeWarping.set_live (drop-in class, one launch)", frames=int(live1.shape[1]), seconds=t,
                    frames_per_s=live1.shape[1] / t))
    t0 = time.perf_counter()
    acc = o2.acc_cost
    out.append(dict(kernel="OnlineTimeWarping.acc_cost (lazy dense replay + D2H of 2N x N float64)", seconds=time.perf_counter() - t0,
                    bytes=int(acc.nbytes)))
    # ---- live use: one frame per stream per launch (rts_otw_insert), 64 streams, c = 500
    ref6, lives6 = synth.synth_batch(2200, 64, seed=1000)
    eng6 = BatchedOTW(ref6, 500, 3, batch=64, dtype=torch.float64)
    T6 = min(l.shape[1] for l in lives6)
    cols6 = torch.from_numpy(np.stack([np.ascontiguousarray(l[:, :T6].T) for l in lives6])).to(dev)
    for i in range(700):          # past the warm-up so every insert evaluates ~2 strips of 500 cells
        eng6.insert(cols6[:, i].contiguous())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(700, 1000):
        eng6.insert(cols6[:, i].contiguous())
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 300
    out.append(dict(kernel="rts_otw_insert (append + otw_advance_kernel, state persisted in HBM)", streams=64, c=500,
                    device_us_per_call=e0.elapsed_time(e1) * 1e3 / 300, host_wall_us_per_call=wall * 1e6,
                    note="real time needs one call per 92.9 ms hop; each call reloads the 100 KB windows + bands into LDS"))
    # ---- PCIe-inclusive headline: host float32 chroma (pinned) -> device -> rts_otw_run -> states back
    eng7 = BatchedOTW(ref6, 500, 3, batch=64, dtype=torch.float32)
    tmax7 = max(l.shape[1] for l in lives6)
    host = torch.zeros((64, tmax7, 12), dtype=torch.float32).pin_memory()
    for b_, l in enumerate(lives6):
        host[b_, :l.shape[1]] = torch.from_numpy(np.ascontiguousarray(l.T)).float()
    lens7 = torch.tensor([l.shape[1] for l in lives6], dtype=torch.int32, device=dev)
    dbuf = torch.empty_like(host, device=dev)

    def pcie_run():
        dbuf.copy_(host, non_blocking=True)
        eng7.run(dbuf, lens7)
    t = timed(pcie_run, reps=10, warm=3)
    frames7 = int(eng7.states()[:, 8].sum())
    out.append(dict(kernel="H2D copy (6.6 MB, pinned) + rts_otw_run", streams=64, seconds=t, frames_per_s=frames7 / t,
                    note="PCIe-inclusive variant of the headline metric; bench.py's value excludes the copy by contract"))

    # ---- the whole path in serving shape: 64 microphones deliver 1-second buffers; device chroma -> device OTW
    from real_time_audio_sync_amd.live import LiveSession
    sess = LiveSession(ref6, batch=64, c=500, max_run_count=3)
    secs = 60
    audio = (np.random.RandomState(5).rand(64, secs * 22050) - 0.5).astype(np.float32)
    for s_ in range(3):   # warm-up feeds
        sess.feed([audio[b, s_ * 22050:(s_ + 1) * 22050] for b in range(64)])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(3, secs):
        sess.feed([audio[b, s_ * 22050:(s_ + 1) * 22050] for b in range(64)])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = int(sess.otw.states()[:, 8].sum())
    out.append(dict(kernel="LiveSession.feed(list of 64 arrays) via rts_live_* (one H2D + append + chroma + OTW push + compact per feed; tools/bench_live.py has the other modes)",
                    streams=64, audio_seconds_per_stream=secs - 3, wall_seconds=dt,
                    realtime_factor=(secs - 3) / dt, frames_per_s=64 * (secs - 3) * 22050 / 2048 / dt,
                    frames_total=frames,
                    note="random audio (worst case for the tracker); per feed: 64 numpy copies into the pinned staging slot, nothing read back"))
    for o in out:
        print(json.dumps(o))


if __name__ == "__main__":
    main()
