#!/usr/bin/env python3
"""Timings of the strip-DP kernels (csrc/sdp.h): offline DTW and windowed time warping on synthetic chroma.
Device time from HIP events on the launch stream.  One JSON object per line.

    python tools/bench_sdp.py [dtw] [wtw] [big] [chroma] [wtw10k]
    (default: dtw wtw; `big` adds the 19 380^2 DTW, `chroma` 30 minutes of audio, `wtw10k` only the W = 10 000 stream)
"""
import json
import os
import sys

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
    from real_time_audio_sync_amd import dtw, synth, wtw
    from real_time_audio_sync_amd.otw_batch import frames_tensor
    what = set(sys.argv[1:]) or {"dtw", "wtw"}
    dev = torch.device("cuda:0")
    cfg = os.environ.get("RTS_SDP_CONFIG", "default")

    def emit(**kw):
        kw["sdp_config"] = cfg
        print(json.dumps(kw), flush=True)

    if "dtw" in what:
        for n in (322, 1289):
            r = synth.synth_ref(n, seed=n)
            l = synth.synth_live(r, seed=n + 1, max_frames=n)
            a, b = frames_tensor(l, dev, torch.float32), frames_tensor(r, dev, torch.float32)
            cells = a.shape[0] * b.shape[0]
            t = timed(lambda: dtw.dtw_batch(a, b, want_back=False))
            assert int(dtw.dtw_batch(a, b, want_back=False, check=True)[4].min()) > 0
            emit(kernel="rts_dtw", M=int(a.shape[0]), N=int(b.shape[0]), pairs=1, seconds=t, cells_per_s=cells / t,
                 algorithmic_bytes=cells * 16.25, hbm_GBps=cells * 16.25 / t / 1e9)
            if n == 322:
                ab = a.unsqueeze(0).repeat(256, 1, 1).contiguous()
                t = timed(lambda: dtw.dtw_batch(ab, b, want_back=False))
                emit(kernel="rts_dtw", M=int(a.shape[0]), N=int(b.shape[0]), pairs=256, seconds=t,
                     cells_per_s=256 * cells / t, hbm_GBps=256 * cells * 16.25 / t / 1e9)
    if "big" in what:
        n = 19380
        ref = synth.synth_ref(n, seed=80)
        live = synth.synth_live(ref, seed=81)
        a, b = frames_tensor(live, dev, torch.float32), frames_tensor(ref, dev, torch.float32)
        cells = a.shape[0] * b.shape[0]
        t = timed(lambda: dtw.dtw_batch(a, b, want_back=False), reps=3, warm=1)
        emit(kernel="rts_dtw", M=int(a.shape[0]), N=int(b.shape[0]), pairs=1, seconds=t, cells_per_s=cells / t,
             algorithmic_bytes=cells * 16.25, hbm_GBps=cells * 16.25 / t / 1e9,
             note="8 B cost + 8 B acc written, 2 bits of step code per cell")
    if "chroma" in what:
        from real_time_audio_sync_amd import chroma
        plan = chroma.ChromaPlan(4096, 2048, 22050)
        wav = torch.from_numpy((np.random.RandomState(0).rand(30 * 60 * 22050) - 0.5).astype(np.float32)).to(dev)
        m = plan.num_frames(wav.numel(), 2048)
        t = timed(lambda: plan.frames(wav, pad_left=2048))
        emit(kernel="chroma_frames4096_kernel", frames=m, seconds=t, frames_per_s=m / t,
             algorithmic_bytes=m * (2048 * 4 + 96), hbm_GBps=m * (2048 * 4 + 96) / t / 1e9)
    if "wtw" in what or "wtw10k" in what:
        ref5 = synth.synth_ref(19380, seed=500)
        live5 = synth.synth_live(ref5, seed=501)
        eng5 = wtw.BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref5.T)).to(dev), 10000, 5000, 1)
        c5 = torch.from_numpy(np.ascontiguousarray(live5.T))[None].to(dev)

        def run5():
            eng5.reset()
            eng5.push(c5, precheck=True)
        t = timed(run5, reps=3, warm=1)
        s5 = eng5.state()
        emit(kernel="rts_wtw_push (W=10000: dp + hops + 2 segment passes + ctl per window)", streams=1, W=10000,
             hop=5000, windows=s5["windows"], seconds=t, seconds_per_window=t / max(1, s5["windows"]),
             cells_per_s=s5["cells"] / t, frames_per_s=s5["chroma_ptr"] / t,
             algorithmic_bytes_per_window=2 * 10000 * 10000 + 2 * 12 * 4 * 10000,
             roofline_GBps=(2 * 10000 * 10000 + 2 * 12 * 4 * 10000) * s5["windows"] / t / 1e9)
        eng5.close()
    if "wtw" in what:
        for W, hop, B, nref in ((700, 350, 8, 2500), (2000, 1000, 4, 5000)):
            ref, lives = synth.synth_batch(nref, B, seed=90 + W)
            eng = wtw.BatchedWTW(torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev), W, hop, B)
            tmax = max(l.shape[1] for l in lives)
            cols = np.zeros((B, tmax, 12))
            for i, l in enumerate(lives):
                cols[i, :l.shape[1]] = l.T
            cols_d = torch.from_numpy(cols).to(dev)
            n_new = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)

            def run():
                eng.reset()
                eng.push(cols_d, n_new, precheck=True)
            t = timed(run, reps=3, warm=1)
            st = eng.states()
            cells = int(sum((int(np.uint32(s[7])) << 32) | int(np.uint32(s[6])) for s in st))
            emit(kernel="rts_wtw_push", streams=B, W=W, hop=hop, windows=int(st[:, 5].sum()), seconds=t,
                 cells_per_s=cells / t)
            eng.close()
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
        emit(kernel="rts_wtw_push (W=100: strip-DP path)", streams=64, W=100, hop=50, frames=frames,
             windows=int(st[:, 5].sum()), seconds=t, frames_per_s=frames / t, cells_per_s=cells / t)


if __name__ == "__main__":
    main()
