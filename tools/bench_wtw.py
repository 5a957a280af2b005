#!/usr/bin/env python3
"""WTW push timings, 64 streams against a 2200-frame reference, per window size and kernel path (A/B in one process):
default (wtw_win_kernel for W <= 128, strip DP above) vs RTS_WTW_WIN=0 (anti-diagonal sweep up to 64 frames, strip DP
above).  Device time from HIP events; one JSON object per line.   python tools/bench_wtw.py [W ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def eng_lib():
    from real_time_audio_sync_amd import _native
    return _native.lib


def main():
    import torch
    from real_time_audio_sync_amd import synth, wtw
    dev = torch.device("cuda:0")
    Ws = [int(a) for a in sys.argv[1:]] or [20, 64, 100, 128]
    ref, lives = synth.synth_batch(2200, 64, seed=3)
    refd = torch.from_numpy(np.ascontiguousarray(ref.T)).to(dev)
    tmax = max(l.shape[1] for l in lives)
    cols = np.zeros((64, tmax, 12))
    for i, l in enumerate(lives):
        cols[i, :l.shape[1]] = l.T
    cols_d = torch.from_numpy(cols).to(dev)
    n_new = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)
    for W in Ws:
        paths = {}
        for mode in ("default", "RTS_WTW_WIN=0"):
            if mode != "default":
                os.environ["RTS_WTW_WIN"] = "0"
            eng = wtw.BatchedWTW(refd, W, max(W // 2, 1), 64)
            os.environ.pop("RTS_WTW_WIN", None)
            ts = []
            for rep in range(7):
                eng.reset()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.push(cols_d, n_new, precheck=True)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
            st = eng.states()
            paths[mode] = [eng.path(b) for b in (0, 31, 63)]
            windows = int(st[:, 5].sum())
            ms = float(np.median(ts[2:]))
            print(json.dumps(dict(W=W, hop=max(W // 2, 1), mode=mode, ms=ms, windows=windows, us_per_window_per_stream=ms * 1e3 / (windows / 64.0),
                                  frames_per_s=int(st[:, 0].sum()) / (ms * 1e-3))), flush=True)
            if mode == "default" and hasattr(eng_lib(), "rts_wtw_read_win_stamps"):
                import ctypes
                buf = (ctypes.c_longlong * 8)()
                eng_lib().rts_wtw_read_win_stamps(buf)
                wn = max(buf[5], 1)
                print(json.dumps(dict(W=W, stamps_cycles_per_window=dict(phase_a_wave0=buf[0] / wn, running_sums=buf[1] / wn, dp=buf[2] / wn,
                                                                          walk_handover_wave0=buf[3] / wn, costs_wave1=buf[4] / wn,
                                                                          dp_blocks_own_wave0=buf[6] / wn, dp_blocks_own_wave1=buf[7] / wn), windows_stream0=int(buf[5]))), flush=True)
            eng.close()
        assert all(np.array_equal(a, b) for a, b in zip(paths["default"], paths["RTS_WTW_WIN=0"])), W


if __name__ == "__main__":
    main()
