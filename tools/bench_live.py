#!/usr/bin/env python3
"""Serving shape: 64 microphones deliver 1-second buffers to live.LiveSession (rts_live_*: one H2D copy per feed, device
pending buffers, HIP chroma, HIP OTW, nothing read back).  Wall time over many feeds, one JSON object per mode:

  feed_list      feed([array per stream])      -- 64 numpy copies into the pinned staging slot per feed
  feed_block     feed_block(block [64][n])     -- one numpy copy per feed
  staged         staging() + submit()          -- the producer writes the pinned slot itself (here: the samples are
                                                   already there, only the counts are rewritten): what the ingestion
                                                   itself costs, PCIe copy included
each for float32 and PCM16 samples.  frames/s counts chroma frames pushed into the trackers.

    python tools/bench_live.py [seconds_of_audio_per_stream=120]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from real_time_audio_sync_amd import synth
    from real_time_audio_sync_amd.live import LiveSession
    secs = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    B, fs = 64, 22050
    ref, _ = synth.synth_batch(2200, 1, seed=1000)
    rs = np.random.RandomState(5)
    pcm = rs.randint(-20000, 20000, size=(B, 8 * fs), dtype=np.int16)      # 8 s of audio per stream, cycled
    f32 = pcm.astype(np.float32) / np.float32(32768.0)
    for dt_name, audio in (("f32", f32), ("i16", pcm)):
        for mode in ("feed_list", "feed_block", "staged"):
            sess = LiveSession(ref, batch=B, c=500, max_run_count=3)
            if mode == "staged":                      # fill all four staging slots once
                for k in range(4):
                    cv, sv = sess.staging(audio.dtype)
                    cv[:] = fs
                    sv[:B * fs] = audio[:, :fs].reshape(-1)
                    sess.submit(audio.dtype)
                sess.sync()
                sess.reset()

            def one(i):
                blk = audio[:, (i % 8) * fs:(i % 8 + 1) * fs]
                if mode == "feed_list":
                    sess.feed([blk[b] for b in range(B)])
                elif mode == "feed_block":
                    sess.feed_block(blk)
                else:
                    cv, _ = sess.staging(audio.dtype)
                    cv[:] = fs
                    sess.submit(audio.dtype)
            for i in range(3):
                one(i)
            sess.sync()
            t0 = time.perf_counter()
            for i in range(3, secs):
                one(i)
            t_submit = time.perf_counter() - t0
            sess.sync()
            dt = time.perf_counter() - t0
            info = sess.poll()
            frames = int(sess.otw.states()[:, 8].sum())
            n_feeds = secs - 3
            fr = B * n_feeds * fs / 2048.0
            print(json.dumps(dict(mode=mode, samples=dt_name, streams=B, feeds=n_feeds, wall_s=dt, host_submit_s=t_submit,
                                  us_per_feed=dt / n_feeds * 1e6, frames_per_s=fr / dt, realtime_factor=n_feeds / dt,
                                  frames_consumed_total=frames, feeds_done=info["feeds_done"],
                                  h2d_MB_per_feed=B * fs * audio.dtype.itemsize / 1e6,
                                  note="random audio: trackers wander (worst case); ref 2200 frames, c=500")), flush=True)
            sess.close()


if __name__ == "__main__":
    main()
