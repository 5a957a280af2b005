#!/usr/bin/env python3
"""Diagnostic (not shipped): build librtsync with -DRTS_OTW_STAMPS into tools/_diag/ and print
the share of wave-0 cycles each phase of otw_advance_kernel takes on the bench workload."""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DIAG = os.path.join(ROOT, "tools", "_diag")


def build():
    os.makedirs(DIAG, exist_ok=True)
    so = os.path.join(DIAG, "librtsync_diag.so")
    src = [os.path.join(ROOT, "real_time_audio_sync_amd", "csrc", f) for f in ("common.cpp", "otw.hip")]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-fno-fast-math", "-mllvm", "-amdgpu-sched-strategy=max-ilp",  # as _build.PER_SOURCE_FLAGS["otw.hip"]
           "-DRTS_OTW_STAMPS=%s" % os.environ.get("RTS_DIAG_LEVEL", "1"), "-o", so]
    for s in src:
        cmd += ["-x", "hip", s]
    subprocess.check_call(cmd)
    return so


def main():
    so = os.path.join(DIAG, "librtsync_diag.so")
    if "--build" in sys.argv or not os.path.exists(so):
        build()
        if "--build" in sys.argv:
            return
    import torch
    from real_time_audio_sync_amd import synth
    L = ctypes.CDLL(so)
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    L.rts_otw_create.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, ctypes.POINTER(vp)]
    L.rts_otw_run.argtypes = [vp, vp, i32, i32, vp, i32, vp]
    L.rts_otw_set_debug.argtypes = [vp, vp]
    L.rts_otw_set_waves.argtypes = [vp, i32]
    L.rts_otw_read_states.argtypes = [vp, vp, vp]
    B, N, c = 64, 2200, 500
    ref = synth.synth_ref(N, seed=1000)
    lives = [synth.synth_live(ref, seed=1001 + b) for b in range(B)]
    dev = torch.device("cuda:0")
    ref_d = torch.from_numpy(np.ascontiguousarray(ref.T)).float().to(dev)
    tmax = max(l.shape[1] for l in lives)
    buf = torch.zeros((B, tmax, 12), dtype=torch.float32)
    for b, l in enumerate(lives):
        buf[b, :l.shape[1]] = torch.from_numpy(l.T.copy()).float()
    live_d = buf.to(dev)
    len_d = torch.tensor([l.shape[1] for l in lives], dtype=torch.int32, device=dev)
    names = ["barrier 2 wait + plan", "-", "-", "phase A (chain / install)", "-", "barrier 1 wait", "-", "-", "control"]
    for waves in (8,):
        h = vp()
        assert L.rts_otw_create(ref_d.data_ptr(), 0, 12, N, B, c, 3, 0, 0, ctypes.byref(h)) == 0
        L.rts_otw_set_waves(h, waves)
        dbg = torch.zeros((B, 16), dtype=torch.int64, device=dev)
        L.rts_otw_set_debug(h, dbg.data_ptr())
        for _ in range(2):
            L.rts_otw_run(h, live_d.data_ptr(), 0, tmax, len_d.data_ptr(), 0, None)
        torch.cuda.synchronize()
        st = np.zeros((B, 16), dtype=np.int32)
        L.rts_otw_read_states(h, st.ctypes.data, None)
        d = dbg.cpu().numpy().astype(np.float64)
        frames = st[:, 8].sum()
        if os.environ.get("RTS_DIAG_LEVEL", "1") == "2":
            for base, label in ((0, "hit steps"), (3, "other steps")):
                n = max(d[:, base + 2].sum(), 1)
                print("%s: %d  wave-0 work %.0f cycles, end-of-step barrier wait %.0f cycles" % (label, n, d[:, base].sum() / n, d[:, base + 1].sum() / n))
            n = max(d[:, 2].sum(), 1)
            print("hit steps, own work: wave 1 (row speculation) %.0f, wave 2 (column speculation) %.0f, helper waves 3-7: %s"
                  % (d[:, 6].sum() / n, d[:, 7].sum() / n, ", ".join("%.0f" % (d[:, 8 + i].sum() / n) for i in range(5))))
            steps = max(d[:, 2].sum() + d[:, 5].sum(), 1)
            print("extra carry rounds per speculative chain: row %.2f, column %.2f" % (d[:, 13].sum() / steps, d[:, 14].sum() / steps))
            return
        nm = ["barrier-2 wait", "phase A (chain / install) | hit: step entry", "barrier-1 wait", "settle", "decide", "plan + refill"]
        for base, label in ((0, "hit steps"), (8, "other steps")):
            n = d[:, base + 6].sum()
            tot = d[:, base:base + 6].sum()
            print("%s: %d (%.1f per frame), %.0f cycles per step" % (label, n, n / frames, tot / max(n, 1)))
            for i, x in enumerate(nm):
                print("   %-28s %7.0f cycles  %5.1f %%" % (x, d[:, base + i].sum() / max(n, 1), 100 * d[:, base + i].sum() / max(tot, 1)))
        print("band argmin recomputes per frame: %.3f" % (st[:, 15].sum() / frames))


if __name__ == "__main__":
    main()
