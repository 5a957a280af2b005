"""Many live audio streams followed against one reference, entirely on the device.

This is the batched form of the reference's microphone loops (livenote_live.py:161-209 for the OTW family,
wtw.py:71-93 for WTW): audio arrives in buffers of arbitrary size; whenever a stream has at least ``fft_len``
pending samples, every complete hop becomes a chroma column (un-padded framing, chroma.py:35-42) and is inserted
into that stream's alignment state; ``hop_size`` samples are dropped per column (livenote_live.py:208).

All of it happens behind ``rts_live_*`` (csrc/live.hip): per feed ONE host-to-device copy from a pinned staging slot
and a fixed chain of launches, nothing read back -- the pending samples live in per-stream device buffers, and the
device publishes status and position of every stream into host-mapped memory, which ``poll()`` reads without
touching the stream.  The host only mirrors the pending-sample counts (integer arithmetic)."""
import ctypes

import numpy as np
import torch

from . import _native as nat
from .chroma import ChromaPlan
from .otw_batch import BatchedOTW

_KINDS = {np.dtype(np.float32): nat.F32, np.dtype(np.int16): nat.I16}


class LiveSession(object):
    def __init__(self, ref_chroma, batch, c=500, max_run_count=3, variant="otw", fft_len=4096, hop_size=2048,
                 fs=22050, max_pending=1 << 16, device="cuda:0", wtw_params=None):
        """``ref_chroma``: (12, N) reference chroma (e.g. chroma.wav_to_chroma(ref_path)).  With ``wtw_params``
        ({'dtw_win_size', 'dtw_hop_size'} in samples, like wtw.py:29-30) the streams are followed by windowed time
        warping instead of ``variant`` ('otw' | 'livenote' | 'livenote_v2')."""
        self.plan = ChromaPlan(fft_len, hop_size, fs, device)
        self.dev = self.device = self.plan.device
        self.B, self.L, self.H = int(batch), int(fft_len), int(hop_size)
        self.cap = int(max_pending)
        ref = np.asarray(ref_chroma, dtype=np.float64)
        self.otw = self.wtw = None
        if wtw_params is None:
            self.otw = BatchedOTW(ref, c, max_run_count, batch=batch, variant=variant, device=device, dtype=torch.float64)
        else:
            from .wtw import BatchedWTW
            self._ref_dev = torch.from_numpy(np.ascontiguousarray(ref.T)).to(self.dev)
            self.wtw = BatchedWTW(self._ref_dev, wtw_params['dtw_win_size'] // self.H, wtw_params['dtw_hop_size'] // self.H,
                                  batch)
        h = ctypes.c_void_p()
        with torch.cuda.device(self.dev):
            nat.check(nat.lib.rts_live_create(self.plan._h, self.otw._h if self.otw else None,
                                              self.wtw._h if self.wtw else None, self.B, self.cap, ctypes.byref(h)))
        self._h = h
        self._status = np.zeros(self.B, dtype=np.int32)
        self._pos = np.zeros((self.B, 2), dtype=np.int32)

    # ---- feeding ----------------------------------------------------------------------------------------------
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def staging(self, dtype=np.float32):
        """The next pinned staging slot as numpy views: (counts int32 [B], samples `dtype` [B * max_pending]).  Write
        the new sample count of every stream and the samples of all streams packed back to back in stream order, then
        call ``submit(dtype)``.  A producer that writes here directly (an audio callback, a socket reader) saves the
        copy ``feed`` makes."""
        counts, samples, capn = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_longlong()
        nat.check(nat.lib.rts_live_staging(self._h, ctypes.byref(counts), ctypes.byref(samples), ctypes.byref(capn)))
        dt = np.dtype(dtype)
        cv = np.ctypeslib.as_array(ctypes.cast(counts.value, ctypes.POINTER(ctypes.c_int32)), shape=(self.B,))
        ctype = ctypes.c_float if dt == np.float32 else ctypes.c_int16
        sv = np.ctypeslib.as_array(ctypes.cast(samples.value, ctypes.POINTER(ctype)), shape=(capn.value,))
        return cv, sv

    @nat.on_device
    def submit(self, dtype=np.float32):
        """Enqueue the feed written into the staging slot (asynchronous; nothing is read back)."""
        nat.check(nat.lib.rts_live_submit(self._h, _KINDS[np.dtype(dtype)], self._stream()))

    def feed_block(self, block):
        """``block``: (B, n) float32 or int16 -- the same number of new samples for every stream (a multi-channel
        interface): one copy into the staging slot, one submit."""
        block = np.asarray(block)
        assert block.ndim == 2 and block.shape[0] == self.B and block.dtype in _KINDS
        cv, sv = self.staging(block.dtype)
        cv[:] = block.shape[1]
        sv[:block.size] = block.reshape(-1)
        self.submit(block.dtype)

    def feed(self, buffers, wait=False):
        """``buffers``: one array of new samples per stream (None / empty = nothing new).  Asynchronous: returns the
        streams known to have reached the end of the reference ("stop", livenote_live.py:188-190) according to what
        the device has published so far -- the result of this very feed shows up in a later call, or at once with
        ``wait=True`` (which synchronises the stream)."""
        assert len(buffers) == self.B
        dt = np.int16 if all(x is None or np.asarray(x).dtype == np.int16 for x in buffers) else np.float32
        cv, sv = self.staging(dt)
        off = 0
        for b, x in enumerate(buffers):
            n = 0 if x is None else len(x)
            cv[b] = n
            if n:
                sv[off:off + n] = x
                off += n
        self.submit(dt)
        if wait:
            self.sync()
        return self.stopped()

    # ---- results ----------------------------------------------------------------------------------------------
    def poll(self):
        """Non-blocking: {'status' [B], 'positions' [B][2] (live frame, reference frame), 'feeds_done',
        'feeds_submitted'} as last published by the device."""
        done, sub = ctypes.c_int(), ctypes.c_int()
        nat.check(nat.lib.rts_live_poll(self._h, self._status.ctypes.data, self._pos.ctypes.data, ctypes.byref(done),
                                        ctypes.byref(sub)))
        return dict(status=self._status.copy(), positions=self._pos.copy(), feeds_done=done.value, feeds_submitted=sub.value)

    def stopped(self):
        st = self.poll()["status"]
        return [int(b) for b in np.nonzero(st == nat.STOP_REF_END)[0]]

    def pending(self):
        out = np.zeros(self.B, dtype=np.int64)
        nat.check(nat.lib.rts_live_pending(self._h, out.ctypes.data))
        return out

    def sync(self):
        torch.cuda.current_stream(self.dev).synchronize()

    @nat.on_device
    def reset(self):
        nat.check(nat.lib.rts_live_reset(self._h, self._stream()))

    def path(self, b=0):
        return (self.otw or self.wtw).path(b)

    def position(self, b=0):
        """(live_frame, ref_frame) of stream b's latest path point, or None."""
        p = self.path(b)
        return (int(p[-1, 0]), int(p[-1, 1])) if len(p) else None

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                torch.cuda.synchronize(self.dev)
            except Exception:
                pass
            nat.destroy_on(self.dev, nat.lib.rts_live_destroy, h)
        for o in (getattr(self, "otw", None), getattr(self, "wtw", None), getattr(self, "plan", None)):
            if o is not None:
                o.close()

    __del__ = close
