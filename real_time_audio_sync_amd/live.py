"""Many live audio streams followed against one reference, entirely on the device.

This is the batched form of the reference's microphone loop (livenote_live.py:161-209): audio
arrives in buffers of arbitrary size; whenever a stream has at least ``fft_len`` pending samples,
every complete hop becomes a chroma column (un-padded framing, chroma.py:35-42) and is inserted
into that stream's online-time-warping state; ``hop_size`` samples are dropped per column
(livenote_live.py:208).  Host code only tracks how many samples are pending per stream."""
import numpy as np
import torch

from . import _native as nat
from .chroma import ChromaPlan
from .otw_batch import BatchedOTW


class LiveSession(object):
    def __init__(self, ref_chroma, batch, c=500, max_run_count=3, variant="otw", fft_len=4096, hop_size=2048,
                 fs=22050, max_pending=1 << 16, device="cuda:0"):
        """``ref_chroma``: (12, N) reference chroma (e.g. chroma.wav_to_chroma(ref_path))."""
        self.plan = ChromaPlan(fft_len, hop_size, fs, device)
        self.dev = self.plan.device
        self.otw = BatchedOTW(np.asarray(ref_chroma, dtype=np.float64), c, max_run_count, batch=batch,
                              variant=variant, device=device, dtype=torch.float64)
        self.B, self.L, self.H = int(batch), int(fft_len), int(hop_size)
        self.cap = int(max_pending)
        self.buf = torch.zeros((self.B, self.cap), dtype=torch.float32, device=self.dev)
        self.pending = np.zeros(self.B, dtype=np.int64)

    def feed(self, buffers):
        """``buffers``: one array of new samples per stream (None / empty = nothing new).  Returns the
        list of stream indices that have reached the end of the reference ("stop")."""
        assert len(buffers) == self.B
        for b, x in enumerate(buffers):
            if x is None or len(x) == 0:
                continue
            x = torch.as_tensor(np.asarray(x, dtype=np.float32))
            n = x.numel()
            if self.pending[b] + n > self.cap:
                raise ValueError("stream %d: more than max_pending=%d samples pending" % (b, self.cap))
            self.buf[b, self.pending[b]:self.pending[b] + n] = x.to(self.dev)
            self.pending[b] += n
        n_cols = np.where(self.pending >= self.L, (self.pending - self.L) // self.H + 1, 0).astype(np.int32)
        n_max = int(n_cols.max())
        if n_max > 0:
            ns = torch.from_numpy(self.pending.astype(np.int32)).to(self.dev)
            nf = torch.from_numpy(n_cols).to(self.dev)
            cols = self.plan.frames_batch(self.buf, ns, nf, n_max, pad_left=0)   # livenote_live.py:186
            self.otw.push(cols, nf)                                               # livenote_live.py:187
            for b in np.nonzero(n_cols)[0]:                                       # livenote_live.py:208
                used = int(n_cols[b]) * self.H
                rem = int(self.pending[b]) - used
                if rem > 0:
                    self.buf[b, :rem] = self.buf[b, used:used + rem].clone()
                self.pending[b] = rem
        st = self.otw.states()
        return [int(b) for b in np.nonzero(st[:, nat.ST_STATUS] == nat.STOP_REF_END)[0]]

    def path(self, b=0):
        return self.otw.path(b)

    def position(self, b=0):
        """(live_frame, ref_frame) of stream b's latest path point, or None."""
        p = self.otw.path(b)
        return (int(p[-1, 0]), int(p[-1, 1])) if len(p) else None

    def close(self):
        self.otw.close()
        self.plan.close()
