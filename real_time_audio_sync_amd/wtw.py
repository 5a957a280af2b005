"""Drop-in for the reference's wtw.py: ``WTW(ref_recording, params, debug_params)`` with
``insert(list_of_samples)`` (reference: wtw.py:19-240), plus ``BatchedWTW`` for many streams.

Host side: only the pending-sample buffer (wtw.py:73, :81-83).  Device side: reference chroma and
per-hop live chroma (csrc/chroma.hip), window cost / DP / backtrack / hand-over (csrc/wtw.hip)."""
import ctypes

import numpy as np
import torch

from . import _native as nat
from . import filters
from .chroma import ChromaPlan


class BatchedWTW(object):
    """B live streams against one reference chroma.  ``chroma_ref_dev``: device tensor [M][12] float64."""

    def __init__(self, chroma_ref_dev, win_frames, hop_frames, batch=1, keep_last_d=False):
        assert chroma_ref_dev.dtype == torch.float64 and chroma_ref_dev.is_contiguous()
        self.device = chroma_ref_dev.device
        torch.cuda.set_device(self.device)
        self.ref = chroma_ref_dev
        self.M = chroma_ref_dev.shape[0]
        self.B, self.W, self.hopf = int(batch), int(win_frames), int(hop_frames)
        h = ctypes.c_void_p()
        nat.check(nat.lib.rts_wtw_create(self.ref.data_ptr(), 12, self.M, self.B, self.W, self.hopf,
                                         int(bool(keep_last_d)), ctypes.byref(h)))
        self._h = h
        self._keep_d = keep_last_d

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            nat.destroy_on(self.device, nat.lib.rts_wtw_destroy, h)

    __del__ = close

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    @nat.on_device
    def reset(self):
        nat.check(nat.lib.rts_wtw_reset(self._h, self._stream()))

    @nat.on_device
    def push(self, cols_dev, n_new_dev=None, precheck=True):
        """cols_dev: [B][n_max][12] float32/float64 device tensor of new live chroma columns."""
        assert cols_dev.is_contiguous() and cols_dev.shape[0] == self.B and cols_dev.shape[2] == 12
        nat.check(nat.lib.rts_wtw_push(self._h, cols_dev.data_ptr(),
                                       nat.F64 if cols_dev.dtype == torch.float64 else nat.F32,
                                       int(cols_dev.shape[1]), n_new_dev.data_ptr() if n_new_dev is not None else None,
                                       int(bool(precheck)), self._stream()))

    @nat.on_device
    def precheck(self):
        nat.check(nat.lib.rts_wtw_push(self._h, None, nat.F64, 0, None, 1, self._stream()))

    @nat.on_device
    def states(self):
        out = np.zeros((self.B, nat.WTW_STATE_LEN), dtype=np.int32)
        nat.check(nat.lib.rts_wtw_read_states(self._h, out.ctypes.data, self._stream()))
        return out

    def state(self, b=0):
        s = self.states()[b]
        return dict(chroma_ptr=int(s[0]), live_ptr=int(s[1]), ref_ptr=int(s[2]), status=int(s[3]), n_path=int(s[4]),
                    windows=int(s[5]), cells=(int(np.uint32(s[7])) << 32) | int(np.uint32(s[6])))

    @nat.on_device
    def path(self, b=0):
        n = ctypes.c_int(0)
        nat.check(nat.lib.rts_wtw_read_path(self._h, b, None, 0, ctypes.byref(n), self._stream()))
        out = np.empty((n.value, 2), dtype=np.int32)
        if n.value:
            nat.check(nat.lib.rts_wtw_read_path(self._h, b, out.ctypes.data, n.value, ctypes.byref(n), self._stream()))
        return out

    @nat.on_device
    def last_d(self, b=0):
        """The last window's accumulated-cost matrix D (W, W) float64 (needs keep_last_d=True)."""
        out = np.empty((self.W, self.W), dtype=np.float64)
        nat.check(nat.lib.rts_wtw_read_last_d(self._h, b, out.ctypes.data, self._stream()))
        return out


class WTW():
    """``WTW(ref_recording, {'fft_len','hop_size','dtw_win_size','dtw_hop_size'}, debug_params)``;
    ``insert(list_of_float_samples)`` returns None or "stop"; ``.path`` is a list of (live, ref)."""

    def __init__(self, ref_recording, params, debug_params, device="cuda:0"):
        # reference audio, fs = 22050 (wtw.py:23-24)
        self.ref, self.fs = filters.load_wav(ref_recording)
        assert (self.fs == 22050)
        self._init_from_samples(self.ref, params, debug_params, device)

    @classmethod
    def from_samples(cls, ref_samples, params, debug_params=None, fs=22050, device="cuda:0"):
        """Same object from already-loaded mono samples (what librosa.load would have returned)."""
        self = cls.__new__(cls)
        self.ref, self.fs = np.asarray(ref_samples), fs
        self._init_from_samples(self.ref, params, debug_params or {}, device)
        return self

    def _init_from_samples(self, ref, params, debug_params, device):
        self.fft_len = params['fft_len']
        self.hop_size = params['hop_size']
        self.dtw_win_size = params['dtw_win_size']
        self.dtw_hop_size = params['dtw_hop_size']
        self.chroma_info = debug_params.get('chroma', False) if isinstance(debug_params, dict) else False
        self._plan = ChromaPlan(self.fft_len, self.hop_size, self.fs, device)
        self.chromafb = self._plan.chromafb
        dev = self._plan.device
        ref_dev = torch.from_numpy(np.ascontiguousarray(ref)).to(dev)
        chroma_ref_dev, _ = self._plan.frames(ref_dev, pad_left=self.fft_len // 2)  # wtw.py:37-41
        self._chroma_ref_dev = chroma_ref_dev
        self.chroma_ref = chroma_ref_dev.t().contiguous().cpu().numpy()
        self.N = self.chroma_ref.shape[1] * 2
        self.M = self.chroma_ref.shape[1]
        self._eng = BatchedWTW(chroma_ref_dev, self.dtw_win_size // self.hop_size, self.dtw_hop_size // self.hop_size, 1)
        self._pending = np.zeros(0, dtype=np.float64)  # self.buf (wtw.py:61)

    def insert(self, live_audio_buf):
        # store incoming music (wtw.py:73)
        self._pending = np.concatenate((self._pending, np.asarray(live_audio_buf, dtype=np.float64)))
        L, H = self.fft_len, self.hop_size
        n_cols = (len(self._pending) - L) // H + 1 if len(self._pending) >= L else 0
        if n_cols > 0:
            used = (n_cols - 1) * H + L
            dev = self._plan.device
            samples = torch.from_numpy(self._pending[:used]).to(dev)
            cols, _ = self._plan.frames(samples, pad_left=0, n_frames=n_cols)  # wtw.py:81-90, all hops at once
            self._eng.push(cols.unsqueeze(0).contiguous(), precheck=True)
            self._pending = self._pending[n_cols * H:]  # wtw.py:83
        else:
            self._eng.precheck()  # wtw.py:76-77
        if int(self._eng.states()[0, 3]) != nat.RUNNING:
            return "stop"
        return None

    @property
    def path(self):
        return [(int(x), int(y)) for x, y in self._eng.path(0)]

    @property
    def live_ptr(self):
        return self._eng.state(0)["live_ptr"]

    @property
    def ref_ptr(self):
        return self._eng.state(0)["ref_ptr"]

    @property
    def chroma_ptr(self):
        return self._eng.state(0)["chroma_ptr"]
