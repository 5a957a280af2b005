"""Drop-in for the reference's chroma.py (module-level API and constants kept):
``wav_to_chroma(path)``, ``wav_to_chroma_col(buf)``, ``create_stft(wav)``,
``create_chroma(ft, normalize=True)``, ``wav_to_chroma_diff(path)``; arrays are feature-major
(12, M) / (2049, M) like the reference's.  Computation: csrc/chroma.hip."""
import ctypes

import numpy as np
import torch

from . import _native as nat
from . import filters

# globals (chroma.py:20-22)
fft_len = 4096
hop_size = 2048
fs = 22050


class ChromaPlan(object):
    """Device-resident window / twiddles / filterbank for one (fft_len, hop, fs)."""

    def __init__(self, fft_len=fft_len, hop=hop_size, fs=fs, device="cuda:0"):
        if not torch.cuda.is_available():
            raise RuntimeError("the chroma kernels need a ROCm GPU (no CPU fallback)")
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.fft_len, self.hop, self.fs = int(fft_len), int(hop), int(fs)
        self.n_bins = self.fft_len // 2 + 1
        self.chromafb = filters.chroma_filterbank(fs, self.fft_len)
        win = filters.hann_window(self.fft_len)
        h = ctypes.c_void_p()
        nat.check(nat.lib.rts_chroma_create(self.fft_len, self.hop, win.ctypes.data, self.chromafb.ctypes.data,
                                            ctypes.byref(h)))
        self._h = h

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            nat.destroy_on(self.device, nat.lib.rts_chroma_destroy, h)

    __del__ = close

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def num_frames(self, n_samples, pad_left):
        return int(nat.lib.rts_chroma_num_frames(int(n_samples), self.fft_len, self.hop, int(pad_left)))

    @nat.on_device
    def frames(self, samples_dev, pad_left, normalize=True, out_dtype=torch.float64, want_stft=False,
               want_chroma=True, n_frames=None):
        """samples_dev: 1-D float32/float64 device tensor.  Returns (chroma [M][12] or None,
        stft [M][n_bins] complex128 or None), asynchronous on the current stream."""
        assert samples_dev.dim() == 1 and samples_dev.is_contiguous()
        n = samples_dev.numel()
        m = self.num_frames(n, pad_left) if n_frames is None else int(n_frames)
        chroma = torch.empty((m, 12), dtype=out_dtype, device=self.device) if want_chroma else None
        stft = torch.empty((m, self.n_bins, 2), dtype=torch.float64, device=self.device) if want_stft else None
        if m > 0:
            nat.check(nat.lib.rts_chroma_frames(
                self._h, samples_dev.data_ptr(), nat.F64 if samples_dev.dtype == torch.float64 else nat.F32, n,
                int(pad_left), m, int(bool(normalize)), chroma.data_ptr() if want_chroma else None,
                nat.F64 if out_dtype == torch.float64 else nat.F32, stft.data_ptr() if want_stft else None,
                self._stream()))
        return chroma, (torch.view_as_complex(stft) if want_stft else None)

    @nat.on_device
    def frames_batch(self, samples_dev, n_samples_dev, n_frames_dev, n_frames_max, pad_left=0, normalize=True,
                     out_dtype=torch.float64):
        """B sample buffers at once: samples_dev [B][stride] float32/64, n_samples_dev / n_frames_dev int32 [B].
        Returns chroma [B][n_frames_max][12] (rows beyond n_frames_dev[b] are left untouched)."""
        assert samples_dev.dim() == 2 and samples_dev.is_contiguous()
        B, stride = samples_dev.shape
        out = torch.zeros((B, int(n_frames_max), 12), dtype=out_dtype, device=self.device)
        if n_frames_max > 0:
            nat.check(nat.lib.rts_chroma_frames_batch(
                self._h, samples_dev.data_ptr(), nat.F64 if samples_dev.dtype == torch.float64 else nat.F32, stride,
                n_samples_dev.data_ptr(), int(pad_left), B, int(n_frames_max), n_frames_dev.data_ptr(),
                int(bool(normalize)), out.data_ptr(), nat.F64 if out_dtype == torch.float64 else nat.F32,
                self._stream()))
        return out

    @nat.on_device
    def project(self, spec_dev, normalize=True, out_dtype=torch.float64):
        """spec_dev: [M][n_bins] float64 power spectrum on the device -> chroma [M][12]."""
        assert spec_dev.dtype == torch.float64 and spec_dev.is_contiguous() and spec_dev.shape[1] == self.n_bins
        m = spec_dev.shape[0]
        out = torch.empty((m, 12), dtype=out_dtype, device=self.device)
        if m > 0:
            nat.check(nat.lib.rts_chroma_project(self._h, spec_dev.data_ptr(), m, int(bool(normalize)),
                                                 out.data_ptr(), nat.F64 if out_dtype == torch.float64 else nat.F32,
                                                 self._stream()))
        return out

    @nat.on_device
    def diff(self, chroma_dev):
        m = chroma_dev.shape[0]
        out = torch.empty((max(m - 1, 0), 12), dtype=chroma_dev.dtype, device=self.device)
        if m >= 2:
            nat.check(nat.lib.rts_chroma_diff(chroma_dev.data_ptr(),
                                              nat.F64 if chroma_dev.dtype == torch.float64 else nat.F32, m,
                                              out.data_ptr(), self._stream()))
        return out


_PLANS = {}


def _plan(device="cuda:0"):
    key = (fft_len, hop_size, fs, str(device))
    if key not in _PLANS:
        _PLANS[key] = ChromaPlan(fft_len, hop_size, fs, device)
    return _PLANS[key]


def _to_dev(x, plan):
    x = np.ascontiguousarray(np.asarray(x))
    if x.dtype not in (np.float32, np.float64):
        x = x.astype(np.float64)
    return torch.from_numpy(x).to(plan.device)


def _load(path_to_wav):
    wav, wav_fs = filters.load_wav(path_to_wav)
    assert (wav_fs == 22050)
    return wav


def wav_to_chroma(path_to_wav):
    plan = _plan()
    chroma, _ = plan.frames(_to_dev(_load(path_to_wav), plan), pad_left=fft_len // 2)
    return chroma.t().contiguous().cpu().numpy()


def wav_to_chroma_col(wav_buf):
    assert (len(wav_buf) == fft_len)
    plan = _plan()
    chroma, _ = plan.frames(_to_dev(np.array(wav_buf), plan), pad_left=0)
    return chroma[0].cpu().numpy()


def create_stft(wav):
    plan = _plan()
    _, stft = plan.frames(_to_dev(wav, plan), pad_left=fft_len // 2, want_stft=True, want_chroma=False)
    return stft.t().contiguous().cpu().numpy()


def create_chroma(ft, normalize=True):
    plan = _plan()
    ft = np.asarray(ft)
    one_col = ft.ndim == 1
    if one_col:
        ft = ft[:, None]
    spec = torch.from_numpy(np.ascontiguousarray((np.abs(ft) ** 2).T.astype(np.float64))).to(plan.device)
    out = plan.project(spec, normalize=normalize).t().contiguous().cpu().numpy()
    return out[:, 0] if one_col else out


def wav_to_chroma_diff(path_to_wav):
    plan = _plan()
    chroma, _ = plan.frames(_to_dev(_load(path_to_wav), plan), pad_left=fft_len // 2)
    return plan.diff(chroma).t().contiguous().cpu().numpy()
