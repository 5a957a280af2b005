"""numpy restatement of the reference's chroma front end -- TEST INFRASTRUCTURE ONLY.

Follows /root/reference/chroma.py (create_stft :44-65, create_chroma :67-75, wav_to_chroma_col
:35-42, wav_to_chroma_diff :77-90) and the audio half of wtw.WTW (wtw.py:21-41, :71-93), using
the same numpy primitives the reference calls (np.hanning, np.fft.rfft, np.dot).

Third-party arithmetic that is NOT in /root/reference: **librosa** (absent from this image, version
unpinned by the reference -- no requirements file; era April 2018 => 0.6.0).  Its three functions
on this path are restated from the library's published algorithm:
  * librosa.filters.chroma(sr, n_fft) -> chroma_filterbank()   (call sites chroma.py:69, wtw.py:39)
  * librosa.util.normalize(S, norm=2, axis=0) -> l2_normalize_columns()   (chroma.py:74, wtw.py:41,:90)
  * librosa.load(path) for 22 050 Hz PCM16 WAVs -> load_wav_mono()   (chroma.py:27, wtw.py:23)
Chroma *values* are therefore parity-unpinned by the reference (it stores no chroma anywhere);
the chain chroma -> WTW is pinned at path level by the reference's one reproducible known-answer
file, Songs/chopin/tests/wtw_test_20b.txt (tests/test_oracle_golden.py::test_wtw_known_answer).
"""
import wave

import numpy as np

FFT_LEN = 4096   # chroma.py:20
HOP_SIZE = 2048  # chroma.py:21
FS = 22050       # chroma.py:22


def chroma_filterbank(sr=FS, n_fft=FFT_LEN, n_chroma=12, a440=440.0, ctroct=5.0, octwidth=2.0,
                      base_c=True, dtype=np.float64):
    """librosa.filters.chroma (defaults norm=2): Gaussian bumps per pitch class around each FFT
    bin's (fractional) chroma position, column L2-normalised, octave-weighted, rolled so row 0 = C.
    Returns (n_chroma, 1 + n_fft//2)."""
    freqs = np.linspace(0, sr, n_fft, endpoint=False)[1:]
    frqbins = n_chroma * np.log2(freqs / (float(a440) / 16))
    # bin 0 (DC) gets a made-up position 1.5 octaves below bin 1
    frqbins = np.concatenate(([frqbins[0] - 1.5 * n_chroma], frqbins))
    binwidth = np.concatenate((np.maximum(frqbins[1:] - frqbins[:-1], 1.0), [1]))
    D = np.subtract.outer(frqbins, np.arange(0, n_chroma, dtype="d")).T
    half = np.round(float(n_chroma) / 2)
    D = np.remainder(D + half + 10 * n_chroma, n_chroma) - half
    wts = np.exp(-0.5 * (2 * D / np.tile(binwidth, (n_chroma, 1))) ** 2)
    wts = l2_normalize_columns(wts)
    if octwidth is not None:
        wts *= np.tile(np.exp(-0.5 * (((frqbins / n_chroma - ctroct) / octwidth) ** 2)), (n_chroma, 1))
    if base_c:
        wts = np.roll(wts, -3, axis=0)
    return np.ascontiguousarray(wts[:, : int(1 + n_fft / 2)], dtype=dtype)


def l2_normalize_columns(S):
    """librosa.util.normalize(S, norm=2, axis=0): columns with norm below the smallest positive
    normal float are left unscaled."""
    S = np.asarray(S, dtype=np.float64)
    length = np.sqrt(np.sum(np.abs(S) ** 2, axis=0, keepdims=True))
    length = np.where(length < np.finfo(S.dtype).tiny, 1.0, length)
    return S / length


def load_wav_mono(path):
    """librosa.load(path) restricted to what the reference's WAVs need: PCM16 at 22 050 Hz,
    scaled by 1/32768, channels averaged, float32; no resampling."""
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2, "PCM16 only"
        fs = w.getframerate()
        nch = w.getnchannels()
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    x = raw.reshape(-1, nch).astype(np.float32) / np.float32(32768.0)
    y = x.mean(axis=1, dtype=np.float32) if nch > 1 else x[:, 0]
    return np.ascontiguousarray(y, dtype=np.float32), fs


def create_stft(wav, fft_len=FFT_LEN, hop_size=HOP_SIZE):
    """chroma.py:44-65 (== wtw.py:137-160): zero-pad L/2 at the start, symmetric Hann, rfft."""
    L, H = fft_len, hop_size
    x = np.concatenate((np.zeros(L // 2), wav))
    n_hops = int((len(x) - L) // H + 1)
    out = np.empty((1 + L // 2, n_hops), dtype=complex)
    win = np.hanning(L)
    for m in range(n_hops):
        out[:, m] = np.fft.rfft(x[m * H: m * H + L] * win)
    return out


_FB_CACHE = {}


def _fb(fs, fft_len):
    key = (fs, fft_len)
    if key not in _FB_CACHE:
        _FB_CACHE[key] = chroma_filterbank(fs, fft_len)
    return _FB_CACHE[key]


def create_chroma(ft, normalize=True, fs=FS, fft_len=FFT_LEN):
    """chroma.py:67-75."""
    spec = np.abs(ft) ** 2
    raw = np.dot(_fb(fs, fft_len), spec)
    return l2_normalize_columns(raw) if normalize else raw


def wav_to_chroma(samples):
    """chroma.py:25-33 from already-loaded mono samples -> (12, M) float64."""
    return create_chroma(create_stft(np.asarray(samples)))


def wav_to_chroma_col(buf):
    """chroma.py:35-42: one un-padded 4096-sample buffer -> (12,)."""
    buf = np.asarray(buf, dtype=np.float64)
    assert len(buf) == FFT_LEN
    return create_chroma(np.fft.rfft(buf * np.hanning(len(buf))))


def wav_to_chroma_diff(samples):
    """chroma.py:77-90: half-wave rectified temporal difference, not renormalised."""
    return np.clip(np.diff(wav_to_chroma(samples)), 0, float("inf"))


class WtwAudioOracle:
    """wtw.WTW from raw audio (wtw.py:21-128): reference chroma in the constructor, per-hop live
    chroma (no zero-pad) in insert(), window DP delegated to the C restatement."""

    def __init__(self, ref_samples, params, fs=FS):
        from . import binding
        self.fft_len = params["fft_len"]
        self.hop_size = params["hop_size"]
        self.win_frames = params["dtw_win_size"] // self.hop_size
        self.hop_frames = params["dtw_hop_size"] // self.hop_size
        self.fs = fs
        stft_ref = create_stft(np.asarray(ref_samples), self.fft_len, self.hop_size)
        self.chromafb = _fb(fs, self.fft_len)
        self.chroma_ref = l2_normalize_columns(np.dot(self.chromafb, np.abs(stft_ref) ** 2))
        self._core = binding.WtwOracle(self.chroma_ref, self.win_frames, self.hop_frames)
        self._win = np.hanning(self.fft_len)
        self.buf = []
        self.chroma_live_cols = []

    def insert(self, samples):
        from . import binding
        self.buf += list(samples)
        if self._core.insert_precheck() == binding.STOP_REF_END:
            return "stop"
        while len(self.buf) >= self.fft_len:
            section = np.array(self.buf[: self.fft_len])
            self.buf = self.buf[self.hop_size:]
            spec = np.abs(np.fft.rfft(section * self._win)) ** 2
            col = l2_normalize_columns(np.dot(self.chromafb, spec)[:, None])[:, 0]
            self.chroma_live_cols.append(col)
            if self._core.push_col(col) != binding.RUNNING:
                return "stop"
        return None

    @property
    def path(self):
        return self._core.path
