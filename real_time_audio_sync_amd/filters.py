"""Constant tables of the chroma front end, built once on the host and uploaded to the device.

``chroma_filterbank`` restates the published algorithm of ``librosa.filters.chroma`` (the
reference calls it at chroma.py:69 and wtw.py:39; librosa itself is not a dependency of this
package and is absent from the build image): a Gaussian bump per pitch class around each FFT bin's
fractional chroma position, columns L2-normalised, a Gaussian octave weighting centred on
``ctroct`` octaves above A0/16, rows rolled so that row 0 is C."""
import wave

import numpy as np


def chroma_filterbank(sr=22050, n_fft=4096, n_chroma=12, a440=440.0, ctroct=5.0, octwidth=2.0, base_c=True):
    bins = np.arange(1, n_fft) * (float(sr) / n_fft)
    pos = n_chroma * np.log2(bins / (float(a440) / 16))
    pos = np.concatenate(([pos[0] - 1.5 * n_chroma], pos))          # DC bin: 1.5 octaves below bin 1
    width = np.concatenate((np.maximum(np.diff(pos), 1.0), [1.0]))
    dist = pos[None, :] - np.arange(n_chroma, dtype=np.float64)[:, None]
    half = np.round(n_chroma / 2.0)
    dist = np.remainder(dist + half + 10 * n_chroma, n_chroma) - half
    w = np.exp(-0.5 * (2.0 * dist / width[None, :]) ** 2)
    w = w / np.sqrt((w * w).sum(axis=0, keepdims=True))
    if octwidth is not None:
        w = w * np.exp(-0.5 * ((pos / n_chroma - ctroct) / octwidth) ** 2)[None, :]
    if base_c:
        w = np.roll(w, -3, axis=0)
    return np.ascontiguousarray(w[:, : n_fft // 2 + 1], dtype=np.float64)


def hann_window(n):
    """The symmetric window the reference multiplies every frame by (np.hanning, chroma.py:39,:62)."""
    return np.hanning(n).astype(np.float64)


def load_wav(path):
    """What ``librosa.load(path)`` returns for the recordings this project uses -- PCM16 WAV, any
    channel count, already at the file's native rate: float32 samples scaled by 1/32768, channels
    averaged.  No resampling: the reference asserts fs == 22050 right after loading
    (chroma.py:28, wtw.py:24), and so do the callers here."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError("only 16-bit PCM WAV files are supported")
        fs = w.getframerate()
        nch = w.getnchannels()
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2")
    if fs != 22050:
        # librosa.load(path) would resample to 22 050 Hz here (chroma.py:27, wtw.py:23); its resampler is a
        # third-party dependency the reference does not pin, so it is not reproduced: refuse instead of drifting
        raise ValueError("%s is sampled at %d Hz; this build takes 22 050 Hz input only (the reference relies on "
                         "librosa.load's resampler for other rates -- resample the file first)" % (path, fs))
    x = raw.reshape(-1, nch).astype(np.float32) / np.float32(32768.0)
    y = x.mean(axis=1, dtype=np.float32) if nch > 1 else x[:, 0]
    return np.ascontiguousarray(y, dtype=np.float32), fs
