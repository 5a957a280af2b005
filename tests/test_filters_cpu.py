"""The chroma filterbank table (host constant uploaded to the device): the product's builder
(real_time_audio_sync_amd/filters.py) against the oracle's independent restatement of
librosa.filters.chroma, plus properties that follow from the published algorithm.  librosa itself is
absent from this image and unpinned by the reference, so there is no reference-generated fixture;
the path-level pin is the WTW known-answer test."""
import numpy as np

from oracle import chroma_oracle
from real_time_audio_sync_amd import filters


def test_filterbank_matches_oracle_and_structure():
    for n_fft in (4096, 2048, 256):
        fb = filters.chroma_filterbank(22050, n_fft)
        ofb = chroma_oracle.chroma_filterbank(22050, n_fft)
        assert fb.shape == ofb.shape == (12, n_fft // 2 + 1) and fb.dtype == np.float64
        assert np.abs(fb - ofb).max() <= 4 * np.finfo(np.float64).eps      # two codings of the same formula
        assert (fb >= 0).all() and np.isfinite(fb).all()
    fb = filters.chroma_filterbank(22050, 4096)
    # per-bin L2 norm over the 12 pitch classes equals the Gaussian octave weight exp(-0.5 ((oct - 5)/2)^2)
    k = np.arange(1, 2049)
    octs = np.log2(k * 22050.0 / 4096 / (440.0 / 16))
    expect = np.exp(-0.5 * ((octs - 5.0) / 2.0) ** 2)
    assert np.allclose(np.sqrt((fb[:, 1:] ** 2).sum(axis=0)), expect, rtol=1e-12)
    # base_c: the bin nearest A4 = 440 Hz peaks in row 9 (C=0, ..., A=9); middle C (261.63 Hz) in row 0
    a4 = int(round(440.0 * 4096 / 22050))
    c4 = int(round(261.6256 * 4096 / 22050))
    assert fb[:, a4].argmax() == 9 and fb[:, c4].argmax() == 0


def test_window_and_wav_loader(tmp_path):
    import wave
    assert np.array_equal(filters.hann_window(4096), np.hanning(4096))
    # 16-bit stereo -> float32 mono (L + R) / 65536, like librosa.load on the project's recordings
    rs = np.random.RandomState(0)
    pcm = rs.randint(-32768, 32767, size=(1000, 2)).astype("<i2")
    path = str(tmp_path / "x.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(2)
        w.setsampwidth(2)
        w.setframerate(22050)
        w.writeframes(pcm.tobytes())
    y, fs = filters.load_wav(path)
    oy, ofs = chroma_oracle.load_wav_mono(path)
    assert fs == ofs == 22050 and y.dtype == np.float32
    assert np.array_equal(y, oy)
    assert np.array_equal(y, (pcm.astype(np.int32).sum(axis=1) / 65536.0).astype(np.float32))
