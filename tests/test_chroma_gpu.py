"""HIP chroma front end (csrc/chroma.hip through the C-ABI) against the numpy oracle
(oracle/chroma_oracle.py, which uses the reference's own primitives np.hanning / np.fft.rfft /
np.dot).  Everything is float64; the kernels' FFT adds in a different order than numpy's pocketfft,
so values are compared with tolerances:
    STFT bins      |delta| <= 1e-11 * max|X|  per frame   (observed ~1e-15)
    chroma (unit-norm columns)  |delta| <= 1e-11          (observed ~1e-15)
Chroma values are parity-unpinned by the reference itself (librosa version unpinned, no stored
chroma); the path-level pin is tests/test_wtw_gpu.py::test_wtw_known_answer_on_gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

STFT_RTOL = 1e-11
CHROMA_ATOL = 1e-11


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(scope="module")
def mods():
    from real_time_audio_sync_amd import chroma
    from oracle import chroma_oracle
    return chroma, chroma_oracle


def test_create_stft_and_chroma_on_real_audio(mods, chopin_audio, otw_golden):
    chroma, co = mods
    wav = chopin_audio["ref"]
    ft = chroma.create_stft(wav)
    oft = co.create_stft(wav)
    assert ft.shape == oft.shape == (2049, 380) and ft.dtype == np.complex128
    scale = np.abs(oft).max(axis=0, keepdims=True)
    assert (np.abs(ft - oft) <= STFT_RTOL * scale).all()
    ch = chroma.create_chroma(ft)
    assert ch.shape == (12, 380)
    assert np.abs(ch - otw_golden["G/ref"]).max() <= CHROMA_ATOL
    raw = chroma.create_chroma(oft, normalize=False)
    oraw = co.create_chroma(oft, normalize=False)
    assert np.abs(raw - oraw).max() <= 1e-11 * np.abs(oraw).max()


def test_fused_wav_to_chroma_matches_oracle(mods, chopin_audio, otw_golden):
    chroma, co = mods
    plan = chroma._plan()
    for key, gold in (("ref", "G/ref"), ("live", "G/live")):
        dev = torch.from_numpy(chopin_audio[key]).to(plan.device)
        ch, _ = plan.frames(dev, pad_left=chroma.fft_len // 2)
        ch = ch.t().cpu().numpy()
        assert ch.shape == otw_golden[gold].shape
        assert np.abs(ch - otw_golden[gold]).max() <= CHROMA_ATOL
        assert np.allclose(np.sqrt((ch ** 2).sum(axis=0)), 1.0, atol=1e-12)
        d = plan.diff(plan.frames(dev, pad_left=chroma.fft_len // 2)[0]).t().cpu().numpy()
        od = np.clip(np.diff(otw_golden[gold]), 0, np.inf)
        assert d.shape == od.shape and np.abs(d - od).max() <= 2 * CHROMA_ATOL and (d >= 0).all()


def test_chroma_col_and_live_framing(mods, chopin_audio):
    chroma, co = mods
    wav = chopin_audio["live"]
    for start in (0, 12345, 400000):
        buf = wav[start:start + 4096]
        col = chroma.wav_to_chroma_col(buf)
        ocol = np.asarray(co.wav_to_chroma_col(buf)).reshape(-1)
        assert col.shape == (12,) and np.abs(col - ocol).max() <= CHROMA_ATOL
    with pytest.raises(AssertionError):
        chroma.wav_to_chroma_col(wav[:100])
    # un-padded hop framing of a live buffer (wtw.py:81-83): frame m = samples [m*2048, +4096)
    plan = chroma._plan()
    seg = wav[50000:50000 + 4096 + 5 * 2048]
    ch, _ = plan.frames(torch.from_numpy(seg).to(plan.device), pad_left=0)
    assert ch.shape[0] == 6
    for m in range(6):
        ocol = np.asarray(co.wav_to_chroma_col(seg[m * 2048:m * 2048 + 4096])).reshape(-1)
        assert np.abs(ch[m].cpu().numpy() - ocol).max() <= CHROMA_ATOL


def test_silence_short_input_and_f32_out(mods):
    chroma, co = mods
    plan = chroma._plan()
    z = torch.zeros(3 * 4096, dtype=torch.float32, device=plan.device)
    ch, _ = plan.frames(z, pad_left=2048)
    assert ch.shape[0] == plan.num_frames(3 * 4096, 2048) == 6
    assert (ch == 0).all()  # near-zero columns are left unscaled (librosa.util.normalize), not NaN
    short = torch.ones(100, dtype=torch.float32, device=plan.device)
    ch, _ = plan.frames(short, pad_left=2048)
    assert ch.shape[0] == 0
    rs = np.random.RandomState(0)
    x = (rs.rand(20000) - 0.5).astype(np.float32)
    ch32, _ = plan.frames(torch.from_numpy(x).to(plan.device), pad_left=2048, out_dtype=torch.float32)
    och = co.wav_to_chroma(x)
    assert ch32.dtype == torch.float32 and np.abs(ch32.t().cpu().numpy() - och).max() <= 1e-6
    # float64 samples are accepted too
    ch64, _ = plan.frames(torch.from_numpy(x.astype(np.float64)).to(plan.device), pad_left=2048)
    assert np.abs(ch64.t().cpu().numpy() - och).max() <= CHROMA_ATOL


def test_other_hop_and_fft_sizes(mods):
    """BASELINE configs[0] quotes hop=512; WTW's params dict carries fft_len/hop_size (wtw.py:27-28)."""
    chroma, co = mods
    rs = np.random.RandomState(1)
    x = (rs.rand(30000) - 0.5).astype(np.float32)
    for L, H in ((4096, 512), (2048, 1024), (1024, 256), (256, 64), (8192, 2048), (8192, 4096)):
        plan = chroma.ChromaPlan(L, H, 22050)
        ch, st = plan.frames(torch.from_numpy(x).to(plan.device), pad_left=L // 2, want_stft=True)
        ost = co.create_stft(x, L, H)
        assert st.shape[0] == ost.shape[1]
        scale = np.abs(ost).max(axis=0, keepdims=True)
        assert (np.abs(st.t().cpu().numpy() - ost) <= STFT_RTOL * scale).all(), (L, H)
        fb = co.chroma_filterbank(22050, L)
        och = co.l2_normalize_columns(np.dot(fb, np.abs(ost) ** 2))
        assert np.abs(ch.t().cpu().numpy() - och).max() <= CHROMA_ATOL, (L, H)
        plan.close()
    from real_time_audio_sync_amd import _native as nat
    with pytest.raises(nat.RtsyncError):
        chroma.ChromaPlan(16384, 2048, 22050)   # 16384-point frames do not fit the LDS-resident FFT


def test_create_stft_against_reference_columns(mods, chopin_audio):
    """Direct pin of a1: STFT columns and per-frame power sums that the reference's own create_stft
    (chroma.py:44-65, executed from its text by tests/golden/make_golden.py) produced for the chopin pair,
    against the HIP FFT at the tolerance stated at the top of this file."""
    import os
    from conftest import GOLDEN
    chroma, _ = mods
    g = np.load(os.path.join(GOLDEN, "stft_golden.npz"))
    for key in ("ref", "live"):
        ft = chroma.create_stft(chopin_audio[key])
        assert ft.shape == tuple(g[key + "/shape"])
        want = g[key + "/stft_cols"]
        got = ft[:, g[key + "/cols"]]
        scale = np.abs(want).max(axis=0, keepdims=True)
        assert (np.abs(got - want) <= STFT_RTOL * scale).all(), key
        ps = (np.abs(ft) ** 2).sum(axis=0)
        assert np.abs(ps - g[key + "/power_sum"]).max() <= 1e-11 * g[key + "/power_sum"].max(), key
