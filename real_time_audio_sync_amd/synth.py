"""Seeded synthetic chroma for parity tests and bench.py.

The WAVs the BASELINE configs name are not in the reference checkout, so every benchmark and most
parity cases run on synthetic 12-bin chroma: a piecewise-constant random "score" as the reference
recording and randomly time-warped, noisy renditions of it as the live streams.  Additive noise
makes the data tie-free (no two accumulated costs are equal), which is what makes alignment-path
indices well defined across summation orders.

All values are rounded to float32 and returned as float64 arrays holding exactly those values, so
the same numbers can be fed to float32 device buffers and to float64 CPU code without a
representation difference.
"""
import numpy as np

N_CHROMA = 12


def _unit_columns(x):
    return x / np.sqrt((x * x).sum(axis=0, keepdims=True))


def _as_f32_values(x):
    return x.astype(np.float32).astype(np.float64)


def synth_ref(n_frames, seed=0):
    """Reference chroma, feature-major (12, n_frames) like chroma.wav_to_chroma's output."""
    rs = np.random.RandomState(seed)
    cols = np.empty((N_CHROMA, n_frames))
    pos = 0
    while pos < n_frames:
        hold = rs.randint(2, 12)
        tpl = rs.rand(N_CHROMA) ** 3
        cols[:, pos:pos + hold] = tpl[:, None]
        pos += hold
    cols = cols + 0.02 * rs.rand(N_CHROMA, n_frames)
    return _as_f32_values(_unit_columns(cols))


def synth_live(ref, seed, max_frames=None, lo=0.8, hi=1.25, noise=0.05):
    """One live stream: a random time warp of ``ref`` (12, N) plus noise, (12, T)."""
    rs = np.random.RandomState(seed)
    n = ref.shape[1]
    steps = rs.uniform(lo, hi, size=int(n / lo) + 8)
    posn = np.cumsum(steps)
    posn = posn[posn < n - 1]
    idx = np.floor(posn).astype(np.int64)
    if max_frames is not None:
        idx = idx[:max_frames]
    live = ref[:, idx] + noise * rs.rand(N_CHROMA, idx.size)
    return _as_f32_values(_unit_columns(live))


def synth_batch(n_ref, batch, seed=0, max_frames=None):
    """(ref (12,N), [live_b (12,T_b)] for b < batch); stream b uses seed+1+b."""
    ref = synth_ref(n_ref, seed)
    lives = [synth_live(ref, seed + 1 + b, max_frames=max_frames) for b in range(batch)]
    return ref, lives


def synth_tie(n_frames, seed=0):
    """A deliberately tie-ridden pair (exactly repeated frames, no noise) documenting the
    reference's tie rules; only meaningful in float64 with the oracle's summation order."""
    rs = np.random.RandomState(seed)
    tpl = _unit_columns(rs.rand(N_CHROMA, 6) ** 2)
    idx_r = np.repeat(np.arange(6), int(np.ceil(n_frames / 6.0)))[:n_frames]
    idx_l = np.repeat(np.arange(6), int(np.ceil(n_frames / 6.0)) + 1)[:n_frames]
    return _as_f32_values(tpl[:, idx_r]), _as_f32_values(tpl[:, idx_l])
