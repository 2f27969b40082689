"""Shared helpers for the parity tests (synthetic inputs per SURVEY.md 8(d))."""
import numpy as np

SSPEED = 343740.0   # mm/s, reference beamformer/beamformer.h:26


def design_prototype(M, m, kind="h"):
    """The filter-bank prototype the tests run on: the reference designer's Nyquist(M) pair where one exists
    (distant_speech_recognition_amd/prototypes: M = 256, 512, 1024, 2048 at m = 4, designed for r = 1 -- the BASELINE
    configs C0..C5); for every other geometry of the sweeps (m != 4, small M) a deterministic Kaiser-windowed sinc of
    m*M taps -- parity tests only need *some* real prototype shared by oracle and GPU."""
    from distant_speech_recognition_amd import prototypes
    if (M, m, 1) in prototypes.available():
        h, g = prototypes.load(M, m, 1)
        return h if kind == "h" else g
    L = m * M
    n = np.arange(L) - (L - 1) / 2.0
    w = np.kaiser(L, 8.0)
    h = np.sinc(n / M) * w
    h = h / np.sqrt(np.sum(h * h)) / np.sqrt(M) * (1.0 if kind == "h" else M / 2.0)
    return h.astype(np.float64)


def ula_positions(N, pitch_mm=20.0):
    x = (np.arange(N) - (N - 1) / 2.0) * pitch_mm
    return np.stack([x, np.zeros(N), np.zeros(N)], axis=1)


def la_delays(mpos, azimuth, sspeed=SSPEED):
    """calc_la_delays (reference lib/pybeamformer.py:41-64)."""
    N = len(mpos)
    d = -mpos[:, 0] * np.cos(azimuth) / sspeed
    return d - d[N // 2]


def synthetic_pcm(S, N, L, seed=20260927, target=True, fs=16000.0, azimuth=-1.306379, pitch_mm=20.0):
    """int16-scale float32 PCM [S][N][L]: iid noise N(0,1000^2) + a common target N(0,3000^2)
    delayed per channel by the far-field linear-array delays (integer-sample approximation)."""
    out = np.zeros((S, N, L), np.float32)
    mpos = ula_positions(N, pitch_mm)
    delays = la_delays(mpos, azimuth)
    for s in range(S):
        tg = np.random.default_rng(seed + 1000 * s + 999).normal(0.0, 3000.0, L + 64) if target else None
        for c in range(N):
            rng = np.random.default_rng(seed + 1000 * s + c)
            x = rng.normal(0.0, 1000.0, L)
            if target:
                sh = int(round(delays[c] * fs))
                x = x + tg[32 + sh: 32 + sh + L]
            out[s, c] = np.clip(np.rint(x), -32767, 32767)
    return out, delays
