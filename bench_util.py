"""Helpers shared by the benchmark scripts (bench.py, bench_stages.py, bench_configs.py, bench_bin_sharded.py, profiles/*.py):
the designed Nyquist(M) prototypes, the array geometry and the synthetic PCM of SURVEY 8(d).  Nothing here touches tests/."""
import numpy as np

from distant_speech_recognition_amd import prototypes
from distant_speech_recognition_amd.pybeamformer import calc_la_delays

FS = 16000.0


def design_prototype(M, m=4, kind="h", r=1):
    """analysis ("h") or synthesis ("g") Nyquist(M) prototype as the reference's designer produces it"""
    h, g = prototypes.load(M, m, r)
    return h if kind == "h" else g


def ula_positions(N, pitch_mm=20.0):
    """uniform linear array, centred (positions in mm)"""
    x = (np.arange(N) - (N - 1) / 2.0) * pitch_mm
    return np.stack([x, np.zeros(N), np.zeros(N)], axis=1)


def la_delays(mpos, azimuth):
    """far-field linear-array delays (reference lib/pybeamformer.py:41-64)"""
    return calc_la_delays(mpos, azimuth)
