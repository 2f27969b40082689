"""Helpers shared by the benchmark scripts (bench.py, bench_stages.py, bench_configs.py, bench_bin_sharded.py, profiles/*.py):
the designed Nyquist(M) prototypes, the array geometry and the synthetic PCM of SURVEY 8(d).  Nothing here touches tests/."""
import hashlib
import os

import numpy as np

from distant_speech_recognition_amd import prototypes
from distant_speech_recognition_amd.pybeamformer import calc_la_delays

FS = 16000.0


def kernel_source_sha(files=("fb_analysis512.hip", "fft_packed.h")):
    """sha256 over the sources of the headline kernel: PMC traffic figures (profiles/*_pmc_traffic.json) carry it, and bench.py
    quotes them only while it still matches -- a changed kernel silently keeping an old traffic number was possible before"""
    h = hashlib.sha256()
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "distant_speech_recognition_amd", "csrc")
    for f in files:
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


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


def gpu_time(torch, fn, n=3, prewarm_ms=250.0, min_ms=60.0, max_calls=40):
    """(seconds per call by HIP events, last result).  The shader clock needs ~0.3 s of uninterrupted load to settle (idle
    at ~150 MHz, ramping through ~2 GHz; profiles/clock_probe.py), and a host synchronisation between calls lets it fall
    back: the pre-warm issues calls back to back for `prewarm_ms` of GPU time and the timed region covers at least `min_ms`
    (at least n, at most max_calls calls each -- stateful stages such as the NLMS step-size schedule must not be run
    hundreds of times)."""
    r = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn()
    e1.record()
    torch.cuda.synchronize()
    t1 = max(e0.elapsed_time(e1), 1e-3)
    for _ in range(int(min(max(prewarm_ms / t1, 1), max_calls))):
        r = fn()
    reps = int(min(max(min_ms / t1, n), max(max_calls, n)))
    e0.record()
    for _ in range(reps):
        r = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps, r
