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


class ClockPowerSampler:
    """Shader clock and package power of one GPU, sampled from a thread while a timed region runs: the hwmon files of the amdgpu
    driver (freq1_input = sclk in Hz, power1_input = package power in microwatts, power1_cap = the cap), readable by an ordinary
    user.  The headline kernel sits at the package power cap, and what clock the cap leaves differs from box to box: a bench line
    that carries the two makes a 0.39-against-0.41 run attributable.  Without the files (no GPU, another driver) the summary
    says so and holds no numbers."""

    def __init__(self, torch=None, device=None, period_s=0.002):
        import glob
        self.period = period_s
        self.dir = None
        want = None
        try:                                            # the card that is the torch device, by PCI address
            p = torch.cuda.get_device_properties(device)
            want = "%04x:%02x:%02x" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        except Exception:
            pass
        cands = []
        for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            if os.path.exists(os.path.join(d, "freq1_input")) and os.path.exists(os.path.join(d, "power1_input")):
                real = os.path.realpath(os.path.join(d, "..", ".."))
                cands.append((d, real))
        for d, real in cands:
            if want and want in real:
                self.dir = d
        if self.dir is None and cands:
            self.dir = cands[0][0]
        self.samples = []
        self._stop = False
        self._th = None

    def _read(self, name):
        try:
            return float(open(os.path.join(self.dir, name)).read())
        except (OSError, ValueError):
            return None

    def _loop(self):
        import time
        while not self._stop:
            f, p = self._read("freq1_input"), self._read("power1_input")
            if f is not None and p is not None:
                self.samples.append((f * 1e-6, p * 1e-6))
            time.sleep(self.period)

    def __enter__(self):
        if self.dir is not None:
            import threading
            self._th = threading.Thread(target=self._loop, daemon=True)
            self._th.start()
        return self

    def begin(self):
        """drop what was sampled so far: the region of interest starts now (the thread is started earlier so that nothing but the
        time stamp lies between the synchronisation in front of a timed region and its first launch)"""
        self.samples = []

    def __exit__(self, *a):
        self._stop = True
        if self._th is not None:
            self._th.join()
        return False

    def summary(self):
        if self.dir is None:
            return {"sclk_MHz": None, "package_power_W": None, "why": "no amdgpu hwmon files (freq1_input / power1_input) on this box"}
        if not self.samples:
            return {"sclk_MHz": None, "package_power_W": None, "why": "the timed region was shorter than one sample"}
        sc = sorted(s[0] for s in self.samples)
        pw = sorted(s[1] for s in self.samples)
        cap = self._read("power1_cap")
        return {"sclk_MHz": {"median": sc[len(sc) // 2], "min": sc[0], "max": sc[-1]},
                "package_power_W": {"median": pw[len(pw) // 2], "max": pw[-1], "cap": cap * 1e-6 if cap else None},
                "samples": len(sc), "source": self.dir}
