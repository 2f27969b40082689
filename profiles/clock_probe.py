#!/usr/bin/env python
"""Shader clock and package power while the fused analysis+beamform kernel runs back to back for a few seconds
(BTK_FUSED_VAR selects the kernel form; rocm-smi sampled from a thread of the same process)."""
import os, sys, json, re, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, ula_positions, la_delays

dev = torch.device("cuda:0")
N, M, S, T = 64, 512, 16, 4096
D, K = M // 2, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
L = (T - afb.processing_delay + afb.lookahead) * D
g = torch.Generator(device=dev).manual_seed(1)
pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_()
wq = eng.weights_mainlobe(M, N, 16000.0, la_delays(ula_positions(N), -1.306379))
W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
for _ in range(3):
    afb.analysis_beamform(pcm, W, out=Y)
torch.cuda.synchronize()
samples, stop = [], False


def sampler():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
        pw = re.search(r"Power \(W\): ([\d.]+)", out)
        samples.append((int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None))


th = threading.Thread(target=sampler)
t_end = time.time() + float(os.environ.get("PROBE_SECONDS", "8"))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
th.start()
times = []
while time.time() < t_end:
    e0.record()
    for _ in range(50):
        afb.analysis_beamform(pcm, W, out=Y)
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) / 50)
stop = True
th.join()
sc = [s[0] for s in samples if s[0]]
pw = [s[1] for s in samples if s[1]]
print(json.dumps({"var": os.environ.get("BTK_FUSED_VAR"), "ms_first": times[0], "ms_last": times[-1], "ms_min": min(times),
                  "sclk_MHz_samples": sc, "power_W_samples": pw}))
