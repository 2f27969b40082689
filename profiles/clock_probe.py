#!/usr/bin/env python
"""Shader clock and package power while one kernel runs back to back for a few seconds: PROBE_WORK = fused (the fused
analysis+beamform kernel, BTK_FUSED_VAR selects the form), nlms or apply (PROBE_S streams), wpe (the estimate at the reference configuration); rocm-smi is sampled from a
thread of the same process."""
import os, sys, json, re, subprocess, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, ula_positions, la_delays

dev = torch.device("cuda:0")
WORK = os.environ.get("PROBE_WORK", "fused")
if WORK == "fused":
    N, M, S, T = 64, 512, 16, 4096
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    L = (T - afb.processing_delay + afb.lookahead) * D
    g = torch.Generator(device=dev).manual_seed(1)
    pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_()
    wq = eng.weights_mainlobe(M, N, 16000.0, la_delays(ula_positions(N), -1.306379))
    W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    step = lambda: afb.analysis_beamform(pcm, W, out=Y)
elif WORK == "wpe":                        # WPE estimate at the reference configuration (8 ch x lags 0..32, 1000 frames): 2/3 lag products on the matrix cores
    S, C, M, T = int(os.environ.get("PROBE_S", "2")), 8, 512, 1000
    K = M // 2 + 1
    g = torch.Generator(device=dev).manual_seed(3)
    X = ((torch.randn((S, K, C, T), device=dev, generator=g) + 1j * torch.randn((S, K, C, T), device=dev, generator=g)) * 300).to(torch.complex64)
    step = lambda: eng.wpe_estimate(X, M, lower_num=0, upper_num=32, iterations_num=2, load_db=-18.0, diagonal_bias=1e-4)
else:                                       # "nlms": PROBE_S streams x 64 mics x 257 bins x 4096 frames; "apply": the same snapshots through bf_apply
    N, M, S, T = 64, 512, int(os.environ.get("PROBE_S", "32")), 4096
    K = M // 2 + 1
    X = (torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 2000
    delays = la_delays(ula_positions(N), -1.306379)
    vs = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
    st = eng.NLMSState(S, M, N, dev)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    def nlms_step():                         # the step size halves every `slowdown_after` frames: without a reset the canceller
        st.reset_stats()                     # stops adapting after a few hundred calls and the kernel gets cheaper
        eng.nlms_process(vs, X, st, out=Y)
    step = nlms_step if WORK == "nlms" else (lambda: eng.bf_apply(vs, X, out=Y))
for _ in range(3):
    step()
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
NREP = 50 if WORK == "fused" else (3 if WORK == "wpe" else 10)
th.start()
times = []
while time.time() < t_end:
    e0.record()
    for _ in range(NREP):
        step()
    e1.record()
    torch.cuda.synchronize()
    times.append(e0.elapsed_time(e1) / NREP)
stop = True
th.join()
sc = [s[0] for s in samples if s[0]]
pw = [s[1] for s in samples if s[1]]
print(json.dumps({"work": WORK, "var": os.environ.get("BTK_FUSED_VAR"), "nlms_alt": os.environ.get("BTK_NLMS_ALT"), "ms_first": times[0], "ms_last": times[-1], "ms_min": min(times),
                  "sclk_MHz_samples": sc, "power_W_samples": pw}))
