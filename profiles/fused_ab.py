#!/usr/bin/env python
"""In-run A/B of the fused analysis+beamform kernel forms (run in separate processes because the switch is read once):
   python profiles/fused_ab.py            -> current kernel
   BTK_FUSED_V1=1 python profiles/fused_ab.py -> per-channel post-pass form (diagnostic)"""
import os, sys, json
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
delays = la_delays(ula_positions(N), -1.306379)
wq = eng.weights_mainlobe(M, N, 16000.0, delays)
W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
for _ in range(3):
    afb.analysis_beamform(pcm, W, out=Y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for rep in range(5):
    e0.record()
    for _ in range(10):
        afb.analysis_beamform(pcm, W, out=Y)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10)
# checksum against the staged path
X = afb.analysis(pcm[:1])
Yr = eng.bf_apply(W, X)
err = float((Y[:1] - Yr).abs().max() / Yr.abs().max())
print(json.dumps({"v1": bool(os.environ.get("BTK_FUSED_V1")), "ms": best, "frames_per_s": S * T / best * 1e3, "rel_err_vs_staged": err}))
