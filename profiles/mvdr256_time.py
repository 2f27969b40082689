#!/usr/bin/env python
"""C4-size MVDR design (256 mics, 2048 bins -> 1025 solves of 256 x 256): diffuse model + loading + solve, by HIP events"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays, gpu_time
dev = torch.device("cuda:0")
out = {}
for N, M in ((256, 2048), (140, 1024), (64, 1024)):
    K = M // 2 + 1
    mpos = ula_positions(N, 10.0)
    wqd = torch.from_numpy(eng.weights_mainlobe(M, N, 16000.0, la_delays(mpos, 0.8))[:K].astype(np.complex64)).to(dev)
    Rd = eng.mvdr_diffuse_model(mpos, M, 16000.0, device=dev)
    eng.mvdr_diagonal_loading(Rd, 0.01)
    t, (W, nfb) = gpu_time(torch, lambda: eng.mvdr_weights(Rd, wqd), n=3)
    out["N%d_K%d" % (N, K)] = {"solve_ms": t * 1e3, "identity_fallbacks": nfb, "TFLOPs": (32.0 / 3) * K * N ** 3 / t / 1e12,
                               "checksum": float(W.abs().sum())}
print(json.dumps(out))
