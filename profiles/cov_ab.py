"""covariance HERK timing at C0 (64 mics) and at 128 / 256 mics"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng
dev = torch.device("cuda:0")
for (S, K, N, T) in ((16, 257, 64, 4096), (4, 257, 128, 2048), (1, 513, 256, 1024), (16, 257, 8, 4096), (16, 257, 4, 4096), (16, 257, 16, 4096)):
    X = (torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 1000
    R = torch.zeros((S, K, N, N), dtype=torch.complex64, device=dev)
    for mf in ((1, 2) if N <= 16 else (1, 0)):
        eng.cov_accumulate(X, R=R, use_mfma=mf); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): eng.cov_accumulate(X, R=R, use_mfma=mf)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 3 * 1e-3
        print("N=%3d S=%2d T=%d %s: %.3f ms  %.1f TFLOP/s  read %.0f GB/s" % (N, S, T, {1: "mfma" if N > 16 else "small-N", 0: "valu", 2: "tiled mfma"}[mf], t * 1e3, 8.0 * K * N * N * S * T / t / 1e12, 8.0 * K * N * S * T / t / 1e9))
    del X, R
