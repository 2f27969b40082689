"""Covariance HERK at the C0 snapshot shape (16 streams x 257 bins x 64 mics x 4096 frames) and at 128 mics: the float32 matrix
instruction (BTK_COV_F32=1, separate process) against the bfloat16-piece form, with the max difference between the two."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from bench_util import gpu_time
dev = torch.device("cuda", 0)
for S, K, N, T in ((16, 257, 64, 4096), (4, 257, 128, 4096), (4, 513, 64, 2048)):
    g = torch.Generator(device=dev).manual_seed(1)
    X = (torch.randn((S, K, N, T), device=dev, generator=g) + 1j * torch.randn((S, K, N, T), device=dev, generator=g)).to(torch.complex64) * 2000
    fw = (torch.rand((S, T), device=dev, generator=g) > 0.3).float()
    R = eng.cov_accumulate(X, frame_weights=fw, use_mfma=True)
    Rv = eng.cov_accumulate(X[:, :3].contiguous(), frame_weights=fw, use_mfma=False)
    err = float((R[:, :3] - Rv).abs().max() / Rv.abs().max())
    Racc = torch.zeros_like(R)
    t = gpu_time(torch, lambda: eng.cov_accumulate(X, R=Racc, frame_weights=fw, use_mfma=True), n=3)[0]
    print("S=%d K=%d N=%d T=%d: %.3f ms  %.1f TFLOP/s-equivalent  %.0f GB/s read  max |R - R_valu| / max |R| = %.2e  (%s)"
          % (S, K, N, T, t * 1e3, 8.0 * K * N * N * S * T / t / 1e12, 8.0 * K * N * S * T / t / 1e9, err,
             "float32 instruction" if os.environ.get("BTK_COV_F32") else "bfloat16 pieces"), flush=True)
