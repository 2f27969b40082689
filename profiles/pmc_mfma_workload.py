"""Workload for a rocprofv3 --pmc pass over the MFMA kernels: covariance HERK at C0 (64 mics, 257 bins, 16 x 4096 frames) and
the WPE normal-equation HERK (8 channels x 33 lags, 1000 frames)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng
dev = torch.device("cuda:0")
S, K, N, T = 16, 257, 64, 4096
X = (torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 1000
R = torch.zeros((S, K, N, N), dtype=torch.complex64, device=dev)
for _ in range(2):
    eng.cov_accumulate(X, R=R, use_mfma=True)
del X, R
Xw = (torch.randn((1, 257, 8, 1000), device=dev) + 1j * torch.randn((1, 257, 8, 1000), device=dev)).to(torch.complex64) * 500
eng.wpe_estimate(Xw, 512, 0, 32, 1, -18.0, 0.0, 1e-4)
torch.cuda.synchronize()
print("done: cov flops per launch %.3e, herk (45 quadrants) %.3e" % (8.0 * K * N * N * S * T, 45 * 1024 * 1000 * 8.0 * 257 * 8))
