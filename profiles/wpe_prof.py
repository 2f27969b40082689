"""WPE estimate workload for rocprofv3 --kernel-trace --stats: 8 channels, 512 bins, lags 0..32, 1000 frames, 2 iterations"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng
dev = torch.device("cuda:0")
C, M, T = 8, 512, 1000
K = M // 2 + 1
S = int(os.environ.get("WPE_S", 1))
Xw = (torch.randn((S, K, C, T), device=dev) + 1j * torch.randn((S, K, C, T), device=dev)).to(torch.complex64) * 500
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    G = eng.wpe_estimate(Xw, M, 0, 32, 2, -18.0, 0.0, 1e-4)
    torch.cuda.synchronize(); print("wpe_estimate S=%d: %.1f ms" % (S, (time.perf_counter() - t0) * 1e3))
