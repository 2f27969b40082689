import sys, torch, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from bench_util import gpu_time
dev = torch.device("cuda:0")
for N, K in ((64, 513), (128, 513), (256, 256)):
    g = torch.Generator(device=dev).manual_seed(N)
    A = torch.randn((K, N, N + 8), device=dev, generator=g) + 1j * torch.randn((K, N, N + 8), device=dev, generator=g)
    R = (A @ A.conj().transpose(1, 2) / (N + 8)).to(torch.complex64).contiguous()
    Dg = torch.diag_embed(torch.rand((K, N), device=dev, generator=g) + 0.5).to(torch.complex64).contiguous()
    t_r = gpu_time(torch, lambda: eng.csvdc_values(R), n=3)[0]
    t_d = gpu_time(torch, lambda: eng.csvdc_values(Dg), n=3)[0]
    print("N=%d K=%d: random Hermitian %.2f ms, diagonal (no QR sweeps) %.2f ms" % (N, K, t_r * 1e3, t_d * 1e3), flush=True)
