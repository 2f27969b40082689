import os, sys, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
from distant_speech_recognition_amd import engine as eng
from bench_util import gpu_time
dev = torch.device("cuda:0")
out = {}
for N in (8, 16, 24, 32, 48, 64, 100, 128, 136):
    for K in (513, 2052):
        g = torch.Generator(device=dev).manual_seed(N)
        A = torch.randn((K, N, N + 8), device=dev, generator=g) + 1j * torch.randn((K, N, N + 8), device=dev, generator=g)
        R = (A @ A.conj().transpose(1, 2) / (N + 8) + 0.05 * torch.eye(N, device=dev)).to(torch.complex64).contiguous()
        d = ((torch.randn((K, N), device=dev, generator=g) + 1j * torch.randn((K, N), device=dev, generator=g)) / N).to(torch.complex64)
        t = gpu_time(torch, lambda: eng.mvdr_weights(R, d), n=5)[0]
        out["N%d_K%d" % (N, K)] = round(t * 1e3, 4)
print(json.dumps(out))
