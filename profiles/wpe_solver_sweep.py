"""WPE estimate time against the order P = C x lags for one solver (BTK_WPE_SOLVE_REG=1 / BTK_WPE_SOLVE_PANEL=1 select it): 2 streams x 257 bins x
1000 frames, 2 iterations.  Finds the P from which the register-resident solver (csrc/chol_reg.h) should take over from the panel solver."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from bench_util import gpu_time
dev = torch.device("cuda:0")
S, M, T = 2, 512, 1000
K = M // 2 + 1
out = {}
for C, L in ((8, 2), (8, 4), (4, 12), (8, 8), (8, 10), (8, 12), (4, 32), (8, 20), (8, 33)):
    g = torch.Generator(device=dev).manual_seed(C * 100 + L)
    X = ((torch.randn((S, K, C, T), device=dev, generator=g) + 1j * torch.randn((S, K, C, T), device=dev, generator=g)) * 300).to(torch.complex64)
    t = gpu_time(torch, lambda: eng.wpe_estimate(X, M, lower_num=0, upper_num=L - 1, iterations_num=2, load_db=-18.0, diagonal_bias=1e-4), n=3)[0]
    out["P%d(C%dxL%d)" % (C * L, C, L)] = round(t * 1e3, 3)
print(json.dumps(out))
