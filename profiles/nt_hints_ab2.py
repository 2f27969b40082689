#!/usr/bin/env python
"""Second set of cache-hint A/Bs (profiles/r06_nt_hints.txt): apply (Y stores), staged banks at M = 1024 / 2048 (analysis stores,
synthesis stores), apply + Zelinski (snapshot loads), one build of the library per process (BTK_LIB_PATH)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, ula_positions, la_delays, gpu_time

dev = torch.device("cuda:0")
res = {"lib": os.path.basename(os.environ.get("BTK_LIB_PATH", "libbtkhip.so"))}
g = torch.Generator(device=dev).manual_seed(1)
# apply + Zelinski at the C0 snapshot shape
S, N, M, T = 16, 64, 512, 4096
K = M // 2 + 1
X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
X.view(torch.float32).normal_(generator=g).mul_(2000.0)
delays = la_delays(ula_positions(N), -1.3)
vd = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
Yc = eng.rows_like(X, (S, K, T))
zs = eng.ZelinskiState(S, K, dev)
for rep in range(2):
    res["apply_ms_%d" % rep] = gpu_time(torch, lambda: eng.bf_apply(vd, X, out=Yc))[0] * 1e3
    res["zelinski_ms_%d" % rep] = gpu_time(torch, lambda: eng.bf_apply_zelinski(vd, vd, X, zs, alpha=0.7))[0] * 1e3
del X, Yc
for M, S, N, T in ((1024, 8, 64, 2048), (2048, 4, 64, 2048)):
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D
    pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_()
    X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
    S2 = 128
    Y = eng.padded_rows((S2, K, T), torch.complex64, dev)
    Y.view(torch.float32).normal_(generator=g)
    out = torch.empty((S2, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)
    for rep in range(2):
        res["analysis%d_ms_%d" % (M, rep)] = gpu_time(torch, lambda: afb.analysis(pcm, out=X))[0] * 1e3
        res["synthesis%d_ms_%d" % (M, rep)] = gpu_time(torch, lambda: sfb.synthesize(Y, out=out))[0] * 1e3
    del pcm, X, Y, out
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
