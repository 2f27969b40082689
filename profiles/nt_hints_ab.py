#!/usr/bin/env python
"""Stage times of the bandwidth-bound staged kernels at the C0 launch (32 streams x 64 mics x 4096 frames, padded rows) for one
build of the library (BTK_LIB_PATH: a -DBTK_EXP=<bit> build makes one global stream non-temporal; profiles/r06_nt_hints.txt)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, ula_positions, la_delays, gpu_time

dev = torch.device("cuda:0")
N, M, S, T = 64, 512, 32, 4096
D, K = M // 2, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
L = (T - afb.processing_delay + afb.lookahead) * D
g = torch.Generator(device=dev).manual_seed(1)
pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_()
delays = la_delays(ula_positions(N), -1.306379)
wq = eng.weights_mainlobe(M, N, 16000.0, delays)
W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
Yc = eng.rows_like(X, (S, K, T))
out = torch.empty((S, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)
vs = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
nst = eng.NLMSState(S, M, N, dev)
res = {"lib": os.path.basename(os.environ.get("BTK_LIB_PATH", "libbtkhip.so"))}
for rep in range(2):
    res["analysis_ms_%d" % rep] = gpu_time(torch, lambda: afb.analysis(pcm, out=X))[0] * 1e3
    res["apply_ms_%d" % rep] = gpu_time(torch, lambda: eng.bf_apply(W, X, out=Yc))[0] * 1e3
    res["nlms_ms_%d" % rep] = gpu_time(torch, lambda: eng.nlms_process(vs, X, nst, out=Yc))[0] * 1e3
    res["nlms_one_launch_ms_%d" % rep] = gpu_time(torch, lambda: eng.nlms_process(vs, X, nst, out=Yc, interleave=(1, T)))[0] * 1e3
    res["synthesis_ms_%d" % rep] = gpu_time(torch, lambda: sfb.synthesize(Yc, out=out))[0] * 1e3
res["checksum"] = float(out.double().abs().sum().item())
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
