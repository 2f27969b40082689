#!/usr/bin/env python
"""Small fixed workload for rocprofv3 --pmc passes: 2 launches each of the C0 analysis, GSC apply and
synthesis kernels (8 streams x 64 mics x 2048 frames, M=512) -- same kernels/geometry as bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distant_speech_recognition_amd import engine as eng
from tests.util import design_prototype

S, N, M, T = 8, 64, 512, 2048
dev = torch.device("cuda:0")
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
L = (T - afb.processing_delay + afb.lookahead) * 256
pcm = (torch.randn((S, N, L), device=dev) * 1000).round_()
W = torch.randn((257, N), dtype=torch.complex64, device=dev) / N
X = torch.empty((S, 257, N, T), dtype=torch.complex64, device=dev)
Y = torch.empty((S, 257, T), dtype=torch.complex64, device=dev)
for _ in range(2):
    afb.analysis(pcm, out=X)
    eng.bf_apply(W, X, out=Y)
    sfb.synthesize(Y)
    afb.analysis_beamform(pcm, W, out=Y)
torch.cuda.synchronize()
print("pmc workload done: algorithmic bytes analysis=%d apply=%d" % ((4 * 256 + 8 * 257) * N * S * T, 8 * 257 * (N + 1) * S * T))
