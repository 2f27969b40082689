#!/usr/bin/env python
"""One launch shape of analysis_bfz_big_kernel, a few calls (workload of profiles/scripts/r05_pmc_fused_big.sh): BIG_SHAPE = M,N,S,T."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng, prototypes
dev = torch.device("cuda:0")
M, N, S, T = [int(v) for v in os.environ.get("BIG_SHAPE", "2048,256,1,4096").split(",")]
D, K = M // 2, M // 2 + 1
h, g = prototypes.load(M, 4, 1)
afb = eng.FilterBank(h, M, 4, 1, 2)
L = (T - afb.processing_delay + afb.lookahead) * D
gen = torch.Generator(device=dev).manual_seed(M + N)
pcm = (torch.randn((S, N, L), device=dev, generator=gen) * 1000.0).round_()
W = ((torch.randn((K, N), device=dev, generator=gen) + 1j * torch.randn((K, N), device=dev, generator=gen)) / N).to(torch.complex64)
Y = eng.padded_rows((S, K, T), torch.complex64, dev)
for _ in range(int(os.environ.get("BIG_CALLS", "6"))):
    afb.analysis_beamform(pcm, W, out=Y)
torch.cuda.synchronize()
