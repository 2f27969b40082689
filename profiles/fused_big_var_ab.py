#!/usr/bin/env python
"""Variants of analysis_bfz_big_kernel (BTK_FUSED_VAR, read once per process) on the five launches of fused_big_ab.py: time of the fused call
and a digest of its output (variants that only move data differently must agree bit for bit)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng, prototypes
from bench_util import gpu_time

dev = torch.device("cuda:0")
var = os.environ.get("BTK_FUSED_VAR", "default")
for name, M, N, S, T in (("C5 block 256x2048 1x512", 2048, 256, 1, 512), ("256x2048 1x4096", 2048, 256, 1, 4096), ("64x2048 8x2048", 2048, 64, 8, 2048),
                         ("C3 apply 64x1024 4x8192", 1024, 64, 4, 8192), ("64x1024 8x2048", 1024, 64, 8, 2048)):
    D, K = M // 2, M // 2 + 1
    h, g = prototypes.load(M, 4, 1)
    afb = eng.FilterBank(h, M, 4, 1, 2)
    L = (T - afb.processing_delay + afb.lookahead) * D
    gen = torch.Generator(device=dev).manual_seed(M + N)
    pcm = (torch.randn((S, N, L), device=dev, generator=gen) * 1000.0).round_()
    W = ((torch.randn((K, N), device=dev, generator=gen) + 1j * torch.randn((K, N), device=dev, generator=gen)) / N).to(torch.complex64)
    Y = eng.padded_rows((S, K, T), torch.complex64, dev)
    afb.analysis_beamform(pcm, W, out=Y)
    dig = hashlib.sha256(torch.view_as_real(Y).contiguous().cpu().numpy().tobytes()).hexdigest()[:12]
    t = min(gpu_time(torch, lambda: afb.analysis_beamform(pcm, W, out=Y), n=5)[0] for _ in range(3))
    frac = (4 * D * N + 8 * K) * S * T / t / 8e12
    print("VAR=%-7s %-26s fused %.4f ms  frac %.3f  sha %s" % (var, name, t * 1e3, frac, dig), flush=True)
    del pcm, Y
