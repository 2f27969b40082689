"""Staged analysis bank at the C0 shape (16 streams x 64 mics x 4096 frames, M = 512, r = 1), row-padded snapshots, at settled
clocks.  (Used for the A/B of an experimental form -- the fused kernel's front end with a per-channel store, DESIGN.md section 8 --
that was selected by BTK_FUSED_VAR and is not in the tree.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, gpu_time
dev = torch.device("cuda:0")
S, N, M, T = 16, 64, 512, 4096
D, K = M // 2, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
L = (T - afb.processing_delay + afb.lookahead) * D
pcm = (torch.randn((S, N, L), device=dev) * 1000.0).round_()
X = afb.analysis(pcm, pad_rows=True)
t = gpu_time(torch, lambda: afb.analysis(pcm, out=X), n=10, prewarm_ms=300.0)[0]
b = S * N * T * (4 * D + 8 * K)
print("BTK_FUSED_VAR=%s analysis %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)  checksum %.6e" % (os.environ.get("BTK_FUSED_VAR", "default"), t * 1e3, b / t / 1e9, 100 * b / t / 8e12, float(X.abs().double().sum())))
