#!/usr/bin/env python
"""Fixed workload for rocprofv3 --pmc passes: 2 launches each of the C0 analysis, GSC apply, synthesis and fused
analysis+apply kernels at bench.py's default launch size (16 streams x 64 mics x 4096 frames, M=512), so the
per-launch counter values compare directly with bench.py's roofline.bytes_per_launch.  PMC_S / PMC_T override."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype

S, N, M, T = int(os.environ.get("PMC_S", 16)), 64, 512, int(os.environ.get("PMC_T", 4096))
dev = torch.device("cuda:0")
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
L = (T - afb.processing_delay + afb.lookahead) * 256
pcm = (torch.randn((S, N, L), device=dev) * 1000).round_()
W = torch.randn((257, N), dtype=torch.complex64, device=dev) / N
X = torch.empty((S, 257, N, T), dtype=torch.complex64, device=dev)
Y = torch.empty((S, 257, T), dtype=torch.complex64, device=dev)
pcm16 = pcm.to(torch.int16)
for _ in range(2):
    afb.analysis(pcm, out=X)
    eng.bf_apply(W, X, out=Y)
    sfb.synthesize(Y)
    afb.analysis_beamform(pcm, W, out=Y)
    afb.analysis_beamform(pcm16, W, out=Y)            # the int16 entry (its instance name ends in ", short>")
torch.cuda.synchronize()
print("pmc workload done: S=%d T=%d algorithmic bytes analysis=%d apply=%d fused(min traffic)=%d"
      % (S, T, (4 * 256 + 8 * 257) * N * S * T, 8 * 257 * (N + 1) * S * T, (4 * 256 * N + 8 * 257) * S * T))
