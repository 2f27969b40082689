"""Fused analysis + apply at the reference's default geometry (M = 256, m = 4, r = 1) against the staged pair."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype
dev = torch.device("cuda:0")
N, M, S, T = 64, 256, 16, 8192
D, K = M // 2, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
L = (T - afb.processing_delay + afb.lookahead) * D
pcm = (torch.randn((S, N, L), device=dev) * 1000.0).round_()
W = ((torch.randn((K, N), device=dev) + 1j * torch.randn((K, N), device=dev)) / N).to(torch.complex64)
def tm(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
Y = eng.padded_rows((S, K, T), torch.complex64, dev)
tf = tm(lambda: afb.analysis_beamform(pcm, W, out=Y))
X = torch.empty((S, K, N, T), dtype=torch.complex64, device=dev)
Yc = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
ta = tm(lambda: afb.analysis(pcm, out=X)); tb = tm(lambda: eng.bf_apply(W, X, out=Yc))
err = float((Y - Yc).abs().max() / Yc.abs().max())
print(json.dumps({"fused_ms": tf, "staged_ms": ta + tb, "analysis_ms": ta, "apply_ms": tb, "frames_per_s_fused": S * T / tf * 1e3,
                  "rel_err": err, "disabled": bool(os.environ.get("BTK_DISABLE_FUSED"))}))
