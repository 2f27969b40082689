"""One WPE estimation at the reference configuration (unit_test/confs/wpe.json: 8 channels, lags 0..32, 2 iterations) for
kernel-trace / PMC passes: WPE_S streams x 257 bins x 1000 frames."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
dev = torch.device("cuda:0")
S, K, C, T, M = int(os.environ.get("WPE_S", 2)), 257, 8, 1000, 512
g = torch.Generator(device=dev).manual_seed(3)
X = (torch.randn((S, K, C, T), device=dev, generator=g) + 1j * torch.randn((S, K, C, T), device=dev, generator=g)).to(torch.complex64) * 300
for _ in range(2):
    G = eng.wpe_estimate(X, M, 0, 32, 2, -18.0, 0.0, 1e-4)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3):
    G = eng.wpe_estimate(X, M, 0, 32, 2, -18.0, 0.0, 1e-4)
e1.record(); torch.cuda.synchronize()
print("wpe_estimate S=%d: %.3f ms per call (%.3f ms per stream)  checksum %.6e" % (S, e0.elapsed_time(e1) / 3, e0.elapsed_time(e1) / 3 / S, float(G[0].abs().sum() if isinstance(G, tuple) else G.abs().sum())))
