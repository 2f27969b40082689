"""A/B timing of the analysis / synthesis kernels for M in {256, 512, 1024, 2048} (same data volume each)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype
dev = torch.device("cuda:0")
for M in (256, 512, 1024, 2048):
    S, N = 8, 64
    D, K = M // 2, M // 2 + 1
    T = 4096 * 512 // M
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D
    pcm = (torch.randn((S, N, L), device=dev) * 1000).round_()
    X = torch.empty((S, K, N, T), dtype=torch.complex64, device=dev)
    Y = (torch.randn((S * 16, K, T), device=dev) + 1j * torch.randn((S * 16, K, T), device=dev)).to(torch.complex64)
    def tm(fn, n=5):
        fn(); fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / n * 1e-3
    ta = tm(lambda: afb.analysis(pcm, out=X))
    ts = tm(lambda: sfb.synthesize(Y))
    ba = (4 * D + 8 * K) * N * S * T
    bs = (8 * K + 4 * D) * S * 16 * T
    print("M=%4d analysis %.3f ms %.0f GB/s (%.1f%%) | synthesis %.3f ms %.0f GB/s (%.1f%%)" %
          (M, ta * 1e3, ba / ta / 1e9, 100 * ba / ta / 8e12, ts * 1e3, bs / ts / 1e9, 100 * bs / ts / 8e12))
    del pcm, X, Y
