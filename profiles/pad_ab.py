"""Does the k-stride of the snapshot block (N * T_stride * 8 B = 2 MiB at T_stride = 4096) hot-spot HBM channels?
Same analysis / apply launches with the time axis padded by a few frames."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype
dev = torch.device("cuda:0")
M, S, N, T = 512, 16, 64, 4096
D, K = M // 2, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
L = (T - afb.processing_delay + afb.lookahead) * D
pcm = (torch.randn((S, N, L), device=dev) * 1000).round_()
W = (torch.randn((K, N), device=dev) + 1j * torch.randn((K, N), device=dev)).to(torch.complex64)
def tm(fn, n=5):
    fn(); fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for pad in (0, 16, 32, 48, 80, 144):
    X = torch.empty((S, K, N, T + pad), dtype=torch.complex64, device=dev)
    ta = tm(lambda: afb.analysis(pcm, tcount=T, out=X))
    Y = torch.empty((S, K, T + pad), dtype=torch.complex64, device=dev)
    tb = tm(lambda: eng.bf_apply(W, X, out=Y))
    print("pad %3d frames: analysis %.3f ms   apply %.3f ms" % (pad, ta, tb))
    del X, Y

# the fused chain: Y [S][K][T] rows are 32 KiB apart at T = 4096
sfb = eng.FilterBank(design_prototype(M, 4, "g"), M, 4, 1, 2, synthesis=True)
for pad in (0, 16, 48):
    Y = torch.empty((S, K, T + pad), dtype=torch.complex64, device=dev)
    tf = tm(lambda: afb.analysis_beamform(pcm, W, tcount=T, out=Y), n=10)
    ts = tm(lambda: sfb.synthesize(Y, nframes=T), n=10)
    print("pad %3d frames: fused %.3f ms   synthesis %.3f ms" % (pad, tf, ts))
