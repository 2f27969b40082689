"""Why bench.py read 1.80 ms for the fused kernel where profiles/fused_ab.py read 1.62: same kernels timed in loops with
bench.py's own PCM generator and with noise, kernel alone and in the chain -- no data dependence; the gap was the GPU
clock ramp after the seconds of host-side weight design (bench.py now pre-warms for 100 ms, --prewarm-ms)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
from distant_speech_recognition_amd import engine as eng
from bench_util import design_prototype, ula_positions, la_delays
dev = torch.device("cuda:0")
N, M, m, r, dct, S, T = 64, 512, 4, 1, 2, 16, 4096
D, K = M >> r, M // 2 + 1
afb = eng.FilterBank(design_prototype(M, m), M, m, r, dct)
sfb = eng.FilterBank(design_prototype(M, m, "g"), M, m, r, dct, synthesis=True)
L = (T - afb.processing_delay + afb.lookahead) * D
delays = la_delays(ula_positions(N), -1.306379)
pcm_speech = bench.synth_pcm_device(torch, dev, S, N, L, delays, seed=1)
pcm_noise = (torch.randn((S, N, L), device=dev) * 1000.0).round_()
wq = eng.weights_mainlobe(M, N, 16000.0, delays)
W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
Y = eng.padded_rows((S, K, T), torch.complex64, dev)
out = torch.empty((S, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)
def tm(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("fused only, speech pcm : %.3f ms" % tm(lambda: afb.analysis_beamform(pcm_speech, W, out=Y)))
print("fused only, noise pcm  : %.3f ms" % tm(lambda: afb.analysis_beamform(pcm_noise, W, out=Y)))
def chain(p):
    afb.analysis_beamform(p, W, out=Y); sfb.synthesize(Y, out=out)
print("chain, speech pcm      : %.3f ms" % tm(lambda: chain(pcm_speech)))
print("chain, noise pcm       : %.3f ms" % tm(lambda: chain(pcm_noise)))
print("pcm stats: speech absmax %.0f, zeros %.3f; dtype %s" % (float(pcm_speech.abs().max()), float((pcm_speech == 0).float().mean()), pcm_speech.dtype))
