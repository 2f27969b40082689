"""Adaptive chain at C0 (analysis -> NLMS canceller -> synthesis), whole block against frame chunks small enough for the
snapshots of one chunk to stay in the 256 MiB Infinity Cache between the analysis kernel that writes them and the canceller
that reads them (the chunk buffer is reused, so its lines are overwritten on die).  Checks that the chunked chain gives the
same output as the whole-block chain, then times both.  CHUNKS = frame counts to try (multiples of 16)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng, prototypes
from distant_speech_recognition_amd.pybeamformer import calc_la_delays
from bench import ula_positions, synth_pcm_device, FS
from bench_util import gpu_time

dev = torch.device("cuda", 0)
N, M, m, r, dct = 64, 512, 4, 1, 2
D, K = M >> r, M // 2 + 1
S, T = int(os.environ.get("S", "32")), int(os.environ.get("T", "4096"))
CHUNKS = [int(c) for c in os.environ.get("CHUNKS", "16,32,64,128,256").split(",")]
h, g = prototypes.load(M, m, r)
afb = eng.FilterBank(h, M, m, r, dct)
sfb = eng.FilterBank(g, M, m, r, dct, synthesis=True)
L = (T - afb.processing_delay + afb.lookahead) * D
delays = calc_la_delays(ula_positions(N), -1.306379)
pcm = synth_pcm_device(torch, dev, S, N, L, delays, seed=7)
vs = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (FS / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
Yc = eng.rows_like(X, (S, K, T))
out = torch.empty((S, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)


def whole(nst):
    afb.analysis(pcm, out=X)
    eng.nlms_process(vs, X, nst, out=Yc)
    sfb.synthesize(Yc, out=out)


nst = eng.NLMSState(S, M, N, dev)
whole(nst)
ref = out.clone()
ref_Y = Yc[..., :T].clone()
t_whole = gpu_time(torch, lambda: whole(nst), n=3)[0]
t_ana = gpu_time(torch, lambda: afb.analysis(pcm, out=X), n=3)[0]
t_nl = gpu_time(torch, lambda: eng.nlms_process(vs, X, nst, out=Yc), n=3)[0]
print("whole block: chain %.3f ms (analysis %.3f, canceller %.3f) = %.2f M frames/s" % (t_whole * 1e3, t_ana * 1e3, t_nl * 1e3, S * T / t_whole / 1e6))
Yfull = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
for Tc in CHUNKS:
    Xc = eng.padded_rows((S, K, N, Tc), torch.complex64, dev)
    Ycc = eng.rows_like(Xc, (S, K, Tc))

    def chunked(nst_):
        for c in range(0, T, Tc):
            n = min(Tc, T - c)
            afb.analysis(pcm, t0=c, tcount=n, out=Xc[..., :n])
            eng.nlms_process(vs, Xc[..., :n], nst_, out=Ycc[..., :n])
            Yfull[..., c:c + n].copy_(Ycc[..., :n])
        sfb.synthesize(Yfull, out=out)

    nst2 = eng.NLMSState(S, M, N, dev)
    chunked(nst2)
    err = (Yfull - ref_Y).abs().max().item() / ref_Y.abs().max().item()
    t_eager = gpu_time(torch, lambda: chunked(nst2), n=2)[0]
    # the same launches replayed from a HIP graph (the eager loop is bound by the host: ~25 us per launch from Python)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chunked(nst2)
    torch.cuda.current_stream().wait_stream(side)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        chunked(nst2)
    t = gpu_time(torch, gr.replay, n=3)[0]
    print("   eager %.3f ms" % (t_eager * 1e3))
    print("chunks of %4d frames (%6.1f MB of snapshots): chain %.3f ms = %.2f M frames/s   max |dY| / max |Y| = %.2e"
          % (Tc, S * K * N * Tc * 8 / 1e6, t * 1e3, S * T / t / 1e6, err))
    del Xc, Ycc
