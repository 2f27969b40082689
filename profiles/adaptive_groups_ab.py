"""Adaptive chain at C0 (32 streams x 4096 frames: analysis -> NLMS canceller -> synthesis): one launch per kernel, the round-5
overlapped form (bank ahead on a second HIP stream, one canceller launch per chunk), and the round-6 form (the canceller of a
chunk as G groups of streams on G HIP streams).  Outputs must be bit-identical."""
import os, sys
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
h, g = prototypes.load(M, m, r)
afb = eng.FilterBank(h, M, m, r, dct)
sfb = eng.FilterBank(g, M, m, r, dct, synthesis=True)
L = (T - afb.processing_delay + afb.lookahead) * D
delays = calc_la_delays(ula_positions(N), -1.306379)
pcm = synth_pcm_device(torch, dev, S, N, L, delays, seed=7)
vs = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (FS / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
Y = eng.rows_like(X, (S, K, T))
out = torch.empty((S, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)


def whole(st):
    afb.analysis(pcm, out=X)
    eng.nlms_process(vs, X, st, out=Y, interleave=(1, T))
    sfb.synthesize(Y, out=out)


st = eng.NLMSState(S, M, N, dev)
whole(st)
ref = out.clone()
t = gpu_time(torch, lambda: whole(st), n=3)[0]
print("one launch per kernel: %.3f ms = %.2f M frames/s" % (t * 1e3, S * T / t / 1e6), flush=True)
X2 = eng.padded_rows((S, K, N, T), torch.complex64, dev)
X2.copy_(X)
for G, Tc in ((1, T), (2, 1024), (4, 1024), (4, 512), (8, 1024)):
    st2 = eng.NLMSState(S, M, N, dev)
    eng.nlms_process(vs, X2, st2, out=Y, interleave=(G, Tc))
    st3 = eng.NLMSState(S, M, N, dev)
    Yr = eng.rows_like(X, (S, K, T))
    eng.nlms_process(vs, X2, st3, out=Yr, interleave=(1, T))
    same = bool(torch.equal(Y[..., :T], Yr[..., :T])) and bool(torch.equal(st2.u, st3.u))
    t = gpu_time(torch, lambda: eng.nlms_process(vs, X2, st2, out=Y, interleave=(G, Tc)), n=5)[0]
    print("canceller alone, %d groups x chunks of %4d: %.3f ms  bit-identical %s" % (G, Tc, t * 1e3, same), flush=True)
for chunk in (512, 1024):
    for G in (1, 2, 4, 8):
        chain = eng.AdaptiveGSCChain(afb, sfb, chunk_frames=chunk, groups=G)
        st2 = eng.NLMSState(S, M, N, dev)
        chain(pcm, vs, st2, X, Y, out=out)
        torch.cuda.synchronize()
        same = bool(torch.equal(out, ref))
        t = gpu_time(torch, lambda: chain(pcm, vs, st2, X, Y, out=out), n=5)[0]
        print("chain, bank ahead in chunks of %4d, canceller in %d groups: %.3f ms = %.2f M frames/s  bit-identical %s"
              % (chunk, G, t * 1e3, S * T / t / 1e6, same), flush=True)
