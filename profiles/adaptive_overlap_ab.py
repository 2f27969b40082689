"""Adaptive chain at C0 (analysis -> NLMS canceller -> synthesis): does running the analysis bank of one group of streams BESIDE the
canceller of another group help?  The canceller keeps two wavefronts per SIMD at 128 VGPRs and reads 3.5 TB/s; the staged analysis
bank needs 64 VGPRs and streams 5 TB/s.  Variants, all inside one "step" (no work carried across steps):
  serial      one launch of each kernel over all S streams (the round-4 chain)
  groups G    the S streams split into G groups, each group's analysis -> canceller -> synthesis on its own HIP stream
  pipelined   two streams, groups of S/G streams: analysis of group g+1 is issued on the second stream while the canceller of group g runs
Outputs are compared with the serial chain (must be bit-identical: streams are independent)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng, prototypes
from distant_speech_recognition_amd.pybeamformer import calc_la_delays
from bench import ula_positions, synth_pcm_device, FS

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
Yc = eng.rows_like(X, (S, K, T))
out = torch.empty((S, sfb.num_blocks(T) * D), dtype=torch.float32, device=dev)


def timeit(fn, reps=12, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def serial(nst):
    afb.analysis(pcm, out=X)
    eng.nlms_process(vs, X, nst, out=Yc)
    sfb.synthesize(Yc, out=out)


nst = eng.NLMSState(S, M, N, dev)
serial(nst)
torch.cuda.synchronize()
ref = out.clone()
t_ser = timeit(lambda: serial(nst))
print("serial: %.3f ms = %.2f M frames/s" % (t_ser * 1e3, S * T / t_ser / 1e6), flush=True)

streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
for G in (2, 4, 8):
    if S % G:
        continue
    Sg = S // G
    states = [eng.NLMSState(Sg, M, N, dev) for _ in range(G)]
    pg = [pcm[i * Sg:(i + 1) * Sg] for i in range(G)]
    Xg = [X[i * Sg:(i + 1) * Sg] for i in range(G)]
    Yg = [Yc[i * Sg:(i + 1) * Sg] for i in range(G)]
    og = [out[i * Sg:(i + 1) * Sg] for i in range(G)]

    def groups(nstreams):
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        done = []
        for i in range(G):
            st = streams[i % nstreams]
            st.wait_event(ev0)
            with torch.cuda.stream(st):
                afb.analysis(pg[i], out=Xg[i])
                eng.nlms_process(vs, Xg[i], states[i], out=Yg[i])
                sfb.synthesize(Yg[i], out=og[i])
                e = torch.cuda.Event()
                e.record(st)
                done.append(e)
        for e in done:
            cur.wait_event(e)

    for ns in sorted({2, G}):
        for stt in states:
            stt.reset_stats()
        out.zero_()
        groups(ns)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        t = timeit(lambda: groups(ns))
        print("groups G=%d on %d HIP streams: %.3f ms = %.2f M frames/s   (first pass bit-identical to serial: %s)"
              % (G, ns, t * 1e3, S * T / t / 1e6, same), flush=True)


    def staggered():
        # group g's analysis starts when group g-1's has finished: its canceller then runs beside the next group's analysis
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        done, prev_ana = [], None
        for i in range(G):
            st = streams[i]
            st.wait_event(ev0)
            if prev_ana is not None:
                st.wait_event(prev_ana)
            with torch.cuda.stream(st):
                afb.analysis(pg[i], out=Xg[i])
                prev_ana = torch.cuda.Event()
                prev_ana.record(st)
                eng.nlms_process(vs, Xg[i], states[i], out=Yg[i])
                sfb.synthesize(Yg[i], out=og[i])
                e = torch.cuda.Event()
                e.record(st)
                done.append(e)
        for e in done:
            cur.wait_event(e)

    for stt in states:
        stt.reset_stats()
    out.zero_()
    staggered()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    t = timeit(staggered)
    print("staggered G=%d (analysis of group g+1 beside the canceller of group g): %.3f ms = %.2f M frames/s   (bit-identical: %s)"
          % (G, t * 1e3, S * T / t / 1e6, same), flush=True)

# ---- frame chunks, all streams: the analysis bank runs ahead on one HIP stream (chunk after chunk into the one snapshot buffer), the
# canceller follows on a second stream as each chunk's snapshots are complete; the canceller keeps its full occupancy (all S streams)
sA, sB = streams[0], streams[1]
for Tc in (512, 1024, 2048):
    nch = (T + Tc - 1) // Tc
    nst2 = eng.NLMSState(S, M, N, dev)

    def chunked():
        cur = torch.cuda.current_stream()
        ev0 = torch.cuda.Event()
        ev0.record(cur)
        sA.wait_event(ev0); sB.wait_event(ev0)
        for c in range(nch):
            a, n = c * Tc, min(Tc, T - c * Tc)
            with torch.cuda.stream(sA):
                afb.analysis(pcm, t0=a, tcount=n, out=X[..., a:a + n])
                e = torch.cuda.Event()
                e.record(sA)
            sB.wait_event(e)
            with torch.cuda.stream(sB):
                eng.nlms_process(vs, X[..., a:a + n], nst2, out=Yc[..., a:a + n])
        with torch.cuda.stream(sB):
            sfb.synthesize(Yc, out=out)
            e = torch.cuda.Event()
            e.record(sB)
        cur.wait_event(e)

    nst2.reset_stats()
    out.zero_()
    chunked()
    torch.cuda.synchronize()
    same = torch.equal(out, ref)
    t = timeit(chunked)
    print("frame chunks of %d (analysis runs ahead on stream A, canceller follows on stream B): %.3f ms = %.2f M frames/s   (bit-identical: %s)"
          % (Tc, t * 1e3, S * T / t / 1e6, same), flush=True)
