"""NLMS canceller, C0 shape, 32 streams x 4096 frames: one launch (2 080 workgroups: a 2 048-workgroup round + a tail of 32) against
the same work cut into frame chunks and G independent groups of streams on G HIP streams -- the groups drift out of step, so a
group's tail runs beside the next chunk of the others and the chip stays full."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays, gpu_time
dev = torch.device("cuda:0")
S, N, M, T = 32, 64, 512, 4096
K = M // 2 + 1
delays = la_delays(ula_positions(N), -1.306379)
vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)
vd = torch.from_numpy(vs).to(dev)
X = (torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 2000
Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
st = eng.NLMSState(S, M, N, dev)
t = gpu_time(torch, lambda: eng.nlms_process(vd, X, st, out=Y), n=5)[0]
print("one launch: %.3f ms" % (t * 1e3), flush=True)
for G in (2, 4):
    for Tc in (256, 512, 1024, 4096):
        streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
        states = [eng.NLMSState(S // G, M, N, dev) for _ in range(G)]
        Xg = [X[g * (S // G):(g + 1) * (S // G)] for g in range(G)]
        Yg = [Y[g * (S // G):(g + 1) * (S // G)] for g in range(G)]

        def run():
            cur = torch.cuda.current_stream()
            e0 = torch.cuda.Event(); e0.record(cur)
            for g in range(G):
                streams[g].wait_event(e0)
            for a in range(0, T, Tc):
                for g in range(G):
                    with torch.cuda.stream(streams[g]):
                        eng.nlms_process(vd, Xg[g][..., a:a + Tc], states[g], out=Yg[g][..., a:a + Tc])
            for g in range(G):
                e = torch.cuda.Event(); e.record(streams[g]); cur.wait_event(e)
        t = gpu_time(torch, run, n=5)[0]
        print("%d groups of %d streams on %d HIP streams, chunks of %4d frames: %.3f ms" % (G, S // G, G, Tc, t * 1e3), flush=True)
