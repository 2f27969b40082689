"""engine.nlms_process(interleave=(G, chunk, stagger)) at C0's shape, each form five times in a row (fresh timing each): how often do
the stream groups fall into step?"""
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
X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
X.copy_((torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 2000)
Y = eng.rows_like(X, (S, K, T))
for plan in ((1, T), (2, 1024, False), (2, 1024, True), (4, 512, False), (4, 512, True), (4, 1024, False), (4, 1024, True), (8, 512, True), (3, 768, True), (2, 512, True)):
    ts = []
    for rep in range(5):
        st = eng.NLMSState(S, M, N, dev)
        ts.append(gpu_time(torch, lambda: eng.nlms_process(vd, X, st, out=Y, interleave=plan), n=5, prewarm_ms=100.0)[0] * 1e3)
    print("%-18s %s ms" % (plan, " ".join("%.2f" % t for t in ts)), flush=True)
