"""NLMS canceller at C0's shape against the number of streams: is 32 streams (2 080 single-wavefront workgroups of 18 KB LDS) one
residency round or two?  A CU holds 8 such workgroups (160 KB / 18 KB): 2 048 on the chip."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays
dev = torch.device("cuda:0")
N, M, T = 64, 512, 4096
K = M // 2 + 1
delays = la_delays(ula_positions(N), -1.306379)
vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)
vd = torch.from_numpy(vs).to(dev)
Xall = (torch.randn((34, K, N, T), device=dev) + 1j * torch.randn((34, K, N, T), device=dev)).to(torch.complex64) * 2000
for S in (8, 16, 24, 30, 31, 32, 33, 34):
    X = Xall[:S]
    st = eng.NLMSState(S, M, N, dev)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    for _ in range(2): eng.nlms_process(vd, X, st, out=Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.nlms_process(vd, X, st, out=Y)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    print("S=%d (%d workgroups): %.3f ms  %.2f M frames/s" % (S, S * ((K + 3) // 4), t * 1e3, S * T / t / 1e6), flush=True)
