import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays
dev = torch.device("cuda:0")
for (S, N, M, T) in ((16, 64, 512, 4096), (32, 64, 512, 4096), (128, 64, 512, 1024), (64, 8, 512, 4096)):
    K = M // 2 + 1
    X = (torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 2000
    delays = la_delays(ula_positions(N), -1.306379)
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)
    vd = torch.from_numpy(vs).to(dev)
    st = eng.NLMSState(S, M, N, dev)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    for _ in range(2): eng.nlms_process(vd, X, st, out=Y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.nlms_process(vd, X, st, out=Y)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    b = 8 * K * (N + 1) * S * T
    print("N=%d S=%d: %.3f ms  %.2f M frames/s  %.0f GB/s (%.1f%% of 8TB/s)" % (N, S, t * 1e3, S * T / t / 1e6, b / t / 1e9, 100 * b / t / 8e12))
    del X, Y
