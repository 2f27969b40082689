"""A/B of two builds of the library on the C0 NLMS launch: python profiles/nlms_lib_ab.py [path/to/libbtkhip.so]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from distant_speech_recognition_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays, gpu_time
dev = torch.device("cuda:0")
for (S, N, M, T) in ((32, 64, 512, 4096), (128, 64, 512, 1024)):
    K = M // 2 + 1
    X = eng.padded_rows((S, K, N, T), torch.complex64, dev)
    X.copy_((torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 2000)
    delays = la_delays(ula_positions(N), -1.306379)
    vd = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
    st = eng.NLMSState(S, M, N, dev)
    Y = eng.rows_like(X, (S, K, T), torch.complex64)
    def step():
        st.reset_stats()
        eng.nlms_process(vd, X, st, out=Y)
    t = gpu_time(torch, step, n=10, prewarm_ms=300.0)[0]
    b = 8 * K * (N + 1) * S * T
    print("%s N=%d S=%d: %.3f ms  %.0f GB/s (%.1f%%)" % (os.path.basename(_lib.LIB_PATH), N, S, t * 1e3, b / t / 1e9, 100 * b / t / 8e12))
    del X, Y
