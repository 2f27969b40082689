"""A/B of the McCowan / Lefkimmiatis statistics kernel row-block size (BTK_PF_JB=8|16) at C0 shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays
dev = torch.device("cuda:0")
for N in [int(v) for v in os.environ.get("PF_NS", "8,64").split(",")]:
    S, M, T = 16, 512, 4096
    K = M // 2 + 1
    X = (torch.randn((S, K, N, T), device=dev) + 1j * torch.randn((S, K, N, T), device=dev)).to(torch.complex64) * 2000
    delays = la_delays(ula_positions(N), -1.3)
    vd = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
    mp = ula_positions(N); mp[:, 2] = 2.0
    R = eng.mvdr_diffuse_model(mp, M, 16000, device=dev); eng.mvdr_diagonal_loading(R, 0.01)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    for lef in (False, True):
        cs = eng.CoherencePostFilterState(S, K, N, dev, lefkimmiatis=lef)
        cs.set_coherence(R, 0.99)
        if lef: cs.set_lambda(R, vd, 1e-4)
        fn = (lambda: eng.bf_apply_lefkimmiatis(vd, vd, X, cs, fbin_x1=100, alpha=0.8, out=Y)) if lef else \
             (lambda: eng.bf_apply_mccowan(vd, vd, X, cs, alpha=0.7, out=Y))
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): fn()
        e1.record(); torch.cuda.synchronize()
        print("JB=%s N=%d %s: %.3f ms" % (os.environ.get("BTK_PF_JB", "default"), N, "lefkimmiatis" if lef else "mccowan", e0.elapsed_time(e1) / 3))
