"""Timing of the RLS sidelobe canceller kernel (btk_rls_process, float64) on synthetic snapshots.
usage: python profiles/rls_bench.py   -> one line per configuration"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng

dev = torch.device("cuda:0")
for (N, S, M, T) in ((4, 16, 512, 4096), (8, 16, 512, 4096), (16, 8, 512, 2048), (32, 4, 512, 1024), (64, 2, 512, 512)):
    K = M // 2 + 1
    g = torch.Generator(device="cpu").manual_seed(N)
    X = (torch.randn((S, K, N, T, 2), generator=g) * 2000.0)
    X = torch.view_as_complex(X).to(dev)
    vs = torch.from_numpy(np.exp(-2j * np.pi * np.random.default_rng(N).random((K, N))) / N).to(dev)
    for mode in (1, 0):
        st = eng.RLSState(mode, S, M, N, vs, **({"min_frames": 0} if mode == 1 else {}))
        if mode == 0:
            st.init_precision_matrix(1.0e-6)
        Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
        eng.rls_process(X, st, out=Y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        reps = 3
        for _ in range(reps):
            eng.rls_process(X, st, out=Y)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = S * K * T * (8.0 * 3 * N * N + 8.0 * 2 * N * N)          # 3 mat-vecs + two rank-1 updates, complex f64
        print("RLS mode %d N=%2d S=%2d K=%d T=%d: %8.2f ms  %7.3f M bin-frames/s  %6.0f kframes/s  ~%5.2f TFLOP/s f64 (finite=%s)"
              % (mode, N, S, K, T, ms, S * K * T / ms / 1e3, S * T / ms, fl / ms / 1e9, bool(torch.isfinite(torch.view_as_real(Y)).all())))
