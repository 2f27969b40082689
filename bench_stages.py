#!/usr/bin/env python
"""bench_stages.py -- per-kernel measurements of the stages that are not in the headline chain
(adaptive NLMS canceller, fused apply+Zelinski, covariance HERK, MVDR solve, WPE), each against its
roofline.  Not the driver's bench (that is bench.py); results are copied into profiles/ and DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM, FP32 = 8.0e12, 157.3e12


def timeit(torch, fn, n=5, warm=2, prewarm_ms=250.0):
    """HIP-event time per call at settled clocks (bench_util.gpu_time)"""
    from bench_util import gpu_time
    return gpu_time(torch, fn, n=n, prewarm_ms=prewarm_ms)[0]


def main():
    import torch
    from distant_speech_recognition_amd import engine as eng
    from bench_util import ula_positions, la_delays
    dev = torch.device("cuda:0")
    out = {}
    # ---- C0 snapshots: 64 mics, 512 bins
    S, N, M, T = 16, 64, 512, 4096
    K = M // 2 + 1
    X = torch.randn((S, K, N, T), dtype=torch.float32, device=dev).to(torch.complex64) * 2000
    X = X + 1j * torch.randn((S, K, N, T), dtype=torch.float32, device=dev) * 2000
    delays = la_delays(ula_positions(N), -1.306379)
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)
    vd = torch.from_numpy(vs).to(dev)
    st = eng.NLMSState(S, M, N, dev)
    Y = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    t = timeit(torch, lambda: eng.nlms_process(vd, X, st, out=Y))
    b = 8 * K * (N + 1) * S * T
    out["nlms_c0"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "GBps": b / t / 1e9, "hbm_frac": b / t / HBM}
    # the recursion is sequential in t: with 16 streams the chip holds one wavefront per SIMD; the same number of frames
    # spread over 128 streams (what a loaded server sees) fills it
    S2, T2 = 128, S * T // 128
    X2 = X.reshape(S, K, N, S2 // S, T2).permute(0, 3, 1, 2, 4).reshape(S2, K, N, T2).contiguous()
    st2 = eng.NLMSState(S2, M, N, dev)
    Y2 = torch.empty((S2, K, T2), dtype=torch.complex64, device=dev)
    t = timeit(torch, lambda: eng.nlms_process(vd, X2, st2, out=Y2))
    out["nlms_c0_128streams"] = {"ms": t * 1e3, "frames_per_s": S2 * T2 / t, "GBps": b / t / 1e9, "hbm_frac": b / t / HBM}
    del X2, Y2
    # the same two kernels on row-padded snapshots (analysis(pad_rows=True): rows 48 frames wider than the 32 KiB power of two)
    Xp = eng.padded_rows((S, K, N, T), torch.complex64, dev)
    Xp.copy_(X)
    stp = eng.NLMSState(S, M, N, dev)
    t = timeit(torch, lambda: eng.nlms_process(vd, Xp, stp))
    out["nlms_c0_padded_rows"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "GBps": b / t / 1e9, "hbm_frac": b / t / HBM}
    t = timeit(torch, lambda: eng.bf_apply(vd, X))
    out["apply_c0"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "GBps": b / t / 1e9, "hbm_frac": b / t / HBM}
    t = timeit(torch, lambda: eng.bf_apply(vd, Xp))
    out["apply_c0_padded_rows"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "GBps": b / t / 1e9, "hbm_frac": b / t / HBM}
    zsp = eng.ZelinskiState(S, K, dev)
    t = timeit(torch, lambda: eng.bf_apply_zelinski(vd, vd, Xp, zsp, alpha=0.7))
    out["apply_zelinski_c0_padded_rows"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "GBps": b / t / 1e9, "hbm_frac": b / t / HBM}
    del Xp
    zs = eng.ZelinskiState(S, K, dev)
    t = timeit(torch, lambda: eng.bf_apply_zelinski(vd, vd, X, zs, alpha=0.7, out=Y))
    out["apply_zelinski_c0"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "GBps": b / t / 1e9, "hbm_frac": b / t / HBM}
    # McCowan / Lefkimmiatis: apply + the per-pair weighted quadratic forms (O(N^2) flops per bin-frame)
    mpos0 = ula_positions(N); mpos0[:, 2] = 2.0
    Rc = eng.mvdr_diffuse_model(mpos0, M, 16000, device=dev)
    eng.mvdr_diagonal_loading(Rc, 0.01)
    for lef, tag in ((False, "apply_mccowan_c0"), (True, "apply_lefkimmiatis_c0")):
        cs = eng.CoherencePostFilterState(S, K, N, dev, lefkimmiatis=lef)
        cs.set_coherence(Rc, 0.99)
        if lef:
            cs.set_lambda(Rc, vd, 1e-4)
            fn = lambda: eng.bf_apply_lefkimmiatis(vd, vd, X, cs, fbin_x1=100, alpha=0.8, out=Y)
        else:
            fn = lambda: eng.bf_apply_mccowan(vd, vd, X, cs, alpha=0.7, out=Y)
        t = timeit(torch, fn, n=3, warm=1)
        fl = 8.0 * (N * (N + 1) / 2) * (2 if lef else 1) * K * S * T
        out[tag] = {"ms": t * 1e3, "frames_per_s": S * T / t, "TFLOPs_quadratic_forms": fl / t / 1e12, "GBps": b / t / 1e9}
    R = torch.zeros((S, K, N, N), dtype=torch.complex64, device=dev)
    for mf, tag in ((True, "cov_mfma_c0"), (False, "cov_valu_c0")):
        t = timeit(torch, lambda: eng.cov_accumulate(X, R=R, use_mfma=mf), n=3, warm=1)
        fl = 8.0 * K * N * N * S * T
        out[tag] = {"ms": t * 1e3, "frames_per_s": S * T / t, "TFLOPs": fl / t / 1e12, "fp32_frac": fl / t / FP32,
                    "GBps_read": 8 * K * N * S * T / t / 1e9}
    del X, R, Y
    # ---- C1-style RLS canceller: 8 mics, 512 bins, 16 streams (float64 recursion, O(N^2) per bin-frame)
    N8 = 8
    X8 = (torch.randn((S, K, N8, T), device=dev) + 1j * torch.randn((S, K, N8, T), device=dev)).to(torch.complex64) * 2000
    d8 = la_delays(ula_positions(N8), -1.306379)
    vs8 = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * d8) / N8 for k in range(K)])).to(dev)
    rs = eng.RLSState(1, S, M, N8, vs8, min_frames=0)
    Y8 = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    t = timeit(torch, lambda: eng.rls_process(X8, rs, out=Y8), n=3, warm=1)
    out["rls_8mic"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "bin_frames_per_s": S * K * T / t,
                       "GFLOPs_f64": 8.0 * 5 * N8 * N8 * S * K * T / t / 1e9}
    # round 3: the packed-Hermitian LDS form: two constraints at 8 mics, and a 100-microphone array (one stream, 1024 frames)
    rs2 = eng.RLSState(1, S, M, N8, vs8, Nc=2, min_frames=0)
    t = timeit(torch, lambda: eng.rls_process(X8, rs2, out=Y8), n=2, warm=1)
    out["rls_8mic_nc2_packed"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "bin_frames_per_s": S * K * T / t}
    N100, T100 = 100, 1024
    X100 = (torch.randn((1, K, N100, T100), device=dev) + 1j * torch.randn((1, K, N100, T100), device=dev)).to(torch.complex64) * 2000
    d100 = la_delays(ula_positions(N100), -1.306379)
    vs100 = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * d100) / N100 for k in range(K)])).to(dev)
    rs100 = eng.RLSState(1, 1, M, N100, vs100, min_frames=0)
    t = timeit(torch, lambda: eng.rls_process(X100, rs100), n=2, warm=1)
    out["rls_100mic_packed"] = {"ms": t * 1e3, "frames_per_s": T100 / t, "bin_frames_per_s": K * T100 / t,
                                "GFLOPs_f64": 8.0 * 3 * N100 * N100 * K * T100 / t / 1e9,
                                "note": "one workgroup per bin, P packed in 81 KB of LDS: 257 workgroups on 256 CUs, sequential in t"}
    del X100
    st8 = eng.NLMSState(S, M, N8, dev)
    v8c = vs8.to(torch.complex64)
    t = timeit(torch, lambda: eng.nlms_process(v8c, X8, st8, out=Y8))
    b8 = 8 * K * (N8 + 1) * S * T
    out["nlms_8mic"] = {"ms": t * 1e3, "frames_per_s": S * T / t, "GBps": b8 / t / 1e9, "hbm_frac": b8 / t / HBM}
    del X8, Y8
    # ---- blind MVDR / GEV weight design from covariances: 64 mics, 257 bins
    A = torch.randn((K, N, 2 * N), device=dev) + 1j * torch.randn((K, N, 2 * N), device=dev)
    Rn = (A @ A.conj().transpose(1, 2) / (2 * N) + 0.05 * torch.eye(N, device=dev)).to(torch.complex64)
    A = torch.randn((K, N, 2), device=dev) + 1j * torch.randn((K, N, 2), device=dev)
    Rt = (A @ A.conj().transpose(1, 2)).to(torch.complex64)
    t = timeit(torch, lambda: eng.bmvdr_weights(Rt, Rn), n=2, warm=1)
    out["bmvdr_weights_64mic"] = {"ms": t * 1e3, "bins": K}
    t = timeit(torch, lambda: eng.gev_weights(Rt, Rn), n=2, warm=1)
    out["gev_weights_64mic"] = {"ms": t * 1e3, "bins": K, "squarings": 32}
    del Rn, Rt, A
    # ---- C3: MVDR solve, 64 mics, 1024 bins
    N, M = 64, 1024
    K = M // 2 + 1
    mpos = ula_positions(N); mpos[:, 2] = 2.0
    Rd = eng.mvdr_diffuse_model(mpos, M, 16000, device=dev)
    eng.mvdr_diagonal_loading(Rd, 0.01)
    wq = eng.weights_mainlobe(M, N, 16000, la_delays(mpos, -1.3))[:K].astype(np.complex64)
    wd = torch.from_numpy(wq).to(dev)
    t = timeit(torch, lambda: eng.mvdr_weights(Rd, wd), n=3, warm=1)
    out["mvdr_solve_c3"] = {"ms": t * 1e3, "bins": K, "N": N, "GFLOPs": (32.0 / 3) * K * N ** 3 / t / 1e9}
    # ---- C4-style WPE: 8 mics, 512 bins, lags 0..32, one stream of 1000 frames
    C, M, T = 8, 512, 1000
    K = M // 2 + 1
    Xw = (torch.randn((1, K, C, T), device=dev) + 1j * torch.randn((1, K, C, T), device=dev)).to(torch.complex64) * 500
    G = eng.wpe_estimate(Xw, M, 0, 32, 2, -18.0, 0.0, 1e-4)                 # (first call: workspace allocation, kernel attributes)
    t = timeit(torch, lambda: eng.wpe_estimate(Xw, M, 0, 32, 2, -18.0, 0.0, 1e-4), n=3, warm=1)
    P = C * 33
    out["wpe_estimate_c4"] = {"ms": t * 1e3, "iterations": 2, "P": P, "herk_TFLOPs": 2 * 8.0 * K * C * T * P * P / 2 / t / 1e12}
    t = timeit(torch, lambda: eng.wpe_apply(Xw, G, M, 0, 32), n=3, warm=1)
    out["wpe_apply_c4"] = {"ms": t * 1e3, "frames_per_s": T / t}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
