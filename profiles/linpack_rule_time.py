"""Time of the MVDR design at BASELINE config C5 (256 microphones, 1025 bins, diffuse model + 1e-2 loading) with and without
the reference's csvdc rule (svd_rule "linpack" / "exact"), and of btk_csvdc_values alone at N = 64 / 128 / 256."""
import json
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays, gpu_time

dev = torch.device("cuda", 0)
res = {}
for N, M in ((64, 1024), (128, 1024), (256, 2048)):
    K = M // 2 + 1
    mpos = ula_positions(N, 20.0)
    R = eng.mvdr_diffuse_model(mpos, M, 16000, device=dev)
    eng.mvdr_diagonal_loading(R, 0.01)
    wq = torch.from_numpy(eng.weights_mainlobe(M, N, 16000.0, la_delays(mpos, 0.8))[:K].astype(np.complex64)).to(dev)
    t_sv = gpu_time(torch, lambda: eng.csvdc_values(R), n=2, prewarm_ms=50.0, min_ms=50.0, max_calls=4)[0]
    t_lp = gpu_time(torch, lambda: eng.mvdr_weights(R, wq, svd_rule="linpack"), n=2, prewarm_ms=50.0, min_ms=50.0, max_calls=4)[0]
    nid = eng.mvdr_weights.last_counts
    t_ex = gpu_time(torch, lambda: eng.mvdr_weights(R, wq, svd_rule="exact"), n=2, prewarm_ms=50.0, min_ms=50.0, max_calls=4)[0]
    res["N%d_K%d" % (N, K)] = {"csvdc_values_ms": t_sv * 1e3, "mvdr_weights_linpack_ms": t_lp * 1e3, "mvdr_weights_exact_ms": t_ex * 1e3,
                                "bins_info_nonzero": nid[0], "bins_sigma_below_threshold": nid[1]}
# the designs of four streams in one call (BASELINE config C3 shape: 4 x 513 bins of 64 x 64): beyond what LDS holds at once
N, M = 64, 1024
K = M // 2 + 1
mpos = ula_positions(N, 20.0)
R1 = eng.mvdr_diffuse_model(mpos, M, 16000, device=dev)
eng.mvdr_diagonal_loading(R1, 0.01)
R4 = R1.unsqueeze(0).expand(4, K, N, N).contiguous()
wq1 = torch.from_numpy(eng.weights_mainlobe(M, N, 16000.0, la_delays(mpos, 0.8))[:K].astype(np.complex64)).to(dev)
wq4 = wq1.unsqueeze(0).expand(4, K, N).contiguous()
t4 = gpu_time(torch, lambda: eng.mvdr_weights(R4, wq4, svd_rule="linpack"), n=2, prewarm_ms=50.0, min_ms=50.0, max_calls=4)[0]
t4e = gpu_time(torch, lambda: eng.mvdr_weights(R4, wq4, svd_rule="exact"), n=2, prewarm_ms=50.0, min_ms=50.0, max_calls=4)[0]
res["N64_4streams_x_K513"] = {"mvdr_weights_linpack_ms": t4 * 1e3, "mvdr_weights_exact_ms": t4e * 1e3}
print(json.dumps(res))
