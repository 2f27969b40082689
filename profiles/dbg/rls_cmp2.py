import os, sys, subprocess
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from distant_speech_recognition_amd import engine as eng
from tests.util import ula_positions, la_delays
dev = torch.device("cuda:0")
N, M, S = 8, 32, 1
K = M // 2 + 1
rng = np.random.default_rng(N * 1000 + M)
X = ((rng.normal(size=(S, K, N, 2500)) + 1j * rng.normal(size=(S, K, N, 2500))) * 1000).astype(np.complex64)
delays = la_delays(ula_positions(N), -1.306379)
vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
for T in (16, 64, 400, 800, 1200):
    st = eng.RLSState(1, S, M, N, torch.from_numpy(vs).to(dev), min_frames=64)
    eng.rls_process(torch.from_numpy(X[..., :T].copy()).to(dev), st)
    P = st.P.cpu().numpy()[0]; w = st.w.cpu().numpy()[0]
    k = 5
    n = vs[k] / np.linalg.norm(vs[k])
    herm = np.abs(P[k] - P[k].conj().T).max() / np.abs(P[k]).max()
    ev = np.linalg.eigvalsh((P[k] + P[k].conj().T) / 2)
    print(os.environ.get("BTK_RLS_PACKED", "reg"), "T", T, "|P n|/|P|", np.linalg.norm(P[k] @ n) / np.linalg.norm(P[k]), "herm", herm, "eig min/max", ev[0], ev[-1], "|w.n|", abs(w[k] @ n))
