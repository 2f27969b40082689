"""McCowan / Lefkimmiatis launch at the C0 shape for rocprofv3 (kernel trace or --pmc): PF_LEF=1 Lefkimmiatis, PF_PAD=1 row-padded
snapshots, PF_S streams (16), PF_REPS launches (4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distant_speech_recognition_amd import engine as eng
from bench_util import ula_positions, la_delays
dev = torch.device("cuda:0")
N, M, T = 64, 512, 4096
S = int(os.environ.get("PF_S", 16)); K = M // 2 + 1
lef = os.environ.get("PF_LEF", "0") == "1"
shape = (S, K, N, T)
X = eng.padded_rows(shape, torch.complex64, dev) if os.environ.get("PF_PAD", "0") == "1" else torch.empty(shape, dtype=torch.complex64, device=dev)
X.copy_((torch.randn(shape, device=dev) + 1j * torch.randn(shape, device=dev)).to(torch.complex64) * 2000)
delays = la_delays(ula_positions(N), -1.3)
vd = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
mp = ula_positions(N); mp[:, 2] = 2.0
R = eng.mvdr_diffuse_model(mp, M, 16000, device=dev); eng.mvdr_diagonal_loading(R, 0.01)
cs = eng.CoherencePostFilterState(S, K, N, dev, lefkimmiatis=lef)
cs.set_coherence(R, 0.99)
if lef: cs.set_lambda(R, vd, 1e-4)
fn = (lambda: eng.bf_apply_lefkimmiatis(vd, vd, X, cs, fbin_x1=100, alpha=0.8)) if lef else (lambda: eng.bf_apply_mccowan(vd, vd, X, cs, alpha=0.7))
for _ in range(int(os.environ.get("PF_REPS", 4))): fn()
torch.cuda.synchronize()
