import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np
if len(sys.argv) > 1:
    import torch
    from distant_speech_recognition_amd import engine as eng
    from tests.util import ula_positions, la_delays
    dev = torch.device("cuda:0")
    N, M, T, S = 8, 32, 2500, 1
    K = M // 2 + 1
    rng = np.random.default_rng(N * 1000 + M)
    X = ((rng.normal(size=(S, K, N, T)) + 1j * rng.normal(size=(S, K, N, T))) * 1000).astype(np.complex64)
    delays = la_delays(ula_positions(N), -1.306379)
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
    st = eng.RLSState(1, S, M, N, torch.from_numpy(vs).to(dev), min_frames=64)
    Xd = torch.from_numpy(X).to(dev)
    split = int(sys.argv[2])
    if split:
        Y = torch.cat([eng.rls_process(Xd[..., :split].contiguous(), st), eng.rls_process(Xd[..., split:].contiguous(), st)], dim=-1)
    else:
        Y = eng.rls_process(Xd, st)
    np.save(sys.argv[1], Y.cpu().numpy())
else:
    for split in (0, 1253):
        subprocess.run([sys.executable, __file__, "/tmp/y_reg.npy", str(split)], check=True)
        subprocess.run([sys.executable, __file__, "/tmp/y_pk.npy", str(split)], check=True, env=dict(os.environ, BTK_RLS_PACKED="1"))
        a, b = np.load("/tmp/y_reg.npy")[0], np.load("/tmp/y_pk.npy")[0]
        err = np.abs(a - b).max(axis=0) / np.abs(a).max()
        first = np.argmax(err > 1e-6) if (err > 1e-6).any() else -1
        print("split", split, "max rel diff", err.max(), "first frame > 1e-6:", first, "err at frames", [float("%.2g" % err[t]) for t in (10, 100, 500, 1000, 1252, 1260, 1300, 2000, 2499)])
