"""GPU parity: RLS sidelobe cancellers (SubbandGSCRLSBeamformer / SubbandGSCRLS) vs the oracle and vs the
golden outputs of the reference's own Python arithmetic (tests/golden/gen_golden_pybeamformer_rls.py)."""
import os

import numpy as np
import pytest

from tests.util import ula_positions, la_delays

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rlsgolden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "pybeamformer_rls_golden.npz"))


def _to_engine_layout(X, K):
    """oracle frames [T][N][M] -> engine [1][K][N][T] complex64"""
    return np.ascontiguousarray(np.transpose(X[:, :, :K], (2, 1, 0))[None]).astype(np.complex64)


def _full(Xe, M):
    """engine [K][N][T] complex64 -> oracle frames [T][N][M] complex128 (what the GPU saw, mirrored)"""
    K = M // 2 + 1
    X = np.transpose(Xe.astype(np.complex128), (2, 1, 0))
    full = np.zeros(X.shape[:2] + (M,), np.complex128)
    full[..., :K] = X
    full[..., K:] = np.conj(X[..., M // 2 - 1:0:-1])
    return full


def _random_frames(rng, S, T, N, M, scale=2000.0):
    K = M // 2 + 1
    Xs = (rng.normal(size=(S, T, N, M)) + 1j * rng.normal(size=(S, T, N, M))) * scale
    # a coherent component so the canceller has something to learn
    d = np.exp(-2j * np.pi * rng.random((N, 1)) * np.arange(M)[None, :] / 7.0)
    Xs = Xs + (rng.normal(size=(S, T, 1, M)) + 1j * rng.normal(size=(S, T, 1, M))) * 3.0 * scale * d
    Xs[..., 0] = Xs[..., 0].real
    Xs[..., M // 2] = Xs[..., M // 2].real
    return np.concatenate([_to_engine_layout(Xs[s], K) for s in range(S)])


@pytest.mark.parametrize("tag", ["rls_default", "rls_constrained", "rls_quadonly", "rlsnc2_default", "rlsnc2_constrained"])
def test_rls_vs_reference_python_golden(orc, dev, proto256, kinect_pcm, pygolden, rlsgolden, tag):
    """Real 4-mic Kinect data: GPU output and state against what the REFERENCE's pybeamformer.py produced (Nc = 1 and, round 3,
    Nc = 2: tests/golden/pybeamformer_rls_nc_golden.npz)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    G = rlsgolden if not tag.startswith("rlsnc") else np.load(os.path.join(os.path.dirname(__file__), "golden", "pybeamformer_rls_nc_golden.npz"))
    Nc = int(G["meta_Nc"][0])
    T, M, N, K = int(G["meta_T"][0]), 256, 4, 129
    h, _ = proto256
    X = np.stack([orc.analysis(h, M, 4, 1, 2, kinect_pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)
    delays = pygolden["delays_kinect"]
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
    p = G[tag + "_params"]
    st = eng.RLSState(1, 1, M, N, torch.from_numpy(vs).to(dev), Nc=Nc, beta=p[0], gamma=p[1], mu=p[2], init_diagonal_load=p[3],
                      regularization_param=p[4], sil_thresh=p[5], constraint_option=int(p[6]), alpha2=p[7],
                      max_wa_l2norm=p[8], min_frames=int(p[9]))
    Y = eng.rls_process(torch.from_numpy(_to_engine_layout(X, K)).to(dev), st).cpu().numpy()[0]       # [K][T]
    ref = G[tag + "_Y"]
    scale = np.max(np.abs(ref))
    # stated tolerance 1e-4 relative (recurrences); the snapshots are rounded to complex64 on the way in
    assert np.max(np.abs(Y[::5].T - ref)) <= 1e-4 * scale
    Pd, wd = st.P.cpu().numpy()[0], st.w.cpu().numpy()[0]
    gw, gP = G[tag + "_waH"], G[tag + "_Pz"]
    for i, k in enumerate(range(0, K, 8)):
        B = orc.blocking_matrix(vs[k], Nc)
        Pz, waH = eng.rls_state_to_reference(1, Pd[k], wd[k], B)
        assert np.max(np.abs(waH - gw[k])) <= 2e-4 * np.max(np.abs(gw))
        assert np.max(np.abs(Pz - gP[i])) <= 2e-3 * np.max(np.abs(gP[i]))
    ss = st.stream_state.cpu().numpy()[0]
    g = G[tag + "_scal"]
    assert ss[2] == T and ss[3] == g[2] and abs(ss[0] - g[0]) <= 1e-5 * g[0]


@pytest.mark.parametrize("N,M,T,S,kw", [
    (4, 64, 90, 2, dict(min_frames=4)),
    (8, 128, 80, 2, dict(min_frames=0, gamma=0.3, alpha2=1e-4, max_wa_l2norm=5e-4, init_diagonal_load=1e3)),
    (7, 64, 60, 3, dict(min_frames=10, constraint_option=1, alpha2=1e-5, regularization_param=0.0)),
    (16, 64, 50, 1, dict(min_frames=2, constraint_option=2, max_wa_l2norm=1e-4)),
    (33, 32, 40, 1, dict(min_frames=2)),
    (8, 32, 2500, 1, dict(min_frames=64)),                       # long run: the per-tile re-projection must not drift
    (64, 32, 100, 1, dict(min_frames=2, gamma=0.2, constraint_option=2, max_wa_l2norm=0.05)),
    # round 3: more than 64 channels (precision matrix packed in LDS) and more than one constraint
    (100, 16, 70, 1, dict(min_frames=2)),
    (128, 8, 40, 1, dict(min_frames=0, gamma=0.2, constraint_option=2, max_wa_l2norm=0.05)),
    (8, 32, 90, 2, dict(min_frames=4, Nc=2)),
    (9, 32, 80, 1, dict(min_frames=0, Nc=3, gamma=0.3, alpha2=1e-4, max_wa_l2norm=5e-4, init_diagonal_load=1e3)),
    (100, 8, 50, 1, dict(min_frames=2, Nc=2)),
    (8, 16, 1500, 1, dict(min_frames=64, Nc=2)),                 # long run with two blocked directions
    # round 4: more than 128 channels (the precision matrix stays in global memory, full [N][N], both triangles updated)
    (129, 8, 30, 1, dict(min_frames=2)),
    (200, 8, 40, 1, dict(min_frames=0, gamma=0.2, constraint_option=2, max_wa_l2norm=0.05)),
    (256, 4, 36, 2, dict(min_frames=2, Nc=2)),
])
def test_rls_py_matches_oracle_synthetic(orc, dev, N, M, T, S, kw):
    """mode 1 (pybeamformer) at other array sizes, two consecutive blocks continuing the recursion."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    kw = dict(kw)
    Nc = kw.pop("Nc", 1)
    rng = np.random.default_rng(N * 1000 + M)
    K = M // 2 + 1
    delays = la_delays(ula_positions(N), -1.306379)
    Xe = _random_frames(rng, S, T, N, M)
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
    st = eng.RLSState(1, S, M, N, torch.from_numpy(vs).to(dev), Nc=Nc, **kw)
    Xd = torch.from_numpy(Xe).to(dev)
    T1 = T // 2 + 3
    Y = torch.cat([eng.rls_process(Xd[..., :T1].contiguous(), st), eng.rls_process(Xd[..., T1:].contiguous(), st)],
                  dim=-1).cpu().numpy()
    Pd, wd = st.P.cpu().numpy(), st.w.cpu().numpy()
    for s in range(S):
        o = orc.RLSPy(M, N, Nc, **kw)
        o.calc_beamformer_weights(16000, delays)
        ref = o.run(_full(Xe[s], M))
        scale = np.max(np.abs(ref))
        # stated tolerance: recurrences <= 1e-4 relative; float64 state, complex64 in/out
        err = np.abs(Y[s].T - ref[:, :K])
        assert np.max(err) <= 1e-4 * scale, (np.max(err) / scale, np.unravel_index(np.argmax(err), err.shape))
        for k in (0, 1, K // 2, K - 1):
            B = orc.blocking_matrix(vs[k], Nc)
            Pz, waH = eng.rls_state_to_reference(1, Pd[s, k], wd[s, k], B)
            assert np.max(np.abs(waH - o.waH[k])) <= 1e-4 * max(np.max(np.abs(o.waH)), 1e-30)
            assert np.max(np.abs(Pz - o.Pz[k])) <= 1e-3 * np.max(np.abs(o.Pz[k]))
        assert st.stream_state.cpu().numpy()[s, 2] == T


@pytest.mark.parametrize("N,M,T,S,opts", [
    (4, 64, 80, 2, dict(mu=0.9, sigma2=0.0)),
    (8, 128, 70, 1, dict(mu=0.95, sigma2=0.01)),
    (8, 64, 60, 2, dict(mu=0.9, sigma2=0.0, qc=(0.05, 1))),
    (6, 64, 60, 1, dict(mu=0.9, sigma2=0.001, qc=(1e-3, 2), normalize=True)),
    (16, 64, 40, 1, dict(mu=0.98, sigma2=0.0)),
    (40, 32, 30, 1, dict(mu=0.9, sigma2=0.0)),
    # round 3: more than 64 channels, more than one constraint (the blocking matrix of calc_gsc_weights_n keeps N - NC columns)
    (100, 16, 30, 1, dict(mu=0.9, sigma2=0.0)),
    (8, 64, 60, 2, dict(mu=0.9, sigma2=0.01, Nc=2)),
    (9, 32, 50, 1, dict(mu=0.95, sigma2=0.0, Nc=3, qc=(0.05, 1))),
    (100, 8, 24, 1, dict(mu=0.9, sigma2=0.0, Nc=2, normalize=True)),
    (160, 8, 24, 1, dict(mu=0.9, sigma2=0.0)),                   # round 4: N > 128
    (256, 4, 20, 1, dict(mu=0.95, sigma2=0.01, Nc=2, qc=(0.05, 1))),
])
def test_rls_cc_matches_oracle_synthetic(orc, dev, N, M, T, S, opts):
    """mode 0 (C++ SubbandGSCRLS, beamformer.cc:1514-1645)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N * 77 + M)
    K = M // 2 + 1
    delays = la_delays(ula_positions(N), 0.6)
    # unit-scale snapshots: with the reference's default Pz_0 = 100 I (init_precision_matrix(0.01)) int16-scale data
    # makes the first updates cancel ~10 digits in the reference itself (P: 1e2 -> 1e-8), which no restatement survives
    Xe = _random_frames(rng, S, T, N, M, scale=0.5)
    Nc = opts.get("Nc", 1)
    o0 = orc.RLSCc(M, N, delays, 16000.0, mu=opts["mu"], sigma2=opts["sigma2"], Nc=Nc)
    kw = dict(mu=o0.mu, diagonal_weight=o0.diag_w, normalize_weight=bool(opts.get("normalize", False)))
    if "qc" in opts:
        kw.update(alpha=float(np.float32(opts["qc"][0])), qctype=opts["qc"][1])
    st = eng.RLSState(0, S, M, N, torch.from_numpy(np.ascontiguousarray(o0.wq[:K])).to(dev), Nc=Nc, **kw)
    st.init_precision_matrix(float(np.float32(1) / np.float32(0.01)))
    Xd = torch.from_numpy(Xe).to(dev)
    T1 = T // 3
    Y = torch.cat([eng.rls_process(Xd[..., :T1].contiguous(), st), eng.rls_process(Xd[..., T1:].contiguous(), st)],
                  dim=-1).cpu().numpy()
    Pd, wd = st.P.cpu().numpy(), st.w.cpu().numpy()
    for s in range(S):
        o = orc.RLSCc(M, N, delays, 16000.0, mu=opts["mu"], sigma2=opts["sigma2"], Nc=Nc)
        o.init_precision_matrix(0.01)
        o.normalize = int(bool(opts.get("normalize", False)))
        if "qc" in opts:
            o.set_quadratic_constraint(*opts["qc"])
        ref = o.run(_full(Xe[s], M))
        scale = np.max(np.abs(ref))
        # stated tolerance: recurrences <= 1e-4 relative; float64 state, complex64 in/out
        assert np.max(np.abs(Y[s].T - ref[:, :K])) <= 1e-4 * scale
        for k in (1, K // 2, K - 1):
            assert np.max(np.abs(wd[s, k] - o.wl[k])) <= 1e-4 * max(np.max(np.abs(o.wl)), 1e-30)
            Pz, wa = eng.rls_state_to_reference(0, Pd[s, k], wd[s, k], o.B[k])
            assert np.max(np.abs(wa - o.wa[k])) <= 1e-4 * max(np.max(np.abs(o.wa)), 1e-30)
            assert np.max(np.abs(Pz - o.Pz[k])) <= 1e-3 * np.max(np.abs(o.Pz[k]))


def test_rls_hold_and_errors(dev):
    """update flag off == fixed GSC; unsupported sizes fail loudly"""
    import torch
    from distant_speech_recognition_amd import engine as eng, _lib
    rng = np.random.default_rng(5)
    N, M, T = 8, 64, 20
    K = M // 2 + 1
    Xe = _random_frames(rng, 1, T, N, M)
    v = (rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))) / N
    vd = torch.from_numpy(v).to(dev)
    st = eng.RLSState(0, 1, M, N, vd, update=False)
    st.init_precision_matrix(100.0)
    Y = eng.rls_process(torch.from_numpy(Xe).to(dev), st).cpu().numpy()[0]
    ref = np.einsum("kn,knt->kt", np.conj(v), Xe[0].astype(np.complex128))
    assert np.max(np.abs(Y - ref)) <= 2e-6 * np.max(np.abs(ref))
    st257 = eng.RLSState(0, 1, 8, 257, torch.zeros((5, 257), dtype=torch.complex128, device=dev))
    with pytest.raises(_lib.BtkError):                           # beyond two lanes per matrix row in a 512-thread workgroup (N <= 256)
        eng.rls_process(torch.zeros((1, 5, 257, 4), dtype=torch.complex64, device=dev), st257)
