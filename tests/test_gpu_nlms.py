"""GPU parity: adaptive GSC canceller (SubbandGSCLMSBeamformer) vs the oracle and vs the golden
outputs of the reference's own Python arithmetic."""
import numpy as np
import pytest

from tests.util import design_prototype, synthetic_pcm

pytestmark = pytest.mark.gpu


def _frames_from_kinect(orc, proto256, kinect_pcm, T):
    h, _ = proto256
    return np.stack([orc.analysis(h, 256, 4, 1, 2, kinect_pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)


def _to_engine_layout(X, K):
    """oracle frames [T][N][M] -> engine [1][K][N][T] complex64"""
    return np.ascontiguousarray(np.transpose(X[:, :, :K], (2, 1, 0))[None]).astype(np.complex64)


@pytest.mark.parametrize("tag,kw", [("nlms_default", {}), ("nlms_fast", dict(min_frames=16, gamma=0.05, slowdown_after=64))])
def test_nlms_vs_reference_python_golden(orc, dev, proto256, kinect_pcm, pygolden, tag, kw):
    """Real 4-mic Kinect data, confs/gsclms.json parameters: GPU output against what the REFERENCE's
    pybeamformer.py produced (tests/golden/gen_golden_pybeamformer.py)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    G = pygolden
    T, M, N, K = int(G["meta_T"][0]), 256, 4, 129
    X = _frames_from_kinect(orc, proto256, kinect_pcm, T)
    delays = G["delays_kinect"]
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
    st = eng.NLMSState(1, M, N, dev, **kw)
    Y = eng.nlms_process(torch.from_numpy(vs.astype(np.complex64)).to(dev),
                         torch.from_numpy(_to_engine_layout(X, K)).to(dev), st).cpu().numpy()[0]      # [K][T]
    ref = G[tag + "_Y"]                                                                              # [T][every 5th bin]
    scale = np.max(np.abs(ref))
    # stated tolerance: NLMS recurrences <= 1e-4 relative; float32 state vs float64 reference
    assert np.max(np.abs(Y[::5].T - ref)) <= 1e-4 * scale
    # active weights: change of basis back to wa^H
    u = st.u.cpu().numpy()[0].astype(np.complex128)
    for k in (5, 40, 128):
        B = orc.blocking_matrix(vs[k], 1)
        wa = eng.nlms_u_to_wa(u[k], B)
        assert np.max(np.abs(wa - G[tag + "_waH"][k])) <= 2e-4 * max(1.0, np.max(np.abs(G[tag + "_waH"])))
    ss = st.stream_state.cpu().numpy()[0]
    assert ss[2] == T and ss[3] == G[tag + "_energy"][2]
    assert abs(ss[0] - G[tag + "_energy"][0]) <= 1e-5 * G[tag + "_energy"][0]
    assert ss[1] == G[tag + "_energy"][1]
    se = st.sigma2.cpu().numpy()[0]
    assert np.max(np.abs(se - G[tag + "_subband_energy"]) / G[tag + "_subband_energy"]) < 1e-4


@pytest.mark.parametrize("N,M,T,S", [(8, 512, 150, 2), (64, 128, 70, 1), (5, 64, 40, 3), (16, 64, 33, 2), (100, 64, 20, 1)])
def test_nlms_matches_oracle_synthetic(orc, dev, N, M, T, S):
    import torch
    from distant_speech_recognition_amd import engine as eng
    from tests.util import ula_positions, la_delays
    rng = np.random.default_rng(N + M)
    K = M // 2 + 1
    delays = la_delays(ula_positions(N), -1.306379)
    kw = dict(min_frames=8, gamma=0.05, slowdown_after=32, max_wa_l2norm=0.5)     # exercises halving + clamp
    Xs = (rng.normal(size=(S, T, N, M)) + 1j * rng.normal(size=(S, T, N, M))) * 2000.0
    Xs[..., K:] = np.conj(Xs[..., M // 2 - 1:0:-1])
    Xs[..., 0] = Xs[..., 0].real
    Xs[..., M // 2] = Xs[..., M // 2].real
    Xe = np.concatenate([_to_engine_layout(Xs[s], K) for s in range(S)])
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
    st = eng.NLMSState(S, M, N, dev, **kw)
    Xd = torch.from_numpy(Xe).to(dev)
    vd = torch.from_numpy(vs.astype(np.complex64)).to(dev)
    # two consecutive blocks must continue the recursion exactly like one pass
    T1 = T // 2 + 1
    Y = torch.cat([eng.nlms_process(vd, Xd[..., :T1].contiguous(), st),
                   eng.nlms_process(vd, Xd[..., T1:].contiguous(), st)], dim=-1).cpu().numpy()
    for s in range(S):
        o = orc.NLMS(M, N, **kw)
        o.calc_beamformer_weights(16000, delays)
        ref = o.run(_full(Xe[s], M))
        scale = np.max(np.abs(ref))
        assert np.max(np.abs(Y[s].T - ref[:, :K])) <= 2e-4 * scale
        assert np.abs(o.wa()).max() > 1e-4


def _full(Xe, M):
    """engine [K][N][T] complex64 -> oracle [T][N][M] complex128 with mirror bins (what the GPU saw)"""
    K, N, T = Xe.shape
    full = np.zeros((T, N, M), np.complex128)
    full[:, :, :K] = np.transpose(Xe.astype(np.complex128), (2, 1, 0))
    full[:, :, K:] = np.conj(full[:, :, M // 2 - 1:0:-1])
    return full


@pytest.mark.parametrize("Nc", [2, 3])
def test_nlms_nc_constraints_vs_reference_python_golden(orc, dev, proto256, kinect_pcm, Nc):
    """SubbandGSCLMSBeamformer(..., Nc) with Nc > 1 (lib/pybeamformer.py:588-607, 742): the blocking matrix keeps the first
    N - Nc Gram-Schmidt columns, the canceller's projector loses Nc - 1 more directions.  GPU output and exported wa^H
    against what the REFERENCE's own Python produced (tests/golden/gen_golden_pybeamformer_nc.py)."""
    import os
    import torch
    from distant_speech_recognition_amd import engine as eng
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pybeamformer_nc_golden.npz"))
    T, M, N, K = int(G["meta_T"][0]), 256, 4, 129
    X = _frames_from_kinect(orc, proto256, kinect_pcm, T)
    delays = G["delays"]
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
    kw = dict(min_frames=16, gamma=0.05, slowdown_after=64)
    st = eng.NLMSState(1, M, N, dev, Nc=Nc, **kw)
    st.set_constraints(vs)
    assert st.cextra.shape == (K, Nc - 1, N)
    Y = eng.nlms_process(torch.from_numpy(vs.astype(np.complex64)).to(dev),
                         torch.from_numpy(_to_engine_layout(X, K)).to(dev), st).cpu().numpy()[0]
    tag = "nlms_nc%d" % Nc
    ref = G[tag + "_Y"]
    assert np.max(np.abs(Y[::5].T - ref)) <= 1e-4 * np.max(np.abs(ref))
    u = st.u.cpu().numpy()[0].astype(np.complex128)
    B40 = eng.weights_blocking_matrix(vs[40], Nc)
    assert np.max(np.abs(B40 - G[tag + "_blockmat_k40"])) < 1e-12                     # the host designer == the reference's
    for k in (5, 40, 128):
        wa = eng.nlms_u_to_wa(u[k], eng.weights_blocking_matrix(vs[k], Nc))
        assert wa.shape == (N - Nc,)
        assert np.max(np.abs(wa - G[tag + "_waH"][k])) <= 2e-4 * max(1.0, np.max(np.abs(G[tag + "_waH"])))
    se = st.sigma2.cpu().numpy()[0]
    assert np.max(np.abs(se - G[tag + "_subband_energy"]) / G[tag + "_subband_energy"]) < 1e-4
    # the constraint directions: orthonormal, orthogonal to vs and to span(conj(B))
    cx = st.cextra.cpu().numpy().astype(np.complex128)
    for k in (1, 40, 128):
        B = eng.weights_blocking_matrix(vs[k], Nc)
        assert np.max(np.abs(cx[k].conj() @ cx[k].T - np.eye(Nc - 1))) < 1e-6
        assert np.max(np.abs(cx[k].conj() @ vs[k])) < 1e-6 and np.max(np.abs(cx[k].conj() @ B.conj())) < 1e-6


@pytest.mark.parametrize("N,Nc", [(8, 2), (64, 2), (16, 4), (100, 3), (16, 6), (64, 8), (12, 5), (130, 7)])
def test_nlms_nc_matches_oracle_synthetic(orc, dev, N, Nc):
    import torch
    from distant_speech_recognition_amd import engine as eng
    from tests.util import ula_positions, la_delays
    M, T, S = 64, 40, 2
    rng = np.random.default_rng(N * 10 + Nc)
    K = M // 2 + 1
    delays = la_delays(ula_positions(N), -1.306379)
    kw = dict(min_frames=8, gamma=0.05, slowdown_after=32, max_wa_l2norm=0.5)
    Xs = (rng.normal(size=(S, T, N, M)) + 1j * rng.normal(size=(S, T, N, M))) * 2000.0
    Xs[..., K:] = np.conj(Xs[..., M // 2 - 1:0:-1])
    Xs[..., 0] = Xs[..., 0].real
    Xs[..., M // 2] = Xs[..., M // 2].real
    Xe = np.concatenate([_to_engine_layout(Xs[s], K) for s in range(S)])
    vs = np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)])
    st = eng.NLMSState(S, M, N, dev, Nc=Nc, **kw)
    st.set_constraints(vs)
    Y = eng.nlms_process(torch.from_numpy(vs.astype(np.complex64)).to(dev), torch.from_numpy(Xe).to(dev), st).cpu().numpy()
    for s in range(S):
        o = orc.NLMS(M, N, Nc=Nc, **kw)
        o.calc_beamformer_weights(16000, delays)
        ref = o.run(_full(Xe[s], M))
        assert np.max(np.abs(Y[s].T - ref[:, :K])) <= 2e-4 * np.max(np.abs(ref))
