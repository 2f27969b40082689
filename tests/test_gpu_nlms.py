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
