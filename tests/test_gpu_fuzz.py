"""GPU: seeded random sweep over filter-bank geometries, decimations, delay-compensation types, lengths that land on and
next to every tile boundary of the tuned kernels (16 / 32 / 128 / 256 frames), channel and stream counts -- analysis,
fused analysis+apply, synthesis and the chunked launches against the float64 oracle."""
import numpy as np
import pytest

from tests.util import design_prototype

pytestmark = pytest.mark.gpu

_RNG = np.random.default_rng(20260927)
_CASES = []
for M in (64, 128, 256, 512, 1024, 2048):
    for _ in range(10 if M <= 512 else 6):
        m = int(_RNG.choice([2, 3, 4])) if M < 256 else 4 if _RNG.random() < 0.8 else int(_RNG.choice([2, 3]))
        r = int(_RNG.choice([0, 1, 2]))
        dct = int(_RNG.choice([0, 1, 2]))
        D = M >> r
        # frame counts on / next to tile boundaries
        T = int(_RNG.choice([1, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 130, 255, 256, 257]))
        if M >= 1024:
            T = min(T, 65)
        L = max(0, T * D + int(_RNG.integers(-D + 1, D)))
        _CASES.append((M, m, r, dct, L, int(_RNG.integers(1, 4)), int(_RNG.integers(1, 6))))


@pytest.mark.parametrize("M,m,r,dct,L,S,N", _CASES)
def test_filterbank_random_geometry(orc, dev, M, m, r, dct, L, S, N):
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(M * 131 + L)
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    K, D = M // 2 + 1, M >> r
    pcm = np.rint(rng.normal(0, 2000, (S, N, L))).astype(np.float32)
    afb = eng.FilterBank(h, M, m, r, dct)
    sfb = eng.FilterBank(g, M, m, r, dct, synthesis=True)
    T = afb.num_frames(L)
    assert T == orc.analysis_num_frames(L, M, m, r, dct)
    if T == 0:
        return
    p = torch.from_numpy(pcm).to(dev)
    X = afb.analysis(p)
    assert X.shape == (S, K, N, T)
    Xh = X.cpu().numpy()
    refX = np.zeros((S, N, T, M), np.complex128)
    for s in range(S):
        for c in range(N):
            refX[s, c] = orc.analysis(h, M, m, r, dct, pcm[s, c])
    scale = max(np.max(np.abs(refX)), 1.0)
    assert np.max(np.abs(np.transpose(Xh, (0, 2, 3, 1)) - refX[..., :K])) <= 1e-5 * scale
    # chunked analysis == whole
    c0 = max(1, T // 3)
    parts = [afb.analysis(p, t0=a, tcount=min(c0, T - a)) for a in range(0, T, c0)]
    assert torch.equal(torch.cat(parts, dim=-1), X)
    # fused analysis + apply == staged (and the oracle's frame-by-frame beamformer)
    W = ((rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))) / N).astype(np.complex64)
    Wd = torch.from_numpy(W).to(dev)
    Yf, Ys = afb.analysis_beamform(p, Wd), eng.bf_apply(Wd, X)
    ys = float(Ys.abs().max()) + 1e-30
    assert float((Yf - Ys).abs().max()) <= 2e-6 * np.sqrt(N) * ys + 1e-6 * ys
    Yref = np.einsum("kn,sntk->skt", np.conj(W.astype(np.complex128)), refX[..., :K])
    assert np.max(np.abs(Ys.cpu().numpy() - Yref)) <= 2e-5 * max(np.max(np.abs(Yref)), 1.0)
    # synthesis of the beamformed frames
    nb = sfb.num_blocks(T)
    if nb > 0:
        out = sfb.synthesize(Ys).cpu().numpy()
        Yh = Ys.cpu().numpy().astype(np.complex128)
        for s in range(S):
            full = np.zeros((T, M), np.complex128)
            full[:, :K] = Yh[s].T
            full[:, K:] = np.conj(Yh[s].T[:, M // 2 - 1:0:-1])
            ref = orc.synthesis(g, M, m, r, dct, full)
            assert out[s].shape == ref.shape
            assert np.max(np.abs(out[s] - ref)) <= 2e-6 * np.sqrt(M) * max(np.max(np.abs(ref)), 1.0)


_BIN_CASES = []
for _ in range(24):
    N = int(_RNG.choice([2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 32, 33, 63, 64, 65, 70]))
    M = int(_RNG.choice([16, 32, 64]))
    T = int(_RNG.choice([1, 2, 15, 16, 17, 63, 64, 65, 100, 255, 256, 257, 300]))
    _BIN_CASES.append((N, M, T, int(_RNG.integers(1, 4))))


@pytest.mark.parametrize("N,M,T,S", _BIN_CASES)
def test_per_bin_kernels_random_shapes(orc, dev, N, M, T, S):
    """apply, apply+Zelinski statistics, covariance (MFMA and VALU), frame energy at random channel / frame counts"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N * 1009 + M * 31 + T)
    K = M // 2 + 1
    X = ((rng.normal(size=(S, K, N, T)) + 1j * rng.normal(size=(S, K, N, T))) * 1000).astype(np.complex64)
    X[:, 0] = X[:, 0].real
    X[:, K - 1] = X[:, K - 1].real
    W = ((rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))) / N).astype(np.complex64)
    Xd, Wd = torch.from_numpy(X).to(dev), torch.from_numpy(W).to(dev)
    X128, W128 = X.astype(np.complex128), W.astype(np.complex128)
    Yref = np.einsum("kn,sknt->skt", np.conj(W128), X128)
    Y = eng.bf_apply(Wd, Xd).cpu().numpy()
    assert np.max(np.abs(Y - Yref)) <= 4e-6 * np.sqrt(N) * np.max(np.abs(Yref))
    # energy of channel 0 over all M bins (mirror bins counted)
    e = eng.frame_energy(Xd, M).cpu().numpy()
    x0 = X128[:, :, 0, :]
    eref = (np.sum(np.abs(x0) ** 2, axis=1) * 2 - np.abs(x0[:, 0]) ** 2 - np.abs(x0[:, K - 1]) ** 2) / M
    assert np.max(np.abs(e - eref)) <= 1e-5 * np.max(eref)
    # covariance with a frame gate: both kernels against numpy
    fw = (rng.random((S, T)) > 0.3).astype(np.float32)
    Rref = np.einsum("sknt,st,skmt->sknm", X128, fw.astype(np.float64), np.conj(X128))
    for mfma in (True, False):
        R = eng.cov_accumulate(Xd, frame_weights=torch.from_numpy(fw).to(dev), use_mfma=mfma).cpu().numpy()
        for s in range(S):
            for k in range(K):
                assert np.linalg.norm(R[s, k] - Rref[s, k]) <= 2e-5 * np.linalg.norm(Rref[s, k]) + 1e-3
    # Zelinski through the oracle for one stream (N >= 2)
    if N >= 2 and T <= 100:
        full = np.zeros((T, N, M), np.complex128)
        full[:, :, :K] = np.transpose(X128[0], (2, 1, 0))
        full[:, :, K:] = np.conj(full[:, :, M // 2 - 1:0:-1])
        Wf = np.zeros((M, N), np.complex128)
        Wf[:K] = W128
        Wf[K:] = np.conj(W128[M // 2 - 1:0:-1])
        st = eng.ZelinskiState(S, K, dev)
        Yz = eng.bf_apply_zelinski(Wd, Wd, Xd, st, alpha=0.6, type_=2).cpu().numpy()[0]
        refz, _ = orc.zelinski_frames(full, orc.gsc_frames(full, Wf, None), Wf, 0.6, 2)
        assert np.max(np.abs(Yz.T - refz[:, :K])) <= 1e-4 * np.max(np.abs(refz))
