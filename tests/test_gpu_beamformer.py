"""GPU parity: fixed-weight beamformer apply (SubbandDS / SubbandGSC / SubbandMVDR ::next)."""
import numpy as np
import pytest

from tests.util import design_prototype, synthetic_pcm, ula_positions, la_delays

pytestmark = pytest.mark.gpu


def _snapshots(rng, S, K, N, T):
    return ((rng.normal(size=(S, K, N, T)) + 1j * rng.normal(size=(S, K, N, T))) * 3000.0).astype(np.complex64)


def _to_orc(Xs, M):
    """[K][N][T] (bins 0..M/2) -> [T][N][M] with mirror bins, as the analysis banks would deliver."""
    K, N, T = Xs.shape
    full = np.zeros((T, N, M), np.complex128)
    full[:, :, :K] = np.transpose(Xs, (2, 1, 0))
    full[:, :, K:] = np.conj(full[:, :, M // 2 - 1:0:-1])
    return full


@pytest.mark.parametrize("N,M,T", [(2, 256, 33), (8, 512, 100), (64, 512, 64), (5, 64, 7), (16, 128, 257)])
def test_gsc_apply_matches_oracle(orc, dev, N, M, T):
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N * 7 + M)
    K, S = M // 2 + 1, 2
    delays = la_delays(ula_positions(N), -1.306379)
    wq, B, _ = orc.gsc_weights(M, N, 16000, delays)
    wa = (rng.normal(size=(M, N - 1)) + 1j * rng.normal(size=(M, N - 1))) * 0.05
    wl = np.zeros((M, N), np.complex128)
    for k in range(K):
        wl[k] = orc.sidelobe_canceller(B[k], wa[k])
    X = _snapshots(rng, S, K, N, T)
    for normalize in (False, True):
        w = eng.weights_gsc_effective(wq, wl, M, normalize)
        Y = eng.bf_apply(torch.from_numpy(w).to(dev), torch.from_numpy(X).to(dev)).cpu().numpy()
        for s in range(S):
            ref = orc.gsc_frames(_to_orc(X[s], M), wq, wl, normalize)          # [T][M]
            got = Y[s].T
            # tolerance: <= 2e-6 * sqrt(N) relative, complex64 weights+data vs float64 oracle
            tol = 2e-6 * np.sqrt(N) * np.max(np.abs(ref)) + 1e-6 * np.max(np.abs(ref))
            assert np.max(np.abs(got - ref[:, :K])) <= tol


def test_ds_apply_and_per_stream_weights(orc, dev):
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(1)
    N, M, T, S = 4, 256, 50, 3
    K = M // 2 + 1
    X = _snapshots(rng, S, K, N, T)
    ws = []
    for s in range(S):
        wq = orc.calc_mainlobe(M, N, 16000, la_delays(ula_positions(N), 0.3 * (s + 1)))
        ws.append((wq, eng.weights_gsc_effective(wq, None, M)))
    W = torch.from_numpy(np.stack([w for _, w in ws])).to(dev)
    Y = eng.bf_apply(W, torch.from_numpy(X).to(dev)).cpu().numpy()
    for s in range(S):
        ref = orc.gsc_frames(_to_orc(X[s], M), ws[s][0], None)               # SubbandDS::next
        assert np.max(np.abs(Y[s].T - ref[:, :K])) <= 4e-6 * np.max(np.abs(ref))


def test_linearity_full_size(dev):
    """Size-independent property at the headline size (64 mics, 512 bins): apply is linear in X and
    the all-ones/N weight returns the channel mean."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    g = torch.Generator(device="cpu").manual_seed(0)
    S, K, N, T = 2, 257, 64, 1024
    X1 = torch.randn((S, K, N, T), generator=g, dtype=torch.float32).to(dev).to(torch.complex64)
    X2 = torch.randn((S, K, N, T), generator=g, dtype=torch.float32).to(dev).to(torch.complex64) * 1j
    W = torch.full((K, N), 1.0 / N, dtype=torch.complex64, device=dev)
    Y1, Y2, Y12 = eng.bf_apply(W, X1), eng.bf_apply(W, X2), eng.bf_apply(W, X1 + X2)
    assert torch.allclose(Y12, Y1 + Y2, atol=1e-4)
    assert torch.allclose(Y1, X1.mean(dim=2), atol=1e-5)


def test_full_chain_vs_oracle_c2(orc, dev):
    """BASELINE config C2 shape (8 mics, 512 bins, one stream): analysis -> SubbandGSC -> synthesis
    against the frame-by-frame pull graph of the oracle (src/beamformerDS.cc:184-191 order)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    M, m, r, N, dct = 512, 4, 1, 8, 2
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    L = 60 * 256
    pcm, delays = synthetic_pcm(1, N, L, seed=3)
    wq, B, _ = orc.gsc_weights(M, N, 16000, delays)
    rng = np.random.default_rng(0)
    wl = np.zeros((M, N), np.complex128)
    for k in range(M // 2 + 1):
        wl[k] = orc.sidelobe_canceller(B[k], (rng.normal(size=N - 1) + 1j * rng.normal(size=N - 1)) * 0.02)
    ref, nbf = orc.pipeline_gsc(h, g, M, m, r, dct, pcm[0], wq, wl)
    afb = eng.FilterBank(h, M, m, r, dct)
    sfb = eng.FilterBank(g, M, m, r, dct, synthesis=True)
    X = afb.analysis(torch.from_numpy(pcm).to(dev))
    assert X.shape[-1] == nbf
    Y = eng.bf_apply(torch.from_numpy(eng.weights_gsc_effective(wq, wl, M)).to(dev), X)
    out = sfb.synthesize(Y).cpu().numpy()[0]
    assert out.shape == ref.shape
    # synthesis PCM tolerance: <= 0.5 LSB at int16 scale
    assert np.max(np.abs(out - ref)) < 0.5


@pytest.mark.parametrize("N,r,S,L", [(8, 1, 2, 40 * 256 + 31), (64, 1, 1, 20 * 256), (5, 0, 2, 9 * 512 + 100), (3, 2, 1, 70 * 128)])
def test_fused_analysis_beamform_equals_staged(dev, N, r, S, L):
    """btk_fb_analysis_bf (snapshots never written to HBM) == btk_fb_analysis + btk_bf_apply, incl. ragged tails,
    per-stream weights and every supported decimation."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    M, m = 512, 4
    rng = np.random.default_rng(N + r)
    fb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    pcm, _ = synthetic_pcm(S, N, L, seed=N)
    p = torch.from_numpy(pcm).to(dev)
    K = M // 2 + 1
    Wn = ((rng.normal(size=(S, K, N)) + 1j * rng.normal(size=(S, K, N))) / N).astype(np.complex64)
    for W in (torch.from_numpy(Wn).to(dev), torch.from_numpy(Wn[0]).to(dev)):
        X = fb.analysis(p)
        ref = eng.bf_apply(W, X)
        got = fb.analysis_beamform(p, W)
        assert got.shape == ref.shape
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-6 * np.sqrt(N) * scale + 1e-6 * scale


@pytest.mark.parametrize("N,r,S,T", [(6, 1, 3, 300), (4, 2, 2, 517), (64, 1, 2, 160)])
def test_fused_interior_tiles_lds_dma_path(orc, dev, N, r, S, T):
    """Aligned recordings long enough that most 16-frame tiles lie inside them: those take the LDS-DMA staging path of
    the fused kernel (edge tiles go through registers).  Fused == staged on every frame, and both == the oracle on a
    sample of frames from the middle of the recording."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    M, m = 512, 4
    D, K = M >> r, M // 2 + 1
    h = design_prototype(M, m)
    fb = eng.FilterBank(h, M, m, r, 2)
    L = (T - fb.processing_delay + fb.lookahead) * D
    assert L % 4 == 0 and fb.num_frames(L) == T
    pcm, _ = synthetic_pcm(S, N, L, seed=17 + N)
    p = torch.from_numpy(pcm).to(dev)
    rng = np.random.default_rng(N * 7 + r)
    Wn = ((rng.normal(size=(S, K, N)) + 1j * rng.normal(size=(S, K, N))) / N).astype(np.complex64)
    for W in (torch.from_numpy(Wn).to(dev), torch.from_numpy(Wn[1]).to(dev)):
        ref = eng.bf_apply(W, fb.analysis(p))
        got = fb.analysis_beamform(p, W)
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-6 * np.sqrt(N) * scale + 1e-6 * scale
    # oracle on whole 16-frame tiles of BOTH streams: the first tile (its span starts before the recording: edge path), interior
    # tiles (LDS-DMA / direct-window path), the last, ragged tile (edge path), per-stream weights for stream 0 and shared ones for 1
    gotp = fb.analysis_beamform(p, torch.from_numpy(Wn).to(dev)).cpu().numpy()
    gots = got.cpu().numpy()
    tiles = sorted({0, 16 * (T // 32), 96 if T > 112 else 16, 16 * ((T - 1) // 16)})
    for s_, cases in ((0, ((Wn[0], gotp),)), (1, ((Wn[1], gots), (Wn[1], gotp)))):
        Xo = np.stack([orc.analysis(h, M, m, r, 2, pcm[s_, n]) for n in range(N)], axis=1)      # [T][N][M]
        assert Xo.shape[0] == T
        for Wo, G in cases:
            for t0 in tiles:
                t1 = min(t0 + 16, T)
                Yo = np.einsum("kn,tnk->kt", np.conj(Wo.astype(np.complex128)), Xo[t0:t1, :, :K])
                assert np.max(np.abs(G[s_, :, t0:t1] - Yo)) <= 4e-6 * np.sqrt(N) * np.max(np.abs(Yo)), (s_, t0)


def test_fused_chain_with_row_padded_output(dev):
    """Y handed over as a [..., :T] view of a buffer with padded rows (engine.padded_rows: what analysis_beamform
    allocates itself when a contiguous row would be a multiple of 4 KiB): the fused kernel and the synthesis bank take
    the row stride through the C-ABI's T_stride and give the same bits as with contiguous rows."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    M, m, r, N, S, T = 512, 4, 1, 5, 2, 512
    fb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    sfb = eng.FilterBank(design_prototype(M, m, "g"), M, m, r, 2, synthesis=True)
    L = (T - fb.processing_delay + fb.lookahead) * (M >> r)
    pcm, _ = synthetic_pcm(S, N, L, seed=3)
    p = torch.from_numpy(pcm).to(dev)
    rng = np.random.default_rng(5)
    K = M // 2 + 1
    W = torch.from_numpy(((rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))) / N).astype(np.complex64)).to(dev)
    Yc = torch.empty((S, K, T), dtype=torch.complex64, device=dev)
    fb.analysis_beamform(p, W, out=Yc)
    Yp = fb.analysis_beamform(p, W)                       # default allocation: T * 8 B = 4 KiB rows -> padded
    assert not Yp.is_contiguous() and Yp.stride(1) > T and Yp.shape == Yc.shape
    assert torch.equal(Yp, Yc)
    assert torch.equal(sfb.synthesize(Yp), sfb.synthesize(Yc))
    with pytest.raises(ValueError):
        sfb.synthesize(Yc.transpose(1, 2))                # rows must be contiguous


@pytest.mark.parametrize("N,r,S,T", [(6, 1, 2, 300), (4, 2, 3, 133), (5, 0, 1, 70), (64, 1, 1, 96)])
def test_fused_default_geometry_m256(orc, dev, N, r, S, T):
    """The reference's default geometry (M = 256, m = 4, unit_test/test_online_beamforming.py:259-262) has its own fused
    analysis -> apply kernel (fb_fast.hip): equal to the staged pair on every frame (aligned and odd-length recordings,
    shared and per-stream weights) and to the oracle on a block of frames."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    M, m = 256, 4
    D, K = M >> r, M // 2 + 1
    h = design_prototype(M, m)
    fb = eng.FilterBank(h, M, m, r, 2)
    for extra in (0, 37):
        L = (T - fb.processing_delay + fb.lookahead) * D + extra
        pcm, _ = synthetic_pcm(S, N, L, seed=29 + N + extra)
        p = torch.from_numpy(pcm).to(dev)
        rng = np.random.default_rng(N * 11 + r)
        Wn = ((rng.normal(size=(S, K, N)) + 1j * rng.normal(size=(S, K, N))) / N).astype(np.complex64)
        for W in (torch.from_numpy(Wn).to(dev), torch.from_numpy(Wn[0]).to(dev)):
            ref = eng.bf_apply(W, fb.analysis(p))
            got = fb.analysis_beamform(p, W)
            assert got.shape == ref.shape
            scale = float(ref.abs().max())
            assert float((got - ref).abs().max()) <= 2e-6 * np.sqrt(N) * scale + 1e-6 * scale
    Xo = np.stack([orc.analysis(h, M, m, r, 2, pcm[0, n]) for n in range(N)], axis=1)      # [T'][N][M]
    t0 = min(32, Xo.shape[0] - 16)
    Yo = np.einsum("kn,tnk->kt", np.conj(Wn[0].astype(np.complex128)), Xo[t0:t0 + 16, :, :K])
    g = got[0, :, t0:t0 + 16].cpu().numpy()
    assert np.max(np.abs(g - Yo)) <= 4e-6 * np.sqrt(N) * np.max(np.abs(Yo))


def test_row_padded_snapshots_give_identical_results(dev):
    """analysis(pad_rows=True) spaces the snapshot rows 48 frames wider when a row is a multiple of 4 KiB (engine.padded_rows);
    bf_apply and nlms_process take such a view (the C-ABI's T_stride) and their outputs share the row stride -- bit for bit
    the contiguous results."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    from tests.util import design_prototype, ula_positions, la_delays
    N, M, S, T = 8, 512, 2, 512                                   # T * 8 B = 4 KiB rows -> padded
    D = M // 2
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    L = (T - afb.processing_delay + afb.lookahead) * D
    g = torch.Generator(device=dev).manual_seed(2)
    pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000.0).round_()
    Xc = afb.analysis(pcm)
    Xp = afb.analysis(pcm, pad_rows=True)
    assert Xc.is_contiguous() and not Xp.is_contiguous() and Xp.stride(-2) == T + 48
    assert torch.equal(Xc, Xp)
    delays = la_delays(ula_positions(N), 0.4)
    wq = eng.weights_mainlobe(M, N, 16000.0, delays)
    W = torch.from_numpy(eng.weights_gsc_effective(wq, np.zeros_like(wq), M)).to(dev)
    Yc, Yp = eng.bf_apply(W, Xc), eng.bf_apply(W, Xp)
    assert Yp.stride(-2) == T + 48 and torch.equal(Yc, Yp)
    vs = torch.from_numpy(wq[: M // 2 + 1].astype(np.complex64)).to(dev)
    sc, sp = eng.NLMSState(S, M, N, dev), eng.NLMSState(S, M, N, dev)
    Zc, Zp = eng.nlms_process(vs, Xc, sc), eng.nlms_process(vs, Xp, sp)
    assert torch.equal(Zc, Zp) and torch.equal(sc.u, sp.u)
    with pytest.raises(Exception):
        eng.bf_apply(W, Xp, out=torch.empty_like(Yc))             # a contiguous Y cannot share the padded T_stride
    assert torch.equal(eng.cov_accumulate(Xc), eng.cov_accumulate(Xp))
    zc, zp = eng.ZelinskiState(S, M // 2 + 1, dev), eng.ZelinskiState(S, M // 2 + 1, dev)
    assert torch.equal(eng.bf_apply_zelinski(W, W, Xc, zc, alpha=0.7), eng.bf_apply_zelinski(W, W, Xp, zp, alpha=0.7))
    with pytest.raises(Exception):
        eng.frame_energy(Xp, M)                                   # the remaining consumers want contiguous snapshots


@pytest.mark.parametrize("M,N,S,T,extra", [(1024, 6, 2, 100, 0), (1024, 64, 1, 40, 0), (1024, 5, 3, 37, 301), (2048, 4, 2, 70, 0),
                                           (2048, 40, 1, 24, 0), (2048, 7, 1, 19, 1023), (2048, 256, 1, 16, 0)])
def test_fused_large_geometries(orc, dev, M, N, S, T, extra):
    """analysis_bfz_big_kernel (fb_fused_big.hip: M = 1024 / 2048, the three-pass 16 x Q x 16 transform with the beamformer sum on
    the FFT lanes): equal to the staged pair btk_fb_analysis + btk_bf_apply on every frame -- interior tiles (unguarded window
    loads), the edge tiles at both ends, recordings whose length is odd (element-wise guarded loads everywhere), ragged last tiles,
    shared and per-stream weights, few tiles (the channels are split over workgroups and the partial sums added in a second
    kernel) -- and to the oracle on whole tiles."""
    import torch
    from distant_speech_recognition_amd import engine as eng, _lib
    m, r = 4, 1
    D, K = M >> r, M // 2 + 1
    h = design_prototype(M, m)
    fb = eng.FilterBank(h, M, m, r, 2)
    L = (T - fb.processing_delay + fb.lookahead) * D + extra
    pcm, _ = synthetic_pcm(S, N, L, seed=M + N)
    p = torch.from_numpy(pcm).to(dev)
    rng = np.random.default_rng(M + 3 * N)
    Wn = ((rng.normal(size=(S, K, N)) + 1j * rng.normal(size=(S, K, N))) / N).astype(np.complex64)
    nb = _lib.lib().btk_fb_analysis_bf_scratch_bytes(fb._h, S, N, 0, fb.num_frames(L))
    assert nb < 8 * S * K * N * fb.num_frames(L) or N <= 8                   # the fused form: weight pairs (+ partial blocks), not the snapshots
    for W in (torch.from_numpy(Wn).to(dev), torch.from_numpy(Wn[0]).to(dev)):
        ref = eng.bf_apply(W, fb.analysis(p))
        got = fb.analysis_beamform(p, W)
        assert got.shape == ref.shape
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= 2e-6 * np.sqrt(N) * scale + 1e-6 * scale
    if N <= 8:
        gots = got.cpu().numpy()
        Tn = gots.shape[-1]
        Xo = np.stack([orc.analysis(h, M, m, r, 2, pcm[0, n]) for n in range(N)], axis=1)      # [T][N][M]
        assert Xo.shape[0] == Tn
        for t0 in sorted({0, 8 * (Tn // 16), 8 * ((Tn - 1) // 8)}):
            t1 = min(t0 + 8, Tn)
            Yo = np.einsum("kn,tnk->kt", np.conj(Wn[0].astype(np.complex128)), Xo[t0:t1, :, :K])
            assert np.max(np.abs(gots[0, :, t0:t1] - Yo)) <= 4e-6 * np.sqrt(N) * np.max(np.abs(Yo)), t0


@pytest.mark.parametrize("seed", range(8))
def test_fused_large_geometries_fuzz(dev, seed):
    """random launches of the M = 1024 / 2048 fused kernel -- channel counts that do not divide by the channel-group split, stream counts,
    recording lengths (odd ones: element-wise guarded window loads), frame sub-ranges [t0, t0 + tcount) as the frame-sharded path issues
    them, shared and per-stream weights -- against the staged pair on the same frames"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(4000 + seed)
    M = int(rng.choice([1024, 2048]))
    N = int(rng.choice([1, 2, 3, 5, 9, 17, 33, 70]))
    S = int(rng.integers(1, 4))
    T = int(rng.integers(1, 90))
    m, r = 4, 1
    D, K = M >> r, M // 2 + 1
    fb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    L = (T - fb.processing_delay + fb.lookahead) * D + int(rng.integers(0, 3)) * int(rng.integers(0, D))
    pcm, _ = synthetic_pcm(S, N, L, seed=seed)
    p = torch.from_numpy(pcm).to(dev)
    Tn = fb.num_frames(L)
    per_stream = bool(rng.integers(0, 2))
    Wn = ((rng.normal(size=(S if per_stream else 1, K, N)) + 1j * rng.normal(size=(S if per_stream else 1, K, N))) / N).astype(np.complex64)
    W = torch.from_numpy(Wn if per_stream else Wn[0]).to(dev)
    ref = eng.bf_apply(W, fb.analysis(p))
    scale = float(ref.abs().max())
    tol = 2e-6 * np.sqrt(N) * scale + 1e-6 * scale
    got = fb.analysis_beamform(p, W)
    assert got.shape == ref.shape == (S, K, Tn)
    assert float((got - ref).abs().max()) <= tol, (M, N, S, T, per_stream)
    for _ in range(3):
        t0 = int(rng.integers(0, Tn))
        tc = int(rng.integers(1, Tn - t0 + 1))
        part = fb.analysis_beamform(p, W, t0=t0, tcount=tc)
        assert part.shape == (S, K, tc)
        assert float((part - ref[:, :, t0:t0 + tc]).abs().max()) <= tol, (M, N, S, T, t0, tc)


def test_fused_large_geometry_is_bit_reproducible(dev):
    """a channel-split launch (one stream, few tiles: eight channel groups per tile at this size) adds its partial sums in a fixed
    order in a second kernel -- no atomics, the same bits every run"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    M, N, T = 2048, 64, 40
    fb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    L = (T - fb.processing_delay + fb.lookahead) * (M // 2)
    pcm, _ = synthetic_pcm(1, N, L, seed=9)
    p = torch.from_numpy(pcm).to(dev)
    rng = np.random.default_rng(2)
    W = torch.from_numpy(((rng.normal(size=(M // 2 + 1, N)) + 1j * rng.normal(size=(M // 2 + 1, N))) / N).astype(np.complex64)).to(dev)
    a = fb.analysis_beamform(p, W).clone()
    for _ in range(3):
        assert torch.equal(fb.analysis_beamform(p, W), a)


_ROWSWAP_CHILD = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from distant_speech_recognition_amd import engine as eng
from tests.util import design_prototype, synthetic_pcm
dev = torch.device("cuda", 0)
for M, N, S, T, extra in ((1024, 12, 2, 100, 0), (2048, 9, 2, 70, 0), (2048, 64, 1, 40, 7)):
    fb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    L = (T - fb.processing_delay + fb.lookahead) * (M // 2) + extra
    pcm, _ = synthetic_pcm(S, N, L, seed=M + N)
    rng = np.random.default_rng(M)
    W = ((rng.normal(size=(M // 2 + 1, N)) + 1j * rng.normal(size=(M // 2 + 1, N))) / N).astype(np.complex64)
    Y = fb.analysis_beamform(torch.from_numpy(pcm).to(dev), torch.from_numpy(W).to(dev))
    print("DIGEST", M, N, hashlib.sha256(torch.view_as_real(Y).contiguous().cpu().numpy().tobytes()).hexdigest())
"""


def test_fused_large_row_swap_form_equals_lds_form_bit_for_bit():
    """analysis_bfz_big_kernel hands pass 1a's result to pass 1b through v_permlane32_swap / v_permlane16_swap (lane row against two bits
    of the register index) instead of an LDS round trip: pure data movement, so the output is the LDS form's bit for bit -- interior and
    edge tiles, odd recording length, channel-split launch.  (The forms are chosen once per process by BTK_FUSED_VAR: two children.)"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for var in ("3", "7"):
        env = dict(os.environ)
        env["BTK_FUSED_VAR"] = var
        r = subprocess.run([sys.executable, "-c", _ROWSWAP_CHILD, root], env=env, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, (r.stdout[-800:], r.stderr[-800:])
        outs.append([l for l in r.stdout.splitlines() if l.startswith("DIGEST")])
    assert len(outs[0]) == 3 and outs[0] == outs[1], outs
