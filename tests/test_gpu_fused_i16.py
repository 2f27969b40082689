"""GPU: the int16-PCM entry of the fused analysis -> apply kernels (btk_fb_analysis_bf_i16).  SampleFeature turns a WAV's 16-bit
samples into un-normalised floats (feature/feature.cc:265-269): the float kernel and the int16 kernel see the same values, and
the widening (v_cvt_f32_i32 on sign-extended halves) is exact -- the outputs must be BIT-identical, on interior tiles, edge
tiles, ragged launches, every geometry with an int16 kernel, shared and per-stream weights; and equal the oracle like the float
path does."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(M, N, S, T, r=1, seed=0, dev=None):
    from distant_speech_recognition_amd import engine as eng
    from tests.util import design_prototype, synthetic_pcm
    m, dct = 4, 2
    D = M >> r
    h = design_prototype(M, m)
    afb = eng.FilterBank(h, M, m, r, dct)
    pcm, delays = synthetic_pcm(S, N, T * D + 37, seed=seed)
    assert np.array_equal(pcm, np.rint(pcm)) and np.max(np.abs(pcm)) <= 32767
    rng = np.random.default_rng(seed + 1)
    W = ((rng.normal(size=(S, afb.K, N)) + 1j * rng.normal(size=(S, afb.K, N))) / N).astype(np.complex64)
    return afb, pcm, torch.from_numpy(W).to(dev), delays


def _bits(t):
    return t.contiguous().view(torch.float32).view(torch.int32)


@pytest.mark.parametrize("M,N,S,T,r", [(512, 8, 2, 100, 1), (512, 64, 3, 75, 1), (512, 5, 1, 40, 0), (512, 6, 2, 90, 2),
                                       (256, 8, 2, 120, 1), (256, 5, 1, 70, 0), (256, 12, 2, 90, 2),
                                       (1024, 16, 2, 50, 1), (2048, 24, 1, 44, 1), (2048, 256, 1, 24, 1)])
def test_i16_entry_equals_f32_entry_bit_for_bit(dev, M, N, S, T, r):
    afb, pcm, W, _ = _setup(M, N, S, T, r, seed=M + N, dev=dev)
    assert afb.fused_i16()
    pf = torch.from_numpy(pcm).to(dev)
    pi = torch.from_numpy(pcm.astype(np.int16)).to(dev)
    nfr = afb.num_frames(pf.shape[-1])
    for Wx in (W, W[:1]):                                   # per-stream and shared weights
        Yf = afb.analysis_beamform(pf, Wx)[..., :nfr]
        Yi = afb.analysis_beamform(pi, Wx)[..., :nfr]
        assert float(Yf.abs().max()) > 10
        assert torch.equal(_bits(Yf), _bits(Yi)), float((Yf - Yi).abs().max())
    # a ragged piece in the middle (edge handling of t0 / tcount) and a launch whose sample count ends inside the last tile
    for (t0, tc) in ((5, 23), (nfr - 21, 21)):
        Yf = afb.analysis_beamform(pf, W, t0=t0, tcount=tc)[..., :tc]
        Yi = afb.analysis_beamform(pi, W, t0=t0, tcount=tc)[..., :tc]
        assert torch.equal(_bits(Yf), _bits(Yi))
    L2 = pf.shape[-1] - 301
    Yf = afb.analysis_beamform(pf, W, nsamples=L2)
    Yi = afb.analysis_beamform(pi, W, nsamples=L2)
    n2 = afb.num_frames(L2)
    assert torch.equal(_bits(Yf[..., :n2]), _bits(Yi[..., :n2]))


def test_i16_entry_unaligned_rows_and_odd_pitch(dev):
    """rows that are not 8-byte aligned / an odd row pitch take the guarded loads: still the float path's bits"""
    from distant_speech_recognition_amd import engine as eng
    afb, pcm, W, _ = _setup(512, 8, 2, 60, seed=3, dev=dev)
    pf = torch.from_numpy(pcm).to(dev)
    S, N, L = pf.shape
    buf = torch.zeros((S, N, L + 3), dtype=torch.int16, device=dev)
    buf[..., :L] = torch.from_numpy(pcm.astype(np.int16)).to(dev)
    nfr = afb.num_frames(L)
    Yf = afb.analysis_beamform(pf, W)[..., :nfr]
    Y = eng.padded_rows((S, afb.K, nfr), torch.complex64, dev)
    nb = eng._lib.lib().btk_fb_analysis_bf_scratch_bytes(afb._h, S, N, 1, nfr)
    scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
    eng.check(eng._lib.lib().btk_fb_analysis_bf_i16(afb._h, buf.data_ptr(), L, L + 3, S, N, W.data_ptr(), 1, Y.data_ptr(), Y.stride(1), 0, nfr,
                                                    scratch.data_ptr(), nb, 0))
    torch.cuda.synchronize()
    assert torch.equal(_bits(Yf), _bits(Y[..., :nfr]))


def test_i16_entry_against_oracle(orc, dev):
    from tests.util import design_prototype
    M, m, r, dct, N = 512, 4, 1, 2, 8
    afb, pcm, W, delays = _setup(M, N, 1, 48, seed=11, dev=dev)
    h = design_prototype(M, m)
    X = np.stack([orc.analysis(h, M, m, r, dct, pcm[0, c]) for c in range(N)], axis=1)       # [T][N][M]
    Wn = W.cpu().numpy()[0].astype(np.complex128)
    nfr = afb.num_frames(pcm.shape[-1])
    ref = np.einsum("kn,tnk->kt", np.conj(Wn), X[:nfr, :, :afb.K])
    Yi = afb.analysis_beamform(torch.from_numpy(pcm.astype(np.int16)).to(dev), W)[0, :, :nfr].cpu().numpy()
    assert np.max(np.abs(Yi - ref)) <= 2e-6 * np.sqrt(N) * 4 * np.max(np.abs(ref))


def test_geometries_without_an_i16_kernel_say_so(dev):
    from distant_speech_recognition_amd import engine as eng, _lib
    from tests.util import design_prototype
    afb = eng.FilterBank(design_prototype(128, 4), 128, 4, 1, 2)
    assert not afb.fused_i16()
    pcm = torch.zeros((1, 4, 64 * 40), dtype=torch.int16, device=dev)
    W = torch.zeros((1, afb.K, 4), dtype=torch.complex64, device=dev)
    with pytest.raises(_lib.BtkError):
        afb.analysis_beamform(pcm, W)


@pytest.mark.parametrize("M,N,S,T,r", [(512, 8, 2, 100, 1), (512, 64, 2, 300, 1), (512, 5, 1, 40, 0), (512, 6, 2, 90, 2),
                                       (256, 8, 2, 150, 1), (256, 5, 1, 60, 0), (1024, 6, 2, 70, 1), (1024, 4, 1, 50, 2), (2048, 5, 1, 40, 1)])
def test_staged_i16_analysis_equals_f32_analysis_bit_for_bit(dev, M, N, S, T, r):
    """btk_fb_analysis_i16 (the STAGED bank on 16-bit PCM, M = 256 ... 2048): the snapshots of btk_fb_analysis on the float copies of
    the same samples, bit for bit -- whole launches (runs of tiles and the ragged last tile), pieces in the middle, a sample count
    that ends inside a tile, contiguous and row-padded snapshot blocks"""
    afb, pcm, _, _ = _setup(M, N, S, T, r, seed=7 + N + M, dev=dev)
    assert afb.analysis_i16()
    pf = torch.from_numpy(pcm).to(dev)
    pi = torch.from_numpy(pcm.astype(np.int16)).to(dev)
    nfr = afb.num_frames(pf.shape[-1])
    Xf = afb.analysis(pf)
    Xi = afb.analysis(pi)
    assert Xf.shape == Xi.shape == (S, afb.K, N, nfr) and float(Xf.abs().max()) > 100
    assert torch.equal(_bits(Xf), _bits(Xi))
    Xp = afb.analysis(pi, pad_rows=True)
    assert torch.equal(_bits(Xf), _bits(Xp[..., :nfr]))
    for (t0, tc) in ((5, 23), (nfr - 21, 21), (16, 32)):
        assert torch.equal(_bits(afb.analysis(pf, t0=t0, tcount=tc)), _bits(afb.analysis(pi, t0=t0, tcount=tc)))
    L2 = pf.shape[-1] - 301
    assert torch.equal(_bits(afb.analysis(pf, nsamples=L2)), _bits(afb.analysis(pi, nsamples=L2)))


def test_staged_i16_analysis_unaligned_rows_and_other_geometries(dev):
    """rows that are not 8-byte aligned / an odd row pitch take the guarded loads (same bits); a bin range or a geometry without
    the int16 form is refused with a message that says what to do"""
    from distant_speech_recognition_amd import engine as eng, _lib
    from tests.util import design_prototype
    afb, pcm, _, _ = _setup(512, 8, 2, 60, seed=5, dev=dev)
    pf = torch.from_numpy(pcm).to(dev)
    S, N, L = pf.shape
    nfr = afb.num_frames(L)
    Xf = afb.analysis(pf)
    for pad, off in ((3, 0), (2, 1)):                        # odd pitch; even pitch with a 2-byte offset of the first row
        buf = torch.zeros(S * N * (L + pad) + 8, dtype=torch.int16, device=dev)
        view = buf[off:off + S * N * (L + pad)].view(S, N, L + pad)
        view[..., :L] = torch.from_numpy(pcm.astype(np.int16)).to(dev)
        X = torch.empty((S, afb.K, N, nfr), dtype=torch.complex64, device=dev)
        eng.check(_lib.lib().btk_fb_analysis_i16(afb._h, view.data_ptr(), L, L + pad, S, N, X.data_ptr(), nfr, 0, nfr, 0))
        torch.cuda.synchronize()
        assert torch.equal(_bits(Xf), _bits(X)), (pad, off)
    with pytest.raises(_lib.BtkError):
        afb.analysis(torch.from_numpy(pcm.astype(np.int16)).to(dev), bins=(3, 40))
    a128 = eng.FilterBank(design_prototype(128, 4), 128, 4, 1, 2)
    assert not a128.analysis_i16()
    with pytest.raises(_lib.BtkError):
        a128.analysis(torch.zeros((1, 4, 64 * 40), dtype=torch.int16, device=dev))
