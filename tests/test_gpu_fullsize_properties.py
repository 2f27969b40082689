"""GPU: size-independent properties at BASELINE's full launch sizes (the float64 oracle cannot run these in seconds):
C0 = 32 streams x 64 mics x 4096 frames at M = 512 (exactly the launch bench.py times: 8 192 workgroups of the fused kernel), and a
256-mic / 2048-bin C4 launch."""
import numpy as np
import pytest

from tests.util import design_prototype

pytestmark = pytest.mark.gpu


def test_c0_full_launch_properties(dev):
    import torch
    from distant_speech_recognition_amd import engine as eng
    S, N, M, m, r, T = 32, 64, 512, 4, 1, 4096            # bench.py's launch (streams_per_gpu = 32)
    D, K = M >> r, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    sfb = eng.FilterBank(design_prototype(M, m, "g"), M, m, r, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D
    g = torch.Generator(device=dev).manual_seed(1234)
    pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000).round_()
    W = (torch.randn((K, N), device=dev, generator=g) + 1j * torch.randn((K, N), device=dev, generator=g)).to(torch.complex64) / N
    # 1. fused analysis+apply == staged analysis -> apply at the full size
    X = afb.analysis(pcm)
    assert X.shape == (S, K, N, T)
    Ys = eng.bf_apply(W, X)
    Yf = afb.analysis_beamform(pcm, W)
    scale = float(Ys.abs().max())
    assert float((Yf - Ys).abs().max()) <= 2e-6 * np.sqrt(N) * scale + 1e-6 * scale
    # 2. streams are independent: a stream computed alone equals its slice of the batched launch, bit for bit
    for s in (0, 7, 31):
        assert torch.equal(afb.analysis(pcm[s:s + 1].contiguous()), X[s:s + 1])
        assert torch.equal(afb.analysis_beamform(pcm[s:s + 1].contiguous(), W), Yf[s:s + 1])
    # 3. Hermitian symmetry of a real input: bins 0 and M/2 are real (up to the FFT's rounding)
    assert float(X[:, 0].imag.abs().max()) <= 1e-5 * float(X[:, 0].abs().max())
    assert float(X[:, K - 1].imag.abs().max()) <= 1e-5 * float(X[:, K - 1].abs().max())
    # 4. linearity of the whole chain in the PCM (analysis, apply and synthesis are linear maps)
    pcm2 = (torch.randn((S, N, L), device=dev, generator=g) * 700).round_()
    o1, o2 = sfb.synthesize(Yf), sfb.synthesize(afb.analysis_beamform(pcm2, W))
    o12 = sfb.synthesize(afb.analysis_beamform(pcm + pcm2, W))
    assert float((o12 - (o1 + o2)).abs().max()) <= 2e-5 * float(o12.abs().max())
    # 5. chunked launches (t0/tcount) tile the full launch exactly; a checksum over chunks equals the checksum of the whole
    parts = [afb.analysis_beamform(pcm, W, t0=a, tcount=min(1000, T - a)) for a in range(0, T, 1000)]
    assert torch.equal(torch.cat(parts, dim=-1), Yf)
    assert abs(sum(float(p.abs().double().sum()) for p in parts) - float(Yf.abs().double().sum())) <= 1e-9 * float(Yf.abs().double().sum())
    # 5b. the int16 entry at the full launch: the same samples as 16-bit PCM -> the same bits (btk_fb_analysis_bf_i16)
    Yi = afb.analysis_beamform(pcm.to(torch.int16), W)
    assert torch.equal(Yi.contiguous().view(torch.float32).view(torch.int32), Yf.contiguous().view(torch.float32).view(torch.int32))
    del Yi
    # 6. synthesis blocks: number and stream independence
    assert o1.shape == (S, sfb.num_blocks(T) * D)
    assert torch.equal(sfb.synthesize(Yf[3:4].contiguous()), o1[3:4])


def test_reconstruction_at_scale_reference_prototypes(dev, proto256):
    """analysis -> (channel 0) -> synthesis with the reference's Nyquist(M) prototypes reproduces the input at lag 0 for
    every stream of a large launch (the identity tools/filterbank/test_oversampled_dft_filter.py measures)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    h, g = proto256
    S, N, M, T = 32, 16, 256, 2048
    afb = eng.FilterBank(h, M, 4, 1, 2)
    sfb = eng.FilterBank(g, M, 4, 1, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * 128
    gen = torch.Generator(device=dev).manual_seed(5)
    pcm = (torch.randn((S, N, L), device=dev, generator=gen) * 3000).round_()
    W = torch.zeros((129, N), dtype=torch.complex64, device=dev)
    W[:, 0] = 1.0                                                        # pick channel 0
    y = sfb.synthesize(afb.analysis_beamform(pcm, W))
    n = min(y.shape[1], L)
    x = pcm[:, 0, :n]
    err = (y[:, :n] - x)[:, 1024: n - 1024]
    snr = 10 * torch.log10((x[:, 1024: n - 1024] ** 2).sum(dim=1) / (err ** 2).sum(dim=1))
    assert float(snr.min()) > 50.0, snr


@pytest.mark.parametrize("M,S,N,T", [(1024, 16, 4, 2048), (2048, 8, 4, 1024)])
def test_reconstruction_at_scale_designed_prototypes(dev, M, S, N, T):
    """The same identity for the BASELINE geometries M = 1024 / 2048 (prototypes of the reference's designer, shipped as data) at a
    launch that fills the chip: analysis -> pick one channel -> synthesis (the register-history synthesis kernel of round 3) gives the
    input back at lag 0 for every stream, and a stream synthesised alone equals its rows of the batched launch."""
    import torch
    from distant_speech_recognition_amd import engine as eng, prototypes
    h, g = prototypes.load(M, 4, 1)
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(h, M, 4, 1, 2)
    sfb = eng.FilterBank(g, M, 4, 1, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D
    gen = torch.Generator(device=dev).manual_seed(M)
    noise = torch.randn((S, N, L + 7), device=dev, generator=gen) * 3000
    pcm = noise.unfold(-1, 8, 1).mean(dim=-1).round_().contiguous()           # smoothed noise at int16 scale (the designer's stop band is finite)
    X = afb.analysis(pcm)
    W = torch.zeros((K, N), dtype=torch.complex64, device=dev)
    W[:, 1] = 1.0                                                              # pick channel 1
    Y = eng.bf_apply(W, X)
    y = sfb.synthesize(Y)
    n = min(y.shape[1], L)
    a, b = 8 * M, n - 8 * M
    x = pcm[:, 1, a:b]
    err = y[:, a:b] - x
    snr = 10 * torch.log10((x ** 2).sum(dim=1) / (err ** 2).sum(dim=1))
    assert float(snr.min()) > 45.0, snr
    assert torch.equal(sfb.synthesize(Y[2:3].contiguous()), y[2:3])


def test_c4_full_launch_properties(dev):
    """256 mics, 2048 bins: apply is linear, the all-ones/N weight returns the channel mean, bin shards tile the launch"""
    import torch
    from distant_speech_recognition_amd import engine as eng, sharding
    S, N, M, T = 1, 256, 2048, 512
    K = M // 2 + 1
    g = torch.Generator(device=dev).manual_seed(9)
    X = (torch.randn((S, K, N, T), device=dev, generator=g) + 1j * torch.randn((S, K, N, T), device=dev, generator=g)).to(torch.complex64)
    W = torch.full((K, N), 1.0 / N, dtype=torch.complex64, device=dev)
    Y = eng.bf_apply(W, X)
    assert torch.allclose(Y, X.mean(dim=2), atol=2e-5)
    Wr = (torch.randn((K, N), device=dev, generator=g) + 1j * torch.randn((K, N), device=dev, generator=g)).to(torch.complex64) / N
    Yr = eng.bf_apply(Wr, X)
    shards = []
    for rank in range(8):
        k0, k1 = sharding.bin_range_for_rank(K, rank, 8)
        shards.append(eng.bf_apply(Wr[k0:k1].contiguous(), X[:, k0:k1].contiguous()))
    assert torch.equal(torch.cat(shards, dim=1), Yr)


def test_c0_coherence_postfilter_properties(dev):
    """McCowan / Lefkimmiatis at the C0 snapshot shape (8 streams x 257 bins x 64 mics x 4096 frames; the matrix-core statistics
    kernel): streams are independent and a block split in time continues exactly (the recursion carries its state), and the
    pair-weighted sums are quadratic in the snapshots -- doubling X (exact in float32) leaves every gain, hence Y / 2, unchanged
    bit for bit."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    from tests.util import ula_positions
    S, N, M, T = 8, 64, 512, 4096
    K = M // 2 + 1
    g = torch.Generator(device=dev).manual_seed(77)
    X = (torch.randn((S, K, N, T), device=dev, generator=g) + 1j * torch.randn((S, K, N, T), device=dev, generator=g)).to(torch.complex64) * 1500
    X += ((torch.randn((S, K, 1, T), device=dev, generator=g) + 1j * torch.randn((S, K, 1, T), device=dev, generator=g)) * 2500).to(torch.complex64)
    d = torch.polar(torch.full((K, N), 1.0 / N, device=dev), torch.rand((K, N), device=dev, generator=g) * 6.2831853).to(torch.complex64)
    R = eng.mvdr_diffuse_model(ula_positions(N, 20.0), M, 16000.0, device=dev)
    eng.mvdr_diagonal_loading(R, 0.01)

    def run(Xin, lef, splits=(T,)):
        st = eng.CoherencePostFilterState(Xin.shape[0], K, N, dev, lefkimmiatis=lef)
        st.set_coherence(R, 0.99)
        if lef:
            st.set_lambda(R, d, 1.0e-4)
        outs, a = [], 0
        for n in splits:
            blk = Xin[..., a:a + n].contiguous()
            outs.append(eng.bf_apply_lefkimmiatis(d, d, blk, st, fbin_x1=100, alpha=0.8) if lef else eng.bf_apply_mccowan(d, d, blk, st, alpha=0.7))
            a += n
        return torch.cat(outs, dim=-1), st.w_last.clone()

    for lef in (False, True):
        Y, wl = run(X, lef)
        assert torch.isfinite(Y.abs()).all()
        assert 1e-4 < float(wl.mean()) < 1.0                                    # gains neither all at the floor nor all at one
        Y3, wl3 = run(X[3:4].contiguous(), lef)                                 # a stream alone == its slice of the batch
        assert torch.equal(Y3, Y[3:4]) and torch.equal(wl3, wl[3:4])
        Ys, wls = run(X, lef, splits=(1024, 2048, 1024))                        # blocks continue the recursion exactly (64-frame scan chunks)
        assert torch.equal(Ys, Y) and torch.equal(wls, wl)
        Y2, wl2 = run(X * 2, lef)                                               # gains are ratios of quadratic forms
        assert torch.equal(Y2, Y * 2) and torch.equal(wl2, wl)


def test_c0_adaptive_chain_full_launch_properties(dev):
    """The ADAPTIVE chain at bench.py's launch (32 streams x 64 mics x 4096 frames, M = 512: analysis -> snapshots in HBM -> NLMS
    sidelobe canceller (lib/pybeamformer.py:659-762) -> synthesis), properties that need no float64 oracle run:
    the two-HIP-stream pipeline (engine.AdaptiveGSCChain) and a split in time give the one-launch chain's bits incl. the state;
    a stream alone is its slice of the batch; with a zero step size the canceller is the quiescent beamformer vs^H x; and with
    adaptation on, a target that lies in the look direction passes while the output power does not exceed the quiescent one's."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    from tests.util import la_delays, ula_positions
    S, N, M, m, r, T = 32, 64, 512, 4, 1, 4096
    D, K = M >> r, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, m), M, m, r, 2)
    sfb = eng.FilterBank(design_prototype(M, m, "g"), M, m, r, 2, synthesis=True)
    L = (T - afb.processing_delay + afb.lookahead) * D
    g = torch.Generator(device=dev).manual_seed(4321)
    pcm = (torch.randn((S, N, L), device=dev, generator=g) * 1000).round_()
    pcm += (torch.randn((S, 1, L), device=dev, generator=g) * 3000).round_()           # a broadside target, common to all channels
    delays = la_delays(ula_positions(N), 0.5 * np.pi)                                 # look direction: broadside (all delays 0)
    assert np.max(np.abs(delays)) < 1e-12
    vs = torch.from_numpy(np.stack([np.exp(-2j * np.pi * k * (16000.0 / M) * delays) / N for k in range(K)]).astype(np.complex64)).to(dev)
    kw = dict(min_frames=16, slowdown_after=1024)

    def fresh():
        return eng.NLMSState(S, M, N, dev, **kw)

    # one launch per kernel
    st = fresh()
    X = afb.analysis(pcm)
    Y = eng.nlms_process(vs, X, st)
    out = sfb.synthesize(Y)
    assert torch.isfinite(out).all() and float(st.u.abs().max()) > 0                   # the canceller adapted
    # 1. analysis running ahead of the canceller on a second HIP stream: same bits, same state
    for chunk in (512, 1024):
        st2, X2 = fresh(), torch.empty_like(X)
        Y2 = torch.empty_like(Y)
        out2 = eng.AdaptiveGSCChain(afb, sfb, chunk_frames=chunk)(pcm, vs, st2, X2, Y2)
        torch.cuda.synchronize()
        assert torch.equal(out2, out) and torch.equal(Y2, Y) and torch.equal(st2.u, st.u) and torch.equal(st2.sigma2, st.sigma2)
        assert torch.equal(st2.stream_state, st.stream_state)
    del X2, Y2
    # 1b. round 6: this launch (2 080 single-wavefront workgroups, 2 048 resident) runs by default as two staggered groups of streams
    #     on two HIP streams in 512-frame chunks (engine._nlms_interleave_plan) -- the one-launch form and other groupings: same bits
    assert eng._nlms_interleave_plan(S, K, N, T, fresh())[0] == 2
    for plan in ((1, T), (4, 256, True), (2, 1024, False), (3, 448, True)):
        st2 = fresh()
        Y2 = eng.nlms_process(vs, X, st2, interleave=plan)
        torch.cuda.synchronize()
        assert torch.equal(Y2, Y) and torch.equal(st2.u, st.u) and torch.equal(st2.sigma2, st.sigma2) and torch.equal(st2.stream_state, st.stream_state), plan
    del Y2
    # 2. a split in time (blocks of 1024 + 3072 frames) continues the recursion exactly
    st3 = fresh()
    Y3 = torch.cat([eng.nlms_process(vs, X[..., :1024].contiguous(), st3), eng.nlms_process(vs, X[..., 1024:].contiguous(), st3)], dim=-1)
    assert torch.equal(Y3, Y) and torch.equal(st3.u, st.u)
    del Y3
    # 3. streams are independent
    st4 = eng.NLMSState(1, M, N, dev, **kw)
    assert torch.equal(eng.nlms_process(vs, X[5:6].contiguous(), st4), Y[5:6]) and torch.equal(st4.u, st.u[5:6])
    # 4. zero step size: the quiescent beamformer Yc = vs^H x
    st5 = eng.NLMSState(S, M, N, dev, gamma=0.0, **kw)
    Yq = eng.nlms_process(vs, X, st5)
    Yref = eng.bf_apply(vs, X)
    assert float((Yq - Yref).abs().max()) <= 2e-6 * np.sqrt(N) * float(Yref.abs().max()) and not bool(st5.u.any())
    # 5. the canceller removes what is not in the look direction: less output power than the quiescent beamformer, and the target
    #    (the common signal: channel mean of the snapshots, which the blocking matrix cannot see) is still there
    p_q, p_a = float(Yq[..., 64:].abs().double().pow(2).sum()), float(Y[..., 64:].abs().double().pow(2).sum())
    assert p_a < p_q
    tgt = X.mean(dim=2)                                                                # broadside: vs^H x is this mean
    corr = float((Y * tgt.conj()).real.double().sum() / tgt.abs().double().pow(2).sum())
    assert 0.8 < corr < 1.1, corr


@pytest.mark.parametrize("M,S,N,T", [(1024, 8, 64, 16384), (2048, 8, 64, 8192)])
def test_large_geometry_fused_kernel_beyond_4GB_of_samples(dev, M, S, N, T):
    """analysis_bfz_big_kernel on 17 GB of PCM (C3's channel count; its window rows and weight pairs are buffer loads whose resource is the
    channel's row base -- a 48-bit address -- and whose offsets stay inside one tile's span): equal to the staged pair on every frame of
    every stream, streams independent bit for bit, linear in the samples."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    D, K = M // 2, M // 2 + 1
    afb = eng.FilterBank(design_prototype(M, 4), M, 4, 1, 2)
    L = (T - afb.processing_delay + afb.lookahead) * D
    assert S * N * L * 4 > 15 * 2 ** 30
    g = torch.Generator(device=dev).manual_seed(M)
    pcm = torch.empty((S, N, L), dtype=torch.float32, device=dev)
    for s in range(S):
        pcm[s] = (torch.randn((N, L), device=dev, generator=g) * 1000).round_()
    W = (torch.randn((K, N), device=dev, generator=g) + 1j * torch.randn((K, N), device=dev, generator=g)).to(torch.complex64) / N
    Yf = afb.analysis_beamform(pcm, W)
    assert Yf.shape == (S, K, T)
    for s in (0, S - 1):                                                      # the staged pair one stream at a time (34 GB of snapshots for all of them)
        Ys = eng.bf_apply(W, afb.analysis(pcm[s:s + 1]))
        scale = float(Ys.abs().max())
        assert float((Yf[s:s + 1] - Ys).abs().max()) <= 2e-6 * np.sqrt(N) * scale + 1e-6 * scale
        assert torch.equal(afb.analysis_beamform(pcm[s:s + 1], W), Yf[s:s + 1])
        del Ys
    pcm2 = torch.empty_like(pcm)
    for s in range(S):
        pcm2[s] = (torch.randn((N, L), device=dev, generator=g) * 700).round_()
    Y2 = afb.analysis_beamform(pcm2, W)
    pcm2 += pcm
    Y12 = afb.analysis_beamform(pcm2, W)
    assert float((Y12 - (Yf + Y2)).abs().max()) <= 2e-5 * float(Y12.abs().max())
