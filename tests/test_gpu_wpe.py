"""GPU parity: multi-channel WPE dereverberation (estimate + apply) vs the oracle restatement of
dereverberation/dereverberation.cc:312-698."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _reverberant(rng, T, C, M):
    """AR-ish reverberant subband signals so that the prediction filters are non-trivial."""
    K = M // 2 + 1
    src = (rng.normal(size=(T + 16, K)) + 1j * rng.normal(size=(T + 16, K))) * 500.0
    Y = np.zeros((T, C, M), np.complex128)
    for c in range(C):
        taps = (rng.normal(size=(8, K)) + 1j * rng.normal(size=(8, K))) * (0.6 ** np.arange(8))[:, None]
        for t in range(T):
            Y[t, c, :K] = sum(taps[d] * src[t + 16 - d] for d in range(8))
    Y[:, :, 0] = Y[:, :, 0].real
    Y[:, :, M // 2] = Y[:, :, M // 2].real
    Y[:, :, K:] = np.conj(Y[:, :, M // 2 - 1:0:-1])
    return Y


@pytest.mark.parametrize("C,M,T,lower,upper,iters", [(2, 64, 120, 0, 5, 2), (3, 64, 90, 1, 4, 2), (4, 64, 150, 0, 15, 1), (1, 64, 100, 2, 6, 2),
                                                    # the lag-product kernel (4 / 8 channels): delayed prediction (own r-vector kernel), lag counts that leave partial row
                                                    # blocks, frame counts that are not a multiple of its 64-frame tile
                                                    (8, 32, 200, 1, 10, 2), (4, 64, 130, 2, 11, 2), (8, 16, 331, 0, 12, 1),
                                                    # the matrix-core prediction kernel with a channel count that is not a multiple of four (block HERK + MFMA prediction)
                                                    (5, 32, 300, 1, 6, 2), (12, 16, 270, 0, 4, 1), (8, 16, 530, 3, 9, 1)])
def test_wpe_matches_oracle(orc, dev, C, M, T, lower, upper, iters):
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(C * 100 + T)
    K = M // 2 + 1
    Y = _reverberant(rng, T, C, M)
    Xe = np.ascontiguousarray(np.transpose(Y[:, :, :K], (2, 1, 0))[None]).astype(np.complex64)      # [1][K][C][T]
    Yo = np.zeros((T, C, M), np.complex128)                                                        # what the GPU saw
    Yo[:, :, :K] = np.transpose(Xe[0].astype(np.complex128), (2, 1, 0))
    Yo[:, :, K:] = np.conj(Yo[:, :, M // 2 - 1:0:-1])
    Gref = orc.wpe_estimate(Yo, lower, upper, iters, -18.0, 0.0, 1e-4)                               # [C][M][P]
    ref = orc.wpe_apply(Yo, Gref, lower, upper)
    Xd = torch.from_numpy(Xe).to(dev)
    G = eng.wpe_estimate(Xd, M, lower_num=lower, upper_num=upper, iterations_num=iters, load_db=-18.0, diagonal_bias=1e-4)
    out = eng.wpe_apply(Xd, G, M, lower_num=lower, upper_num=upper).cpu().numpy()[0]                # [K][C][T]
    Gg = G.cpu().numpy()[0]                                                                        # [C][K][P]
    gscale = np.max(np.abs(Gref[:, :K]))
    assert gscale > 1e-2
    # filters: normal equations solved in float32 vs float64 -> 2e-3 of the largest tap
    assert np.max(np.abs(Gg - Gref[:, :K])) <= 2e-3 * gscale
    got = np.transpose(out, (2, 1, 0))                                                             # [T][C][K]
    assert np.max(np.abs(got - ref[:, :, :K])) <= 1e-3 * np.max(np.abs(ref))
    # dereverberation actually removes energy
    assert np.sum(np.abs(got) ** 2) < 0.99 * np.sum(np.abs(Yo[:, :, :K]) ** 2)


def test_wpe_reference_configuration_8ch_lags0to32(orc, dev):
    """unit_test/confs/wpe.json as shipped: lower_num 0, upper_num 32 (33 lags incl. the current frame), 2 iterations,
    load_db -18, diagonal_bias 1e-4 -- with 8 channels that is a 264 x 264 system per bin and channel (the blocked
    Cholesky path of wpe_solve_kernel, several 64 x 64 HERK tiles).  M = 16 keeps the float64 oracle at ~10 s."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    C, M, T, lower, upper, iters = 8, 16, 520, 0, 32, 2
    rng = np.random.default_rng(2026)
    K = M // 2 + 1
    Y = _reverberant(rng, T, C, M)
    Xe = np.ascontiguousarray(np.transpose(Y[:, :, :K], (2, 1, 0))[None]).astype(np.complex64)
    Yo = np.zeros((T, C, M), np.complex128)
    Yo[:, :, :K] = np.transpose(Xe[0].astype(np.complex128), (2, 1, 0))
    Yo[:, :, K:] = np.conj(Yo[:, :, M // 2 - 1:0:-1])
    Gref = orc.wpe_estimate(Yo, lower, upper, iters, -18.0, 0.0, 1e-4)
    ref = orc.wpe_apply(Yo, Gref, lower, upper)
    Xd = torch.from_numpy(Xe).to(dev)
    G = eng.wpe_estimate(Xd, M, lower_num=lower, upper_num=upper, iterations_num=iters, load_db=-18.0, diagonal_bias=1e-4)
    assert G.shape == (1, C, K, C * (upper - lower + 1))
    out = eng.wpe_apply(Xd, G, M, lower_num=lower, upper_num=upper).cpu().numpy()[0]
    Gg = G.cpu().numpy()[0]
    gscale = np.max(np.abs(Gref[:, :K]))
    # lag 0 is the current frame: its own tap dominates (the filter nearly predicts the frame from itself)
    assert gscale > 0.1
    assert np.max(np.abs(Gg - Gref[:, :K])) <= 5e-3 * gscale             # 264-dim normal equations in float32 vs float64
    got = np.transpose(out, (2, 1, 0))
    # lag 0 predicts the frame from itself, so the output is small: bound the error by the REFERENCE OUTPUT's scale (a bound at the
    # input's scale would pass almost anything), and make sure that output is not just noise around zero
    assert np.max(np.abs(ref)) > 1e-3 * np.max(np.abs(Yo))
    assert np.max(np.abs(got - ref[:, :, :K])) <= 2e-2 * np.max(np.abs(ref[:, :, :K]))
    assert np.linalg.norm(got - ref[:, :, :K]) <= 5e-3 * np.linalg.norm(ref[:, :, :K])


@pytest.mark.parametrize("T", [520, 777])
def test_wpe_delayed_prediction_8ch_lags1to33(orc, dev, T):
    """The reference-size system (8 channels x 33 lags = 264 x 264 per bin and channel) with DELAYED prediction, lower_num = 1
    (lags 1..33, dereverberation.cc:557-690): the lag-product normal equations with their own r-vector kernel, against the oracle --
    round 3 checked this size against the oracle only with lower_num = 0 and the delayed form only kernel against kernel.  The
    output here is the dereverberated signal itself (same order of magnitude as the input), bounded at ITS scale."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    C, M, lower, upper, iters = 8, 16, 1, 33, 2
    rng = np.random.default_rng(4000 + T)
    K = M // 2 + 1
    Y = _reverberant(rng, T, C, M)
    Xe = np.ascontiguousarray(np.transpose(Y[:, :, :K], (2, 1, 0))[None]).astype(np.complex64)
    Yo = np.zeros((T, C, M), np.complex128)
    Yo[:, :, :K] = np.transpose(Xe[0].astype(np.complex128), (2, 1, 0))
    Yo[:, :, K:] = np.conj(Yo[:, :, M // 2 - 1:0:-1])
    Gref = orc.wpe_estimate(Yo, lower, upper, iters, -18.0, 0.0, 1e-4)
    ref = orc.wpe_apply(Yo, Gref, lower, upper)
    Xd = torch.from_numpy(Xe).to(dev)
    G = eng.wpe_estimate(Xd, M, lower_num=lower, upper_num=upper, iterations_num=iters, load_db=-18.0, diagonal_bias=1e-4)
    assert G.shape == (1, C, K, C * (upper - lower + 1))
    out = eng.wpe_apply(Xd, G, M, lower_num=lower, upper_num=upper).cpu().numpy()[0]
    Gg = G.cpu().numpy()[0]
    gscale = np.max(np.abs(Gref[:, :K]))
    assert gscale > 1e-2
    assert np.max(np.abs(Gg - Gref[:, :K])) <= 5e-3 * gscale             # 264-dim normal equations in float32 vs float64
    got = np.transpose(out, (2, 1, 0))
    rscale = np.max(np.abs(ref[:, :, :K]))
    assert rscale > 0.05 * np.max(np.abs(Yo))                             # a real signal, not a residual near zero
    assert np.max(np.abs(got - ref[:, :, :K])) <= 5e-3 * rscale
    assert np.linalg.norm(got - ref[:, :, :K]) <= 2e-3 * np.linalg.norm(ref[:, :, :K])
    assert np.sum(np.abs(got) ** 2) < 0.99 * np.sum(np.abs(Yo[:, :, :K]) ** 2)


def test_wpe_long_utterance_accumulators_are_flushed(orc, dev):
    """20 000 frames (80 s at D = 64) through the float16-split lag-product kernel: the low-part products are 2^-11 of the high-part ones
    and of one sign on the diagonal of the normal equations, so an accumulator that runs over the whole utterance loses them to rounding
    (measured: 4.9e-3 of the largest tap at this length, 0.29 at 100 000 frames -- profiles/r05_wpe_long_utterance.txt).  The kernel
    flushes its accumulators to R every 2048 frames; the filters then stay at the float32-solve level against the float64 oracle
    (6.2e-4 measured; the all-float32 kernel gives 1.4e-3 here)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    M, C, T = 16, 8, 20000
    K = M // 2 + 1
    g = torch.Generator(device=dev).manual_seed(3)
    src = (torch.randn((K, T + 16), device=dev, generator=g) + 1j * torch.randn((K, T + 16), device=dev, generator=g)) * 500
    X = torch.zeros((1, K, C, T), dtype=torch.complex64, device=dev)
    for c in range(C):
        for dd in range(6):
            X[0, :, c] += (0.6 ** dd) * np.exp(1j * (c + dd)) * src[:, 16 - dd: 16 - dd + T]
    X += 5 * (torch.randn(X.shape, device=dev, generator=g) + 1j * torch.randn(X.shape, device=dev, generator=g))
    G = eng.wpe_estimate(X, M, lower_num=0, upper_num=7, iterations_num=2, load_db=-18.0, diagonal_bias=1e-4).cpu().numpy()[0]
    Y = np.zeros((T, C, M), np.complex128)
    Y[:, :, :K] = X[0].cpu().numpy().transpose(2, 1, 0)
    Y[:, :, K:] = np.conj(Y[:, :, M // 2 - 1:0:-1])
    Gref = orc.wpe_estimate(Y, 0, 7, 2, -18.0, 0.0, 1e-4)[:, :K]
    assert np.max(np.abs(Gref)) > 1e-2
    assert np.max(np.abs(G - Gref)) <= 2e-3 * np.max(np.abs(Gref))


@pytest.mark.parametrize("db", [40, 60])
def test_wpe_non_stationary_envelope(orc, dev, db):
    """Speech-like dynamics: loud and quiet segments `db` apart, each longer than the lag span.  The weights are 1 / |y|^2, so a
    quiet segment pairs large weights with tiny products and a loud one the reverse; the float16-split lag-product kernel keeps
    both operands in float16's normal range by trading a power of two between them per 64-frame tile (wpe_kernels.hip,
    lagprod16_task).  With one scale per (stream, bin) the taps were 1.4e-3 / 1.9e-3 of the largest off the float64 oracle at
    40 / 60 dB -- the test bound of the stationary cases -- against 3e-5 / 1.5e-4 for float32 products; balanced: 1.3e-4 / 2.3e-4."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    C, M, T, lower, upper = 8, 16, 2400, 1, 10
    K = M // 2 + 1
    rng = np.random.default_rng(7)
    src = (rng.normal(size=(T + 16, K)) + 1j * rng.normal(size=(T + 16, K)))
    env = np.ones(T + 16)
    seg = 150
    for i in range(0, T + 16, seg):
        env[i:i + seg] = 1.0 if (i // seg) % 2 == 0 else 10.0 ** (-db / 20.0)
    src *= (env * 8000.0)[:, None]                          # int16-scale loud segments
    Y = np.zeros((T, C, M), np.complex128)
    for c in range(C):
        taps = (rng.normal(size=(8, K)) + 1j * rng.normal(size=(8, K))) * (0.6 ** np.arange(8))[:, None]
        for t in range(T):
            Y[t, c, :K] = sum(taps[d] * src[t + 16 - d] for d in range(8))
    Y[:, :, :K] += (rng.normal(size=(T, C, K)) + 1j * rng.normal(size=(T, C, K))) * 0.05      # sensor noise, well above the 1e-3 floor
    Xe = np.ascontiguousarray(np.transpose(Y[:, :, :K], (2, 1, 0))[None]).astype(np.complex64)
    Yo = np.zeros((T, C, M), np.complex128)
    Yo[:, :, :K] = np.transpose(Xe[0].astype(np.complex128), (2, 1, 0))
    Yo[:, :, K:] = np.conj(Yo[:, :, M // 2 - 1:0:-1])
    Gref = orc.wpe_estimate(Yo, lower, upper, 2, -18.0, 0.0, 1e-4)[:, :K]
    G = eng.wpe_estimate(torch.from_numpy(Xe).to(dev), M, lower_num=lower, upper_num=upper, iterations_num=2, load_db=-18.0,
                         diagonal_bias=1e-4).cpu().numpy()[0]
    assert np.max(np.abs(Gref)) > 1e-4                      # (delayed prediction of a nearly white source: small taps, but not noise)
    assert np.max(np.abs(G - Gref)) <= 5e-4 * np.max(np.abs(Gref))


def test_wpe_near_silent_bin_stays_finite(dev):
    """a bin whose samples are ~1e-20 must not drive the power-of-two scales of the float16-split kernel to inf (NaN in R)"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    C, M, T = 8, 16, 300
    K = M // 2 + 1
    g = torch.Generator(device=dev).manual_seed(5)
    X = ((torch.randn((1, K, C, T), device=dev, generator=g) + 1j * torch.randn((1, K, C, T), device=dev, generator=g)) * 300).to(torch.complex64)
    X[0, 3] *= 1e-22
    X[0, 5] = 0
    G = eng.wpe_estimate(X, M, lower_num=1, upper_num=6, iterations_num=2, load_db=-18.0, diagonal_bias=1e-4)
    assert bool(torch.isfinite(torch.view_as_real(G)).all())
