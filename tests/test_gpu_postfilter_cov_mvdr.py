"""GPU parity: Zelinski post-filter, covariance accumulation (MFMA + VALU), MVDR weight design."""
import numpy as np
import pytest

from tests.util import ula_positions, la_delays

pytestmark = pytest.mark.gpu


def _rand_snapshots(rng, S, K, N, T, scale=2000.0):
    return ((rng.normal(size=(S, K, N, T)) + 1j * rng.normal(size=(S, K, N, T))) * scale).astype(np.complex64)


def _full(Xe, M):
    K, N, T = Xe.shape
    full = np.zeros((T, N, M), np.complex128)
    full[:, :, :K] = np.transpose(Xe.astype(np.complex128), (2, 1, 0))
    full[:, :, K:] = np.conj(full[:, :, M // 2 - 1:0:-1])
    return full


@pytest.mark.parametrize("N,M,T,type_,minf", [(4, 256, 150, 2, 0), (8, 64, 70, 1, 0), (64, 64, 40, 2, 5), (3, 128, 200, 2, 0)])
def test_zelinski_matches_oracle(orc, dev, N, M, T, type_, minf):
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N + M + T)
    K, S = M // 2 + 1, 2
    X = _rand_snapshots(rng, S, K, N, T)
    # correlated target so the gain is neither 1e-4 nor 1 everywhere
    X += (rng.normal(size=(S, K, 1, T)) + 1j * rng.normal(size=(S, K, 1, T))).astype(np.complex64) * 1500.0
    delays = la_delays(ula_positions(N), 0.4)
    wq = orc.calc_mainlobe(M, N, 16000, delays)
    w = eng.weights_gsc_effective(wq, None, M)
    d = wq[:K].astype(np.complex64)
    st = eng.ZelinskiState(S, K, dev)
    Xd = torch.from_numpy(X).to(dev)
    Wd, Dd = torch.from_numpy(w).to(dev), torch.from_numpy(d).to(dev)
    T1 = T // 3
    Y = torch.cat([eng.bf_apply_zelinski(Wd, Dd, Xd[..., :T1].contiguous(), st, alpha=0.7, type_=type_, min_frames=minf),
                   eng.bf_apply_zelinski(Wd, Dd, Xd[..., T1:].contiguous(), st, alpha=0.7, type_=type_, min_frames=minf)],
                  dim=-1).cpu().numpy()
    for s in range(S):
        Xo = _full(X[s], M)
        ref, Wref = orc.zelinski_frames(Xo, orc.gsc_frames(Xo, wq), wq, alpha=0.7, type_=type_, min_frames=minf)
        # stated tolerance: Zelinski recurrence <= 1e-4 relative
        assert np.max(np.abs(Y[s].T - ref[:, :K])) <= 1e-4 * np.max(np.abs(ref))
        wl = st.w_last.cpu().numpy()[s]
        assert np.max(np.abs(wl - Wref[-1, :K].real)) <= 1e-4
    assert 1e-3 < np.mean(Wref.real[:, :K]) < 0.999


@pytest.mark.parametrize("N,M,T,S", [(4, 256, 100, 2), (64, 64, 300, 1), (8, 128, 77, 2), (100, 64, 50, 1), (130, 64, 40, 1)])
@pytest.mark.parametrize("mfma", [True, False])
def test_covariance_matches_oracle(orc, dev, N, M, T, S, mfma):
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N * 3 + T)
    K = M // 2 + 1
    X = _rand_snapshots(rng, S, K, N, T, scale=300.0)
    fw = (rng.random((S, T)) > 0.3).astype(np.float32)
    tf = (rng.random((S, K, T)) > 0.5).astype(np.float32) * rng.integers(1, 3, size=(S, K, T)).astype(np.float32)
    Xd = torch.from_numpy(X).to(dev)
    R1 = eng.cov_accumulate(Xd, frame_weights=torch.from_numpy(fw).to(dev), use_mfma=mfma)
    R2 = eng.cov_accumulate(Xd, tf_weights=torch.from_numpy(tf).to(dev), use_mfma=mfma)
    # accumulate in two halves into the same R (+=)
    T1 = T // 2
    R3 = eng.cov_accumulate(Xd[..., :T1].contiguous(), use_mfma=mfma)
    R3 = eng.cov_accumulate(Xd[..., T1:].contiguous(), R=R3, use_mfma=mfma)
    R1, R2, R3 = R1.cpu().numpy(), R2.cpu().numpy(), R3.cpu().numpy()
    for s in range(S):
        Xo = _full(X[s], M)
        ref1 = orc.cov_accumulate(Xo, frame_weights=fw[s])
        ref2 = orc.cov_accumulate(Xo, masks=tf[s].T)
        ref3 = orc.cov_accumulate(Xo)
        for got, ref in ((R1[s], ref1), (R2[s], ref2), (R3[s], ref3)):
            # stated tolerance: <= 1e-5 relative Frobenius, fp32 accumulate
            for k in range(K):
                assert np.linalg.norm(got[k] - ref[k]) <= 1e-5 * np.linalg.norm(ref[k]) + 1e-3


def test_covariance_golden_from_reference_python(orc, dev, proto256, kinect_pcm, pygolden):
    """accu_stats_from_label / finalize_stats on the Kinect fixture vs the REFERENCE's numpy output."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    G = pygolden
    T, M, K = int(G["meta_T"][0]), 256, 129
    h, _ = proto256
    afb = eng.FilterBank(h, M, 4, 1, 2)
    X = afb.analysis(torch.from_numpy(kinect_pcm[None, :, : (T + 8) * 128]).to(dev))[..., :T].contiguous()
    en = eng.frame_energy(X, M)
    # VAD label (0.5 s .. 1.0 s is target) -> noise frames; gating as pybeamformer.py:967-985
    el, dt, label = 0.0, 128 / 16000.0, []
    labs, labx = [(0.5, 1.0)], 0
    for t in range(T):
        tgt = False
        if labx < len(labs):
            if el >= labs[labx][0] and (el <= labs[labx][1] or labs[labx][1] < 0):
                tgt = True
            elif el > labs[labx][1]:
                labx += 1
        label.append(0.0 if tgt else 1.0)
        el += dt
    w, cnt = eng.cov_frame_gate(en, torch.tensor([label], dtype=torch.float32, device=dev), 10.0)
    assert int(cnt.item()) == int(G["smi_noise_frames"][0])
    R = eng.cov_accumulate(X, frame_weights=w)
    raw = R.cpu().numpy()[0]
    ref = G["smi_cov_raw"]
    for k in range(K):
        assert np.linalg.norm(raw[k] - ref[k]) <= 2e-5 * np.linalg.norm(ref[k])
    fin = eng.cov_finalize(R, cnt).cpu().numpy()[0]
    for k in range(K):
        assert np.linalg.norm(fin[k] - G["smi_cov_final"][k]) <= 2e-5 * np.linalg.norm(G["smi_cov_final"][k])
    # improve_matrix_condition known answer
    R9 = torch.from_numpy((G["tf_cov_j"][9] / 50.0).astype(np.complex64)[None, None]).to(dev).contiguous()
    out = eng.cov_finalize(R9, torch.ones(1, dtype=torch.float32, device=dev), gamma=1e-3).cpu().numpy()[0, 0]
    assert np.linalg.norm(out - G["imc"]) <= 1e-5 * np.linalg.norm(G["imc"])


@pytest.mark.parametrize("N,M", [(4, 256), (8, 64), (64, 64), (100, 64), (140, 64)])
def test_mvdr_weights_match_oracle(orc, dev, N, M):
    import torch
    from distant_speech_recognition_amd import engine as eng
    K = M // 2 + 1
    mpos = ula_positions(N, 20.0)
    mpos[:, 2] = 2.0
    delays = la_delays(mpos, -1.306379)
    wq = orc.calc_mainlobe(M, N, 16000, delays)
    mu = 0.01                                               # confs/sd.json diagonal_load
    Rd = eng.mvdr_diffuse_model(mpos, M, 16000, device=dev)
    Rref = orc.diffuse_noise_model(mpos, M, 16000)
    assert np.max(np.abs(Rd.cpu().numpy() - Rref)) < 2e-6
    eng.mvdr_diagonal_loading(Rd, mu)
    Rref = orc.diagonal_loading(Rref, M, mu)
    wqd = torch.from_numpy(wq[:K].astype(np.complex64)).to(dev)
    W, nfb = eng.mvdr_weights(Rd, wqd)                              # default rule "linpack": the reference's pseudoinverse() decision
    We, nfe = eng.mvdr_weights(Rd, wqd, svd_rule="exact")           # every positive definite bin solved
    W, We = W.cpu().numpy(), We.cpu().numpy()
    assert nfe == 0
    assert np.allclose(W[0], 1.0) and np.allclose(We[0], 1.0)
    inv = np.linalg.inv(Rref[1:])
    nonconv = 0
    for k in range(1, K):
        z = inv[k - 1].conj().T @ wq[k]
        exact = z / (N * np.vdot(z, wq[k]))
        # stated tolerance: MVDR weights <= 1e-3 relative (the reference itself uses a float32 SVD)
        assert np.linalg.norm(We[k] - exact) <= 1e-3 * np.linalg.norm(exact)
        # the oracle's pinned path: pseudoinverse() through the reference's own compiled csvdc (oracle/_ref); where it returns
        # INFO != 0 the reference substitutes the identity and so does the default rule -- no bin is stepped around
        ref, info = _oracle_mvdr_bin(orc, Rref[k], wq[k], with_info=True)
        nonconv += info != 0
        assert np.linalg.norm(W[k] - ref) <= (3e-3 if info == 0 else 1e-6) * np.linalg.norm(ref), (k, info)
        if info == 0:
            assert np.array_equal(W[k], We[k])
        # distortionless known answer: w^H d = 1/N
        assert abs(np.vdot(W[k], wq[k]) - 1.0 / N) < 1e-3 / N + 1e-5
    assert nfb == nonconv
    assert N > 64 or nonconv == 0        # LINPACK's float32 SVD only gives up on the large, highly degenerate diffuse matrices


def _oracle_mvdr_bin(orc, Rk, d, threshold=1.0e-8, with_info=False):
    """calc_mvdr_weights for one bin, literally (beamformer.cc:2372-2397), on the oracle's pseudoinverse(): the identity replaces
    the inverse whenever it returns false -- a singular value under the threshold or csvdc INFO != 0 (:253-270, 2381-2383)"""
    N = d.shape[0]
    inv, ok, info = orc.pseudoinverse(Rk, threshold, return_info=True)
    if not ok:
        inv = np.eye(N, dtype=np.complex128)
    t = inv.conj().T @ d
    w = t / (np.vdot(t, d) * N)
    return (w, info) if with_info else w


@pytest.mark.parametrize("N", [4, 64])
def test_mvdr_pinv_fallback_matches_oracle(orc, dev, N):
    """Bins the Cholesky solve cannot take -- an indefinite (but non-singular) Hermitian R, a dead channel (exact zero
    row / column), a barely loaded covariance of a few frames -- get the reference's rule: float32-SVD pseudo-inverse,
    identity only when a singular value is below the threshold (beamformer.cc:232-289, 2381-2383)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N)
    K = 6
    d = (np.exp(-2j * np.pi * rng.uniform(size=(K, N))) / N).astype(np.complex64)
    R = np.zeros((K, N, N), np.complex64)
    H = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
    R[1] = ((H + H.conj().T) / 2).astype(np.complex64)                     # indefinite, non-singular: pinv weights
    X = rng.normal(size=(N, 3 * N)) + 1j * rng.normal(size=(N, 3 * N))
    R[2] = (X @ X.conj().T / (3 * N) + 0.01 * np.eye(N)).astype(np.complex64)   # positive definite: Cholesky path
    # dead channel: sigma = 0 -> identity.  (Channel 0 or N-1: there LINPACK's Householder steps keep the zero exact.  With
    # the dead channel in the middle the reference's float32 csvdc returns sigma ~ 2e-8 > 1e-8 and inverts that rounding
    # noise -- 1/sigma ~ 4e7 -- instead of thresholding it; the engine's sigma is exactly 0 for every position.)
    R[3] = R[2]; R[3][:, 0] = 0; R[3][0, :] = 0
    Xf = rng.normal(size=(N, max(N // 2, 2))) + 1j * rng.normal(size=(N, max(N // 2, 2)))
    R[4] = (Xf @ Xf.conj().T / Xf.shape[1] + 1e-3 * np.eye(N)).astype(np.complex64)   # few frames, barely loaded
    R[5] = -R[2]                                                            # negative definite: pinv = inverse
    R[0] = R[2]
    W, nident = eng.mvdr_weights(torch.from_numpy(R).to(dev), torch.from_numpy(d).to(dev))
    W = W.cpu().numpy()
    assert nident == 1                                                      # only the dead-channel bin ends with the identity
    assert np.allclose(W[0], 1.0)
    for k in range(1, K):
        ref = _oracle_mvdr_bin(orc, R[k].astype(np.complex128), d[k].astype(np.complex128))
        cond = np.linalg.cond(R[k].astype(np.complex128)) if k != 3 else 1.0
        assert np.linalg.norm(W[k] - ref) <= (2e-6 * cond + 1e-5) * np.linalg.norm(ref), (k, cond)
    assert np.allclose(W[3], d[3] / (N * np.vdot(d[3], d[3])), atol=1e-6)   # invR = I


@pytest.mark.parametrize("N", [8, 64])
def test_dead_channel_in_the_middle_of_the_array_is_a_stated_deviation(orc, dev, N):
    """DESIGN.md 3.7's one stated deviation of the MVDR design, pinned.  A dead channel (zero row and column of R) at the END of
    the array leaves LINPACK's Householder steps an exact zero: sigma = 0 < threshold, pseudoinverse() returns false, identity --
    reference and engine agree (test above).  In the MIDDLE of the array the reference's float32 csvdc returns sigma ~ 2e-8, just
    above the 1e-8 threshold, pseudoinverse() returns TRUE and the reference inverts that rounding noise (1 / sigma ~ 5e7,
    beamformer/beamformer.cc:262-270); the engine's solve finds the exact zero and takes the identity.  The engine's answer is the
    one a user wants; the reference's is an accident of rounding -- kept as a deviation, not reproduced (it would take csvdc's
    singular VECTORS in float32 rounding order, not only its values)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N)
    X = rng.normal(size=(N, 3 * N)) + 1j * rng.normal(size=(N, 3 * N))
    R0 = (X @ X.conj().T / (3 * N) + 0.01 * np.eye(N)).astype(np.complex64)
    d = (np.exp(-2j * np.pi * rng.uniform(size=(3, N))) / N).astype(np.complex64)
    R = np.stack([R0, R0, R0]).copy()
    mid = N // 2
    R[1][:, mid] = 0; R[1][mid, :] = 0                                      # dead channel in the middle
    R[2][:, N - 1] = 0; R[2][N - 1, :] = 0                                  # ... at the end
    # the reference (oracle pseudoinverse == the compiled csvdc): middle -> "ok", inverse of rounding noise; end -> false
    inv_mid, ok_mid = orc.pseudoinverse(R[1].astype(np.complex128), 1.0e-8)
    inv_end, ok_end = orc.pseudoinverse(R[2].astype(np.complex128), 1.0e-8)
    assert ok_mid and not ok_end
    assert np.max(np.abs(inv_mid)) > 1e6                                    # 1 / sigma of the dead direction
    W, nident = eng.mvdr_weights(torch.from_numpy(R).to(dev), torch.from_numpy(d).to(dev))
    W = W.cpu().numpy()
    assert nident == 2                                                      # the engine: identity for BOTH dead-channel bins
    for k in (1, 2):
        assert np.allclose(W[k], d[k] / (N * np.vdot(d[k], d[k])), atol=1e-6)
    # what the reference would have used for the middle case is a different vector altogether
    w_ref = _oracle_mvdr_bin(orc, R[1].astype(np.complex128), d[1].astype(np.complex128))
    assert np.linalg.norm(W[1] - w_ref) > 0.1 * np.linalg.norm(w_ref)


def _pinv_fallback_both(eng, R, d, threshold, first_bin=0):
    """(W_gpu, nident_gpu, ms_gpu), (W_host, nident_host): every bin flagged, the batched GPU Jacobi solve against the
    bin-by-bin host solve (btk_mvdr_pinv_fallback_host, the round-2 product path, itself pinned by the compiled csvdc)"""
    import ctypes as C
    import torch
    from distant_speech_recognition_amd import _lib
    L = _lib.lib()
    K, N, _ = R.shape
    flags = torch.ones(K, dtype=torch.int32, device=R.device)
    Wg = torch.zeros((K, N), dtype=torch.complex64, device=R.device)
    Wh = torch.zeros_like(Wg)
    cnt = torch.zeros(2, dtype=torch.int32, device=R.device)          # [identity, not converged]
    sb = L.btk_mvdr_pinv_scratch_bytes(K, N)
    scratch = torch.empty(max(sb, 16), dtype=torch.uint8, device=R.device)
    st = torch.cuda.current_stream().cuda_stream
    args = (R.data_ptr(), d.data_ptr(), Wg.data_ptr(), K, N, first_bin, threshold, flags.data_ptr(), cnt.data_ptr(),
            scratch.data_ptr() if sb else None, st)
    _lib.check(L.btk_mvdr_pinv_fallback_async(*args))                      # warm-up (module load, attribute)
    torch.cuda.synchronize()
    cnt.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(L.btk_mvdr_pinv_fallback_async(*args))
    e1.record()
    torch.cuda.synchronize()
    ni = C.c_int(0)
    _lib.check(L.btk_mvdr_pinv_fallback_host(R.data_ptr(), d.data_ptr(), Wh.data_ptr(), K, N, first_bin, threshold,
                                             flags.data_ptr(), C.byref(ni), st))
    assert int(cnt[1].item()) == 0                                         # every Jacobi solve converged
    return (Wg.cpu().numpy(), int(cnt[0].item()), e0.elapsed_time(e1)), (Wh.cpu().numpy(), ni.value)


def test_mvdr_pinv_gpu_all_bins_rank_deficient_n64(orc, dev):
    """an SMI-MVDR covariance from fewer than N frames (unit_test/confs/smimvdr.json, barely loaded) sends
    ALL K = 513 bins of a 64-mic array to the pseudo-inverse.  The batched GPU Jacobi solve does them in < 50 ms (the host
    loop needed 15 ms per bin = 8 s) and agrees with the host solve and, on sampled bins, with the oracle's compiled-csvdc path."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    N, K, T = 64, 513, 40
    rng = np.random.default_rng(64)
    X = (rng.normal(size=(K, N, T)) + 1j * rng.normal(size=(K, N, T))) * 1000.0
    R = (np.einsum("knt,kmt->knm", X, X.conj()) / T)
    R[:, np.arange(N), np.arange(N)] += 1.0e-4 * np.trace(R, axis1=1, axis2=2).real[:, None] / N     # mu = 1e-4 loading
    R[7] = 0.0                                                              # an empty bin: sigma = 0 -> identity
    R[9][:, 0] = 0; R[9][0, :] = 0                                          # a dead channel: identity
    R = R.astype(np.complex64)
    d = (np.exp(-2j * np.pi * rng.uniform(size=(K, N))) / N).astype(np.complex64)
    Rd, dd = torch.from_numpy(R).to(dev), torch.from_numpy(d).to(dev)
    (Wg, nig, ms), (Wh, nih) = _pinv_fallback_both(eng, Rd, dd, 1.0e-8)
    assert ms < 50.0, ms
    assert nig == nih == 2
    assert np.allclose(Wg[0], 0.0) and np.allclose(Wh[0], 0.0)              # global bin 0 is not touched (all-ones set by the solve kernel)
    for k in range(1, K):
        assert np.linalg.norm(Wg[k] - Wh[k]) <= 2e-6 * np.linalg.norm(Wh[k]), k
    for k in (7, 9):
        assert np.allclose(Wg[k], d[k] / (N * np.vdot(d[k], d[k])), atol=1e-6)       # invR = I
    for k in (1, 2, 256, 512):
        ref, info = _oracle_mvdr_bin(orc, R[k].astype(np.complex128), d[k].astype(np.complex128), with_info=True)
        if info == 0:                 # (this entry is the pseudo-inverse solve itself; the INFO != 0 rule is btk_mvdr_linpack_rule's)
            cond = np.linalg.cond(R[k].astype(np.complex128))
            assert np.linalg.norm(Wg[k] - ref) <= (2e-6 * cond + 1e-5) * np.linalg.norm(ref), (k, cond)
    # and through the product entry: the Cholesky kernel takes what it can, the rest goes to the GPU fall-back
    W, nident = eng.mvdr_weights(Rd, dd)
    W = W.cpu().numpy()
    assert nident == 2 and np.allclose(W[0], 1.0)
    for k in range(1, K):
        assert np.linalg.norm(W[k] - Wh[k]) <= (2e-6 * np.linalg.cond(R[k].astype(np.complex128)) + 1e-5) * np.linalg.norm(Wh[k]) or k in (7, 9), k


@pytest.mark.parametrize("N,K", [(1, 3), (2, 4), (3, 5), (8, 9), (33, 5), (65, 4), (100, 3), (256, 2)])
def test_mvdr_pinv_gpu_sizes(dev, N, K):
    """every storage form of the GPU pseudo-inverse solve (LDS up to N = 64, global scratch above; odd N: the tournament's dummy
    player) against the host solve: indefinite, positive definite, rank-deficient and zero matrices"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(1000 + N)
    R = np.zeros((K, N, N), np.complex64)
    for k in range(K):
        H = rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N))
        if k % 3 == 0:
            R[k] = ((H + H.conj().T) / 2).astype(np.complex64)               # indefinite
        elif k % 3 == 1:
            R[k] = (H @ H.conj().T / N + 0.01 * np.eye(N)).astype(np.complex64)
        else:
            Hr = H[:, : max(N // 2, 1)]
            R[k] = (Hr @ Hr.conj().T).astype(np.complex64) if N > 1 else 0   # exactly rank deficient in float64; float32 rounding noise decides
    d = (np.exp(-2j * np.pi * rng.uniform(size=(K, N))) / N).astype(np.complex64)
    (Wg, nig, ms), (Wh, nih) = _pinv_fallback_both(eng, torch.from_numpy(R).to(dev), torch.from_numpy(d).to(dev), 1.0e-8, first_bin=1)
    assert nig == nih
    for k in range(K):
        cond = min(np.linalg.cond(R[k].astype(np.complex128)), 1e7) if N > 1 and np.any(R[k]) else 1.0
        assert np.all(np.isfinite(Wg[k]))
        assert np.linalg.norm(Wg[k] - Wh[k]) <= (1e-9 * cond + 2e-6) * np.linalg.norm(Wh[k]), (k, cond)


def test_mvdr_divide_nondiagonal(orc, dev):
    """divide_all_nondiagonal_elements (beamformer.h:357-362) on the diffuse model, then MVDR == the oracle's pinned path"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    N, M, mu = 8, 64, 0.01
    K = M // 2 + 1
    mpos = ula_positions(N, 20.0)
    wq = orc.calc_mainlobe(M, N, 16000, la_delays(mpos, 0.4))
    Rd = eng.mvdr_divide_nondiagonal(eng.mvdr_diffuse_model(mpos, M, 16000, device=dev), mu)
    Rref = orc.diffuse_noise_model(mpos, M, 16000)
    off = ~np.eye(N, dtype=bool)
    Rref[:, off] = Rref[:, off] / (1.0 + mu)
    assert np.max(np.abs(Rd.cpu().numpy()[:K] - Rref[:K])) < 2e-6
    W, nident = eng.mvdr_weights(Rd, torch.from_numpy(wq[:K].astype(np.complex64)).to(dev))
    assert nident == 0
    for k in range(1, K):
        ref = _oracle_mvdr_bin(orc, Rref[k], wq[k])
        assert np.linalg.norm(W[k].cpu().numpy() - ref) <= 3e-3 * np.linalg.norm(ref)


def test_mvdr_identity_fallback(dev):
    """A singular R (no loading) trips the threshold and falls back to invR = I like the reference."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    N, K = 4, 3
    R = torch.ones((K, N, N), dtype=torch.complex64, device=dev)          # rank one
    d = torch.full((K, N), 0.25 + 0.0j, dtype=torch.complex64, device=dev)
    W, nfb = eng.mvdr_weights(R, d, threshold=1e-6)
    assert nfb == K - 1
    # invR = I: w = d / (N d^H d) = 0.25 / (4 * 0.25) = 0.25
    assert torch.allclose(W[1:], torch.full((K - 1, N), 0.25 + 0.0j, dtype=torch.complex64, device=dev), atol=1e-6)


@pytest.mark.parametrize("N", [137, 160, 255, 256, 257, 271, 272])
def test_mvdr_register_resident_solver_sizes(dev, N):
    """round 4: 136 < N <= 271 run on the register-resident Cholesky (csrc/chol_reg.h: 16 x 16 tiles in the matrix cores' accumulators,
    the right-hand side as an extra matrix row) -- sizes around its tile-row boundaries (N + 1 = 256, 257, 272) and the first size that
    falls back to the panel solver (272), random Hermitian positive definite R, plus the distortionless answer.  The expected value
    here is a numpy float64 SOLVE of the float32-rounded system (an analysis of the solver's accuracy), not the oracle: the oracle's
    pseudoinverse() path (the reference's compiled float32 csvdc) covers N <= 140 in test_mvdr_weights_match_oracle and N = 256 in
    tests/test_gpu_linpack_rule.py / tests/test_gpu_configs.py; svd_rule "exact" keeps the rule out of a test of the solver."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    K = 4
    rng = np.random.default_rng(N)
    A = rng.normal(size=(K, N, N + 40)) + 1j * rng.normal(size=(K, N, N + 40))
    R = (A @ A.conj().transpose(0, 2, 1)) / (N + 40) + 0.05 * np.eye(N)
    d = (rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))) / N
    W, nfb = eng.mvdr_weights(torch.from_numpy(R.astype(np.complex64)).to(dev), torch.from_numpy(d.astype(np.complex64)).to(dev), svd_rule="exact")
    W = W.cpu().numpy()
    assert nfb == 0
    assert np.allclose(W[0], 1.0)                                  # bin 0 is the reference's all-ones vector
    R32 = R.astype(np.complex64).astype(np.complex128)
    for k in range(1, K):
        z = np.linalg.solve(R32[k], d[k].astype(np.complex64).astype(np.complex128))
        exact = z / (N * np.vdot(d[k], z))
        assert np.linalg.norm(W[k] - exact) <= 1e-3 * np.linalg.norm(exact), (N, k)
        assert abs(np.vdot(W[k], d[k]) - 1.0 / N) < 1e-3 / N + 1e-5


def test_mvdr_register_resident_solver_every_size(dev):
    """every N the register-resident solver takes (137 .. 271: every padding count of the last tile row, every number of tile rows from 9
    to 17), one random positive definite system each, against a numpy float64 solve of the float32-rounded matrix (solver accuracy; the
    oracle comparison of the design as a whole is in test_mvdr_weights_match_oracle and tests/test_gpu_linpack_rule.py)"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(2026)
    worst = 0.0
    for N in range(137, 272):
        A = rng.normal(size=(N, N + 24)) + 1j * rng.normal(size=(N, N + 24))
        R = np.stack([np.eye(N), (A @ A.conj().T) / (N + 24) + 0.1 * np.eye(N)]).astype(np.complex64)
        d = ((rng.normal(size=(2, N)) + 1j * rng.normal(size=(2, N))) / N).astype(np.complex64)
        W, nfb = eng.mvdr_weights(torch.from_numpy(R).to(dev), torch.from_numpy(d).to(dev), svd_rule="exact")
        assert nfb == 0, N
        z = np.linalg.solve(R[1].astype(np.complex128), d[1].astype(np.complex128))
        exact = z / (N * np.vdot(d[1].astype(np.complex128), z))
        err = np.linalg.norm(W[1].cpu().numpy() - exact) / np.linalg.norm(exact)
        worst = max(worst, err)
        assert err <= 1e-3, (N, err)
    assert worst < 1e-3


@pytest.mark.parametrize("N", [200, 256])
def test_mvdr_register_resident_solver_flags_singular_bins(dev, N):
    """a rank-deficient bin trips the pivot threshold inside the register-resident solver and takes the reference's identity answer
    (beamformer.cc:2381-2383); its well-conditioned neighbours are solved"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    K = 4
    rng = np.random.default_rng(7 * N)
    A = rng.normal(size=(K, N, N + 16)) + 1j * rng.normal(size=(K, N, N + 16))
    R = (A @ A.conj().transpose(0, 2, 1)) / (N + 16) + 0.05 * np.eye(N)
    B = rng.normal(size=(N, 5)) + 1j * rng.normal(size=(N, 5))
    R[2] = B @ B.conj().T                                           # rank 5
    d = (rng.normal(size=(K, N)) + 1j * rng.normal(size=(K, N))) / N
    W, nfb = eng.mvdr_weights(torch.from_numpy(R.astype(np.complex64)).to(dev), torch.from_numpy(d.astype(np.complex64)).to(dev), threshold=1e-6)
    W = W.cpu().numpy()
    assert nfb == 1
    ident = d[2] / (N * np.vdot(d[2], d[2]))
    assert np.linalg.norm(W[2] - ident) <= 1e-5 * np.linalg.norm(ident)
    for k in (1, 3):
        z = np.linalg.solve(R[k], d[k])
        exact = z / (N * np.vdot(d[k], z))
        assert np.linalg.norm(W[k] - exact) <= 1e-3 * np.linalg.norm(exact)


@pytest.mark.parametrize("N", [8, 64, 160])
def test_mvdr_weights_of_stacked_streams_equal_the_per_stream_design(dev, N):
    """btk_mvdr_weights_streams (engine.mvdr_weights with R [S][K][N][N]): S designs in one launch, bit for bit what S calls give -- every
    stream's bin 0 is the all-ones vector, a singular bin of one stream takes the identity rule without touching its neighbours"""
    import torch
    from distant_speech_recognition_amd import engine as eng
    S, K = 3, 5
    rng = np.random.default_rng(100 + N)
    A = rng.normal(size=(S, K, N, N + 8)) + 1j * rng.normal(size=(S, K, N, N + 8))
    R = (A @ A.conj().transpose(0, 1, 3, 2)) / (N + 8) + 0.05 * np.eye(N)
    B = rng.normal(size=(N, 2)) + 1j * rng.normal(size=(N, 2))
    R[1, 3] = B @ B.conj().T                                        # rank 2: flagged, identity answer
    d = (rng.normal(size=(S, K, N)) + 1j * rng.normal(size=(S, K, N))) / N
    Rt = torch.from_numpy(R.astype(np.complex64)).to(dev)
    dt = torch.from_numpy(d.astype(np.complex64)).to(dev)
    W, nfb = eng.mvdr_weights(Rt, dt, threshold=1e-6)
    assert W.shape == (S, K, N) and nfb == 1
    for s in range(S):
        Ws, n1 = eng.mvdr_weights(Rt[s].contiguous(), dt[s].contiguous(), threshold=1e-6)
        assert n1 == (1 if s == 1 else 0)
        assert torch.equal(W[s], Ws), s
        assert torch.all(W[s, 0] == 1.0)


def test_mvdr_lds_kernel_still_matches_the_oracle_where_the_register_solver_took_over(dev):
    """round 4 moved 64 <= N <= 136 to the register-resident solver (4-8 x faster there); the LDS kernel stays selectable
    (BTK_MVDR_REG_MIN, read once per process) and runs the same oracle tests in a child process"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env["BTK_MVDR_REG_MIN"] = "1000"
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "mvdr_weights_match_oracle or mvdr_pinv or mvdr_identity or stacked_streams"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-500:])
    assert " passed" in r.stdout


def _coherent_snapshots(rng, S, K, N, T):
    X = _rand_snapshots(rng, S, K, N, T)
    X += (rng.normal(size=(S, K, 1, T)) + 1j * rng.normal(size=(S, K, 1, T))).astype(np.complex64) * 2500.0
    return X


@pytest.mark.parametrize("N,M,T,type_,minf,load", [(4, 256, 120, 2, 0, 0.01), (8, 64, 70, 1, 0, 0.01), (64, 64, 40, 2, 5, 0.0),
                                                  (5, 128, 90, 10, 0, 0.05), (20, 64, 50, 2, 0, 0.01), (32, 64, 45, 2, 0, 0.01), (16, 64, 60, 2, 0, 0.01),
                                                  (100, 32, 30, 2, 0, 0.01), (12, 64, 40, 2, 0, 0.01)])
def test_mccowan_matches_oracle(orc, dev, N, M, T, type_, minf, load):
    """McCowanPostFilter (postfilter.cc:798-935) over a delay-and-sum beamformer, diffuse-noise coherence
    (confs/sd_and_mccowan.json shape), two consecutive blocks."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N + M + T)
    K, S = M // 2 + 1, 2
    X = _coherent_snapshots(rng, S, K, N, T)
    mpos = ula_positions(N, 40.0)
    delays = la_delays(mpos, 0.4)
    wq = orc.calc_mainlobe(M, N, 16000, delays)
    w = eng.weights_gsc_effective(wq, None, M)
    d = wq[:K].astype(np.complex64)
    R = eng.mvdr_diffuse_model(mpos, M, 16000.0, device=dev)
    if load > 0:
        eng.mvdr_diagonal_loading(R, load)
    st = eng.CoherencePostFilterState(S, K, N, dev)
    st.set_coherence(R, 0.99)
    Xd = torch.from_numpy(X).to(dev)
    Wd, Dd = torch.from_numpy(w).to(dev), torch.from_numpy(d).to(dev)
    T1 = T // 3
    Y = torch.cat([eng.bf_apply_mccowan(Wd, Dd, Xd[..., :T1].contiguous(), st, alpha=0.7, type_=type_, min_frames=minf),
                   eng.bf_apply_mccowan(Wd, Dd, Xd[..., T1:].contiguous(), st, alpha=0.7, type_=type_, min_frames=minf)],
                  dim=-1).cpu().numpy()
    Ro = orc.diffuse_noise_model(mpos, M, 16000.0)
    if load > 0:
        Ro = orc.diagonal_loading(Ro, M, load)
    for s in range(S):
        Xo = _full(X[s], M)
        ref, Wref = orc.mccowan_frames(Xo, orc.gsc_frames(Xo, wq), wq, Ro, alpha=0.7, type_=type_, min_frames=minf)
        # stated tolerance: post-filter recurrences <= 1e-4 relative
        assert np.max(np.abs(Y[s].T - ref[:, :K])) <= 1e-4 * np.max(np.abs(ref))
        wl = st.w_last.cpu().numpy()[s]
        assert np.max(np.abs(wl - Wref[-1, :K].real)) <= 1e-4
    assert 1e-3 < np.mean(Wref.real[:, :K]) < 0.999


@pytest.mark.parametrize("N,M,T,type_,minf,x1", [(4, 256, 120, 2, 0, 100), (8, 64, 70, 1, 0, 10), (33, 64, 40, 2, 3, 0),
                                                (6, 128, 90, 2, 0, 200), (64, 64, 40, 2, 0, 5), (32, 64, 50, 2, 2, 5),
                                                (100, 32, 25, 2, 0, 3), (20, 64, 40, 2, 0, 5)])
def test_lefkimmiatis_matches_oracle(orc, dev, N, M, T, type_, minf, x1):
    """LefkimmiatisPostFilter (postfilter.cc:967-1190), confs/sd_and_lefkimmiatis.json shape (alpha 0.8, min_sv 1e-4)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(N * 5 + M + T)
    K, S = M // 2 + 1, 2
    X = _coherent_snapshots(rng, S, K, N, T)
    mpos = ula_positions(N, 40.0)
    delays = la_delays(mpos, -0.8)
    wq = orc.calc_mainlobe(M, N, 16000, delays)
    w = eng.weights_gsc_effective(wq, None, M)
    d = wq[:K].astype(np.complex64)
    R = eng.mvdr_diffuse_model(mpos, M, 16000.0, device=dev)
    eng.mvdr_diagonal_loading(R, 0.1)
    st = eng.CoherencePostFilterState(S, K, N, dev, lefkimmiatis=True)
    st.set_coherence(R, 0.99)
    Dd = torch.from_numpy(d).to(dev)
    Ro = orc.diagonal_loading(orc.diffuse_noise_model(mpos, M, 16000.0), M, 0.1)
    # bins whose pseudoinverse() returns false in the reference (csvdc INFO != 0 at N = 100): identity, Lambda = d^H d (:971-977)
    nfalse = sum(not orc.pseudoinverse(Ro[k], 1.0e-4)[1] for k in range(K))
    assert st.set_lambda(R, Dd, 1.0e-4) == nfalse and (N > 64 or nfalse == 0)
    Xd = torch.from_numpy(X).to(dev)
    Wd = torch.from_numpy(w).to(dev)
    T1 = T // 2
    kw = dict(fbin_x1=x1, alpha=0.8, type_=type_, min_frames=minf)
    Y = torch.cat([eng.bf_apply_lefkimmiatis(Wd, Dd, Xd[..., :T1].contiguous(), st, **kw),
                   eng.bf_apply_lefkimmiatis(Wd, Dd, Xd[..., T1:].contiguous(), st, **kw)], dim=-1).cpu().numpy()
    for s in range(S):
        Xo = _full(X[s], M)
        ref, Wref = orc.lefkimmiatis_frames(Xo, orc.gsc_frames(Xo, wq), wq, Ro, min_sv=1.0e-4, fbin_x1=x1, alpha=0.8,
                                            type_=type_, min_frames=minf)
        # Lambda goes through the reference's float32 SVD vs a float32 Cholesky here: gains agree to ~1e-3 like MVDR weights
        assert np.max(np.abs(Y[s].T - ref[:, :K])) <= 1e-3 * np.max(np.abs(ref))
        wl = st.w_last.cpu().numpy()[s]
        assert np.max(np.abs(wl - Wref[-1, :K].real)) <= 1e-3
    assert 1.5e-4 < np.mean(Wref.real[:, :K]) < 0.999          # neither clamped to the floor nor to 1 everywhere


def test_cov_accumulate_on_row_padded_snapshots(dev):
    """cov_accumulate takes row-padded snapshots (engine.padded_rows) and per-frame weights of the same row stride: identical
    to the contiguous result for the MFMA, VALU and small-N kernels."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    for N in (4, 64):
        S, K, T = 2, 5, 512                                   # 512 frames x 8 B = 4 KiB rows -> padded
        g = torch.Generator(device=dev).manual_seed(N)
        Xc = (torch.randn((S, K, N, T), device=dev, generator=g) + 1j * torch.randn((S, K, N, T), device=dev, generator=g)).to(torch.complex64)
        Xp = eng.padded_rows((S, K, N, T), torch.complex64, dev)
        Xp.copy_(Xc)
        assert not Xp.is_contiguous()
        for mf in (True, False):
            assert torch.equal(eng.cov_accumulate(Xc, use_mfma=mf), eng.cov_accumulate(Xp, use_mfma=mf))
        fw = torch.rand((S, T), device=dev, generator=g)
        fwp = eng.rows_like(Xp, (S, T), dtype=torch.float32); fwp.copy_(fw)
        assert torch.equal(eng.cov_accumulate(Xc, frame_weights=fw), eng.cov_accumulate(Xp, frame_weights=fwp))
        with pytest.raises(Exception):
            eng.cov_accumulate(Xp, frame_weights=fw)             # contiguous weights cannot share the padded T_stride


@pytest.mark.parametrize("N,T,nq,cplx,pad", [(64, 37, 1, False, False), (64, 530, 2, False, True), (32, 100, 2, False, False),
                                             (64, 75, 2, True, True), (32, 16, 1, True, False), (20, 50, 2, True, False),
                                             (16, 40, 2, True, False), (48, 100, 1, False, True), (48, 33, 2, True, False),
                                             # round 3: any channel count on the matrix cores (padded blocks, > 64 channels in one or two passes)
                                             (8, 70, 2, True, False), (12, 33, 1, False, True), (21, 50, 2, True, True), (63, 40, 2, False, False),
                                             (100, 45, 1, True, False), (100, 36, 2, False, True), (128, 20, 2, True, False), (7, 60, 2, True, False),
                                             (130, 18, 1, False, False)])
def test_stats2_quadratic_forms_against_their_definition(dev, N, T, nq, cplx, pad):
    """btk_bf_apply_stats2 (the per-frame sums behind McCowan / Lefkimmiatis, postfilter.cc:798-829, 1041-1077) against
    u_t = sum_{i<=j} Cs[j][i] x'_i conj(x'_j) evaluated in float64: the matrix-core kernel (8 <= N <= 128, any N: real and complex
    pair weights, one and two forms, ragged tiles, row-padded snapshots, channel counts padded to 16) and the VALU kernel (N < 8, N > 128)."""
    import torch
    from distant_speech_recognition_amd import _lib, engine as eng
    rng = np.random.default_rng(N * 7 + T + nq)
    S, K = 2, 5
    Xh = ((rng.normal(size=(S, K, N, T)) + 1j * rng.normal(size=(S, K, N, T))) * 1500.0).astype(np.complex64)
    Xh += ((rng.normal(size=(S, K, 1, T)) + 1j * rng.normal(size=(S, K, 1, T))) * 2500.0).astype(np.complex64)
    if pad:
        X = torch.zeros((S, K, N, T + 48), dtype=torch.complex64, device=dev)[..., :T]      # rows as engine.padded_rows lays them out
        X.copy_(torch.from_numpy(Xh))
    else:
        X = torch.from_numpy(Xh).to(dev)
    Ts = X.stride(2)
    W = (rng.normal(size=(1, K, N)) + 1j * rng.normal(size=(1, K, N))).astype(np.complex64) / N
    D = np.exp(1j * rng.uniform(0, 2 * np.pi, size=(1, K, N))).astype(np.complex64)
    # pair weights as btk_pf_coherence_coeffs lays them out: row j holds i <= j, zeros right of the diagonal
    C = [np.tril(rng.normal(size=(K, N, N)) + (1j * rng.normal(size=(K, N, N)) if cplx else 0)).astype(np.complex64) for _ in range(nq)]
    Wd, Dd = torch.from_numpy(W).to(dev), torch.from_numpy(D).to(dev)
    Cd = [torch.from_numpy(c).to(dev) for c in C]
    Y = torch.zeros((S, K, Ts), dtype=torch.complex64, device=dev)
    U = torch.zeros_like(Y)
    V = torch.zeros_like(Y) if nq == 2 else None
    E = torch.zeros((S, K, Ts), dtype=torch.float32, device=dev)
    p = lambda t: None if t is None else t.data_ptr()
    _lib.check(_lib.lib().btk_bf_apply_stats2(p(Wd), p(Dd), 0, p(X), p(Y), p(Cd[0]), p(Cd[1]) if nq == 2 else None, p(U), p(V), p(E),
                                              S, K, N, Ts, T, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    X64 = Xh.astype(np.complex128)
    xp = np.conj(D[0].astype(np.complex128))[None, :, :, None] * X64                         # x' = conj(d) x
    yref = np.einsum("kn,sknt->skt", np.conj(W[0].astype(np.complex128)), X64)
    eref = np.sum(np.abs(xp) ** 2, axis=2)
    assert np.max(np.abs(Y.cpu().numpy()[..., :T] - yref)) <= 2e-6 * np.sqrt(N) * np.max(np.abs(yref))
    assert np.max(np.abs(E.cpu().numpy()[..., :T] - eref)) <= 1e-5 * np.max(eref)
    for got, c in zip((U, V), C):
        L = np.tril(c.astype(np.complex128))                                                 # row j holds i <= j
        ref = np.einsum("skjt,kji,skit->skt", np.conj(xp), L, xp)
        # fp32 inner products of N terms of magnitude |C| |x'|^2, float64 outer sums
        scale = np.max(np.einsum("skjt,kji,skit->skt", np.abs(xp), np.abs(L), np.abs(xp)))
        assert np.max(np.abs(got.cpu().numpy()[..., :T] - ref)) <= 1e-6 * scale
    if Ts > T:
        assert float(torch.abs(U[..., T:]).max()) == 0.0                                     # nothing written past the last frame


@pytest.mark.parametrize("N", [8, 64])
def test_coherence_postfilters_on_row_padded_snapshots(dev, N):
    """bf_apply_mccowan / bf_apply_lefkimmiatis take row-padded snapshots (engine.padded_rows, analysis(pad_rows=True)): the same
    numbers as on contiguous rows, Y sharing the row stride of X."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    rng = np.random.default_rng(99 + N)
    S, M, T = 2, 64, 512                                     # 512 frames x 8 B = 4 KiB rows: padded_rows pads them
    K = M // 2 + 1
    Xh = _coherent_snapshots(rng, S, K, N, T)
    Xc = torch.from_numpy(Xh).to(dev)
    Xp = eng.padded_rows((S, K, N, T), torch.complex64, dev)
    assert Xp.stride(2) > T
    Xp.copy_(Xc)
    mpos = ula_positions(N, 40.0)
    d = torch.from_numpy(np.exp(-2j * np.pi * rng.uniform(size=(K, N))).astype(np.complex64) / N).to(dev)
    R = eng.mvdr_diffuse_model(mpos, M, 16000.0, device=dev)
    eng.mvdr_diagonal_loading(R, 0.05)
    for lef in (False, True):
        outs = []
        for X in (Xc, Xp):
            st = eng.CoherencePostFilterState(S, K, N, dev, lefkimmiatis=lef)
            st.set_coherence(R, 0.99)
            if lef:
                st.set_lambda(R, d, 1.0e-4)
                outs.append(eng.bf_apply_lefkimmiatis(d, d, X, st, fbin_x1=3, alpha=0.8))
            else:
                outs.append(eng.bf_apply_mccowan(d, d, X, st, alpha=0.7))
        assert outs[1].stride(1) == Xp.stride(2) and outs[0].is_contiguous()
        assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("N", [96, 200])
def test_mvdr_pinv_gpu_threshold_close_calls(dev, N):
    """The Gauss-Jordan + power-iteration form of the pseudo-inverse rule (N > 70) on matrices whose smallest singular value sits
    just under / just above the threshold, alone and hidden in a cluster of values slightly above it (the iterate approaches
    1 / sigma_min^2 from below: a loose stop would over-estimate sigma_min): identity exactly where sigma_min < threshold."""
    import torch
    rng = np.random.default_rng(N)
    thr = 1.0e-3
    K = 7
    Q, _ = np.linalg.qr(rng.normal(size=(N, N)) + 1j * rng.normal(size=(N, N)))
    base = np.linspace(0.5, 3.0, N)
    base[0] = -1.0                                                    # indefinite: the Cholesky solve never takes these bins
    want = []
    R = np.zeros((K, N, N), np.complex128)
    for k, (smin, cluster) in enumerate([(None, 0), (0.8 * thr, 0), (1.25 * thr, 0), (0.8 * thr, 12), (1.25 * thr, 12), (0.97 * thr, 3), (1.03 * thr, 3)]):
        sv = base.copy()
        if smin is not None:
            sv[-1] = smin
            sv[-1 - cluster:-1] = thr * np.linspace(1.3, 1.6, cluster) if cluster else sv[-1 - cluster:-1]
        R[k] = (Q * sv) @ Q.conj().T
        want.append(smin is not None and smin < thr)
    from distant_speech_recognition_amd import engine as eng
    d = (np.exp(-2j * np.pi * rng.uniform(size=(K, N))) / N).astype(np.complex64)
    (Wg, nig, _), _ = _pinv_fallback_both(eng, torch.from_numpy(R.astype(np.complex64)).to(dev), torch.from_numpy(d).to(dev), thr, first_bin=1)
    got = [bool(np.allclose(Wg[k], d[k] / (N * np.vdot(d[k], d[k])), atol=1e-7 / N)) for k in range(K)]
    assert got == want, (got, want)
    assert nig == sum(want)
