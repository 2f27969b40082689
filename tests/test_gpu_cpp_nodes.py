"""GPU: the C++ node layer (distant_speech_recognition_amd/host) driven like the reference's
src/beamformerDS.cc:144-223 -- SampleFeature -> OverSampledDFTAnalysisBank xN -> SubbandGSC
(-> ZelinskiPostFilter) -> OverSampledDFTSynthesisBank, pulled until jiterator_error."""
import os
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "beamformer_ds")
MPOS = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
M, m, r, FS = 256, 4, 1, 16000


@pytest.mark.parametrize("pf,alpha,kind", [(0, 0.0, ""), (2, 0.7, ""), (2, 0.7, "mccowan"), (2, 0.8, "lefkimmiatis"),
                                           (0, 0.0, "gscrls")])
def test_beamformer_ds_binary_matches_oracle(orc, dev, tmp_path, proto256, kinect_pcm, pf, alpha, kind):
    from tests.util import la_delays
    assert os.path.exists(EXE), "build the host layer: make -C distant_speech_recognition_amd/host"
    h, g = proto256
    coeffs = str(tmp_path / "coeffs.f64")
    np.concatenate([h, g]).astype(np.float64).tofile(coeffs)
    delays = la_delays(MPOS, -1.306379)
    L = 30000
    args = [EXE, coeffs, str(M), str(m), str(r), str(pf), str(alpha), str(tmp_path / "out.f32")]
    for c in range(4):
        p = str(tmp_path / ("c%d.wav" % c))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:L].astype(np.int16).tobytes())
        w.close()
        args += [repr(float(delays[c])), p]
    env = dict(os.environ)
    if kind in ("mccowan", "lefkimmiatis"):
        env["BTK_EXAMPLE_PF"] = kind
        env["BTK_EXAMPLE_MPOS"] = ";".join(",".join(repr(float(v)) for v in row) for row in MPOS)
    if kind == "gscrls":
        env["BTK_EXAMPLE_BF"] = "gscrls"
    res = subprocess.run(args, capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    out = np.fromfile(str(tmp_path / "out.f32"), np.float32)
    # oracle: the C++ mains use delayCompensationType = 0 (src/beamformerDS.cc:155,172)
    X = np.stack([orc.analysis(h, M, m, r, 0, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    wq, B, wl = orc.gsc_weights(M, 4, FS, delays)
    tol = 0.5                                          # <= 0.5 LSB at int16 scale
    if kind == "gscrls":
        o = orc.RLSCc(M, 4, delays, FS, mu=0.97, sigma2=0.001)
        o.init_precision_matrix(1.0e6)
        o.set_quadratic_constraint(0.1, 2)
        Y = o.run(X)
        tol = 1e-4 * np.max(np.abs(kinect_pcm[:, :L])) + 0.5      # recurrences: 1e-4 relative
    else:
        Y = orc.gsc_frames(X, wq, wl)
    if kind == "mccowan":
        R = orc.diagonal_loading(orc.diffuse_noise_model(MPOS, M, FS), M, 0.01)
        Y, _ = orc.mccowan_frames(X, Y, wq, R, alpha=alpha, type_=pf)
    elif kind == "lefkimmiatis":
        R = orc.diagonal_loading(orc.diffuse_noise_model(MPOS, M, FS), M, 0.1)
        Y, _ = orc.lefkimmiatis_frames(X, Y, wq, R, min_sv=1e-4, fbin_x1=100, alpha=alpha, type_=pf)
        tol = 2e-3 * np.max(np.abs(kinect_pcm[:, :L])) + 0.5      # Lambda: float32 SVD vs float32 Cholesky (like MVDR)
    elif pf:
        Y, _ = orc.zelinski_frames(X, Y, wq, alpha, pf)
    ref = orc.synthesis(g, M, m, r, 0, Y)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < tol


@pytest.mark.parametrize("M_,m_,r_", [(512, 4, 2), (128, 2, 0), (1024, 4, 1)])
def test_beamformer_ds_binary_other_geometries(orc, dev, tmp_path, kinect_pcm, M_, m_, r_):
    """the C++ nodes are geometry-agnostic: other FFT lengths / decimations through the same example binary"""
    from tests.util import la_delays, design_prototype
    h, g = design_prototype(M_, m_), design_prototype(M_, m_, "g")
    coeffs = str(tmp_path / "coeffs.f64")
    np.concatenate([h, g]).astype(np.float64).tofile(coeffs)
    delays = la_delays(MPOS, 0.5)
    L = 20000
    args = [EXE, coeffs, str(M_), str(m_), str(r_), "2", "0.7", str(tmp_path / "out.f32")]
    for c in range(4):
        p = str(tmp_path / ("c%d.wav" % c))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:L].astype(np.int16).tobytes())
        w.close()
        args += [repr(float(delays[c])), p]
    res = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    out = np.fromfile(str(tmp_path / "out.f32"), np.float32)
    X = np.stack([orc.analysis(h, M_, m_, r_, 0, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    wq, B, wl = orc.gsc_weights(M_, 4, FS, delays)
    Y, _ = orc.zelinski_frames(X, orc.gsc_frames(X, wq, wl), wq, 0.7, 2)
    ref = orc.synthesis(g, M_, m_, r_, 0, Y)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 1e-4 * np.max(np.abs(ref)) + 0.5


EXE_MVDRGSC = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "beamformer_mvdrgsc")


def _write_inputs(tmp_path, proto256, kinect_pcm, L, delays):
    h, g = proto256
    coeffs = str(tmp_path / "coeffs.f64")
    np.concatenate([h, g]).astype(np.float64).tofile(coeffs)
    chan_args = []
    for c in range(4):
        p = str(tmp_path / ("c%d.wav" % c))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:L].astype(np.int16).tobytes())
        w.close()
        chan_args += [repr(float(delays[c])), p]
    return coeffs, chan_args


def _mvdrgsc_oracle(orc, X, delays, bm):
    K = M // 2 + 1
    wq_ds = orc.calc_mainlobe(M, 4, FS, delays)
    R = orc.diagonal_loading(orc.diffuse_noise_model(MPOS, M, FS), M, 0.01)
    wfull = np.zeros((M, 4), np.complex128)
    wfull[:K] = orc.mvdr_weights(R, wq_ds, M)
    base = wq_ds if bm == 1 else wfull
    B = np.zeros((M, 4, 3), np.complex128)
    for k in range(M):
        if bm == 1 or 1 <= k <= M // 2:
            B[k] = orc.blocking_matrix(base[k], 1)
    wl = np.zeros((M, 4), np.complex128)
    for k in range(1, M // 2 + 1):
        i = np.arange(3)
        wl[k] = orc.sidelobe_canceller(B[k], 0.05 * (np.cos(0.37 * k + i) + 1j * np.sin(0.11 * k * (i + 1))))
    return wq_ds, wfull, wl, B


@pytest.mark.parametrize("bm", [1, 2])
def test_beamformer_mvdrgsc_binary_matches_oracle(orc, dev, tmp_path, proto256, kinect_pcm, bm):
    """C++ SubbandMVDRGSC node (reference beamformer.h:385-437): MVDR quiescent + GSC lower branch with the blocking
    matrix of calc_blocking_matrix1 / calc_blocking_matrix2, through the synthesis bank."""
    from tests.util import la_delays
    assert os.path.exists(EXE_MVDRGSC), "build the host layer: make -C distant_speech_recognition_amd/host"
    h, g = proto256
    delays = la_delays(MPOS, -1.306379)
    L = 30000
    coeffs, chan_args = _write_inputs(tmp_path, proto256, kinect_pcm, L, delays)
    mpos = ";".join(",".join(repr(float(v)) for v in row) for row in MPOS)
    env = dict(os.environ, BTK_EXAMPLE_BM=str(bm))
    res = subprocess.run([EXE_MVDRGSC, coeffs, str(M), str(m), str(r), "0.01", str(tmp_path / "out.f32"), mpos] + chan_args,
                         capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    assert "0 identity fall-backs" in res.stderr
    out = np.fromfile(str(tmp_path / "out.f32"), np.float32)
    X = np.stack([orc.analysis(h, M, m, r, 0, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    wq_ds, wfull, wl, B = _mvdrgsc_oracle(orc, X, delays, bm)
    ref = orc.synthesis(g, M, m, r, 0, orc.gsc_frames(X, wfull, wl))
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 2e-3 * np.max(np.abs(kinect_pcm[:, :L])) + 0.5      # float32 SVD vs Cholesky MVDR


def test_mvdrgsc_blocking_matrix_output_and_upgrade(orc, dev, tmp_path, proto256, kinect_pcm):
    """blocking_matrix_output(0) frame by frame after upgrade_blocking_matrix(): b_0^H x with B_k orthogonal to wq_k - wl_k"""
    from tests.util import la_delays
    h, g = proto256
    delays = la_delays(MPOS, -1.306379)
    L = 12000
    coeffs, chan_args = _write_inputs(tmp_path, proto256, kinect_pcm, L, delays)
    mpos = ";".join(",".join(repr(float(v)) for v in row) for row in MPOS)
    env = dict(os.environ, BTK_EXAMPLE_BM="1", BTK_EXAMPLE_UPGRADE="1", BTK_EXAMPLE_BMOUT=str(tmp_path / "bo.f64"))
    res = subprocess.run([EXE_MVDRGSC, coeffs, str(M), str(m), str(r), "0.01", str(tmp_path / "out.f32"), mpos] + chan_args,
                         capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    K = M // 2 + 1
    bo = np.fromfile(str(tmp_path / "bo.f64"), np.float64).reshape(-1, K, 2)
    bo = bo[..., 0] + 1j * bo[..., 1]
    X = np.stack([orc.analysis(h, M, m, r, 0, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    wq_ds, wfull, wl, B = _mvdrgsc_oracle(orc, X, delays, 1)
    assert bo.shape[0] == X.shape[0]
    for k in range(1, K):
        B[k] = orc.blocking_matrix(wq_ds[k] - wl[k], 1)
    want = np.stack([[np.vdot(B[k][:, 0], X[t, :, k]) for k in range(K)] for t in range(X.shape[0])])
    assert np.max(np.abs(bo - want)) < 1e-5 * np.max(np.abs(X))


EXE_WPE = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "subband_dereverberator")


@pytest.mark.parametrize("nchan,lower,upper,bias", [(2, 0, 7, 1e-4), (1, 1, 12, 0.0), (3, 2, 5, 1e-4)])
def test_subband_dereverberator_binary_matches_oracle(orc, dev, tmp_path, proto256, kinect_pcm, nchan, lower, upper, bias):
    """C++ WPE nodes (reference dereverberation.h:31-190) in the flow of unit_test/test_subband_dereverberator.py:
    estimate_filter() on the utterance, then lock-step pull of the dereverberated channels through synthesis banks."""
    assert os.path.exists(EXE_WPE), "build the host layer: make -C distant_speech_recognition_amd/host"
    h, g = proto256
    L = 40000
    coeffs, chan_args = _write_inputs(tmp_path, proto256, kinect_pcm, L, np.zeros(4))
    wavs = chan_args[1::2][:nchan]
    prefix = str(tmp_path / "dereverb")
    res = subprocess.run([EXE_WPE, coeffs, str(M), str(m), str(r), str(lower), str(upper), "2", "-18.0", repr(bias), prefix] + wavs,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    X = np.stack([orc.analysis(h, M, m, r, 0, kinect_pcm[c][:L]) for c in range(nchan)], axis=1)
    assert "%d frames used for the estimate" % X.shape[0] in res.stderr
    G = orc.wpe_estimate(X, lower, upper, 2, -18.0, 0.0, bias)
    Yd = orc.wpe_apply(X, G, lower, upper)
    for c in range(nchan):
        out = np.fromfile(prefix + ".c%d.f32" % c, np.float32)
        ref = orc.synthesis(g, M, m, r, 0, Yd[:, c])
        assert out.shape == ref.shape
        assert np.max(np.abs(out - ref)) < 1e-3 * np.max(np.abs(ref)) + 0.5


EXE_SD = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "beamformer_sd")


@pytest.mark.parametrize("NC,pf,alpha", [(1, 2, 0.7), (1, 0, 0.0), (2, 2, 0.6), (3, 0, 0.0)])
def test_beamformer_sd_binary_matches_oracle(orc, dev, tmp_path, proto256, kinect_pcm, NC, pf, alpha):
    """The reference's superdirective main (src/superdirectiveBeamformer.cc:151-248) through the C++ nodes and the legacy
    camelCase API it uses: SubbandMVDR with setDiffuseNoiseModel + divideAllNonDiagonalElements(0.01) + calcMVDRWeights,
    ZelinskiPostFilter::setBeamformer, synthesis; LCMV quiescent vectors (calcArrayManifoldVectors2 / N) for NC > 1."""
    from tests.util import la_delays
    assert os.path.exists(EXE_SD), "build the host layer: make -C distant_speech_recognition_amd/host"
    h, g = proto256
    delays = la_delays(MPOS, -1.306379)
    L = 30000
    mu = 0.01
    coeffs, chan_args = _write_inputs(tmp_path, proto256, kinect_pcm, L, delays)
    mpos = ";".join(",".join(repr(float(v)) for v in row) for row in MPOS)
    env = dict(os.environ, BTK_EXAMPLE_NC=str(NC))
    res = subprocess.run([EXE_SD, coeffs, str(M), str(m), str(r), str(pf), str(alpha), str(mu), str(tmp_path / "out.f32"), mpos] + chan_args,
                         capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stderr
    assert "0 identity fall-backs" in res.stderr
    out = np.fromfile(str(tmp_path / "out.f32"), np.float32)
    K = M // 2 + 1
    X = np.stack([orc.analysis(h, M, m, r, 0, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    ta = orc.calc_mainlobe(M, 4, FS, delays)
    if NC == 1:
        wq = ta
    elif NC == 2:
        wq = orc.calc_mainlobe_2(M, 4, FS, delays, -0.5 * delays)
    else:
        wq = orc.calc_mainlobe_n(M, 4, FS, delays, np.stack([-0.5 * delays, 0.25 * delays]), 3)
    R = orc.diffuse_noise_model(MPOS, M, FS)
    off = ~np.eye(4, dtype=bool)
    R[:, off] = R[:, off] / (1.0 + mu)
    got_r = float(res.stderr.split("R_10[0][1] = ")[1].split()[0])               # getNoiseSpatialSpectralMatrix(10)
    assert abs(got_r - R[10, 0, 1].real) < 2e-6
    wfull = np.zeros((M, 4), np.complex128)
    wfull[:K] = orc.mvdr_weights(R, wq, M)
    Y = orc.gsc_frames(X, wfull, np.zeros((M, 4), np.complex128))
    if pf:
        Y, _ = orc.zelinski_frames(X, Y, ta, alpha, pf)
    ref = orc.synthesis(g, M, m, r, 0, Y)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 2e-3 * np.max(np.abs(kinect_pcm[:, :L])) + 0.5      # float32 SVD vs Cholesky MVDR


def test_write_fir_coeff_matches_reference_formula(orc, dev, tmp_path, proto256, kinect_pcm):
    """SubbandGSC::writeFIRCoeff (beamformer.cc:775-828): per channel the windowed real part of the inverse DFT of
    e^{j pi (k+1)} conj(wq_k - wl_k), Hermitian-extended; getBlockingMatrix returns the N x (N-1) matrix of a bin."""
    from tests.util import la_delays
    delays = la_delays(MPOS, -1.306379)
    coeffs, chan_args = _write_inputs(tmp_path, proto256, kinect_pcm, 4000, delays)
    mpos = ";".join(",".join(repr(float(v)) for v in row) for row in MPOS)
    fir = str(tmp_path / "fir.txt")
    res = subprocess.run([EXE_SD, coeffs, str(M), str(m), str(r), "0", "0", "0", str(tmp_path / "out.f32"), mpos] + chan_args,
                         capture_output=True, text=True, timeout=120, env=dict(os.environ, BTK_EXAMPLE_FIR=fir))
    assert res.returncode == 0, res.stderr
    assert "blocking matrix of bin 5 is 4 x 3" in res.stderr
    lines = open(fir).read().strip().split("\n")
    assert lines[0].split() == ["4", str(M)]
    got = np.array([[float(v) for v in l.split()] for l in lines[1:]])
    wq, B, _ = orc.gsc_weights(M, 4, FS, delays)
    wl = np.zeros((M, 4), np.complex128)
    i = np.arange(3)
    for k in range(1, M // 2 + 1):
        wl[k] = orc.sidelobe_canceller(B[k], 0.05 * (np.cos(0.37 * k + i) + 1j * np.sin(0.11 * k * (i + 1))))
    win = 0.54 - 0.46 * np.cos(2.0 * np.pi / (M - 1) * np.arange(M))          # winType 1 -> Hamming (modulated.cc:62-67)
    want = np.zeros((4, M))
    for c in range(4):
        val = np.zeros(M, np.complex128)
        for k in range(M // 2 + 1):
            v = np.exp(1j * np.pi * (k + 1)) * np.conj(wq[k, c] - wl[k, c])
            val[k] = v
            if 0 < k < M // 2:
                val[M - k] = np.conj(v)
        want[c] = win * np.real(np.fft.ifft(val))                              # gsl_fft_complex_radix2_inverse: e^{+j}, 1/M
    assert got.shape == want.shape and np.max(np.abs(got - want)) < 1e-6 * np.max(np.abs(want)) + 1e-12


def test_moving_look_direction_through_cpp_nodes(orc, dev, proto256, kinect_pcm, tmp_path):
    """The look direction changes between two output blocks (unit_test/test_online_beamforming.py:209-226) with C++ nodes only:
    frames the synthesis bank of a per-frame graph had already pulled (pd + block + 1, modulated.cc:574-578) keep the old weights,
    later frames use the new ones -- the BlockSource protocol between the engine's batching nodes (modulated/modulated.h)."""
    import wave
    from distant_speech_recognition_amd import btk20cpp as B
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    M, m, r, D, FS = 256, 4, 1, 128, 16000
    MPOS = [[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]]
    h, g = proto256
    L = 40000

    def graph(with_pf):
        afbs = []
        for c in range(4):
            p = str(tmp_path / ("m%d.wav" % c))
            w = wave.open(p, "wb")
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
            w.writeframes(kinect_pcm[c][:L].astype(np.int16).tobytes())
            w.close()
            sf = B.SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
            sf.read(p, FS)
            afbs.append(B.OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2))
        bf = B.SubbandDSPtr(fftlen=M)
        for a in afbs:
            bf.set_channel(a)
        top = bf
        if with_pf:
            top = B.ZelinskiPostFilterPtr(bf, M, 0.7, 2)
            top.set_beamformer(bf)
        return bf, B.OverSampledDFTSynthesisBankPtr(top, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)

    dA = calc_delays("linear", MPOS, [-1.306379, None, None])
    dB = calc_delays("linear", MPOS, [0.4, None, None])
    b_switch, pd_syn = 60, 4
    bf, sfb = graph(False)
    bf.calc_array_manifold_vectors(FS, dA)
    blocks = []
    for b in range(b_switch + 1):
        blocks.append(np.array(sfb.next()))
    bf.calc_array_manifold_vectors(FS, dB)                              # between block b_switch and b_switch + 1
    while True:
        try:
            blocks.append(np.array(sfb.next()))
        except StopIteration:
            break
    out = np.concatenate(blocks)
    X = np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    wa, wb = orc.calc_mainlobe(M, 4, FS, dA), orc.calc_mainlobe(M, 4, FS, dB)
    Ya, Yb = orc.gsc_frames(X, wa, np.zeros_like(wa)), orc.gsc_frames(X, wb, np.zeros_like(wb))
    n_old = pd_syn + b_switch + 1
    ref = orc.synthesis(g, M, m, r, 2, np.concatenate([Ya[:n_old], Yb[n_old:]]))
    assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 0.5
    assert np.max(np.abs(orc.synthesis(g, M, m, r, 2, Ya) - ref)) > 5.0          # the switch is visible in the expectation itself
    # with a post-filter between beamformer and synthesis: the blocks before the switch equal the static run bit for bit
    bf2, sfb2 = graph(True)
    bf2.calc_array_manifold_vectors(FS, dA)
    stat = np.concatenate([np.array(v) for v in sfb2])
    bf3, sfb3 = graph(True)
    bf3.calc_array_manifold_vectors(FS, dA)
    mv = [np.array(sfb3.next()) for _ in range(b_switch + 1)]
    bf3.calc_array_manifold_vectors(FS, dB)
    while True:
        try:
            mv.append(np.array(sfb3.next()))
        except StopIteration:
            break
    mv = np.concatenate(mv)
    nb = (b_switch + 1) * D
    assert mv.shape == stat.shape and np.array_equal(mv[:nb], stat[:nb]) and np.max(np.abs(mv[nb:] - stat[nb:])) > 5.0
