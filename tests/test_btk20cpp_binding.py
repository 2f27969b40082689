"""The pybind11 binding of the C++ node layer (distant_speech_recognition_amd.btk20cpp): the reference's SWIG semantics --
numpy views of the node's vector (include/vector.i:290-305), iterator protocol with StopIteration at jiterator_error,
Python objects as source nodes (stream/pyStream.h:25-168), the j_error family -- and, on the GPU, the reference's D&S +
Zelinski flow (unit_test/test_online_beamforming.py:51-116) through C++ nodes only."""
import wave

import numpy as np
import pytest

M, m, r, D, FS = 256, 4, 1, 128, 16000
MPOS = [[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]]      # confs/ds.json:2-5
AZIMUTH = -1.306379


class _Source(object):
    """what the reference's Python algorithm classes look like to a C++ node: size(), __iter__, next(), reset()"""

    def __init__(self, frames):
        self.frames, self.i, self.resets = frames, 0, 0

    def size(self):
        return self.frames.shape[1]

    def __iter__(self):
        return self

    def next(self):
        if self.i >= len(self.frames):
            raise StopIteration
        self.i += 1
        return self.frames[self.i - 1]

    __next__ = next

    def reset(self):
        self.i = 0
        self.resets += 1


def test_python_objects_source_cpp_nodes_and_views_are_zero_copy():
    from distant_speech_recognition_amd import btk20cpp as B
    fr = np.arange(12).reshape(3, 4) + 1j * np.arange(12).reshape(3, 4)[::-1]
    src = _Source(fr)
    s = B.PyVectorComplexFeatureStreamPtr(src)
    assert s.size() == 4 and s.frame_no() == -1
    a = s.next()
    assert np.array_equal(a, fr[0]) and not a.flags.owndata and a.base is s          # a view of the node's vector_
    assert np.shares_memory(a, s.next(0)) and s.frame_no() == 0                      # same frame -> same buffer, no pull
    assert np.array_equal(s.current(), fr[0])
    frames = [np.array(v) for v in s]                                                # __iter__ = reset() + self
    assert src.resets == 1 and len(frames) == 3 and np.array_equal(frames[2], fr[2])
    with pytest.raises(StopIteration):
        s.next()
    assert s.is_end()
    # float streams and the block reader
    f = B.PyVectorFloatFeatureStreamPtr(_Source(np.arange(6, dtype=np.float32).reshape(2, 3)))
    assert [list(np.array(v)) for v in f] == [[0.0, 1.0, 2.0], [3.0, 4.0, 5.0]]
    sf = B.SampleFeaturePtr(block_len=4, shift_len=4, pad_zeros=True)
    sf.set_samples(np.arange(10, dtype=np.float32))
    assert [list(np.array(v)) for v in sf] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 0, 0]]
    # a source that returns too short a vector is a dimension error, not a crash
    bad = B.PyVectorComplexFeatureStreamPtr(_Source(np.zeros((1, 4), np.complex128)))
    bad_src = _Source(np.zeros((1, 2), np.complex128)); bad_src.size = lambda: 4
    with pytest.raises(B.jdimension_error):
        B.PyVectorComplexFeatureStreamPtr(bad_src).next()
    assert bad.size() == 4


def test_exception_family_and_host_side_methods():
    from distant_speech_recognition_amd import btk20cpp as B
    from distant_speech_recognition_amd import engine
    with pytest.raises(B.jallocation_error) as e:
        B.SubbandMVDRPtr(fftlen=64, half_band_shift=True)                            # beamformer.cc:2283-2285
    assert isinstance(e.value, B.j_error) and "halfBandShift" in str(e.value)
    ds = B.SubbandDSPtr(fftlen=64)
    with pytest.raises(B.j_error):
        ds.next()                                                                    # no weights yet
    # weights are host code: the node's quiescent vectors are the C-ABI's
    chans = [B.PyVectorComplexFeatureStreamPtr(_Source(np.zeros((1, 64), np.complex128))) for _ in range(4)]
    for c in chans:
        ds.set_channel(c)
    assert ds.chan_num() == 4 and ds.fftlen() == 64
    delays = np.array([0.0, 1e-4, 2.5e-4, -1e-4])
    ds.calc_array_manifold_vectors(16000.0, delays)
    wq = engine.weights_mainlobe(64, 4, 16000.0, delays)
    assert np.max(np.abs(np.stack([ds.get_weights(k) for k in range(64)]) - wq)) < 1e-15
    arr = B.SpectralMatrixArrayPtr(8, 3, 0.9)
    x = np.arange(8) + 1j
    for c in range(3):
        arr.set_samples(x * (c + 1), c)
    arr.update()
    assert np.allclose(arr.matrix_f(2), (1 - float(np.float32(0.9))) * np.outer(x[2] * np.arange(1, 4), x[2] * np.arange(1, 4)))
    # no noise matrix yet: None like the reference's NULL R_[fbinX] (the binding once dereferenced it)
    assert B.SubbandMVDRPtr(fftlen=64).noise_spatial_spectral_matrix(3) is None


@pytest.mark.gpu
def test_ds_zelinski_flow_through_cpp_nodes(orc, dev, proto256, kinect_pcm, tmp_path):
    from distant_speech_recognition_amd import btk20cpp as B
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, g = proto256
    afbs = []
    for c in range(4):
        p = str(tmp_path / ("c%d.wav" % c))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:40000].astype(np.int16).tobytes())
        w.close()
        sf = B.SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.read(p, FS)
        afbs.append(B.OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2))
    bf = B.SubbandGSCPtr(fftlen=M, half_band_shift=False)
    for a in afbs:
        bf.set_channel(a)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    bf.calc_gsc_weights(FS, delays)
    pf = B.ZelinskiPostFilterPtr(bf, M, 0.7, 2)                                        # confs/ds_and_zelinski.json
    pf.set_beamformer(bf)
    sfb = B.OverSampledDFTSynthesisBankPtr(pf, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
    out = np.concatenate([np.array(buf) for buf in sfb])
    X = np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:40000]) for c in range(4)], axis=1)
    wq = orc.calc_mainlobe(M, 4, FS, delays)
    Yf, _ = orc.zelinski_frames(X, orc.gsc_frames(X, wq, np.zeros_like(wq)), wq, 0.7, 2)
    ref = orc.synthesis(g, M, m, r, 2, Yf)
    assert out.shape == ref.shape == (313 * D,)
    assert np.max(np.abs(out - ref)) < 0.5                                             # <= 0.5 LSB at int16 scale
    with pytest.raises(StopIteration):
        sfb.next()
    # a Python object feeding a C++ node: the oracle's beamformed frames through the C++ synthesis bank
    Yb = orc.gsc_frames(X, wq, np.zeros_like(wq))
    py_src = B.PyVectorComplexFeatureStreamPtr(_Source(Yb))
    sfb2 = B.OverSampledDFTSynthesisBankPtr(py_src, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
    out2 = np.concatenate([np.array(buf) for buf in sfb2])
    ref2 = orc.synthesis(g, M, m, r, 2, Yb)
    assert out2.shape == ref2.shape and np.max(np.abs(out2 - ref2)) < 0.5


@pytest.mark.gpu
def test_noise_matrix_accessors_are_range_checked(dev):
    """set / get / load / divide of one bin's noise matrix: bins beyond M/2 are jindex_error, not device memory beyond the
    allocation (advisor finding, round 2); unset matrices read as None"""
    from distant_speech_recognition_amd import btk20cpp as B
    Mh, N = 64, 4
    mv = B.SubbandMVDRPtr(fftlen=Mh)
    for _ in range(N):
        mv.set_channel(B.PyVectorComplexFeatureStreamPtr(_Source(np.zeros((1, Mh), np.complex128))))
    R = np.eye(N, dtype=np.complex128) * 2.0
    assert mv.set_noise_spatial_spectral_matrix(Mh // 2, R)
    assert np.allclose(mv.noise_spatial_spectral_matrix(Mh // 2), R) and np.allclose(mv.noise_spatial_spectral_matrix(1), 0.0)
    mv.set_diagonal_looading(Mh // 2, 0.5)
    mv.divide_nondiagonal_elements(Mh // 2, 1.0)
    assert np.allclose(mv.noise_spatial_spectral_matrix(Mh // 2), np.eye(N) * 2.5)
    for call in (lambda: mv.noise_spatial_spectral_matrix(Mh // 2 + 1), lambda: mv.set_diagonal_looading(Mh, 0.1),
                 lambda: mv.divide_nondiagonal_elements(Mh // 2 + 1, 0.1), lambda: mv.set_noise_spatial_spectral_matrix(Mh, R)):
        with pytest.raises(B.jindex_error):
            call()
    src = B.PyVectorComplexFeatureStreamPtr(_Source(np.zeros((1, Mh), np.complex128)))
    pf = B.McCowanPostFilterPtr(src, Mh)
    assert pf.noise_spatial_spectral_matrix(0) is None
    assert pf.set_noise_spatial_spectral_matrix(3, R)
    assert np.allclose(pf.noise_spatial_spectral_matrix(3), R)
    for call in (lambda: pf.noise_spatial_spectral_matrix(Mh // 2 + 1), lambda: pf.set_diagonal_looading(Mh, 0.1),
                 lambda: pf.divide_nondiagonal_elements(Mh, 0.1), lambda: pf.set_noise_spatial_spectral_matrix(Mh // 2 + 1, R)):
        with pytest.raises(B.jindex_error):
            call()
    # the ENABLE_LEGACY_BTK_API spellings of the same calls (reference postfilter/postfilter.h:140-148)
    R2 = R + 0.5 * (np.ones((N, N)) - np.eye(N))
    assert pf.setNoiseSpatialSpectralMatrix(5, R2) and np.allclose(pf.getNoiseSpatialSpectralMatrix(5), R2)
    pf.setLevelOfDiagonalLoading(5, 0.25)
    pf.divideNonDiagonalElements(5, 1.0)
    assert np.allclose(pf.getNoiseSpatialSpectralMatrix(5), np.eye(N) * 2.25 + 0.25 * (np.ones((N, N)) - np.eye(N)))
    pf.setAllLevelsOfDiagonalLoading(0.5)
    pf.divideAllNonDiagonalElements(0.0)
    assert np.allclose(pf.getNoiseSpatialSpectralMatrix(3), R + 0.5 * np.eye(N))


@pytest.mark.gpu
def test_half_band_shift_through_cpp_nodes(orc, dev, proto256, kinect_pcm, tmp_path):
    """halfBandShift == true in the C++ node layer (SubbandDS / SubbandGSC with one constraint, reference beamformer.cc:515-527,
    1113-1128, 1276-1285): analysis-bank channels (conjugate-mirrored upper bins) and generic sources (all M bins as supplied);
    SubbandMVDR refuses it in the constructor, SubbandGSCRLS in next(), post-filters when bound."""
    from distant_speech_recognition_amd import btk20cpp as B
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, _ = proto256
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])

    def banks():
        afbs = []
        for c in range(4):
            p = str(tmp_path / ("h%d.wav" % c))
            w = wave.open(p, "wb")
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
            w.writeframes(kinect_pcm[c][:40000].astype(np.int16).tobytes())
            w.close()
            sf = B.SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
            sf.read(p, FS)
            afbs.append(B.OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2))
        return afbs
    X = np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:40000]) for c in range(4)], axis=1)      # [T][N][M]
    wq = orc.calc_mainlobe_halfband(M, 4, FS, delays)
    ds = B.SubbandDSPtr(fftlen=M, half_band_shift=True)
    assert ds.is_half_band_shift()
    for a in banks():
        ds.set_channel(a)
    ds.calc_array_manifold_vectors(FS, delays)
    assert np.max(np.abs(np.stack([ds.get_weights(k) for k in range(M)]) - wq)) < 1e-15
    out = np.stack([np.array(v) for v in ds])
    ref = orc.gsc_frames_halfband(X, wq, np.zeros_like(wq))
    assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 2e-5 * np.max(np.abs(ref))
    gsc = B.SubbandGSCPtr(fftlen=M, half_band_shift=True)
    for a in banks():
        gsc.set_channel(a)
    gsc.calc_gsc_weights(FS, delays)
    gsc.normalize_weight(True)
    rng = np.random.default_rng(5)
    wl = np.zeros_like(wq)
    for k in range(M):
        wa = (rng.standard_normal(3) + 1j * rng.standard_normal(3)) * 0.05
        packed = np.empty(6); packed[0::2] = wa.real; packed[1::2] = wa.imag
        gsc.set_active_weights_f(k, packed)
        wl[k] = orc.blocking_matrix(wq[k], 1) @ wa
    out = np.stack([np.array(v) for v in gsc])
    ref = orc.gsc_frames_halfband(X, wq, wl, normalize=True)
    assert np.max(np.abs(out - ref)) < 2e-5 * np.max(np.abs(ref))
    # generic sources: no symmetry between the bins, every bin as supplied
    Mh, T = 64, 29
    Xg = (rng.standard_normal((T, 4, Mh)) + 1j * rng.standard_normal((T, 4, Mh))) * 1000.0
    wqh = orc.calc_mainlobe_halfband(Mh, 4, FS, delays)
    g2 = B.SubbandGSCPtr(fftlen=Mh, half_band_shift=True)
    for n in range(4):
        g2.set_channel(B.PyVectorComplexFeatureStreamPtr(_Source(Xg[:, n, :])))
    g2.calc_gsc_weights(FS, delays)
    out = np.stack([np.array(v) for v in g2])
    ref = orc.gsc_frames_halfband(Xg, wqh, np.zeros_like(wqh))
    assert out.shape == ref.shape == (T, Mh) and np.max(np.abs(out - ref)) < 2e-5 * np.max(np.abs(ref))
    # the classes that refuse it
    with pytest.raises(B.jallocation_error):
        B.SubbandMVDRPtr(fftlen=M, half_band_shift=True)
    rls = B.SubbandGSCRLSPtr(fftlen=M, half_band_shift=True, mu=0.9)
    for a in banks():
        rls.set_channel(a)
    rls.calc_gsc_weights(FS, delays)
    rls.init_precision_matrix(0.01)
    with pytest.raises(B.j_error):
        rls.next()
    pf = B.ZelinskiPostFilterPtr(g2, Mh, 0.7, 2)
    pf.set_beamformer(g2)
    with pytest.raises(B.j_error):
        pf.next()


@pytest.mark.gpu
def test_cpp_gscrls_with_two_constraints(orc, dev):
    """C++ SubbandGSCRLS after calc_gsc_weights_2 (look direction + one null: the blocking matrix keeps N - 2 columns): the
    recursion of btk_rls_process_nc (mode 0, the further blocked direction from btk_nlms_constraint_vectors) against the oracle's
    frame-by-frame restatement of beamformer.cc:1514-1645 run on the same LCMV quiescent weights."""
    from distant_speech_recognition_amd import btk20cpp as B
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    Mh, N, T = 32, 6, 70
    K = Mh // 2 + 1
    rng = np.random.default_rng(321)
    X = (rng.standard_normal((T, N, Mh)) + 1j * rng.standard_normal((T, N, Mh))) * 0.5
    X[:, :, 0] = X[:, :, 0].real; X[:, :, Mh // 2] = X[:, :, Mh // 2].real
    X[:, :, K:] = np.conj(X[:, :, Mh // 2 - 1:0:-1])
    mpos = [[-100.0 + 40.0 * i, 0.0, 2.0] for i in range(N)]
    dT, dJ = calc_delays("linear", mpos, [-1.0, None, None]), calc_delays("linear", mpos, [0.5, None, None])
    rls = B.SubbandGSCRLSPtr(fftlen=Mh, half_band_shift=False, mu=0.9, sigma2=0.01)
    for n in range(N):
        rls.set_channel(B.PyVectorComplexFeatureStreamPtr(_Source(X[:, n, :])))
    rls.calc_gsc_weights_2(FS, dT, dJ)
    rls.init_precision_matrix(0.01)
    out = np.stack([np.array(v) for v in rls])
    o = orc.RLSCc(Mh, N, dT, FS, mu=0.9, sigma2=0.01, Nc=2)
    o.wq = orc.calc_mainlobe_2(Mh, N, FS, dT, dJ)                                   # the node's LCMV quiescent weights
    for k in range(Mh):
        o.B[k] = orc.blocking_matrix(o.wq[k], 2)
    o.init_precision_matrix(0.01)
    ref = o.run(X)
    assert out.shape == ref.shape
    assert np.max(np.abs(out[:, :K] - ref[:, :K])) <= 1e-4 * np.max(np.abs(ref))
