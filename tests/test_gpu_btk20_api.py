"""GPU: the reference's own integration flows (unit_test/test_online_beamforming.py:51-228,
unit_test/test_sos_batch_beamforming.py:95-233) written against the btk20 mirror, checked against
the oracle pull graph.  Geometry / parameters from unit_test/confs/{ds,ds_and_zelinski,sd,gsclms,smimvdr}.json."""
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MPOS = [[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]]      # confs/ds.json:2-5
AZIMUTH = -1.306379                                                                      # confs/ds.json:6
M, m, r, D, FS = 256, 4, 1, 128, 16000


@pytest.fixture(scope="module")
def wavs(tmp_path_factory, kinect_pcm):
    d = tmp_path_factory.mktemp("wav")
    paths = []
    for c in range(4):
        p = str(d / ("c%d.wav" % (c + 1)))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:40000].astype(np.int16).tobytes())
        w.close()
        paths.append(p)
    return paths


def _build(wavs, h):
    from distant_speech_recognition_amd.btk20 import SampleFeaturePtr, OverSampledDFTAnalysisBankPtr
    sample_feats, afbs = [], []
    for p in wavs:
        sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.read(p, FS)
        afbs.append(OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2))
        sample_feats.append(sf)
    return sample_feats, afbs


def _oracle_X(orc, h, kinect_pcm):
    return np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:40000]) for c in range(4)], axis=1)      # [T][N][M]


def test_delay_and_sum_with_zelinski_flow(orc, dev, proto256, kinect_pcm, wavs):
    from distant_speech_recognition_amd.btk20 import (PyVectorComplexFeatureStreamPtr, ZelinskiPostFilterPtr,
                                                      OverSampledDFTSynthesisBankPtr)
    from distant_speech_recognition_amd.pybeamformer import SubbandGSCBeamformer, calc_delays
    h, g = proto256
    _, afbs = _build(wavs, h)
    beamformer = SubbandGSCBeamformer(afbs, Nc=1)
    pybf = PyVectorComplexFeatureStreamPtr(beamformer)
    spatial_filter = ZelinskiPostFilterPtr(pybf, M, 0.7, 2)                       # confs/ds_and_zelinski.json
    sfb = OverSampledDFTSynthesisBankPtr(spatial_filter, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    beamformer.calc_beamformer_weights(FS, delays)
    spatial_filter.set_beamformer(beamformer.beamformer())
    out = np.concatenate([np.array(buf) for buf in sfb])
    # oracle: same graph frame by frame
    X = _oracle_X(orc, h, kinect_pcm)
    wq = orc.calc_mainlobe(M, 4, FS, delays)
    Yf, _ = orc.zelinski_frames(X, orc.gsc_frames(X, wq, np.zeros_like(wq)), wq, 0.7, 2)
    ref = orc.synthesis(g, M, m, r, 2, Yf)
    assert out.shape == ref.shape == (313 * D,)
    assert np.max(np.abs(out - ref)) < 0.5                     # <= 0.5 LSB at int16 scale
    # iterator protocol: exhausted stream raises StopIteration, same-frame caching
    with pytest.raises(StopIteration):
        sfb.next()


@pytest.mark.parametrize("kind", ["mccowan", "lefkimmiatis"])
def test_sd_with_coherence_postfilter_flow(orc, dev, proto256, kinect_pcm, wavs, kind):
    """unit_test/test_online_beamforming.py:118-154 with confs/sd_and_mccowan.json / sd_and_lefkimmiatis.json"""
    from distant_speech_recognition_amd.btk20 import (PyVectorComplexFeatureStreamPtr, McCowanPostFilterPtr,
                                                      LefkimmiatisPostFilterPtr, OverSampledDFTSynthesisBankPtr, j_error)
    from distant_speech_recognition_amd.pybeamformer import SubbandMVDRBeamformer, calc_delays
    h, g = proto256
    _, afbs = _build(wavs, h)
    beamformer = SubbandMVDRBeamformer(afbs)
    pybf = PyVectorComplexFeatureStreamPtr(beamformer)
    if kind == "mccowan":
        spatial_filter = McCowanPostFilterPtr(pybf, M, 0.7, 2)
    else:
        spatial_filter = LefkimmiatisPostFilterPtr(pybf, M, 1e-4, 100, 0.8, 2)
    with pytest.raises(j_error):
        spatial_filter.set_all_diagonal_loading(0.01)          # "Construct/set first a noise coherence matrix"
    spatial_filter.set_diffuse_noise_model(MPOS, FS, 343740.0)
    spatial_filter.set_all_diagonal_loading(0.01 if kind == "mccowan" else 0.1)
    if kind == "lefkimmiatis":
        spatial_filter.calc_inverse_noise_spatial_spectral_matrix()
    sfb = OverSampledDFTSynthesisBankPtr(spatial_filter, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    beamformer.calc_sd_beamformer_weights(FS, delays, MPOS, mu=0.01)
    spatial_filter.set_beamformer(beamformer.beamformer())
    out = np.concatenate([np.array(buf) for buf in sfb])
    X = _oracle_X(orc, h, kinect_pcm)
    wq = orc.calc_mainlobe(M, 4, FS, delays)
    Rbf = orc.diagonal_loading(orc.diffuse_noise_model(np.array(MPOS), M, FS), M, 0.01)
    Ybf = orc.mvdr_frames(X, orc.mvdr_weights(Rbf, wq, M))
    if kind == "mccowan":
        Rpf = orc.diagonal_loading(orc.diffuse_noise_model(np.array(MPOS), M, FS), M, 0.01)
        Yf, _ = orc.mccowan_frames(X, Ybf, wq, Rpf, alpha=0.7, type_=2)
    else:
        Rpf = orc.diagonal_loading(orc.diffuse_noise_model(np.array(MPOS), M, FS), M, 0.1)
        Yf, _ = orc.lefkimmiatis_frames(X, Ybf, wq, Rpf, min_sv=1e-4, fbin_x1=100, alpha=0.8, type_=2)
    ref = orc.synthesis(g, M, m, r, 2, Yf)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 2e-3 * np.max(np.abs(ref)) + 0.5


def test_node_semantics(dev, proto256, wavs):
    from distant_speech_recognition_amd.btk20 import SubbandGSCPtr, j_error, jdimension_error, jconsistency_error, \
        OverSampledDFTAnalysisBankPtr, SampleFeaturePtr
    h, _ = proto256
    _, afbs = _build(wavs, h)
    a0 = afbs[0]
    f0 = np.array(a0.next())
    assert a0.frame_no() == 0 and np.array_equal(np.array(a0.next(0)), f0)       # frame_no == frame_no_ -> cached
    f1 = np.array(a0.next())
    assert a0.frame_no() == 1 and not np.array_equal(f0, f1)
    assert np.array_equal(np.array(a0.current()), f1)
    bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
    for a in afbs:
        bf.set_channel(a)
    with pytest.raises(j_error):
        bf.next()                                                                # "call calc_gsc_weights_X() once"
    with pytest.raises(jdimension_error):
        bf.calc_gsc_weights(FS, np.zeros(3))
    with pytest.raises(jconsistency_error):
        OverSampledDFTAnalysisBankPtr(SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True), prototype=h[:-1], M=M, m=m, r=r)
    with pytest.raises(jdimension_error):
        OverSampledDFTAnalysisBankPtr(SampleFeaturePtr(block_len=64, shift_len=64, pad_zeros=True), prototype=h, M=M, m=m, r=r)
    # __iter__ = reset + self; frame count of the fixture slice: ceil(40000/128)=313 blocks -> 317 frames
    _, afbs2 = _build(wavs[:1], h)
    assert sum(1 for _ in afbs2[0]) == 317


def test_super_directive_flow(orc, dev, proto256, kinect_pcm, wavs):
    from distant_speech_recognition_amd.btk20 import PyVectorComplexFeatureStreamPtr, OverSampledDFTSynthesisBankPtr
    from distant_speech_recognition_amd.pybeamformer import SubbandMVDRBeamformer, calc_delays
    h, g = proto256
    _, afbs = _build(wavs, h)
    beamformer = SubbandMVDRBeamformer(afbs)
    sfb = OverSampledDFTSynthesisBankPtr(PyVectorComplexFeatureStreamPtr(beamformer), prototype=g, M=M, m=m, r=r,
                                         delay_compensation_type=2)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    beamformer.calc_sd_beamformer_weights(FS, delays, MPOS, mu=0.01)              # confs/sd.json
    out = np.concatenate([np.array(buf) for buf in sfb])
    X = _oracle_X(orc, h, kinect_pcm)
    wq = orc.calc_mainlobe(M, 4, FS, delays)
    R = orc.diagonal_loading(orc.diffuse_noise_model(np.array(MPOS), M, FS), M, 0.01)
    w = orc.mvdr_weights(R, wq, M)
    ref = orc.synthesis(g, M, m, r, 2, orc.mvdr_frames(X, w))
    assert out.shape == ref.shape
    # MVDR weights agree to ~1e-3 (float32 SVD in the reference vs float32 Cholesky): scale the PCM tolerance
    assert np.max(np.abs(out - ref)) < 2e-3 * np.max(np.abs(ref)) + 0.5


def test_gsclms_flow(orc, dev, proto256, kinect_pcm, wavs):
    from distant_speech_recognition_amd.btk20 import PyVectorComplexFeatureStreamPtr, OverSampledDFTSynthesisBankPtr
    from distant_speech_recognition_amd.pybeamformer import SubbandGSCLMSBeamformer, calc_delays
    h, g = proto256
    _, afbs = _build(wavs, h)
    beamformer = SubbandGSCLMSBeamformer(afbs, min_frames=32)                      # other values = confs/gsclms.json
    sfb = OverSampledDFTSynthesisBankPtr(PyVectorComplexFeatureStreamPtr(beamformer), prototype=g, M=M, m=m, r=r,
                                         delay_compensation_type=2)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    beamformer.calc_beamformer_weights(FS, delays)
    out = np.concatenate([np.array(buf) for buf in sfb])
    X = _oracle_X(orc, h, kinect_pcm)
    o = orc.NLMS(M, 4, min_frames=32)
    o.calc_beamformer_weights(FS, delays)
    ref = orc.synthesis(g, M, m, r, 2, o.run(X))
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 1e-4 * np.max(np.abs(ref)) + 0.5
    assert np.max(np.abs(beamformer._waH - o.wa())) < 2e-4


def test_gscrls_flow(orc, dev, proto256, kinect_pcm, wavs):
    """unit_test/test_online_beamforming.py with confs/gscrls.json (beamformer type "gscrls"): same script shape"""
    from distant_speech_recognition_amd.btk20 import PyVectorComplexFeatureStreamPtr, OverSampledDFTSynthesisBankPtr
    from distant_speech_recognition_amd.pybeamformer import SubbandGSCRLSBeamformer, calc_delays
    h, g = proto256
    _, afbs = _build(wavs, h)
    beamformer = SubbandGSCRLSBeamformer(afbs, min_frames=32)
    sfb = OverSampledDFTSynthesisBankPtr(PyVectorComplexFeatureStreamPtr(beamformer), prototype=g, M=M, m=m, r=r,
                                         delay_compensation_type=2)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    beamformer.calc_beamformer_weights(FS, delays)
    out = np.concatenate([np.array(buf) for buf in sfb])
    X = _oracle_X(orc, h, kinect_pcm)
    o = orc.RLSPy(M, 4, 1, min_frames=32)
    o.calc_beamformer_weights(FS, delays)
    ref = orc.synthesis(g, M, m, r, 2, o.run(X))
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 1e-4 * np.max(np.abs(ref)) + 0.5
    assert np.max(np.abs(beamformer._waH - o.waH)) < 2e-4 * max(1.0, np.max(np.abs(o.waH)))


def test_subband_gscrls_node(orc, dev, proto256, kinect_pcm, wavs):
    """C++-style node SubbandGSCRLS (beamformer.h:224-263): calc_gsc_weights -> init_precision_matrix -> next()"""
    from distant_speech_recognition_amd.btk20 import SubbandGSCRLSPtr, j_error
    h, g = proto256
    _, afbs = _build(wavs, h)
    bf = SubbandGSCRLSPtr(fftlen=M, half_band_shift=False, mu=0.97, sigma2=0.001)
    for a in afbs:
        bf.set_channel(a)
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    with pytest.raises(j_error):
        bf.init_precision_matrix(1.0e6)                 # "call calc_gsc_weights_x() once"
    bf.calc_gsc_weights(FS, delays)
    with pytest.raises(j_error):
        bf.next()                                       # precision matrix not set
    bf.init_precision_matrix(1.0e6)
    bf.set_quadratic_constraint(0.1, 2)
    frames = np.stack([np.array(f) for f in bf])
    X = _oracle_X(orc, h, kinect_pcm)
    o = orc.RLSCc(M, 4, delays, FS, mu=0.97, sigma2=0.001)
    o.init_precision_matrix(1.0e6)
    o.set_quadratic_constraint(0.1, 2)
    ref = o.run(X)
    assert frames.shape == ref.shape
    assert np.max(np.abs(frames - ref)) <= 1e-4 * np.max(np.abs(ref))
    assert np.max(np.abs(bf.beamformer_weight_object(0).wl[: M // 2 + 1] - o.wl[: M // 2 + 1])) <= 1e-4 * max(np.max(np.abs(o.wl)), 1e-30)


def test_smimvdr_batch_flow(orc, dev, proto256, kinect_pcm, wavs):
    from distant_speech_recognition_amd.btk20 import PyVectorComplexFeatureStreamPtr, OverSampledDFTSynthesisBankPtr
    from distant_speech_recognition_amd.pybeamformer import SubbandSMIMVDRBeamformer, calc_delays
    h, g = proto256
    sample_feats, afbs = _build(wavs, h)
    beamformer = SubbandSMIMVDRBeamformer(afbs, Nc=1)
    sfb = OverSampledDFTSynthesisBankPtr(PyVectorComplexFeatureStreamPtr(beamformer), prototype=g, M=M, m=m, r=r,
                                         delay_compensation_type=2)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    labs = [(1.0, 2.0)]
    beamformer.accu_stats_from_label(FS, target_labs=labs, energy_threshold=10)
    beamformer.finalize_stats()
    beamformer.calc_beamformer_weights(FS, delays, mu=1e-4)                        # confs/smimvdr.json
    for c, p in enumerate(wavs):                                                   # reload (test_sos_batch_beamforming.py:221-222)
        sample_feats[c].read(p, FS)
    out = np.concatenate([np.array(buf) for buf in sfb])
    # oracle
    X = _oracle_X(orc, h, kinect_pcm)
    T = X.shape[0]
    en = np.array([orc.frame_energy(X[t, 0]) for t in range(T)])
    el, labx, fw = 0.0, 0, []
    for t in range(T):
        tgt = False
        if labx < len(labs):
            if el >= labs[labx][0] and (el <= labs[labx][1] or labs[labx][1] < 0):
                tgt = True
            elif el > labs[labx][1]:
                labx += 1
        fw.append((not tgt) and en[t] > 10)
        el += D / float(FS)
    R = orc.cov_accumulate(X, frame_weights=fw) / sum(fw)
    wq = orc.calc_mainlobe(M, 4, FS, delays)
    w = orc.mvdr_weights(orc.diagonal_loading(R, M, 1e-4), wq, M)
    ref = orc.synthesis(g, M, m, r, 2, orc.mvdr_frames(X, w))
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 5e-3 * np.max(np.abs(ref)) + 0.5


def test_wpe_chain_flow(orc, dev, proto256, kinect_pcm, wavs):
    """unit_test/test_subband_dereverberator.py:92-170 (multi-channel WPE, confs/wpe.json with a shorter
    prediction order) followed by the beamformer chain of BASELINE config C4 (WPE -> D&S -> synthesis)."""
    from distant_speech_recognition_amd.btk20 import (MultiChannelWPEDereverberationPtr, MultiChannelWPEDereverberationFeaturePtr,
                                                      OverSampledDFTSynthesisBankPtr, PyVectorComplexFeatureStreamPtr)
    from distant_speech_recognition_amd.pybeamformer import SubbandGSCBeamformer, calc_delays
    h, g = proto256
    sample_feats, afbs = _build(wavs[:2], h)
    pre = MultiChannelWPEDereverberationPtr(subbands_num=M, channels_num=2, lower_num=0, upper_num=7, iterations_num=2,
                                            load_db=-18.0, band_width=0.0, diagonal_bias=1e-4, samplerate=FS)
    for a in afbs:
        pre.set_input(a)
    nfr = pre.estimate_filter()
    assert nfr == 317
    for c, p in enumerate(wavs[:2]):
        sample_feats[c].read(p, FS)
    sfbs = [OverSampledDFTSynthesisBankPtr(MultiChannelWPEDereverberationFeaturePtr(pre, channel_no=c), prototype=g, M=M, m=m, r=r,
                                           delay_compensation_type=2) for c in range(2)]
    bufs = [[], []]
    while True:                                    # lock-step pull exactly like test_subband_dereverberator.py:160-170
        try:
            for c in range(2):
                bufs[c].append(np.array(sfbs[c].next()))
        except StopIteration:
            break
    outs = [np.concatenate(b) for b in bufs]
    X = _oracle_X(orc, h, kinect_pcm)[:, :2]
    G = orc.wpe_estimate(X, 0, 7, 2, -18.0, 0.0, 1e-4)
    Yd = orc.wpe_apply(X, G, 0, 7)
    for c in range(2):
        ref = orc.synthesis(g, M, m, r, 2, Yd[:, c])
        assert outs[c].shape == ref.shape
        assert np.max(np.abs(outs[c] - ref)) < 1e-3 * np.max(np.abs(ref)) + 0.5
    # C4-style chain: WPE outputs as beamformer channels, all on the device
    for c, p in enumerate(wavs[:2]):
        sample_feats[c].read(p, FS)
    pre.reset()
    chans = [MultiChannelWPEDereverberationFeaturePtr(pre, channel_no=c) for c in range(2)]
    from distant_speech_recognition_amd.btk20 import SubbandGSCPtr
    bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)            # node-level API: any complex stream is a channel
    for ch in chans:
        bf.set_channel(ch)
    delays = calc_delays("linear", MPOS[:2], [AZIMUTH, None, None])
    bf.calc_gsc_weights(FS, delays)
    sfb = OverSampledDFTSynthesisBankPtr(bf, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
    out = np.concatenate([np.array(b) for b in sfb])
    wq = orc.calc_mainlobe(M, 2, FS, delays)
    ref = orc.synthesis(g, M, m, r, 2, orc.gsc_frames(Yd, wq, None))
    assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 1e-3 * np.max(np.abs(ref)) + 0.5


def test_single_channel_wpe_flow(orc, dev, proto256, kinect_pcm, wavs):
    """unit_test/test_subband_dereverberator.py single-channel branch: estimate on the utterance, re-read, dereverberate."""
    from distant_speech_recognition_amd.btk20 import SingleChannelWPEDereverberationFeaturePtr, OverSampledDFTSynthesisBankPtr
    h, g = proto256
    sample_feats, afbs = _build(wavs[:1], h)
    dereverb = SingleChannelWPEDereverberationFeaturePtr(afbs[0], lower_num=1, upper_num=12, iterations_num=2, load_db=-18.0,
                                                        band_width=0.0, samplerate=FS)
    assert dereverb.estimate_filter() == 317
    sample_feats[0].read(wavs[0], FS)
    sfb = OverSampledDFTSynthesisBankPtr(dereverb, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
    out = np.concatenate([np.array(b) for b in sfb])
    X = _oracle_X(orc, h, kinect_pcm)[:, :1]
    G = orc.wpe_estimate(X, 1, 12, 2, -18.0, 0.0, 0.0)
    ref = orc.synthesis(g, M, m, r, 2, orc.wpe_apply(X, G, 1, 12)[:, 0])
    assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 1e-3 * np.max(np.abs(ref)) + 0.5


def _mvdrgsc_oracle(orc, X, delays, bm):
    """SubbandMVDRGSC by the oracle: MVDR quiescent vector, blocking matrix against the delay-and-sum (bm = 1) or the
    MVDR weights (bm = 2, bins 1..M/2), the deterministic active weights of the tests, y = (w_mvdr - B wa)^H x."""
    K = M // 2 + 1
    wq_ds = orc.calc_mainlobe(M, 4, FS, delays)
    R = orc.diagonal_loading(orc.diffuse_noise_model(np.array(MPOS), M, FS), M, 0.01)
    wfull = np.zeros((M, 4), np.complex128)
    wfull[:K] = orc.mvdr_weights(R, wq_ds, M)
    base = wq_ds if bm == 1 else wfull
    wa = np.zeros((M, 3), np.complex128)
    wl = np.zeros((M, 4), np.complex128)
    B = np.zeros((M, 4, 3), np.complex128)
    for k in range(M):
        if bm == 1 or 1 <= k <= M // 2:
            B[k] = orc.blocking_matrix(base[k], 1)
    for k in range(1, M // 2 + 1):
        i = np.arange(3)
        wa[k] = 0.05 * (np.cos(0.37 * k + i) + 1j * np.sin(0.11 * k * (i + 1)))
        wl[k] = orc.sidelobe_canceller(B[k], wa[k])
    return orc.gsc_frames(X, wfull, wl), wq_ds, wfull, wl, B, wa


@pytest.mark.parametrize("bm", [1, 2])
def test_subband_mvdrgsc_node(orc, dev, proto256, kinect_pcm, wavs, bm):
    """SubbandMVDRGSC (beamformer.cc:2604-2773) used as its header prescribes: calc_array_manifold_vectors ->
    set_diffuse_noise_model -> calc_mvdr_weights -> calc_blocking_matrix1/2 -> set_active_weights_f, then
    upgrade_blocking_matrix / blocking_matrix_output on live frames."""
    from distant_speech_recognition_amd.btk20 import SubbandMVDRGSCPtr
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, g = proto256
    _, afbs = _build(wavs, h)
    bf = SubbandMVDRGSCPtr(fftlen=M, half_band_shift=False)
    for a in afbs:
        bf.set_channel(a)
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    bf.calc_array_manifold_vectors(FS, delays)
    bf.set_diffuse_noise_model(np.array(MPOS), FS)
    bf.set_all_diagonal_loading(0.01)
    assert bf.calc_mvdr_weights(FS, 1.0e-8)
    assert bf.calc_blocking_matrix1(FS, delays) if bm == 1 else bf.calc_blocking_matrix2()
    X = _oracle_X(orc, h, kinect_pcm)
    ref, wq_ds, wfull, wl, B, wa = _mvdrgsc_oracle(orc, X, delays, bm)
    for k in range(1, M // 2 + 1):
        packed = np.empty(6)
        packed[0::2], packed[1::2] = wa[k].real, wa[k].imag
        bf.set_active_weights_f(k, packed)
    out = []
    bo = {}
    for t, v in enumerate(bf):
        out.append(np.array(v))
        if t in (3, 57):
            bo[t] = np.array(bf.blocking_matrix_output(1))[: M // 2 + 1].copy()
    out = np.stack(out)
    assert out.shape == ref.shape
    # MVDR weights agree to ~1e-3 (float32 SVD in the reference vs float32 Cholesky)
    assert np.max(np.abs(out - ref)) < 2e-3 * np.max(np.abs(ref))
    for t, v in bo.items():
        want = np.array([np.vdot(B[k][:, 1], X[t, :, k]) for k in range(M // 2 + 1)])
        assert np.max(np.abs(v - want)) < 1e-5 * np.max(np.abs(X[t]))
    # upgrade_blocking_matrix: B_k <- calc_blocking_matrix_(wq_k - wl_k) (wq = the weight object's quiescent vector)
    bf.upgrade_blocking_matrix()
    bw = bf.beamformer_weight_object(0)
    # (calc_blocking_matrix2 leaves wq = 0 above M/2: the reference's -1/|w|^2 projector turns those bins into NaN there too)
    for k in (1, 17, M // 2) + ((M - 3,) if bm == 1 else ()):
        wk = bw.wq[k] - bw.wl[k]
        assert np.max(np.abs(wk @ bw.B[k])) < 1e-10 * max(1.0, np.max(np.abs(wk)))      # calc_blocking_matrix_: w^T B = 0
        assert np.max(np.abs(bw.B[k] - orc.blocking_matrix(wk, 1))) < 1e-12


def test_wpe_estimate_filter_frame_range_counts_from_the_start(orc, dev, proto256, kinect_pcm, wavs):
    """estimate_filter(start, end): fill_buffer_ (dereverberation.cc:74-94, 506-529) counts frX from 0 and pulls one frame
    per frX in [start, end) from the input's current position, i.e. the estimate sees the FIRST end - start frames."""
    from distant_speech_recognition_amd.btk20 import SingleChannelWPEDereverberationFeaturePtr
    h, g = proto256
    sample_feats, afbs = _build(wavs[:1], h)
    dereverb = SingleChannelWPEDereverberationFeaturePtr(afbs[0], lower_num=1, upper_num=6, iterations_num=2, load_db=-18.0,
                                                        band_width=0.0, samplerate=FS)
    assert dereverb.estimate_filter(10, 110) == 100
    sample_feats[0].read(wavs[0], FS)
    frames = np.stack([np.array(f) for f in dereverb])
    X = _oracle_X(orc, h, kinect_pcm)[:, :1]
    G = orc.wpe_estimate(X[:100], 1, 6, 2, -18.0, 0.0, 0.0)
    ref = orc.wpe_apply(X, G, 1, 6)[:, 0]
    assert frames.shape == ref.shape
    assert np.max(np.abs(frames - ref)) < 1e-3 * np.max(np.abs(ref))


def test_half_band_shift_ds_and_gsc(orc, dev, proto256, kinect_pcm, wavs):
    """halfBandShift == true (SubbandDS / SubbandGSC with one constraint, beamformer.cc:515-527, 1113-1128, 1276-1285): all M
    bins have their own weights and outputs; SubbandMVDR refuses it in the constructor (:2283-2285), SubbandGSCRLS in next()
    (:1528-1530)."""
    from distant_speech_recognition_amd.btk20 import (SubbandDSPtr, SubbandGSCPtr, SubbandGSCRLSPtr, SubbandMVDRPtr,
                                                      jallocation_error, j_error)
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, _ = proto256
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    X = _oracle_X(orc, h, kinect_pcm)                                            # [T][N][M]
    wq = orc.calc_mainlobe_halfband(M, 4, FS, delays)
    # --- D&S
    _, afbs = _build(wavs, h)
    ds = SubbandDSPtr(fftlen=M, half_band_shift=True)
    for a in afbs:
        ds.set_channel(a)
    ds.calc_array_manifold_vectors(FS, delays)
    assert np.max(np.abs(np.stack([ds.get_weights(k) for k in range(M)]) - wq)) < 1e-15
    out = np.stack([np.array(v) for v in ds])
    ref = orc.gsc_frames_halfband(X, wq, np.zeros_like(wq))
    assert out.shape == ref.shape
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(out - ref)) < 2e-5 * scale
    # --- GSC with active weights in every bin and the weight normalisation
    _, afbs = _build(wavs, h)
    gsc = SubbandGSCPtr(fftlen=M, half_band_shift=True)
    for a in afbs:
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
    # --- the classes that refuse it
    with pytest.raises(jallocation_error):
        SubbandMVDRPtr(fftlen=M, half_band_shift=True)
    _, afbs = _build(wavs, h)
    rls = SubbandGSCRLSPtr(fftlen=M, half_band_shift=True, mu=0.9)
    for a in afbs:
        rls.set_channel(a)
    rls.calc_gsc_weights(FS, delays)
    rls.init_precision_matrix(0.01)
    with pytest.raises(j_error):
        rls.next()


class _FrameSource(object):
    """a Python algorithm object as the reference's C++ nodes see it: size(), __iter__, next(), reset()"""

    def __init__(self, frames):
        self.frames, self.i = frames, 0

    def size(self):
        return self.frames.shape[1]

    def __iter__(self):
        self.i = 0
        return self

    def next(self):
        if self.i >= len(self.frames):
            raise StopIteration
        self.i += 1
        return self.frames[self.i - 1]

    __next__ = next

    def reset(self):
        self.i = 0


def test_half_band_shift_over_generic_sources_and_postfilter_guard(orc, dev):
    """halfBandShift over sources that are NOT analysis banks (PyVectorComplexFeatureStream): the reference dots every one of the
    M snapshots as supplied (beamformer.cc:1113-1128) -- no conjugate symmetry between bins may be assumed -- and a post-filter
    over such a beamformer is refused (this engine's post-filters work on the M/2+1 bins of a non-shifted bank)."""
    from distant_speech_recognition_amd.btk20 import (PyVectorComplexFeatureStreamPtr, SubbandGSCPtr, ZelinskiPostFilterPtr, j_error)
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    Mh, N, T = 64, 4, 37
    rng = np.random.default_rng(99)
    X = (rng.standard_normal((T, N, Mh)) + 1j * rng.standard_normal((T, N, Mh))) * 1000.0        # no symmetry between bins k and M-1-k
    delays = calc_delays("linear", MPOS, [AZIMUTH, None, None])
    wq = orc.calc_mainlobe_halfband(Mh, N, FS, delays)
    gsc = SubbandGSCPtr(fftlen=Mh, half_band_shift=True)
    for n in range(N):
        gsc.set_channel(PyVectorComplexFeatureStreamPtr(_FrameSource(X[:, n, :])))
    gsc.calc_gsc_weights(FS, delays)
    wl = np.zeros_like(wq)
    for k in range(Mh):
        wa = (rng.standard_normal(N - 1) + 1j * rng.standard_normal(N - 1)) * 0.05
        packed = np.empty(2 * (N - 1)); packed[0::2] = wa.real; packed[1::2] = wa.imag
        gsc.set_active_weights_f(k, packed)
        wl[k] = orc.blocking_matrix(wq[k], 1) @ wa
    out = np.stack([np.array(v) for v in gsc])
    ref = orc.gsc_frames_halfband(X, wq, wl)
    assert out.shape == ref.shape == (T, Mh)
    assert np.max(np.abs(out - ref)) < 2e-5 * np.max(np.abs(ref))
    assert np.max(np.abs(out[:, Mh // 2 + 1:] - ref[:, Mh // 2 + 1:])) < 2e-5 * np.max(np.abs(ref))     # the bins a mirror would get wrong
    pf = ZelinskiPostFilterPtr(gsc, Mh, 0.7, 2)
    pf.set_beamformer(gsc)
    with pytest.raises(j_error):
        pf.next()
