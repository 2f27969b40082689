"""GPU: the push-style corners of the node API that SURVEY 8(b) names --
  * OverSampledDFTSynthesisBank without a source: input_source_vector() / no_stream_feature() (modulated/modulated.h:320-334),
  * ZelinskiPostFilter without a beamformer object: set_snapshot_array() / set_array_manifold_vector() (postfilter/postfilter.h:83-94,
    postfilter.cc:384-421),
  * BeamformerWeights::CSDs() live (beamformer.cc:874-887, postfilter.cc:77-116),
each through the Python binding AND through a C++ program (tests/cpp/push_nodes.cc) against the oracle."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "distant_speech_recognition_amd", "host")
CSRC = os.path.join(ROOT, "distant_speech_recognition_amd", "csrc")
M, m, r, D, FS, N = 256, 4, 1, 128, 16000, 4
MPOS = [[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]]


@pytest.fixture(scope="module")
def material(orc, proto256, kinect_pcm):
    """snapshots X [T][N][M], D&S weights, beamformed frames Y [T][M] of 60 frames of the Kinect recording (oracle)"""
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, g = proto256
    X = np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:8000]) for c in range(N)], axis=1)[:60]
    delays = calc_delays("linear", MPOS, [-1.306379, None, None])
    wq = orc.calc_mainlobe(M, N, FS, delays)
    Y = orc.gsc_frames(X, wq, np.zeros_like(wq))
    return dict(h=h, g=g, X=X, wq=wq, Y=Y, delays=delays)


def _pushed_reference(orc, g, Y, pd):
    """what a per-frame graph hears: block j is synthesised from the ring after j + 1 pushes == block j of the stream that has pd
    zero frames in front (nothing is primed without a source, modulated.cc:536-549, 574-578)"""
    Z = np.concatenate([np.zeros((pd, Y.shape[1]), np.complex128), Y])
    return orc.synthesis(g, M, m, r, 2, Z).reshape(-1, D)


def test_sourceless_synthesis_bank_python(orc, dev, material):
    from distant_speech_recognition_amd import btk20
    g, Y = material["g"], material["Y"]
    pd = orc.fb_delays(m, r, True, 2)[0]
    ref = _pushed_reference(orc, g, Y, pd)
    assert ref.shape[0] == len(Y)
    sfb = btk20.OverSampledDFTSynthesisBankPtr(prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
    out = []
    for t, y in enumerate(Y):
        (sfb.input_source_vector if t % 2 else sfb.inputSourceVector)(block=y)
        out.append(np.array(sfb.next()))
        assert sfb.frame_no() == t
    out = np.stack(out)
    assert np.max(np.abs(out - ref)) < 0.5                                  # int16 scale
    # the pushed bank IS the sourced bank, pd blocks late -- once the R overlapping polyphase outputs of a block were all computed
    # from the same rings (the sourced bank's first blocks have no earlier next() behind them)
    sourced = orc.synthesis(g, M, m, r, 2, Y).reshape(-1, D)
    R = 1 << r
    assert np.max(np.abs(out[pd + R - 1:] - sourced[R - 1:])) < 0.5
    # not a per-frame graph: no frame, or two, between two next() calls
    with pytest.raises(btk20.jconsistency_error):
        sfb.next()
    sfb.input_source_vector(Y[0]); sfb.input_source_vector(Y[1])
    with pytest.raises(btk20.jconsistency_error):
        sfb.next()
    with pytest.raises(btk20.jdimension_error):
        sfb.input_source_vector(Y[0][:100])
    # reset() empties the ring (buffer_.zero(), modulated.cc:614-622): the same frames give the same blocks again
    sfb.reset()
    again = []
    for y in Y[:12]:
        sfb.input_source_vector(y)
        again.append(np.array(sfb.next()))
    assert np.array_equal(np.stack(again), out[:12])
    # a sourced bank switched to pushing, and gain_factor
    src = btk20.PyVectorComplexFeatureStreamPtr(_Frames(Y))
    sfb2 = btk20.OverSampledDFTSynthesisBankPtr(src, prototype=g, M=M, m=m, r=r, delay_compensation_type=2, gain_factor=2)
    sfb2.no_stream_feature(True)
    sfb2.input_source_vector(Y[0])
    assert np.max(np.abs(np.array(sfb2.next()) - 2 * ref[0])) < 1.0
    sfb2.doNotUseStreamFeature(False)
    first_sourced = np.array(sfb2.next())
    assert first_sourced.shape == (D,)


class _Frames(object):
    def __init__(self, frames):
        self.frames, self.i = frames, 0

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


@pytest.mark.parametrize("type_,alpha", [(2, 0.7), (1, 0.6), (2 | 8, 0.5)])
def test_zelinski_without_beamformer_python(orc, dev, material, type_, alpha):
    """the caller keeps the SnapShotArray current and hands the alignment vectors over bin by bin (the usage of postfilter.h:66-72)"""
    from distant_speech_recognition_amd import btk20
    X, wq, Y = material["X"], material["wq"], material["Y"]
    ref, W, csd = orc.zelinski_frames(X, Y, wq, alpha, type_, return_csd=True)
    pf = btk20.ZelinskiPostFilterPtr(btk20.PyVectorComplexFeatureStreamPtr(_Frames(Y)), M, alpha, type_)
    with pytest.raises(btk20.j_error):
        pf.next()                                                           # "set beamformer's weights"
    pf.reset()
    snap = btk20.SnapShotArrayPtr(M, N)
    pf.set_snapshot_array(snapShotArray=snap)
    for k in range(M):
        pf.set_array_manifold_vector(fbinX=k, arrayManifoldVector=wq[k], halfBandShift=False, NC=1)
    with pytest.raises(btk20.jdimension_error):
        pf.set_array_manifold_vector(M, wq[0], False)
    out = []
    for t in range(len(Y)):
        for c in range(N):
            snap.set_samples(X[t, c], c)
        snap.update()
        out.append(np.array(pf.next()))
    out = np.stack(out)
    scale = np.max(np.abs(ref))
    assert np.max(np.abs(out - ref)) < 2e-5 * scale
    assert np.allclose(pf.postfilter_weights()[:M // 2 + 1].real, W[-1][:M // 2 + 1].real, rtol=2e-4, atol=2e-6)
    # the densities: the filter's own weight object, rebuilt on demand from the frames seen so far
    w = pf.weights_object()
    for k in (0, 1, 17, M // 2):
        got = w.CSDs(k).reshape(-1)
        assert np.max(np.abs(got - csd[k])) < 3e-5 * np.max(np.abs(csd[k]))
    assert np.all(w.CSDs(M // 2 + 5) == 0)                                  # bins beyond M/2 are never touched (postfilter.cc:184)
    assert np.allclose(w.wp1()[:M // 2 + 1].real, W[-1][:M // 2 + 1].real, rtol=2e-4, atol=2e-6)
    with pytest.raises(StopIteration):
        pf.next()


def test_csds_are_live_behind_a_beamformer(orc, dev, material, proto256, kinect_pcm, tmp_path):
    """CSDs() of the beamformer's weight object after k frames of a bound post-filter == the reference's per-pair recursion
    (round 3 handed out zeroed vectors); zero again where the reference re-allocates the weight object"""
    import wave
    from distant_speech_recognition_amd import btk20
    h = material["h"]
    afbs = []
    for c in range(N):
        p = str(tmp_path / ("c%d.wav" % c))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:8000].astype(np.int16).tobytes())
        w.close()
        sf = btk20.SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.read(p, FS)
        afbs.append(btk20.OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2))
    bf = btk20.SubbandDSPtr(fftlen=M, half_band_shift=False)
    for a in afbs:
        bf.set_channel(a)
    bf.calc_array_manifold_vectors(FS, material["delays"])
    w0 = bf.beamformer_weight_object(0)
    assert np.all(w0.CSDs(3) == 0)                                          # no post-filter yet: the reference's zeros
    pf = btk20.ZelinskiPostFilterPtr(bf, M, 0.7, 2)
    pf.set_beamformer(bf)
    served = 25
    for _ in range(served):
        pf.next()
    X, wq = material["X"], material["wq"]
    _, _, csd = orc.zelinski_frames(X[:served], material["Y"][:served], wq, 0.7, 2, return_csd=True)
    for k in (0, 2, 40, M // 2):
        got = bf.beamformer_weight_object(0).CSDs(k).reshape(-1)
        assert np.max(np.abs(got - csd[k])) < 3e-5 * np.max(np.abs(csd[k]))
    # two more frames: the densities move on
    pf.next(); pf.next()
    _, _, csd2 = orc.zelinski_frames(X[:served + 2], material["Y"][:served + 2], wq, 0.7, 2, return_csd=True)
    got = bf.beamformer_weight_object(0).CSDs(40).reshape(-1)
    assert np.max(np.abs(got - csd2[40])) < 3e-5 * np.max(np.abs(csd2[40])) and np.max(np.abs(got - csd[40])) > 1e-3 * np.max(np.abs(csd[40]))
    # new weights = a new weight object (beamformer.cc:1082-1092): zeros until the post-filter serves its next frame, then a history
    # that starts at that frame
    bf.calc_array_manifold_vectors(FS, material["delays"] * 0.5)
    assert np.all(bf.beamformer_weight_object(0).CSDs(40) == 0)
    pf.next(); pf.next(); pf.next()
    wq2 = orc.calc_mainlobe(M, N, FS, material["delays"] * 0.5)
    t0 = served + 2
    ref = np.zeros((N, N), np.complex128)
    for t in range(t0, t0 + 3):                                             # alpha stays 0.7 (the frame counter keeps counting), state from zero
        x = np.conj(wq2[40]) * X[t, :, 40]
        ref = 0.7 * ref + 0.3 * np.outer(x, np.conj(x))
    ref = np.triu(ref)
    ref[np.diag_indices(N)] = ref[np.diag_indices(N)].real
    got = bf.beamformer_weight_object(0).CSDs(40)
    assert np.max(np.abs(got - ref)) < 3e-5 * np.max(np.abs(ref))


def test_push_nodes_through_cpp(orc, dev, material, tmp_path):
    exe = str(tmp_path / "push_nodes")
    cmd = ["g++", "-O1", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(HOST, "include"), "-I" + os.path.join(ROOT, "include"),
           "-I/opt/rocm/include", os.path.join(ROOT, "tests", "cpp", "push_nodes.cc"), "-o", exe,
           "-L" + HOST, "-lbtk20hip", "-L" + CSRC, "-lbtkhip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + HOST, "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-3000:]
    g, X, wq, Y = material["g"], material["X"], material["wq"], material["Y"]
    T = len(Y)
    np.asarray(g, np.float64).tofile(str(tmp_path / "g.f64"))
    np.ascontiguousarray(Y, np.complex128).tofile(str(tmp_path / "Y.c128"))
    np.ascontiguousarray(X, np.complex128).tofile(str(tmp_path / "X.c128"))
    np.ascontiguousarray(wq, np.complex128).tofile(str(tmp_path / "d.c128"))
    alpha, type_ = 0.65, 2
    run = subprocess.run([exe, str(tmp_path), str(M), str(m), str(r), str(T), str(N), repr(alpha), str(type_)],
                         capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ok" in run.stdout, run.stdout + run.stderr
    pd = orc.fb_delays(m, r, True, 2)[0]
    blocks = np.fromfile(str(tmp_path / "blocks.f32"), np.float32).reshape(T, D)
    assert np.max(np.abs(blocks - _pushed_reference(orc, g, Y, pd))) < 0.5
    ref, W, csd = orc.zelinski_frames(X, Y, wq, alpha, type_, return_csd=True)
    Z = np.fromfile(str(tmp_path / "Z.c128"), np.complex128).reshape(T, M)
    assert np.max(np.abs(Z - ref)) < 2e-5 * np.max(np.abs(ref))
    got = np.fromfile(str(tmp_path / "csd.c128"), np.complex128).reshape(M // 2 + 1, N * N)
    for k in (0, 5, 77, M // 2):
        assert np.max(np.abs(got[k] - csd[k])) < 3e-5 * np.max(np.abs(csd[k]))
    wp1 = np.fromfile(str(tmp_path / "wp1.c128"), np.complex128)
    assert np.allclose(wp1[:M // 2 + 1].real, W[-1][:M // 2 + 1].real, rtol=2e-4, atol=2e-6)


def _wav_channels(btk20, kinect_pcm, h, tmp_path, L=8000):
    import wave
    afbs = []
    for c in range(N):
        p = str(tmp_path / ("c%d.wav" % c))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:L].astype(np.int16).tobytes())
        w.close()
        sf = btk20.SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.read(p, FS)
        afbs.append(btk20.OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2))
    return afbs


def test_block_protocol_leaves_the_frame_counter_alone(orc, dev, material, kinect_pcm, tmp_path):
    """A synthesis bank that took the beamformer's block tells it how far a per-frame graph would have pulled (advance_to); that
    mark is the block protocol's own: a second consumer pulling the SAME node frame by frame still starts at frame 0 and sees every
    frame (round 3 moved the node's frame counter and silently skipped them)."""
    from distant_speech_recognition_amd import btk20
    afbs = _wav_channels(btk20, kinect_pcm, material["h"], tmp_path)
    bf = btk20.SubbandDSPtr(fftlen=M, half_band_shift=False)
    for a in afbs:
        bf.set_channel(a)
    bf.calc_array_manifold_vectors(FS, material["delays"])
    sfb = btk20.OverSampledDFTSynthesisBankPtr(bf, prototype=material["g"], M=M, m=m, r=r, delay_compensation_type=2)
    for _ in range(6):
        sfb.next()
    assert bf.frame_no() == -1
    Y = material["Y"]
    for t in range(5):
        fr = np.array(bf.next())
        assert bf.frame_no() == t
        assert np.max(np.abs(fr - Y[t])) <= 2e-5 * np.max(np.abs(Y))
    # the mark still does its job: new weights recompute only what was not handed over to the synthesis bank
    first = np.concatenate([np.array(sfb.next()) for _ in range(3)])
    assert np.all(np.isfinite(first))


def test_gscrls_refuses_a_stale_block(orc, dev, material, kinect_pcm, tmp_path):
    """SubbandGSCRLS runs its recursion over the whole utterance once; new weights after frames were served cannot continue it from
    that frame (the state of that frame is gone) -- jconsistency_error instead of the cached block of the old weights; before any
    frame was served the recursion simply runs again"""
    from distant_speech_recognition_amd import btk20

    def graph(delays):          # (a SampleFeature frees its samples at end of stream, feature.cc:612-620: a new graph per utterance)
        bf = btk20.SubbandGSCRLSPtr(fftlen=M, half_band_shift=False, myu=0.95, sigma2=0.0)
        for a in _wav_channels(btk20, kinect_pcm, material["h"], tmp_path):
            bf.set_channel(a)
        bf.calc_gsc_weights(FS, delays)
        bf.init_precision_matrix(0.01)
        return bf, btk20.OverSampledDFTSynthesisBankPtr(bf, prototype=material["g"], M=M, m=m, r=r, delay_compensation_type=2)

    bf, sfb = graph(material["delays"])
    a0 = np.array(sfb.next()).copy()
    bf.calc_gsc_weights(FS, material["delays"] * 0.5)                        # mid-stream
    with pytest.raises(btk20.jconsistency_error):
        sfb.next()
    # no frame served yet: changing the weights just reruns the recursion with them
    bf, sfb = graph(material["delays"] * 0.5)
    ref = np.array(sfb.next()).copy()
    bf, sfb = graph(material["delays"])
    bf.calc_gsc_weights(FS, material["delays"] * 0.5)
    bf.init_precision_matrix(0.01)
    c0 = np.array(sfb.next())
    assert np.all(np.isfinite(c0)) and not np.array_equal(a0, c0)
    assert np.max(np.abs(c0 - ref)) <= 1e-4 * max(np.max(np.abs(ref)), 1.0)
