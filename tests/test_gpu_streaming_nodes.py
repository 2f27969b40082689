"""GPU: bounded-block streaming of the node layer (host/include/modulated/modulated.h, BlockSource).  The reference is frame-in /
frame-out (modulated/modulated.cc:375-469, 569-612; stream/pyStream.h:44-160); the engine computes blocks of at most
block_frames frames and must (a) give the SAME output bit for bit for every block size, (b) deliver the first output after one
block of input, (c) run an endless source in bounded memory."""
import os
import resource
import subprocess
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "beamformer_ds")
EXE_MVDRGSC = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "beamformer_mvdrgsc")
MPOS = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
AZIMUTH = -1.306379
M, m, r, D, FS = 256, 4, 1, 128, 16000


def _wav(path, pcm):
    w = wave.open(path, "wb")
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
    w.writeframes(pcm.astype(np.int16).tobytes())
    w.close()


@pytest.mark.parametrize("pf,alpha,kind", [(0, 0.0, ""), (2, 0.7, ""), (2, 0.7, "mccowan"), (2, 0.8, "lefkimmiatis"), (0, 0.0, "gscrls")])
def test_example_binary_same_bits_for_every_block_size(dev, tmp_path, proto256, kinect_pcm, pf, alpha, kind):
    """src/beamformerDS.cc's graph (SampleFeature -> analysis x 4 -> GSC / GSC-RLS (-> post-filter) -> synthesis) through the C++
    example: one block for the whole utterance (BTK_BLOCK_FRAMES=0) against 16-, 37- and 100-frame blocks, byte for byte."""
    from tests.util import la_delays
    h, g = proto256
    coeffs = str(tmp_path / "coeffs.f64")
    np.concatenate([h, g]).astype(np.float64).tofile(coeffs)
    delays = la_delays(MPOS, AZIMUTH)
    L = 30000
    chan = []
    for c in range(4):
        p = str(tmp_path / ("c%d.wav" % c))
        _wav(p, kinect_pcm[c][:L])
        chan += [repr(float(delays[c])), p]
    outs = {}
    for bf in (0, 16, 37, 100):
        out = str(tmp_path / ("out%d.f32" % bf))
        env = dict(os.environ, BTK_BLOCK_FRAMES=str(bf))
        if kind in ("mccowan", "lefkimmiatis"):
            env["BTK_EXAMPLE_PF"] = kind
            env["BTK_EXAMPLE_MPOS"] = ";".join(",".join(repr(float(v)) for v in row) for row in MPOS)
        if kind == "gscrls":
            env["BTK_EXAMPLE_BF"] = "gscrls"
        res = subprocess.run([EXE, coeffs, str(M), str(m), str(r), str(pf), str(alpha), out] + chan, capture_output=True, text=True,
                             timeout=300, env=env)
        assert res.returncode == 0, res.stderr
        outs[bf] = np.fromfile(out, np.float32)
    assert outs[0].size > 200 * D and np.max(np.abs(outs[0])) > 100
    for bf in (16, 37, 100):
        assert outs[bf].shape == outs[0].shape, (bf, outs[bf].shape, outs[0].shape)
        assert np.array_equal(outs[bf].view(np.uint32), outs[0].view(np.uint32)), (kind, bf, float(np.max(np.abs(outs[bf] - outs[0]))))


def _graph(wavs, h, g, block_frames, postfilter=False, dct=2):
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr, ZelinskiPostFilterPtr,
                                                      OverSampledDFTSynthesisBankPtr)
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    afbs = []
    for p in wavs:
        sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.read(p, FS)
        a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=dct)
        a.set_block_frames(block_frames)
        afbs.append(a)
    bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
    for a in afbs:
        bf.set_channel(a)
    delays = calc_delays("linear", MPOS.tolist(), [AZIMUTH, None, None])
    bf.calc_gsc_weights(FS, delays)
    top = bf
    if postfilter:
        top = ZelinskiPostFilterPtr(bf, M, 0.7, 2)
        top.set_beamformer(bf)
    sfb = OverSampledDFTSynthesisBankPtr(top, prototype=g, M=M, m=m, r=r, delay_compensation_type=dct)
    return afbs, bf, top, sfb


@pytest.fixture(scope="module")
def wavs(tmp_path_factory, kinect_pcm):
    d = tmp_path_factory.mktemp("swav")
    paths = []
    for c in range(4):
        p = str(d / ("c%d.wav" % c))
        _wav(p, kinect_pcm[c][:40000])
        paths.append(p)
    return paths


@pytest.mark.parametrize("postfilter", [False, True])
def test_python_graph_same_bits_and_weight_change_mid_stream(dev, proto256, wavs, postfilter):
    """The same graph through the Python binding: every block size gives the whole-utterance bits, also when the look direction
    moves while the stream runs (unit_test/test_online_beamforming.py:209-226): frames already pulled keep the old weights."""
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, g = proto256
    outs = {}
    for bfr in (0, 16, 50):
        _, bf, _, sfb = _graph(wavs, h, g, bfr, postfilter)
        blocks = []
        for i, b in enumerate(sfb):
            blocks.append(np.array(b))
            if i == 120:                                         # the speaker moves: new look direction from the next frame on
                bf.calc_gsc_weights(FS, calc_delays("linear", MPOS.tolist(), [0.3, None, None]))
        outs[bfr] = np.concatenate(blocks)
    assert outs[0].size == 313 * D
    for bfr in (16, 50):
        assert outs[bfr].shape == outs[0].shape
        assert np.array_equal(outs[bfr].view(np.uint32), outs[0].view(np.uint32)), (bfr, float(np.max(np.abs(outs[bfr] - outs[0]))))


def test_gsc_lms_and_rls_python_classes_same_bits(dev, proto256, wavs):
    """pybeamformer.SubbandGSCLMSBeamformer / SubbandGSCRLSBeamformer (lib/pybeamformer.py:588-928) over snapshot blocks: the
    recursion state lives on the device and carries from block to block -- same bits as one block for the utterance."""
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, PyVectorComplexFeatureStreamPtr,
                                                      OverSampledDFTSynthesisBankPtr)
    from distant_speech_recognition_amd.pybeamformer import SubbandGSCLMSBeamformer, SubbandGSCRLSBeamformer, calc_delays
    h, g = proto256
    delays = calc_delays("linear", MPOS.tolist(), [AZIMUTH, None, None])
    for cls, kw in ((SubbandGSCLMSBeamformer, dict(min_frames=16, slowdown_after=64)), (SubbandGSCRLSBeamformer, dict(min_frames=16))):
        outs = {}
        for bfr in (0, 24):
            afbs = []
            for p in wavs:
                sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
                sf.read(p, FS)
                a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2)
                a.set_block_frames(bfr)
                afbs.append(a)
            bf = cls(afbs, **kw)
            bf.calc_beamformer_weights(FS, delays)
            sfb = OverSampledDFTSynthesisBankPtr(PyVectorComplexFeatureStreamPtr(bf), prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
            outs[bfr] = np.concatenate([np.array(b) for b in sfb])
        assert outs[0].size == 313 * D and outs[24].shape == outs[0].shape
        assert np.array_equal(outs[24].view(np.uint32), outs[0].view(np.uint32)), (cls.__name__, float(np.max(np.abs(outs[24] - outs[0]))))


class _CountingSource:
    """A live PCM source for PyVectorFloatFeatureStream (stream/pyStream.h:44-160): counts the blocks pulled from it; endless
    unless `limit` is given."""

    def __init__(self, seed, limit=None):
        self._seed, self._limit = seed, limit
        self.reset()

    def size(self):
        return D

    def __iter__(self):
        return self

    def next(self):
        if self._limit is not None and self.pulled >= self._limit:
            raise StopIteration
        self.pulled += 1
        return np.rint(self._rng.normal(0.0, 1000.0, D)).astype(np.float32)

    __next__ = next

    def reset(self):
        self._rng = np.random.default_rng(self._seed)
        self.pulled = 0


def _live_graph(h, g, block_frames, limit=None, postfilter=True):
    from distant_speech_recognition_amd.btk20 import (PyVectorFloatFeatureStreamPtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr,
                                                      ZelinskiPostFilterPtr, OverSampledDFTSynthesisBankPtr)
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    srcs = [_CountingSource(100 + c, limit) for c in range(4)]
    bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
    for s in srcs:
        a = OverSampledDFTAnalysisBankPtr(PyVectorFloatFeatureStreamPtr(s), prototype=h, M=M, m=m, r=r, delay_compensation_type=2)
        a.set_block_frames(block_frames)
        bf.set_channel(a)
    bf.calc_gsc_weights(FS, calc_delays("linear", MPOS.tolist(), [AZIMUTH, None, None]))
    top = bf
    if postfilter:
        top = ZelinskiPostFilterPtr(bf, M, 0.7, 2)
        top.set_beamformer(bf)
    return srcs, OverSampledDFTSynthesisBankPtr(top, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)


def test_first_output_after_one_block_of_input(dev, proto256):
    """A source that counts its pulls: the first output block arrives after block_frames input blocks per channel (a per-frame
    graph needs laN + pd + 1 = 8 at this geometry), and block i after at most i + block_frames + laN + pd + 1.  A chain with a
    post-filter rounds its blocks to the 64-frame chunks of the density scan (set_block_quantum): first output after 64 + laN."""
    h, g = proto256
    for pf, bfr, first in ((False, 8, 8), (False, 16, 16), (False, 64, 64), (True, 64, 67), (True, 128, 128), (True, 16, 67)):
        srcs, sfb = _live_graph(h, g, bfr, postfilter=pf)
        sfb.next()
        assert all(s.pulled <= first + bfr - 1 and s.pulled >= min(first, 8) for s in srcs), (pf, bfr, [s.pulled for s in srcs])
        p0 = srcs[0].pulled
        for i in range(1, 200):
            sfb.next()
            assert all(s.pulled <= i + p0 + bfr + 64 for s in srcs), (bfr, i, [s.pulled for s in srcs])


def test_live_source_equals_the_finite_one(dev, proto256):
    """the first 300 output blocks of an endless source are the 300 blocks a 400-block recording of the same samples gives"""
    h, g = proto256
    _, live = _live_graph(h, g, 32)
    _, rec = _live_graph(h, g, 0, limit=400)
    a = np.concatenate([np.array(live.next()) for _ in range(300)])
    b = np.concatenate([np.array(v) for v in rec])
    assert b.size >= 300 * D and np.array_equal(a.view(np.uint32), b[:300 * D].view(np.uint32))


def test_endless_source_runs_in_bounded_memory(dev, proto256):
    """20 000 output blocks (160 s of audio at this geometry) from a source that never ends: the resident set does not grow with
    the stream (the sample windows, the block buffers and the synthesis history are all bounded by the block size)"""
    h, g = proto256
    srcs, sfb = _live_graph(h, g, 64)
    for _ in range(4000):
        sfb.next()
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    acc = 0.0
    for _ in range(16000):
        acc += float(np.abs(np.array(sfb.next())).max())
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert acc > 0 and srcs[0].pulled >= 20000
    assert rss1 - rss0 < 64 * 1024, (rss0, rss1)                 # KiB: whole-utterance buffering would add > 100 MB here


def test_rls_node_weight_change_before_and_between_blocks(dev, proto256, wavs):
    """SubbandGSCRLS (beamformer.cc:1447-1699): the recursion of a block ran with the weights of that moment.  New look direction
    before any frame of the block was served -> the block runs again from the state at its start, with the new quiescent vector
    and blocked directions on the device and P / w_a carried into the new blocking matrix's basis (the reference keeps them in
    active-weight space); the result equals a node that had the new weights from the start.  Between two blocks the recursion
    simply continues in the new basis; after frames of the block were served the node refuses."""
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCRLSPtr, jconsistency_error)
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, _ = proto256
    d1 = calc_delays("linear", MPOS.tolist(), [AZIMUTH, None, None])
    d2 = calc_delays("linear", MPOS.tolist(), [0.4, None, None])

    def node(delays, bfr):
        bf = SubbandGSCRLSPtr(fftlen=M, half_band_shift=False, mu=0.97, sigma2=0.001)
        for p in wavs:
            sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
            sf.read(p, FS)
            a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2)
            a.set_block_frames(bfr)
            bf.set_channel(a)
        bf.calc_gsc_weights(FS, delays)
        bf.init_precision_matrix(1.0e6)
        return bf

    fresh = node(d2, 0)
    want = np.array(fresh.device_block())
    moved = node(d1, 0)
    first = np.array(moved.device_block())                          # the whole utterance ran towards d1 ...
    moved.calc_gsc_weights(FS, d2)                                  # ... the look direction changes before a frame is pulled
    got = np.array(moved.device_block())
    scale = float(np.max(np.abs(want)))
    assert np.max(np.abs(first - want)) > 1e-3 * scale              # the two directions do differ
    assert np.max(np.abs(got - want)) <= 1e-5 * scale, float(np.max(np.abs(got - want)) / scale)
    # between blocks: frames 0..63 towards d1, the rest towards d2 == a whole-utterance node switched at frame 64?  That one
    # refuses (frames of its only block were served); the blockwise node carries on
    blk = node(d1, 64)
    n0 = blk.num_frames()                                            # frames of the first block (64 input blocks minus the look-ahead)
    assert 32 <= n0 <= 64
    a = [np.array(blk.next()) for _ in range(n0)]
    blk.calc_gsc_weights(FS, d2)
    b = []
    while True:                                                      # (iter() would reset the node)
        try:
            b.append(np.array(blk.next()))
        except StopIteration:
            break
    assert len(a) + len(b) == first.shape[-1] and np.all(np.isfinite(np.array(b)))
    assert np.max(np.abs(np.array(a)[:, :M // 2 + 1].T - first[0][:, :n0])) <= 1e-5 * scale       # the first block: still towards d1
    whole = node(d1, 0)
    for _ in range(64):
        whole.next()
    whole.calc_gsc_weights(FS, d2)
    with pytest.raises(jconsistency_error):
        whole.next()


def _sample_graph(pcms, h, g, block_frames, dct=2, half_band=False):
    """analysis x N -> SubbandDS -> synthesis over in-memory PCM (SampleFeature.set_samples); returns the synthesis node"""
    from distant_speech_recognition_amd.btk20 import SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandDSPtr, OverSampledDFTSynthesisBankPtr
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    bf = SubbandDSPtr(fftlen=M, half_band_shift=half_band)
    feats = []
    for x in pcms:
        sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.setSamples(np.asarray(x, np.float64), FS)
        a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=dct)
        a.set_block_frames(block_frames)
        bf.set_channel(a)
        feats.append(sf)
    bf.calc_array_manifold_vectors(FS, calc_delays("linear", MPOS[:len(pcms)].tolist(), [AZIMUTH, None, None]))
    return feats, bf, OverSampledDFTSynthesisBankPtr(bf, prototype=g, M=M, m=m, r=r, delay_compensation_type=dct)


@pytest.mark.parametrize("dct", [0, 2])
def test_blocks_edge_cases_equal_whole_utterance(dev, proto256, kinect_pcm, dct):
    """Inputs a block protocol can trip over -- empty, shorter than the look-ahead, one sample, a length that is not a multiple
    of the shift, channels of unequal length (the shortest ends the stream) -- give the same number of output blocks and the same
    bits at block sizes 1, 3, 8 and 64 as with one block per utterance; both delay-compensation types; reset() replays the stream."""
    h, g = proto256
    cases = [[np.zeros(0), np.zeros(0)], [kinect_pcm[0][:1], kinect_pcm[1][:1]], [kinect_pcm[0][:3 * D], kinect_pcm[1][:3 * D]],
             [kinect_pcm[0][:10 * D + 17], kinect_pcm[1][:10 * D + 17]], [kinect_pcm[0][:9000], kinect_pcm[1][:5000], kinect_pcm[2][:7001]],
             [kinect_pcm[c][:20011] for c in range(4)]]
    for pcms in cases:
        _, _, ref_node = _sample_graph(pcms, h, g, 0, dct)
        ref = [np.array(b) for b in ref_node]
        for bfr in (1, 3, 8, 64):
            feats, bf, sfb = _sample_graph(pcms, h, g, bfr, dct)
            got = [np.array(b) for b in sfb]
            assert len(got) == len(ref), (len(pcms[0]), bfr, len(got), len(ref))
            if ref:
                assert np.array_equal(np.concatenate(got).view(np.uint32), np.concatenate(ref).view(np.uint32)), (len(pcms[0]), bfr)
            if bfr == 8 and len(ref) > 4:                       # reset in mid-stream, then the whole stream again
                for f, x in zip(feats, pcms):
                    f.setSamples(np.asarray(x, np.float64), FS)
                it = iter(sfb)
                first = [np.array(next(it)) for _ in range(3)]
                for f, x in zip(feats, pcms):
                    f.setSamples(np.asarray(x, np.float64), FS)
                again = [np.array(b) for b in sfb]
                assert len(again) == len(ref) and np.array_equal(np.concatenate(again), np.concatenate(ref))
                assert np.array_equal(np.concatenate(first), np.concatenate(ref[:3]))


def test_half_band_shift_and_mvdrgsc_example_in_blocks(dev, tmp_path, proto256, kinect_pcm):
    """halfBandShift == true (all M bins carried through the blocks) and the MVDR-GSC example binary: same bits for every block size"""
    from tests.util import la_delays
    h, g = proto256
    pcms = [kinect_pcm[c][:15000] for c in range(4)]
    _, _, ref_node = _sample_graph(pcms, h, g, 0, 2, half_band=True)
    ref = np.concatenate([np.array(b) for b in ref_node])
    for bfr in (5, 32):
        _, _, sfb = _sample_graph(pcms, h, g, bfr, 2, half_band=True)
        got = np.concatenate([np.array(b) for b in sfb])
        assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), bfr
    coeffs = str(tmp_path / "coeffs.f64")
    np.concatenate([h, g]).astype(np.float64).tofile(coeffs)
    delays = la_delays(MPOS, AZIMUTH)
    chan = []
    for c in range(4):
        p = str(tmp_path / ("c%d.wav" % c))
        _wav(p, kinect_pcm[c][:30000])
        chan += [repr(float(delays[c])), p]
    outs = {}
    for bfr in (0, 20):
        out = str(tmp_path / ("mg%d.f32" % bfr))
        env = dict(os.environ, BTK_BLOCK_FRAMES=str(bfr))
        mpos = ";".join(",".join(repr(float(v)) for v in row) for row in MPOS)
        res = subprocess.run([EXE_MVDRGSC, coeffs, str(M), str(m), str(r), "0.01", out, mpos] + chan, capture_output=True, text=True, timeout=300, env=env)
        assert res.returncode == 0, res.stderr
        outs[bfr] = np.fromfile(out, np.float32)
    assert outs[0].size > 100 * D and np.array_equal(outs[20].view(np.uint32), outs[0].view(np.uint32))


def test_synthesis_next_blocks_equals_next(dev):
    """OverSampledDFTSynthesisBank.next_blocks (engine extension): the blocks next() would hand out one by one, a round at a time --
    the same bits in the same order, also when the two kinds of call alternate and with a bound on the blocks per call; an empty
    array and is_end() at the end of the stream"""
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr, OverSampledDFTSynthesisBankPtr)
    from tests.util import design_prototype, synthetic_pcm
    M, m, r, N = 512, 4, 1, 8
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, 130 * D + 21, seed=77)

    def graph(block_frames):
        keep = []
        bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
        for c in range(N):
            sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
            sf.set_samples(np.ascontiguousarray(pcm[0][c], np.float32))
            a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2)
            a.set_block_frames(block_frames)
            bf.set_channel(a)
            keep += [sf, a]
        bf.calc_gsc_weights(16000, delays)
        return keep, bf, OverSampledDFTSynthesisBankPtr(bf, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)

    k0, b0, s0 = graph(48)
    ref = np.stack([np.array(b) for b in s0])
    assert ref.shape[1] == D and ref.shape[0] > 100
    # whole rounds
    k1, b1, s1 = graph(48)
    parts = []
    while True:
        a = s1.next_blocks()
        assert a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == D
        if a.shape[0] == 0:
            break
        parts.append(a)
    assert s1.is_end()
    got = np.concatenate(parts)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert 2 <= len(parts) <= ref.shape[0] // 48 + 2                     # a call per round, not per block
    # bounded calls alternating with next()
    k2, b2, s2 = graph(37)
    rows = []
    it = iter(s2)
    try:
        while True:
            a = s2.next_blocks(5)
            if a.shape[0] == 0:
                break
            assert a.shape[0] <= 5
            rows.extend(a)
            rows.append(np.array(next(it)))
    except StopIteration:
        pass
    got = np.stack(rows)
    assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
