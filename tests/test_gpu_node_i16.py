"""GPU: the node API streams an utterance as the 16-bit PCM it is.  SampleFeature::read turns a WAV's int16 samples into
un-normalised floats (feature/feature.cc:265-269) and the analysis banks of a beamformer graph pull them block by block
(modulated/modulated.cc:419-438).  When every source of a beamformer node holds 16-bit PCM (SampleFeature::pcm16) the node takes
the samples where they lie -- no host copies, 2 bytes per sample over PCIe -- and widens them on the device: inside the fused
kernel (btk_fb_analysis_bf_i16) where the geometry has that entry, by btk_pcm_i16_to_f32 for every other consumer.  The widening
is exact, so every graph must give THE SAME BITS as with BTK_NODE_I16=0 (the float path of rounds 1-5)."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FS = 16000


class _float_path:
    """BTK_NODE_I16=0 while the graph streams (read at the start of every stream)"""

    def __enter__(self):
        self.old = os.environ.get("BTK_NODE_I16")
        os.environ["BTK_NODE_I16"] = "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["BTK_NODE_I16"]
        else:
            os.environ["BTK_NODE_I16"] = self.old


def _graph(pcm, h, g, M, m, r, delays, block_frames, dct=2, staged=False, postfilter=False):
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr,
                                                      OverSampledDFTSynthesisBankPtr, ZelinskiPostFilterPtr)
    D = M >> r
    keep = []
    bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
    for c in range(pcm.shape[0]):
        sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.set_samples(np.ascontiguousarray(pcm[c], np.float32))
        a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=dct)
        a.set_block_frames(block_frames)
        bf.set_channel(a)
        keep += [sf, a]
    bf.calc_gsc_weights(FS, delays)
    if staged:
        bf.want_snapshots()
    src = bf
    if postfilter:
        pf = ZelinskiPostFilterPtr(bf, M, 0.7, 2)
        pf.set_beamformer(bf)
        keep.append(pf)
        src = pf
    sfb = OverSampledDFTSynthesisBankPtr(src, prototype=g, M=M, m=m, r=r, delay_compensation_type=dct)
    return keep, bf, sfb


def _pull(sfb):
    return np.concatenate([np.array(b) for b in sfb])


def _same_bits(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("M,N,block_frames", [(256, 4, 0), (256, 4, 48), (512, 8, 64), (512, 8, 37), (512, 64, 0), (1024, 8, 40), (2048, 8, 0)])
def test_i16_stream_same_bits_as_float_stream_and_oracle(orc, dev, M, N, block_frames):
    from tests.util import design_prototype, synthetic_pcm
    m, r, dct = 4, 1, 2
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    nfr = 150 if N <= 8 else 60
    pcm, delays = synthetic_pcm(1, N, nfr * D + 57, seed=11 + M + N)          # (a ragged last block: zero padding of the source)
    pcm = pcm[0]
    keep, bf_i, sfb_i = _graph(pcm, h, g, M, m, r, delays, block_frames)
    assert all(k.holds_pcm16() for k in keep[0::2])
    out_i = _pull(sfb_i)
    assert bf_i.i16_stream() and bf_i.fused_path() and not bf_i.snapshots_materialised()
    with _float_path():
        _, bf_f, sfb_f = _graph(pcm, h, g, M, m, r, delays, block_frames)
        out_f = _pull(sfb_f)
        assert not bf_f.i16_stream() and bf_f.fused_path()
    assert _same_bits(out_i, out_f), float(np.max(np.abs(out_i - out_f)))
    wq, B, wl = orc.gsc_weights(M, N, FS, delays)
    ref, _ = orc.pipeline_gsc(h, g, M, m, r, dct, pcm, wq, wl)
    assert ref.shape == out_i.shape and np.max(np.abs(out_i - ref)) < 0.5     # <= 0.5 LSB at int16 scale


@pytest.mark.parametrize("M,postfilter", [(256, False), (512, False), (512, True)])
def test_i16_stream_with_snapshot_consumers_same_bits(dev, M, postfilter):
    """the staged pair (somebody wants the snapshots) and a Zelinski post-filter behind the beamformer: the int16 block is widened
    once on the device (btk_pcm_i16_to_f32) and everything downstream sees the float samples it always saw"""
    from tests.util import design_prototype, synthetic_pcm
    m, r, N = 4, 1, 8
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, 200 * D, seed=3 + M)
    pcm = pcm[0]
    _, bf_i, sfb_i = _graph(pcm, h, g, M, m, r, delays, 64, staged=not postfilter, postfilter=postfilter)
    out_i = _pull(sfb_i)
    assert bf_i.i16_stream() and bf_i.snapshots_materialised() and not bf_i.fused_path()
    with _float_path():
        _, bf_f, sfb_f = _graph(pcm, h, g, M, m, r, delays, 64, staged=not postfilter, postfilter=postfilter)
        out_f = _pull(sfb_f)
        assert not bf_f.i16_stream()
    assert float(np.max(np.abs(out_f))) > 100
    assert _same_bits(out_i, out_f), float(np.max(np.abs(out_i - out_f)))


def test_snapshots_mid_stream_in_an_i16_stream(orc, dev, proto256, kinect_pcm):
    """device_snapshots() in the middle of a fused 16-bit stream: the current block's snapshots come from the widened samples"""
    from tests.util import la_delays
    M, m, r, dct = 256, 4, 1, 2
    h, g = proto256
    mpos = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
    delays = la_delays(mpos, -1.306379)
    pcm = kinect_pcm[:, :24000]
    _, bf, sfb = _graph(pcm, h, g, M, m, r, delays, 64)
    X = np.stack([orc.analysis(h, M, m, r, dct, pcm[c]) for c in range(4)], axis=1)        # [T][N][M]
    out = []
    for i, b in enumerate(sfb):
        out.append(np.array(b))
        if i == 70:
            assert bf.i16_stream() and bf.fused_path()
            Xd = bf.device_snapshots().cpu().numpy()[0]
            base, T = bf.chunk_base(), bf.num_frames()
            ref = np.transpose(X[base:base + T, :, :M // 2 + 1], (2, 1, 0))
            assert Xd.shape == ref.shape and np.max(np.abs(Xd - ref)) <= 1e-5 * np.max(np.abs(ref))
    out = np.concatenate(out)
    wq, B, wl = orc.gsc_weights(M, 4, FS, delays)
    ref, _ = orc.pipeline_gsc(h, g, M, m, r, dct, pcm, wq, wl)
    assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 0.5


def test_sources_that_are_not_16_bit_pcm_keep_the_float_path(dev):
    from distant_speech_recognition_amd.btk20 import SampleFeaturePtr
    from tests.util import design_prototype, synthetic_pcm
    M, m, r, N = 512, 4, 1, 4
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, 80 * D, seed=9)
    pcm = pcm[0]
    for what in ("fraction", "overflow", "nan", "one_channel"):
        x = pcm.copy()
        if what == "fraction":
            x += 0.25
        elif what == "overflow":
            x[2, 1000] = 40000.0
        elif what == "nan":
            x[1, 77] = np.nan
        else:
            x[3, 5] = 0.5                                      # one source without a 16-bit view: the whole node stays float
        keep, bf, sfb = _graph(x, h, g, M, m, r, delays, 32)
        flags = [k.holds_pcm16() for k in keep[0::2]]
        assert not all(flags), what
        out = _pull(sfb)
        assert not bf.i16_stream() and bf.fused_path(), what
        assert out.size == 80 * D and (what == "nan" or np.isfinite(out).all())
    # the sample-level helpers: zeroMean leaves integers of the int16 range (feature.cc:556-570), randomize does not
    sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
    sf.set_samples(pcm[0])
    assert sf.holds_pcm16()
    sf.zeroMean()
    assert sf.holds_pcm16()
    sf.randomize(0, 10, 3.0)
    assert not sf.holds_pcm16()
    # overlapping blocks (shiftLen != blockLen) have no contiguous 16-bit view
    sf2 = SampleFeaturePtr(block_len=D, shift_len=D // 2, pad_zeros=True)
    sf2.set_samples(pcm[0])
    assert not sf2.holds_pcm16()


def _write_wav(path, x, fs=FS):
    x = np.asarray(x, np.int16)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + 2 * x.size) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, fs, 2 * fs, 2, 16))
        f.write(b"data" + struct.pack("<I", 2 * x.size) + x.tobytes())


def test_wav_sources_stream_as_16_bit_pcm(orc, dev, tmp_path):
    """the reference's own entry: SampleFeature.read(wav) per channel (unit_test/test_online_beamforming.py:80-88); norm == 0 gives
    int16-scale floats -> 16-bit stream; a normalised read (norm != 0) gives fractions -> float stream"""
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr,
                                                      OverSampledDFTSynthesisBankPtr)
    from tests.util import design_prototype, synthetic_pcm
    M, m, r, N, dct = 512, 4, 1, 4, 2
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, 90 * D + 31, seed=21)
    pcm = pcm[0]
    paths = []
    for c in range(N):
        paths.append(str(tmp_path / ("ch%d.wav" % c)))
        _write_wav(paths[-1], pcm[c])

    def run(norm):
        bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
        keep = []
        for c in range(N):
            sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
            sf.read(paths[c], FS, FS, 1, 1, 0, -1, -1, norm)
            a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=dct)
            a.set_block_frames(32)
            bf.set_channel(a)
            keep += [sf, a]
        bf.calc_gsc_weights(FS, delays)
        sfb = OverSampledDFTSynthesisBankPtr(bf, prototype=g, M=M, m=m, r=r, delay_compensation_type=dct)
        out = _pull(sfb)
        return out, bf.i16_stream(), keep

    out, was_i16, _ = run(0.0)
    assert was_i16
    wq, B, wl = orc.gsc_weights(M, N, FS, delays)
    ref, _ = orc.pipeline_gsc(h, g, M, m, r, dct, pcm, wq, wl)
    assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 0.5
    out_n, was_i16_n, _ = run(1.0)
    assert not was_i16_n
    assert np.max(np.abs(out_n * 32768.0 - ref)) < 0.5


def test_graph_pool_16_bit_streams_same_bits(dev):
    """SubbandGraphPool: G graphs as one S = G launch of btk_fb_analysis_bf_i16 == the same pool on float samples, ragged lengths"""
    from distant_speech_recognition_amd.btk20 import SubbandGraphPoolPtr
    from tests.util import design_prototype, synthetic_pcm, la_delays, ula_positions
    M, m, r, N = 512, 4, 1, 8
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    lens = [130 * D, 97 * D + 19, 64 * D]
    utts = [synthetic_pcm(1, N, L, seed=40 + i)[0][0] for i, L in enumerate(lens)]
    dels = [la_delays(ula_positions(N), a) for a in (-1.3, 0.2, 0.9)]

    def run():
        pool = SubbandGraphPoolPtr()
        keep, bfs = [], []
        for u, d in zip(utts, dels):
            k, bf, sfb = _graph(u, h, g, M, m, r, d, 48)
            pool.add(bf, sfb)
            keep.append((k, bf, sfb)); bfs.append(bf)
        outs = [[] for _ in utts]
        for blocks in pool:
            for i, b in enumerate(blocks):
                if b is not None:
                    outs[i].append(np.array(b))
        return [np.concatenate(o) for o in outs], [bf.i16_stream() for bf in bfs]

    o_i, f_i = run()
    with _float_path():
        o_f, f_f = run()
    assert all(f_i) and not any(f_f)
    for a, b in zip(o_i, o_f):
        assert float(np.max(np.abs(b))) > 100
        assert _same_bits(a, b), float(np.max(np.abs(a - b)))


def test_changing_the_samples_under_a_16_bit_stream_is_refused(dev):
    """a 16-bit stream reads the utterance where the source keeps it: new samples without a reset() of the graph raise
    jconsistency_error instead of mixing two utterances; after reset() the new utterance streams"""
    from tests.util import design_prototype, synthetic_pcm
    M, m, r, N = 512, 4, 1, 4
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, 200 * D, seed=5)
    pcm2, _ = synthetic_pcm(1, N, 120 * D, seed=6)
    keep, bf, sfb = _graph(pcm[0], h, g, M, m, r, delays, 32)
    it = iter(sfb)
    for _ in range(40):
        next(it)
    for c in range(N):
        keep[2 * c].set_samples(pcm2[0][c])
    with pytest.raises(Exception) as e:
        for _ in range(400):
            next(it)
    assert "16-bit" in str(e.value)
    sfb.reset()
    out = _pull(sfb)
    _, bf2, sfb2 = _graph(pcm2[0], h, g, M, m, r, delays, 32)
    ref = _pull(sfb2)
    assert bf.i16_stream() and _same_bits(out, ref)


class _copies_per_row:
    """BTK_NODE_GATHER=0: the rows of a block go up by one hipMemcpyAsync each instead of the gather kernel (read per upload)"""

    def __enter__(self):
        self.old = os.environ.get("BTK_NODE_GATHER")
        os.environ["BTK_NODE_GATHER"] = "0"

    def __exit__(self, *a):
        if self.old is None:
            del os.environ["BTK_NODE_GATHER"]
        else:
            os.environ["BTK_NODE_GATHER"] = self.old


@pytest.mark.parametrize("i16", [True, False])
@pytest.mark.parametrize("M,N,block_frames", [(512, 8, 37), (512, 64, 0), (1024, 8, 40)])
def test_gather_upload_same_bits_as_copies_per_row(dev, M, N, block_frames, i16):
    """the upload of a block's sample rows -- every row in its source's own pinned allocation -- by btk_gather_rows (one kernel that
    reads the host memory through a table) and by one copy per row: the same bits, 16-bit and float streams, ragged last blocks"""
    from tests.util import design_prototype, synthetic_pcm
    m, r = 4, 1
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, (70 if N <= 8 else 40) * D + 33, seed=5 + M + N)
    pcm = pcm[0]

    def run():
        _, bf, sfb = _graph(pcm, h, g, M, m, r, delays, block_frames)
        out = _pull(sfb)
        assert bf.i16_stream() == i16
        return out
    if i16:
        a = run()
        with _copies_per_row():
            b = run()
    else:
        with _float_path():
            a = run()
            with _copies_per_row():
                b = run()
    assert _same_bits(a, b)
