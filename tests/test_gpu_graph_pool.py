"""GPU: SubbandGraphPool (host/include/beamformer/beamformer.h) -- many utterance graphs advanced as ONE launch.  The reference
builds one graph per utterance (unit_test/test_online_beamforming.py:80-88); pulled one by one, G utterances are G independent
S = 1 launch sequences.  The pool's results must be those of the graphs pulled on their own."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
FS = 16000


def _graph(pcm, h, g, M, m, r, delays, block_frames, dct=2):
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr, OverSampledDFTSynthesisBankPtr)
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
    sfb = OverSampledDFTSynthesisBankPtr(bf, prototype=g, M=M, m=m, r=r, delay_compensation_type=dct)
    return keep, bf, sfb


@pytest.mark.parametrize("M,N,block_frames,dct", [(256, 4, 32, 2), (512, 8, 64, 0), (512, 8, 0, 2), (1024, 8, 48, 2)])
def test_pool_equals_graphs_pulled_one_by_one(dev, M, N, block_frames, dct):
    """three utterances of different lengths and look directions: per graph the pool's blocks are the single graph's, bit for bit
    (same kernels, same frames; an utterance that ends early just stops yielding), in ceil(frames / block) launches instead of 3 x"""
    from distant_speech_recognition_amd.btk20 import SubbandGraphPoolPtr
    from tests.util import design_prototype, synthetic_pcm, la_delays, ula_positions
    m, r = 4, 1
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    lens = [150 * D, 97 * D + 13, 201 * D]
    az = [-1.306379, 0.3, 1.1]
    singles, keepalive = [], []
    pool = SubbandGraphPoolPtr()
    for i in range(3):
        pcm, _ = synthetic_pcm(1, N, lens[i], seed=100 + i)
        delays = la_delays(ula_positions(N), az[i])
        k1, bf1, sfb1 = _graph(pcm[0], h, g, M, m, r, delays, block_frames, dct)
        singles.append(np.concatenate([np.array(b) for b in sfb1]))
        k2, bf2, sfb2 = _graph(pcm[0], h, g, M, m, r, delays, block_frames, dct)
        pool.add(bf2, sfb2)
        keepalive += [k1, bf1, sfb1, k2, bf2, sfb2]
    assert len(pool) == 3
    got = [[] for _ in range(3)]
    for outs in pool:
        assert len(outs) == 3
        for i, o in enumerate(outs):
            if o is not None:
                got[i].append(np.array(o))
    for i in range(3):
        assert pool.is_end(i)
        y = np.concatenate(got[i])
        assert y.shape == singles[i].shape, (i, y.shape, singles[i].shape)
        assert np.max(np.abs(singles[i])) > 100
        assert np.array_equal(y.view(np.uint32), singles[i].view(np.uint32)), (i, float(np.max(np.abs(y - singles[i]))))
    if block_frames:
        assert pool.rounds() <= -(-(201 + 16) // block_frames) + 1
    # a second pass over the same pool (reset through __iter__) after reloading the samples gives the same blocks
    for i in range(3):
        pcm, _ = synthetic_pcm(1, N, lens[i], seed=100 + i)
        for c in range(N):
            keepalive[6 * i + 3][2 * c].set_samples(np.ascontiguousarray(pcm[0][c], np.float32))
    again = [[] for _ in range(3)]
    for outs in pool:
        for i, o in enumerate(outs):
            if o is not None:
                again[i].append(np.array(o))
    for i in range(3):
        assert np.array_equal(np.concatenate(again[i]).view(np.uint32), singles[i].view(np.uint32))
    # a third pass with the rows of a round going up by one copy each instead of the gather kernel (BTK_NODE_GATHER=0): same blocks
    import os
    old = os.environ.get("BTK_NODE_GATHER")
    os.environ["BTK_NODE_GATHER"] = "0"
    try:
        for i in range(3):
            pcm, _ = synthetic_pcm(1, N, lens[i], seed=100 + i)
            for c in range(N):
                keepalive[6 * i + 3][2 * c].set_samples(np.ascontiguousarray(pcm[0][c], np.float32))
        third = [[] for _ in range(3)]
        for outs in pool:
            for i, o in enumerate(outs):
                if o is not None:
                    third[i].append(np.array(o))
    finally:
        if old is None:
            del os.environ["BTK_NODE_GATHER"]
        else:
            os.environ["BTK_NODE_GATHER"] = old
    for i in range(3):
        assert np.array_equal(np.concatenate(third[i]).view(np.uint32), singles[i].view(np.uint32))

    # a fourth pass a ROUND at a time (next_round: engine extension): per graph the same blocks, in far fewer calls
    for i in range(3):
        pcm, _ = synthetic_pcm(1, N, lens[i], seed=100 + i)
        for c in range(N):
            keepalive[6 * i + 3][2 * c].set_samples(np.ascontiguousarray(pcm[0][c], np.float32))
    pool.reset()
    fourth, calls = [[] for _ in range(3)], 0
    while True:
        outs = pool.next_round()
        if outs is None:
            break
        calls += 1
        assert len(outs) == 3
        for i, a in enumerate(outs):
            assert a.dtype == np.float32 and a.ndim == 2
            if a.shape[0]:
                assert a.shape[1] == D
                fourth[i].append(a)
    for i in range(3):
        assert pool.is_end(i)
        assert np.array_equal(np.concatenate(fourth[i]).reshape(-1).view(np.uint32), singles[i].view(np.uint32))
    assert calls <= pool.rounds() + 1 and (block_frames == 0 or calls < singles[2].size // D // 8)


def test_pool_refuses_what_it_cannot_batch(dev):
    from distant_speech_recognition_amd.btk20 import SubbandGraphPoolPtr, j_error
    from tests.util import design_prototype, synthetic_pcm
    pool = SubbandGraphPoolPtr()
    M, m, r = 512, 4, 1
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, 8, 40 * 256, seed=1)
    k1, bf1, s1 = _graph(pcm[0], h, g, M, m, r, delays, 32)
    pool.add(bf1, s1)
    pcm4, d4 = synthetic_pcm(1, 4, 40 * 256, seed=2)
    k2, bf2, s2 = _graph(pcm4[0], h, g, M, m, r, d4, 32)
    with pytest.raises(j_error):
        pool.add(bf2, s2)                                   # another channel count
    k3, bf3, s3 = _graph(pcm[0], h, g, M, m, r, delays, 16)
    with pytest.raises(j_error):
        pool.add(bf3, s3)                                   # another block size: the graphs would not advance in lock step
    h128, g128 = design_prototype(128, 2), design_prototype(128, 2, "g")
    k4, bf4, s4 = _graph(pcm[0][:, :4000], h128, g128, 128, 2, 0, delays, 32)
    with pytest.raises(j_error):
        SubbandGraphPoolPtr().add(bf4, s4)                  # a geometry without a fused kernel
