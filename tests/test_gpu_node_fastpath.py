"""GPU: the drop-in node API on the engine's fast path.  The reference's hot path is SubbandGSC::next() pulling
OverSampledDFTAnalysisBank::next() (beamformer/beamformer.cc:1251-1316 over modulated/modulated.cc:375-409, driven by
src/beamformerDS.cc:144-223).  A node graph of that shape with no snapshot consumer must run the FUSED analysis -> apply kernel
(btk_fb_analysis_bf), hand its block to the synthesis bank on the device, allocate nothing in steady state, and fall back to
the staged pair the moment somebody asks for the snapshots -- with the same results within the apply tolerance and the oracle's
PCM within 0.5 LSB."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "distant_speech_recognition_amd", "host", "examples", "node_api_bench")
FS = 16000


def _graph(pcm, h, g, M, m, r, delays, block_frames, dct=2, kind="gsc", staged=False):
    """SampleFeature x N -> OverSampledDFTAnalysisBank x N -> SubbandGSC -> OverSampledDFTSynthesisBank over in-memory channels."""
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr, SubbandDSPtr,
                                                      OverSampledDFTSynthesisBankPtr)
    D = M >> r
    keep = []
    bf = SubbandGSCPtr(fftlen=M, half_band_shift=False) if kind == "gsc" else SubbandDSPtr(fftlen=M, half_band_shift=False)
    for c in range(pcm.shape[0]):
        sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
        sf.set_samples(np.ascontiguousarray(pcm[c], np.float32))
        a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=dct)
        a.set_block_frames(block_frames)
        bf.set_channel(a)
        keep += [sf, a]
    if kind == "gsc":
        bf.calc_gsc_weights(FS, delays)
    else:
        bf.calc_array_manifold_vectors(FS, delays)
    if staged:
        # a snapshot consumer: from now on every block brings its snapshots along.  (Asked before anything is loaded -- a
        # SampleFeature drops its samples when it has handed out the last block, like the reference's: feature/feature.cc.)
        bf.want_snapshots()
    sfb = OverSampledDFTSynthesisBankPtr(bf, prototype=g, M=M, m=m, r=r, delay_compensation_type=dct)
    return keep, bf, sfb


def _pull(sfb):
    return np.concatenate([np.array(b) for b in sfb])


@pytest.mark.parametrize("M,N,block_frames", [(256, 4, 0), (256, 4, 48), (512, 8, 64), (512, 64, 0), (1024, 8, 40), (2048, 8, 0)])
def test_node_fused_equals_node_staged_and_oracle(orc, dev, M, N, block_frames):
    from tests.util import design_prototype, synthetic_pcm
    m, r, dct = 4, 1, 2
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    nfr = 150 if N <= 8 else 60
    pcm, delays = synthetic_pcm(1, N, nfr * D, seed=5 + M + N)
    pcm = pcm[0]
    _, bf_f, sfb_f = _graph(pcm, h, g, M, m, r, delays, block_frames)
    out_f = _pull(sfb_f)
    assert bf_f.fused_path() and not bf_f.snapshots_materialised(), "the fused node must never have built its snapshots"
    _, bf_s, sfb_s = _graph(pcm, h, g, M, m, r, delays, block_frames, staged=True)
    out_s = _pull(sfb_s)
    assert not bf_s.fused_path()
    assert bf_s.snapshots_materialised()
    assert out_f.shape == out_s.shape
    scale = float(np.max(np.abs(out_s)))
    assert scale > 100
    # fused and staged differ by the order of the beamformer sum (Z domain against per-channel spectra): apply tolerance 2e-6 sqrt(N)
    assert np.max(np.abs(out_f - out_s)) <= 2e-6 * np.sqrt(N) * 8 * scale + 0.02, float(np.max(np.abs(out_f - out_s)))
    wq, B, wl = orc.gsc_weights(M, N, FS, delays)
    ref, nbf = orc.pipeline_gsc(h, g, M, m, r, dct, pcm, wq, wl)
    assert ref.shape == out_f.shape
    assert np.max(np.abs(out_f - ref)) < 0.5 and np.max(np.abs(out_s - ref)) < 0.5      # <= 0.5 LSB at int16 scale


@pytest.mark.parametrize("M,N", [(256, 4), (512, 8)])
def test_fused_node_every_block_size_same_bits_incl_weight_change(dev, M, N):
    """the fused kernels are partition-exact in (t0, tcount): whole utterance == 16- / 24- / 50-frame blocks byte for byte, also
    when the look direction moves mid-stream (only the frames not yet pulled take the new weights: BlockSource::advance_to)"""
    from tests.util import design_prototype, synthetic_pcm, la_delays, ula_positions
    m, r = 4, 1
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, 260 * D, seed=77)
    pcm = pcm[0]
    d2 = la_delays(ula_positions(N), 0.4)
    outs = {}
    for bfr in (0, 16, 24, 50):
        _, bf, sfb = _graph(pcm, h, g, M, m, r, delays, bfr)
        blocks = []
        for i, b in enumerate(sfb):
            blocks.append(np.array(b))
            if i == 101:
                bf.calc_gsc_weights(FS, d2)
        assert bf.fused_path() and not bf.snapshots_materialised()
        outs[bfr] = np.concatenate(blocks)
    for bfr in (16, 24, 50):
        assert outs[bfr].shape == outs[0].shape
        assert np.array_equal(outs[bfr].view(np.uint32), outs[0].view(np.uint32)), (bfr, float(np.max(np.abs(outs[bfr] - outs[0]))))
    # and the change did change the signal from block 102 on
    _, _, sfb0 = _graph(pcm, h, g, M, m, r, delays, 0)
    same = _pull(sfb0)
    assert np.array_equal(same[:100 * D], outs[0][:100 * D]) and not np.allclose(same[110 * D:], outs[0][110 * D:])


def test_snapshot_consumer_mid_stream_switches_to_staged(orc, dev, proto256, kinect_pcm):
    """a caller that asks for the snapshots in the middle of a fused stream (what a post-filter, an adaptive canceller or a Python
    beamformer class does) gets those of the current block, and the stream goes on staged: output still the oracle's"""
    from tests.util import la_delays
    M, m, r, dct, D = 256, 4, 1, 2, 128
    h, g = proto256
    mpos = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
    delays = la_delays(mpos, -1.306379)
    pcm = kinect_pcm[:, :24000]
    _, bf, sfb = _graph(pcm, h, g, M, m, r, delays, 64)
    out = []
    X = np.stack([orc.analysis(h, M, m, r, dct, pcm[c]) for c in range(4)], axis=1)        # [T][N][M]
    for i, b in enumerate(sfb):
        out.append(np.array(b))
        if i == 70:
            assert bf.fused_path() and not bf.snapshots_materialised()
            Xd = bf.device_snapshots().cpu().numpy()[0]                                    # [K][N][T] of the current block
            base, T = bf.chunk_base(), bf.num_frames()
            assert not bf.fused_path() and bf.snapshots_materialised()
            ref = np.transpose(X[base:base + T, :, :M // 2 + 1], (2, 1, 0))
            assert Xd.shape == ref.shape and np.max(np.abs(Xd - ref)) <= 1e-5 * np.max(np.abs(ref))
    assert bf.snapshots_materialised()                                                     # ... and every later block brought them along
    out = np.concatenate(out)
    wq, B, wl = orc.gsc_weights(M, 4, FS, delays)
    ref, _ = orc.pipeline_gsc(h, g, M, m, r, dct, pcm, wq, wl)
    assert out.shape == ref.shape and np.max(np.abs(out - ref)) < 0.5


def test_steady_state_blocks_allocate_nothing(dev):
    """grow-only node-owned buffers: after the first blocks of a stream neither hipMalloc nor hipHostMalloc is called again"""
    from distant_speech_recognition_amd import btk20cpp
    from tests.util import design_prototype, synthetic_pcm
    M, m, r, N = 512, 4, 1, 8
    D = M >> r
    h, g = design_prototype(M, m), design_prototype(M, m, "g")
    pcm, delays = synthetic_pcm(1, N, 64 * 12 * D, seed=3)
    _, bf, sfb = _graph(pcm[0], h, g, M, m, r, delays, 64)
    counts = []
    for i, b in enumerate(sfb):
        if i % 64 == 63:
            counts.append(btk20cpp.node_alloc_counts())
    assert len(counts) >= 10
    assert counts[2] == counts[-1], counts              # blocks 3 .. 12: not one allocation
    assert bf.fused_path()


def test_bank_shared_by_beamformer_and_frame_puller_raises(dev, proto256, kinect_pcm):
    """a bank whose samples a beamformer node has released cannot also be pulled frame by frame (it would silently start later)"""
    from distant_speech_recognition_amd.btk20 import j_error
    from tests.util import la_delays
    M, m, r = 256, 4, 1
    h, g = proto256
    delays = la_delays(np.array([[-113.0, 0, 2], [36.0, 0, 2], [76.0, 0, 2], [113.0, 0, 2]]), 0.2)
    keep, bf, sfb = _graph(kinect_pcm[:, :20000], h, g, M, m, r, delays, 32)
    for i, b in enumerate(sfb):
        if i == 40:
            break
    bank = keep[1]
    with pytest.raises(j_error):
        bank.next()


def test_lefkimmiatis_lambda_is_designed_once_and_follows_the_rule(dev, proto256, kinect_pcm):
    """Lambda = d^H pinv(R) d is kept across blocks (same bits as one whole-utterance block) and the post-filter has its own
    set_svd_rule (SubbandMVDR::set_svd_rule's twin)"""
    from distant_speech_recognition_amd.btk20 import (SampleFeaturePtr, OverSampledDFTAnalysisBankPtr, SubbandGSCPtr,
                                                      LefkimmiatisPostFilterPtr, OverSampledDFTSynthesisBankPtr)
    from tests.util import la_delays
    M, m, r, D = 256, 4, 1, 128
    h, g = proto256
    mpos = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
    delays = la_delays(mpos, -1.306379)

    def run(block_frames, rule):
        keep = []
        bf = SubbandGSCPtr(fftlen=M, half_band_shift=False)
        for c in range(4):
            sf = SampleFeaturePtr(block_len=D, shift_len=D, pad_zeros=True)
            sf.set_samples(np.ascontiguousarray(kinect_pcm[c, :30000], np.float32))
            a = OverSampledDFTAnalysisBankPtr(sf, prototype=h, M=M, m=m, r=r, delay_compensation_type=2)
            a.set_block_frames(block_frames)
            bf.set_channel(a)
            keep += [sf, a]
        bf.calc_gsc_weights(FS, delays)
        pf = LefkimmiatisPostFilterPtr(bf, M, 1.0e-4, 100, 0.8, 2)
        pf.set_diffuse_noise_model(mpos, FS, 343740.0)
        pf.set_all_diagonal_loading(0.1)
        pf.calc_inverse_noise_spatial_spectral_matrix()
        pf.set_beamformer(bf)
        pf.set_svd_rule(rule)
        assert pf.svd_rule() == rule
        sfb = OverSampledDFTSynthesisBankPtr(pf, prototype=g, M=M, m=m, r=r, delay_compensation_type=2)
        return _pull(sfb)

    a = run(0, "linpack")
    b = run(64, "linpack")
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c = run(64, "exact")
    assert c.shape == a.shape and np.max(np.abs(c - a)) < 2e-3 * np.max(np.abs(a)) + 0.5


def test_node_api_bench_binary_runs_and_reports(dev, tmp_path):
    """host/examples/node_api_bench (what bench.py's stages.node_api runs) on a small job: single graphs and the pool give the same
    checksum, and the timed pass allocates nothing"""
    from tests.util import design_prototype
    M, m, r, N = 512, 4, 1, 8
    coeffs = str(tmp_path / "coeffs.f64")
    np.concatenate([design_prototype(M, m), design_prototype(M, m, "g")]).astype(np.float64).tofile(coeffs)
    res = {}
    for pool in (0, 1):
        out = subprocess.run([BENCH, coeffs, str(M), str(m), str(r), str(N), "600", "3", "128", str(pool)], capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr
        res[pool] = json.loads(out.stdout.strip().splitlines()[-1])
    assert res[0]["output_blocks"] == res[1]["output_blocks"] == 3 * 600
    assert res[0]["hipMalloc_in_timed_pass"] == 0 and res[1]["hipMalloc_in_timed_pass"] == 0
    assert res[0]["hipHostMalloc_in_timed_pass"] == 0 and res[1]["hipHostMalloc_in_timed_pass"] == 0
    assert abs(res[0]["checksum"] - res[1]["checksum"]) <= 1e-3 * (1 + abs(res[0]["checksum"]))
    assert res[1]["rounds"] == 5 and res[0]["frames_per_s"] > 0
