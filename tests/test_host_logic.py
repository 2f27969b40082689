"""CPU: host-side logic of the mirror that needs no GPU -- SampleFeature block reader semantics, delay calculators
against the reference-python goldens, VAD labelling, the tools' configuration handling."""
import json
import os
import pickle
import sys
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _wav(path, x, fs=16000):
    w = wave.open(str(path), "wb")
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(fs)
    w.writeframes(np.asarray(x, np.int16).tobytes())
    w.close()


def test_sample_feature_block_reader(tmp_path):
    """SampleFeature::next (feature/feature.cc:605-649): un-normalised int16 -> float, pad_zeros, the `cur + size >= total`
    end rule (a last block that fits exactly is treated as the padded tail), explicit frame numbers, end of samples."""
    from distant_speech_recognition_amd.btk20 import SampleFeaturePtr
    from distant_speech_recognition_amd.btk20 import jindex_error
    x = (np.arange(1000) - 500).astype(np.int16)
    p = tmp_path / "a.wav"
    _wav(p, x)
    sf = SampleFeaturePtr(block_len=128, shift_len=128, pad_zeros=True)
    assert sf.read(str(p), 16000) == 1000
    blocks = [np.array(b) for b in sf]
    assert len(blocks) == 8                                            # ceil(1000/128)
    cat = np.concatenate(blocks)
    assert np.array_equal(cat[:1000], x.astype(np.float32)) and np.all(cat[1000:] == 0)
    with pytest.raises(StopIteration):
        sf.next()
    # without padding the tail (and an exactly fitting last block) is dropped
    sf = SampleFeaturePtr(block_len=128, shift_len=128, pad_zeros=False)
    sf.read(str(p), 16000)
    assert len([1 for _ in sf]) == 7
    _wav(tmp_path / "b.wav", x[:1024])
    sf.read(str(tmp_path / "b.wav"), 16000)
    assert len([1 for _ in sf]) == 7                                   # 8 * 128 == 1024: cur + size >= total on the 8th
    # overlapping blocks, same-frame caching, out-of-order frame numbers
    sf = SampleFeaturePtr(block_len=320, shift_len=160, pad_zeros=True)
    sf.read(str(p), 16000)
    b0 = np.array(sf.next())
    assert np.array_equal(np.array(sf.next(0)), b0)
    b1 = np.array(sf.next(1))
    assert np.array_equal(b1[:160], b0[160:])
    with pytest.raises(jindex_error):
        sf.next(5)


def test_delay_calculators_match_reference_python(pygolden):
    """calc_la_delays and friends (lib/pybeamformer.py:41-153) against values produced by the reference's own module"""
    from distant_speech_recognition_amd import pybeamformer as pb
    mpos = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
    assert np.allclose(pb.calc_delays("linear", mpos, [-1.306379, None, None]), pygolden["delays_kinect"], rtol=0, atol=1e-18)
    d8 = pb.calc_la_delays(np.array([[20.0 * (i - 3.5), 0, 0] for i in range(8)]), 0.7)
    assert np.allclose(d8, pygolden["delays_ula8"], rtol=0, atol=1e-18)
    vs = pb.calc_array_manifold_f(5, 256, 16000, pygolden["delays_kinect"], False)
    assert np.allclose(vs, pygolden["manifold_k5"], rtol=1e-15)
    B = pb.calc_blocking_matrix(vs, 1)
    assert np.allclose(B, pygolden["blockmat_k5_nc1"], atol=1e-14)
    B2 = pb.calc_blocking_matrix(pb.calc_array_manifold_f(77, 256, 16000, pygolden["delays_kinect"], False), 2)
    assert np.allclose(B2, pygolden["blockmat_k77_nc2"], atol=1e-14)


def test_vad_labelling_walks_segments_like_the_reference():
    """the target/noise decision of accu_stats_from_label (pybeamformer.py:967-985) restated literally: inclusive bounds;
    labx advances as soon as `elapsed > end` -- which is also true for an open end (-1) that has not started yet, so an
    open-ended segment is only honoured if it is active from the first frame it is looked at"""
    from distant_speech_recognition_amd.pybeamformer import _vad_noise_label
    dt = 0.008

    def literal(T, labs):
        el, labx, out = 0.0, 0, []
        for _ in range(T):
            tgt = False
            if labx < len(labs):
                if el >= labs[labx][0] and (el <= labs[labx][1] or labs[labx][1] < 0):
                    tgt = True
                elif el > labs[labx][1]:
                    labx += 1
            out.append(1.0 if tgt else 0.0)
            el += dt
        return np.array(out, np.float32)

    for labs in ([(0.5, 1.0), (1.6, 2.0)], [(0.5, 1.0), (1.6, -1)], [(0.0, -1)], [(0.1, -1)], [], [(0.2, 0.4), (0.3, 0.9)]):
        assert np.array_equal(_vad_noise_label(300, dt, labs), literal(300, labs)), labs
    lab = _vad_noise_label(300, dt, [(0.5, 1.0), (1.6, 2.0)])
    assert lab[0] == 0 and lab[80] == 1 and lab[150] == 0 and lab[220] == 1 and lab[299] == 0
    assert np.all(_vad_noise_label(300, dt, [(0.0, -1)]) == 1)
    assert np.all(_vad_noise_label(300, dt, [(0.1, -1)]) == 0)          # the quirk: skipped before it starts


def test_tools_configuration_handling(tmp_path):
    sys.path.insert(0, ROOT)
    from tools import online_beamforming as ob
    from tools import sos_batch_beamforming as sb
    conf = {"array_type": "linear", "microphone_positions": [[0, 0, 0], [1, 0, 0]],
            "target": {"positions": [[0.0, [0.1, None, None]], [1.0, [0.2, None, None]]]},
            "noises": [{"positions": [[0.0, [1.0, None, None]], [1.0, [1.1, None, None]]]}], "beamformer": {"type": "lcmv"}}
    ob.check_position_data_format(conf)
    bad = json.loads(json.dumps(conf))
    bad["noises"][0]["positions"][1][0] = 2.0
    with pytest.raises(AssertionError):
        ob.check_position_data_format(bad)
    bad = json.loads(json.dumps(conf))
    bad["array_type"] = "planar"
    bad["target"]["positions"][0][1] = [0.1]
    with pytest.raises(AssertionError):
        ob.check_position_data_format(bad)
    # prototypes: the reference's pickles (numpy arrays) and this repo's npz
    h = np.arange(8, dtype=np.float64)
    with open(tmp_path / "h.pickle", "wb") as fp:
        pickle.dump(h, fp, protocol=2)
    assert np.array_equal(ob.load_prototype(str(tmp_path / "h.pickle"), "h"), h)
    np.savez(tmp_path / "p.npz", h=h, g=2 * h)
    assert np.array_equal(ob.load_prototype(str(tmp_path / "p.npz"), "g"), 2 * h)
    # TF masks: a stream of pickled rows (the reference's format) or npy
    rows = (np.random.default_rng(0).random((5, 9)) > 0.5).astype(float)
    with open(tmp_path / "m.pickle", "wb") as fp:
        for r in rows:
            pickle.dump(r, fp, protocol=2)
    assert np.array_equal(sb.load_tfmask(str(tmp_path / "m.pickle")), rows)
    np.save(tmp_path / "m.npy", rows)
    mt, mj = sb.load_tfmasks({"target": {"tfmask_path": str(tmp_path / "m.npy")},
                              "noises": [{"tfmask_path": str(tmp_path / "m.pickle")}, {"tfmask_path": str(tmp_path / "m.npy")}]})
    assert np.array_equal(mt, rows) and np.allclose(mj, rows)


def test_padded_rows_layout():
    """engine.padded_rows: rows a multiple of 4 KiB long get 48 elements of padding (a [..., :T] view), other lengths stay
    contiguous; _row_stride only takes device tensors (no CPU path)."""
    import torch
    from distant_speech_recognition_amd import engine as eng
    y = eng.padded_rows((2, 257, 4096), torch.complex64, "cpu")            # 4096 * 8 B = 32 KiB rows
    assert y.shape == (2, 257, 4096) and y.stride() == (257 * 4144, 4144, 1) and not y.is_contiguous()
    z = eng.padded_rows((2, 257, 1000), torch.complex64, "cpu")
    assert z.is_contiguous()
    e = eng.padded_rows((3, 0), torch.float32, "cpu")
    assert e.shape == (3, 0)
    f = eng.padded_rows((4, 1024), torch.float32, "cpu")                   # 4 KiB rows of floats
    assert f.stride() == (1072, 1)
    with pytest.raises(ValueError):
        eng._row_stride(y, "Y")


def test_spectral_matrix_array_recursion(tmp_path):
    """SpectralMatrixArray (beamformer/spectralinfoarray.h:43-64, beamformer.cc:122-143): R_k <- mu R_k + (1 - mu) x_k x_k^T, the
    outer product WITHOUT conjugation -- Python mirror and C++ node layer against the formula."""
    import subprocess
    from distant_speech_recognition_amd.btk20 import SpectralMatrixArrayPtr
    M, N, mu = 8, 3, 0.9
    rng = np.random.default_rng(11)
    frames = rng.standard_normal((4, N, M)) + 1j * rng.standard_normal((4, N, M))
    arr = SpectralMatrixArrayPtr(M, N, mu)
    R = np.zeros((M, N, N), np.complex128)
    muf = float(np.float32(mu))
    for f in frames:
        for c in range(N):
            arr.set_samples(f[c], c)
        arr.update()
        for k in range(M):
            R[k] = muf * R[k] + (1.0 - muf) * np.outer(f[:, k], f[:, k])          # no conjugate
    for k in range(M):
        assert np.max(np.abs(arr.matrix_f(k) - R[k])) < 1e-14
        assert np.array_equal(arr.snapshot(k), frames[-1][:, k])
    assert np.max(np.abs(arr.matrix_f(2) - arr.matrix_f(2).T)) < 1e-15 and np.max(np.abs(arr.matrix_f(2).imag)) > 1e-3   # symmetric, not Hermitian
    arr.zero()
    assert not arr.matrix_f(0).any() and not arr.snapshot(0).any()
    # the C++ class: same numbers from a small host program linked against the node layer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    host = os.path.join(root, "distant_speech_recognition_amd", "host")
    if not os.path.exists(os.path.join(host, "libbtk20hip.so")):
        pytest.skip("node layer not built")
    src = tmp_path / "sma.cc"
    src.write_text(r'''
#include "beamformer/beamformer.h"
#include <cstdio>
int main() {
  const unsigned M = 8, N = 3;
  SpectralMatrixArrayPtr arr = new SpectralMatrixArray(M, N, 0.9f);
  gsl_vector_complex* v = gsl_vector_complex_calloc(M);
  for (int f = 0; f < 4; f++) {
    for (unsigned c = 0; c < N; c++) {
      for (unsigned k = 0; k < M; k++) gsl_vector_complex_set(v, k, gsl_complex_rect(0.1 * (f + 1) * (c + 1) + k, 0.3 * k - 0.2 * c + f));
      arr->set_samples(v, c);
    }
    arr->update();
  }
  for (unsigned k = 0; k < M; k++)
    for (unsigned i = 0; i < N; i++)
      for (unsigned j = 0; j < N; j++) {
        gsl_complex z = gsl_matrix_complex_get(arr->matrix_f(k), i, j);
        printf("%.17g %.17g\n", GSL_REAL(z), GSL_IMAG(z));
      }
  gsl_vector_complex_free(v);
  return 0;
}
''')
    exe = tmp_path / "sma"
    csrc = os.path.join(root, "distant_speech_recognition_amd", "csrc")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(host, "include"), "-I" + os.path.join(root, "include"), str(src), "-o", str(exe),
                    "-L" + host, "-L" + csrc, "-lbtk20hip", "-lbtkhip", "-Wl,-rpath," + host, "-Wl,-rpath," + csrc], check=True, timeout=120)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60, check=True).stdout.split()
    got = (np.array(out[0::2], np.float64) + 1j * np.array(out[1::2], np.float64)).reshape(M, N, N)
    R = np.zeros((M, N, N), np.complex128)
    for f in range(4):
        x = np.array([[0.1 * (f + 1) * (c + 1) + k + 1j * (0.3 * k - 0.2 * c + f) for k in range(M)] for c in range(N)])
        for k in range(M):
            R[k] = muf * R[k] + (1.0 - muf) * np.outer(x[:, k], x[:, k])
    assert np.max(np.abs(got - R)) < 1e-12


def test_sample_feature_full_class(tmp_path):
    """The rest of SampleFeature (feature/feature.h:153-206, feature.cc:391-680): 16-bit write() / read() round trip, cut (both bounds
    inclusive), zeroMean (int16 clamp, truncation toward zero), copySamples' `to - cfrom` count, setSamples / data / dataDouble,
    getChanN, exit(), normalised reads (libsndfile's x / 32768 times norm) and the write() that undoes them."""
    from distant_speech_recognition_amd.btk20 import SampleFeaturePtr, j_error
    rng = np.random.default_rng(3)
    x = rng.integers(-20000, 20000, 4000).astype(np.int16)
    p = tmp_path / "a.wav"
    _wav(p, x, fs=8000)
    sf = SampleFeaturePtr(block_len=256, shift_len=256, pad_zeros=True)
    assert sf.read(str(p)) == 4000 and sf.getSampleRate() == 8000 and sf.getChanN() == 1
    assert np.array_equal(sf.data(), x.astype(np.float32)) and sf.dataDouble().dtype == np.float64
    # write -> read is the identity on int16 values; the file keeps the sample rate
    sf.write(str(tmp_path / "b.wav"))
    w = wave.open(str(tmp_path / "b.wav"))
    assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 8000, 4000)
    assert np.array_equal(np.frombuffer(w.readframes(4000), np.int16), x)
    with pytest.raises(j_error):
        sf.write(str(tmp_path / "c.au"), format=0x030002)              # only WAV | PCM_16
    # cut: samples cfrom..cto inclusive
    sf.cut(cfrom=100, cto=1099)
    assert sf.samplesN() == 1000 and np.array_equal(sf.data(), x[100:1100].astype(np.float32))
    with pytest.raises(j_error):
        sf.cut(5, 5)
    with pytest.raises(j_error):
        sf.cut(0, 1000)
    # zeroMean: x - mean clamped to int16 and truncated toward zero
    m = float(np.mean(x[100:1100].astype(np.float64)))
    sf.zeroMean()
    want = np.trunc(np.clip(x[100:1100].astype(np.float32).astype(np.float64) - m, -32768, 32767)).astype(np.float32)
    assert np.array_equal(sf.data(), want)
    # copySamples: `to - cfrom` samples from cfrom (the reference's count), to == 0 copies everything
    src = SampleFeaturePtr(block_len=256, shift_len=256)
    src.read(str(p))
    dst = SampleFeaturePtr(block_len=256, shift_len=256)
    dst.copySamples(src, cfrom=10, to=110)
    assert dst.samplesN() == 100 and np.array_equal(dst.data(), x[10:110].astype(np.float32))
    dst.copySamples(src, 0, 0)
    assert dst.samplesN() == 4000
    # setSamples takes doubles and a sample rate, and the node iterates over them
    dst.setSamples(np.arange(600, dtype=np.float64), sampleRate=44100)
    assert dst.getSampleRate() == 44100 and np.array_equal(np.concatenate([np.array(b) for b in dst])[:512], np.arange(512, dtype=np.float32))
    with pytest.raises(StopIteration):
        dst.exit()
    # normalised read (norm != 0): x / 32768 * norm, and write() maps it back to the same int16 values
    nf = SampleFeaturePtr(block_len=256, shift_len=256)
    nf.read(str(p), norm=1.0)
    assert np.allclose(nf.data(), x / 32768.0, rtol=0, atol=1e-7)
    nf.write(str(tmp_path / "n.wav"))
    back = np.frombuffer(wave.open(str(tmp_path / "n.wav")).readframes(4000), np.int16)
    assert np.max(np.abs(back.astype(int) - x.astype(int))) <= 1        # 0x7FFF vs 0x8000 scale, as libsndfile


def test_sample_feature_randomize_is_gsl_mt19937_gaussian():
    """randomize() (feature.cc:589-603) draws gsl_ran_gaussian from gsl_rng_default = mt19937 with GSL's default seed; GSL's manual
    gives the known answer of that generator ("generator type: mt19937, seed = 0, first value = 4293858116"); the polar Box-Muller
    on top is restated here from the published algorithm."""
    from distant_speech_recognition_amd.btk20 import SampleFeaturePtr
    bg = np.random.MT19937()
    bg._legacy_seeding(4357)                                            # GSL: seed 0 means 4357
    raw = bg.random_raw(400).astype(np.float64)
    assert int(raw[0]) == 4293858116
    u = iter(raw / 4294967296.0)
    want = []
    while len(want) < 50:
        xx, yy = -1 + 2 * next(u), -1 + 2 * next(u)
        r2 = xx * xx + yy * yy
        if r2 > 1.0 or r2 == 0:
            continue
        want.append(3.0 * yy * np.sqrt(-2.0 * np.log(r2) / r2))
    sf = SampleFeaturePtr(block_len=16, shift_len=16)
    sf.setSamples(np.zeros(100), 16000)
    sf.randomize(startX=20, endX=69, sigma2=3.0)
    d = sf.data()
    assert np.all(d[:20] == 0) and np.all(d[70:] == 0)
    assert np.allclose(d[20:70], np.array(want, np.float32), rtol=1e-6, atol=0)


def test_sample_feature_add_white_noise_is_the_references_arithmetic():
    """addWhiteNoise (feature.cc:391-427) draws its noise into SHORT integers, so every noise sample is 0 or one negative level;
    the mean absolute noise equals the level the SNR asks for (up to the truncation of that one level to a short)."""
    from distant_speech_recognition_amd.btk20 import SampleFeaturePtr
    x = np.full(20000, 1000.0)
    sf = SampleFeaturePtr(block_len=16, shift_len=16)
    sf.setSamples(x, 16000)
    sf.addWhiteNoise(snr=20.0)
    n = sf.data().astype(np.float64) - x
    vals = np.unique(n)
    assert len(vals) == 2 and vals[1] == 0 and vals[0] < 0
    assert abs(np.mean(np.abs(n)) - 100.0) <= 100.0 * (1.0 / abs(vals[0])) + 1e-9     # |level| truncated by < 1


def test_kwargs_are_the_swig_interface_names():
    """%feature("kwargs") in btk20_src/*/*.i: the parameter names a script may use are the ones the .i files declare (the table
    is generated from them, tools/gen_swig_signatures.py); the spellings earlier versions used still work, and a keyword neither
    table knows reaches the binding's own py::arg names instead of raising here."""
    from distant_speech_recognition_amd import btk20
    from distant_speech_recognition_amd.btk20cpp import _signatures as S
    assert S.METHODS["SubbandGSCPtr"]["set_active_weights_f"] == [("fbinX", "required"), ("packedWeight", "required")]
    assert S.METHODS["MultiChannelWPEDereverberationPtr"]["estimate_filter"] == [("start_frame_no", 0), ("frame_num", -1)]
    assert [p for p, _ in S.METHODS["SubbandGSCPtr"]["calc_gsc_weights_2"]] == ["samplerate", "delaysT", "delaysJ"]
    assert [p for p, _ in S.METHODS["SnapShotArrayPtr"]["set_samples"]] == ["samp", "chanX"]
    a = btk20.SnapShotArrayPtr(fftlen=8, chan_num=2)
    a.set_samples(samp=np.arange(8) + 1j, chanX=1)                      # the .i names
    a.set_samples(samp=np.arange(8) + 2j, chan_no=0)                    # the earlier spelling
    a.update()
    assert a.snapshot(fbinX=3)[1] == 3 + 1j and a.snapshot(fbin_no=3)[0] == 3 + 2j
    sf = btk20.SampleFeaturePtr(block_len=4, shift_len=4)
    sf.setSamples(samples=np.arange(8.0), sampleRate=8000)
    with pytest.raises(TypeError):
        sf.setSamples(samples=np.arange(8.0), no_such_name=1)


def test_snapshot_array_set_snapshots_and_legacy_aliases():
    """SnapShotArray::set_snapshots (beamformer.cc:79-93) and the ENABLE_LEGACY_BTK_API aliases getSnapShot / newSample
    (spectralinfoarray.h:17, 26-27): one bin's snapshot written directly, its conjugate at bin fftLen/2 - fbinX as the reference does."""
    from distant_speech_recognition_amd import btk20
    M, N = 8, 3
    a = btk20.SnapShotArrayPtr(M, N)
    x = np.array([1 + 2j, -3 + 0.5j, 0.25 - 1j])
    a.set_snapshots(x, 1)
    assert np.array_equal(a.snapshot(1), x) and np.array_equal(a.getSnapShot(M // 2 - 1), np.conj(x))
    a.set_snapshots(2 * x, 0)
    assert np.array_equal(a.snapshot(0), 2 * x) and np.array_equal(a.snapshot(M // 2), np.zeros(N))       # bins 0 and fftLen/2: no mirror
    with pytest.raises(Exception):
        a.set_snapshots(x, M // 2 + 1)
    for c in range(N):
        a.newSample(np.arange(M) * (c + 1) + 1j * c, c)
    a.update()
    assert np.array_equal(a.getSnapShot(5), np.array([5 * (c + 1) + 1j * c for c in range(N)]))


def test_bench_measurement_records_match_the_kernel_sources():
    """bench.py quotes two records that were taken on particular sources of the headline kernel -- the ISA count of its interior loop
    (FUSED_ISA) and the PMC traffic passes (profiles/<TRAFFIC_JSON>) -- and refuses either (null in the line) when the sources have
    changed since.  A checkout whose records are stale should fail HERE, not print a line with holes at round end."""
    import json
    import bench
    from bench_util import kernel_source_sha
    sha = kernel_source_sha()
    assert bench.FUSED_ISA["kernel_source_sha256"] == sha, "re-count with tools/isa_loop_count.py and update bench.FUSED_ISA"
    j = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", bench.TRAFFIC_JSON)))
    assert j["kernel_source_sha256"] == sha, "re-take the PMC passes (profiles/scripts/r06_run_all.sh) and commit profiles/%s" % bench.TRAFFIC_JSON
    assert (j["S"], j["T"], j["N"], j["M"]) == (32, 4096, 64, 512)
