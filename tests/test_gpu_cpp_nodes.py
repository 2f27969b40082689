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
