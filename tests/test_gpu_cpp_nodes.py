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


@pytest.mark.parametrize("pf,alpha", [(0, 0.0), (2, 0.7)])
def test_beamformer_ds_binary_matches_oracle(orc, dev, tmp_path, proto256, kinect_pcm, pf, alpha):
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
    res = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    out = np.fromfile(str(tmp_path / "out.f32"), np.float32)
    # oracle: the C++ mains use delayCompensationType = 0 (src/beamformerDS.cc:155,172)
    X = np.stack([orc.analysis(h, M, m, r, 0, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    wq, B, wl = orc.gsc_weights(M, 4, FS, delays)
    Y = orc.gsc_frames(X, wq, wl)
    if pf:
        Y, _ = orc.zelinski_frames(X, Y, wq, alpha, pf)
    ref = orc.synthesis(g, M, m, r, 0, Y)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 0.5            # <= 0.5 LSB at int16 scale
