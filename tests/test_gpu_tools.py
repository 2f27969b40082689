"""GPU: tools/online_beamforming.py (the application-level counterpart of the reference's
unit_test/test_online_beamforming.py) driven with JSON configurations in the reference's schema."""
import json
import os
import subprocess
import sys
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "online_beamforming.py")
MPOS = [[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]]
M, m, r, D, FS, L = 256, 4, 1, 128, 16000, 40000


@pytest.fixture(scope="module")
def wavs(tmp_path_factory, kinect_pcm):
    d = tmp_path_factory.mktemp("toolwav")
    paths = []
    for c in range(4):
        p = str(d / ("c%d.wav" % (c + 1)))
        w = wave.open(p, "wb")
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(FS)
        w.writeframes(kinect_pcm[c][:L].astype(np.int16).tobytes())
        w.close()
        paths.append(p)
    return paths


def _run(tmp_path, wavs, conf, name):
    cpath, opath = str(tmp_path / (name + ".json")), str(tmp_path / (name + ".wav"))
    json.dump(conf, open(cpath, "w"))
    res = subprocess.run([sys.executable, TOOL, "-q", "-c", cpath, "-o", opath, "-i"] + wavs, capture_output=True, text=True,
                         timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "No. frames processed" in res.stdout
    w = wave.open(opath, "rb")
    out = np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(np.float64)
    w.close()
    return out


def _conf(bf, pf=None, positions=None, noises=None):
    c = {"array_type": "linear", "microphone_positions": MPOS,
         "target": {"positions": positions or [[0.0, [-1.306379, None, None]]]}, "beamformer": bf}
    if pf:
        c["postfilter"] = pf
    if noises:
        c["noises"] = noises
    return c


def test_tool_static_configs_match_oracle(orc, dev, proto256, kinect_pcm, wavs, tmp_path):
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, g = proto256
    X = np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    delays = calc_delays("linear", MPOS, [-1.306379, None, None])
    wq = orc.calc_mainlobe(M, 4, FS, delays)
    # confs/ds_and_zelinski.json
    out = _run(tmp_path, wavs, _conf({"type": "delay_and_sum"}, {"type": "zelinski", "subtype": 2, "alpha": 0.7}), "dsz")
    Yf, _ = orc.zelinski_frames(X, orc.gsc_frames(X, wq, np.zeros_like(wq)), wq, 0.7, 2)
    ref = orc.synthesis(g, M, m, r, 2, Yf)
    assert out.shape == ref.shape and np.max(np.abs(out - np.trunc(ref))) <= 1.0          # int16 truncation of the writer
    # confs/gscrls.json (defaults) and lcmv with one jammer run through the same script
    out_rls = _run(tmp_path, wavs, _conf({"type": "gscrls", "min_frames": 32}), "rls")
    o = orc.RLSPy(M, 4, 1, min_frames=32)
    o.calc_beamformer_weights(FS, delays)
    ref = orc.synthesis(g, M, m, r, 2, o.run(X))
    assert out_rls.shape == ref.shape and np.max(np.abs(out_rls - np.trunc(ref))) <= 1e-4 * np.max(np.abs(ref)) + 1.0
    out_l = _run(tmp_path, wavs, _conf({"type": "lcmv"}, {"type": "zelinski", "subtype": 2, "alpha": 0.7},
                                       noises=[{"positions": [[0.0, [0.9, None, None]]]}]), "lcmv")
    dj = calc_delays("linear", MPOS, [0.9, None, None])
    wq2 = orc.calc_mainlobe_2(M, 4, FS, delays, dj)
    Yf, _ = orc.zelinski_frames(X, orc.gsc_frames(X, wq2, np.zeros_like(wq2)), orc.calc_mainlobe(M, 4, FS, delays), 0.7, 2)
    ref = orc.synthesis(g, M, m, r, 2, Yf)
    assert out_l.shape == ref.shape and np.max(np.abs(out_l - np.trunc(ref))) <= 1.0


def test_tool_moving_look_direction(orc, dev, proto256, kinect_pcm, wavs, tmp_path):
    """target.positions with two time stamps: the weights are recomputed between two output blocks; frames the synthesis
    bank had already pulled (pd + block + 1, modulated.cc:574-578) keep the old weights, later frames use the new ones."""
    from distant_speech_recognition_amd.pybeamformer import calc_delays
    h, g = proto256
    X = np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    posA, posB = [-1.306379, None, None], [0.4, None, None]
    out = _run(tmp_path, wavs, _conf({"type": "delay_and_sum"}, positions=[[0.5, posA], [99.0, posB]]), "move")
    wa = orc.calc_mainlobe(M, 4, FS, calc_delays("linear", MPOS, posA))
    wb = orc.calc_mainlobe(M, 4, FS, calc_delays("linear", MPOS, posB))
    Ya, Yb = orc.gsc_frames(X, wa, np.zeros_like(wa)), orc.gsc_frames(X, wb, np.zeros_like(wb))
    b_switch = int(np.floor(0.5 / (D / FS)))                 # elapsed = (b+1) D/fs first exceeds 0.5 after block b
    pd_syn = 4                                               # synthesis processing delay, type 2: m R / 2
    n_old = pd_syn + b_switch + 1
    Y = np.concatenate([Ya[:n_old], Yb[n_old:]])
    ref = orc.synthesis(g, M, m, r, 2, Y)
    assert out.shape == ref.shape and np.max(np.abs(out - np.trunc(ref))) <= 1.0
    static = orc.synthesis(g, M, m, r, 2, Ya)
    assert np.max(np.abs(static - ref)) > 5.0                # the switch is visible in the expectation itself
    # with a post-filter in the chain the switch must go through as well (CSD history restarts, beamformer.cc:1082-1092)
    out_z = _run(tmp_path, wavs, _conf({"type": "delay_and_sum"}, {"type": "zelinski", "subtype": 2, "alpha": 0.7},
                                       positions=[[0.5, posA], [99.0, posB]]), "movez")
    out_s = _run(tmp_path, wavs, _conf({"type": "delay_and_sum"}, {"type": "zelinski", "subtype": 2, "alpha": 0.7}), "statz")
    nb = (b_switch + 1) * D
    assert out_z.shape == out_s.shape and np.array_equal(out_z[:nb], out_s[:nb]) and np.max(np.abs(out_z[nb:] - out_s[nb:])) > 5.0


def test_tool_sos_batch_and_dereverberator(orc, dev, proto256, kinect_pcm, wavs, tmp_path):
    """tools/sos_batch_beamforming.py (confs/gev_vad.json, bmvdr_tfmask.json shapes) and tools/subband_dereverberator.py
    (confs/wpe.json shape): outputs against the oracle graph."""
    import scipy.linalg  # noqa: F401
    h, g = proto256
    X = np.stack([orc.analysis(h, M, m, r, 2, kinect_pcm[c][:L]) for c in range(4)], axis=1)
    T = X.shape[0]
    en = np.array([orc.frame_energy(X[t, 0]) for t in range(T)])
    gate = (en > 10).astype(np.float64)
    # ---- GEV from a VAD label
    conf = {"target": {"vad_label": [[0.4, 1.1]]}, "beamformer": {"type": "gev", "energy_threshold": 10}}
    cpath, opath = str(tmp_path / "gev.json"), str(tmp_path / "gev.wav")
    json.dump(conf, open(cpath, "w"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sos_batch_beamforming.py"), "-q", "-c", cpath, "-o", opath,
                          "-i"] + wavs, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    w = wave.open(opath, "rb"); out = np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(np.float64); w.close()
    from distant_speech_recognition_amd.pybeamformer import _vad_noise_label
    tgt = _vad_noise_label(T, D / FS, [(0.4, 1.1)]).astype(np.float64)
    Rt, Rj = orc.cov_accumulate(X, frame_weights=tgt * gate), orc.cov_accumulate(X, frame_weights=(1 - tgt) * gate)
    ft, fj = orc.sos_finalize(Rt, Rj, np.full(129, (tgt * gate).sum()), np.full(129, ((1 - tgt) * gate).sum()), 1e-6, gev=True)
    wg = orc.gev_weights(ft, fj)
    ref = orc.synthesis(g, M, m, r, 2, orc.sos_frames(X, wg))
    err = min(np.max(np.abs(out - np.trunc(ref))), np.max(np.abs(out - np.trunc(-ref))))      # global sign of the eigenvector
    assert out.shape == ref.shape and err <= 2e-3 * np.max(np.abs(ref)) + 1.0
    # ---- blind MVDR from TF masks (.npy files)
    rng = np.random.default_rng(3)
    mt = (rng.random((T, 129)) > 0.5).astype(np.float64)
    mj = (rng.random((T, 129)) > 0.5).astype(np.float64)
    np.save(str(tmp_path / "mt.npy"), mt); np.save(str(tmp_path / "mj.npy"), mj)
    conf = {"target": {"tfmask_path": str(tmp_path / "mt.npy")}, "noises": [{"tfmask_path": str(tmp_path / "mj.npy")}],
            "beamformer": {"type": "bmvdr", "energy_threshold": 10, "ref_micx": 2, "offset": 0.0}}
    cpath, opath = str(tmp_path / "bm.json"), str(tmp_path / "bm.wav")
    json.dump(conf, open(cpath, "w"))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sos_batch_beamforming.py"), "-q", "-c", cpath, "-o", opath,
                          "-i"] + wavs, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    w = wave.open(opath, "rb"); out = np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(np.float64); w.close()
    Rt, Rj = orc.cov_accumulate(X, masks=mt * gate[:, None]), orc.cov_accumulate(X, masks=mj * gate[:, None])
    ft, fj = orc.sos_finalize(Rt, Rj, (mt * gate[:, None]).sum(0), (mj * gate[:, None]).sum(0), 1e-6)
    ref = orc.synthesis(g, M, m, r, 2, orc.sos_frames(X, orc.blind_mvdr_weights(ft, fj, ref_micx=2)))
    assert out.shape == ref.shape and np.max(np.abs(out - np.trunc(ref))) <= 1e-3 * np.max(np.abs(ref)) + 1.0
    # ---- multi-channel WPE on two channels
    conf = {"lower_num": 0, "upper_num": 7, "iterations_num": 2, "load_db": -18.0, "band_width": 0.0, "diagonal_bias": 0.0001}
    cpath = str(tmp_path / "wpe.json")
    json.dump(conf, open(cpath, "w"))
    outs = [str(tmp_path / "d0.wav"), str(tmp_path / "d1.wav")]
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "subband_dereverberator.py"), "-q", "-c", cpath, "-i"] + wavs[:2]
                         + ["-o"] + outs, capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    G = orc.wpe_estimate(X[:, :2], 0, 7, 2, -18.0, 0.0, 1e-4)
    Yd = orc.wpe_apply(X[:, :2], G, 0, 7)
    for c in range(2):
        w = wave.open(outs[c], "rb"); o = np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(np.float64); w.close()
        ref = orc.synthesis(g, M, m, r, 2, Yd[:, c])
        assert o.shape == ref.shape and np.max(np.abs(o - np.trunc(ref))) <= 1e-3 * np.max(np.abs(ref)) + 1.0
