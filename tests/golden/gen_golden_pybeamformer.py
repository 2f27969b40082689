"""Golden vectors from the REFERENCE's own Python arithmetic (dev container only).

The reference's lib/pybeamformer.py is Python 2 and imports SWIG modules that cannot be built
here (GSL/SWIG absent).  Its numerical classes, however, are pure numpy.  This script

  1. reads /root/reference/btk20_src/lib/pybeamformer.py, translates it to Python 3 IN MEMORY
     with lib2to3 (print statements etc.), drops the five `from btk20.<swig module> import *`
     lines, and execs the result -- nothing of it is written to this repository;
  2. drives the reference classes (bypassing only their SWIG-dependent __init__) with a numpy
     snapshot source that serves analysis frames computed from the committed PCM fixture;
  3. stores inputs' provenance and the reference's OUTPUTS as tests/golden/pybeamformer_golden.npz.

Pinned: calc_la_delays, calc_array_manifold_f, calc_blocking_matrix, improve_matrix_condition,
SubbandGSCLMSBeamformer.__iter__ (NLMS), SubbandSMIMVDRBeamformer.accu_stats_from_label,
SubbandSOSBatchBeamformer.accu_stats_from_tfmask.

Run:  python tests/golden/gen_golden_pybeamformer.py
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF_PY = "/root/reference/btk20_src/lib/pybeamformer.py"


def load_reference_module():
    from lib2to3 import refactor
    rt = refactor.RefactoringTool(refactor.get_fixers_from_package("lib2to3.fixes"))
    src = open(REF_PY).read()
    if not src.endswith("\n"):
        src += "\n"
    py3 = str(rt.refactor_string(src, "pybeamformer"))
    py3 = "\n".join(l for l in py3.splitlines()
                    if not (l.startswith("from btk20.") and l.rstrip().endswith("import *")))
    # numpy aliases removed in numpy>=1.24 that the Python-2 era source uses
    for name, typ in (("float", float), ("complex", complex), ("int", int)):
        if not hasattr(np, name):
            setattr(np, name, typ)
    mod = types.ModuleType("ref_pybeamformer")
    exec(compile(py3, REF_PY, "exec"), mod.__dict__)
    return mod


class NumpySnapshotSource:
    """Stands where MultiChannelSource (pybeamformer.py:241-281) stands: serves frames X[t][chan][bin]."""

    def __init__(self, X):
        self.X = X
        self.t = -1

    def update_snapshot_array(self, chan_no=None):
        self.t += 1
        if self.t >= self.X.shape[0]:
            raise StopIteration
        if chan_no is None:
            return 0.0
        s = self.X[self.t, chan_no]
        return abs(np.dot(np.conjugate(s), s))

    def get_snapshot(self, m):
        return self.X[self.t, :, m]

    def reset(self):
        self.t = -1


def main():
    from oracle import oracle as orc
    ref = load_reference_module()
    proto = np.load(os.path.join(HERE, "prototype_M256_m4_r1.npz"))
    pcm = np.load(os.path.join(HERE, "kinect_4ch_16k.npz"))["pcm"].astype(np.float32)
    M, m, r, fs = 256, 4, 1, 16000
    T = 192
    # analysis frames from the oracle filter bank (input side; deterministic, regenerated in tests)
    X = np.stack([orc.analysis(proto["h"], M, m, r, 2, pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)
    out = {}

    # --- delays / manifold / blocking matrix (confs/ds.json geometry) ---
    mpos = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
    az = -1.306379
    delays = ref.calc_la_delays(mpos, az)
    out["delays_kinect"] = delays
    out["manifold_k5"] = ref.calc_array_manifold_f(5, M, fs, delays, False)
    out["manifold_k128"] = ref.calc_array_manifold_f(128, M, fs, delays, False)
    out["blockmat_k5_nc1"] = ref.calc_blocking_matrix(ref.calc_array_manifold_f(5, M, fs, delays, False), 1)
    out["blockmat_k77_nc2"] = ref.calc_blocking_matrix(ref.calc_array_manifold_f(77, M, fs, delays, False), 2)
    d8 = ref.calc_la_delays(np.array([[20.0 * (i - 3.5), 0, 0] for i in range(8)]), 0.7)
    out["delays_ula8"] = d8
    out["blockmat_ula8_k33"] = ref.calc_blocking_matrix(ref.calc_array_manifold_f(33, 512, fs, d8, False), 1)

    # --- NLMS (SubbandGSCLMSBeamformer), defaults = confs/gsclms.json, except min_frames ---
    for tag, kw in (("nlms_default", dict()), ("nlms_fast", dict(min_frames=16, gamma=0.05, slowdown_after=64))):
        cls = ref.SubbandGSCLMSBeamformer
        bf = cls.__new__(cls)
        bf._array_source = NumpySnapshotSource(X)
        bf._chan_num, bf._fftlen, bf._fftlen2, bf._shiftlen, bf._Nc = 4, M, M // 2, 128, 1
        bf._wqH = np.ones((M // 2 + 1, 4), complex)
        bf._BmH = [np.zeros((3, 4), complex) for _ in range(M // 2 + 1)]
        p = dict(beta=0.97, gamma=0.01, init_diagonal_load=1.0e6, regularization_param=1.0e-4, energy_floor=90,
                 sil_thresh=1.0e8, max_wa_l2norm=100.0, min_frames=128, slowdown_after=4096)
        p.update(kw)
        bf._beta, bf._init_gamma, bf._init_diagonal_load = p["beta"], p["gamma"], p["init_diagonal_load"]
        bf._regularization_param, bf._energy_floor, bf._sil_thresh = p["regularization_param"], p["energy_floor"], p["sil_thresh"]
        bf._max_wa_l2norm, bf._min_frames, bf._slowdown_after = p["max_wa_l2norm"], p["min_frames"], p["slowdown_after"]
        bf._isamp = 0
        bf.reset_stats()
        bf._subband_no_printed = set([])
        bf.calc_beamformer_weights(fs, delays)
        it = iter(bf)
        Y = np.stack([np.array(next(it)) for _ in range(T)])
        out[tag + "_Y"] = Y[:, : M // 2 + 1][:, ::5].copy()     # every 5th bin, all frames
        out[tag + "_Ymirror"] = Y[T - 1].copy()                  # one full frame incl. mirror bins
        out[tag + "_waH"] = np.array(bf._waH)
        out[tag + "_subband_energy"] = np.array(bf._subband_energy)
        out[tag + "_energy"] = np.array([bf._energy, bf._gamma, bf._ttl_updates])

    # --- covariance accumulation from a VAD label ---
    cls = ref.SubbandSMIMVDRBeamformer
    sm = cls.__new__(cls)
    sm._array_source = NumpySnapshotSource(X)
    sm._chan_num, sm._fftlen, sm._fftlen2, sm._shiftlen = 4, M, M // 2, 128
    sm._noise_covariance_matrices, sm._noise_frame_num = None, 0
    sm.accu_stats_from_label(fs, target_labs=[(0.5, 1.0)], energy_threshold=10)
    out["smi_noise_frames"] = np.array([sm._noise_frame_num])
    out["smi_cov_raw"] = sm._noise_covariance_matrices.copy()
    sm.finalize_stats()
    out["smi_cov_final"] = sm._noise_covariance_matrices.copy()

    # --- covariance accumulation from TF masks ---
    rng = np.random.default_rng(7)
    mask_t = (rng.random((T, M // 2 + 1)) > 0.6).astype(np.int64)
    mask_j = (rng.random((T, M // 2 + 1)) > 0.5).astype(np.int64)
    cls = ref.SubbandSOSBatchBeamformer
    so = cls.__new__(cls)
    so._array_source = NumpySnapshotSource(X)
    so._chan_num, so._fftlen, so._fftlen2, so._shiftlen = 4, M, M // 2, 128
    so.reset_stats()
    so.accu_stats_from_tfmask(fs, mask_t, mask_j, energy_threshold=10)
    out["tfmask_t"], out["tfmask_j"] = mask_t.astype(np.int8), mask_j.astype(np.int8)
    out["tf_cov_t"], out["tf_cov_j"] = so._target_covariance_matrices, so._noise_covariance_matrices
    out["tf_cnt_t"], out["tf_cnt_j"] = so._target_frame_counts, so._noise_frame_counts
    out["imc"] = ref.improve_matrix_condition(so._noise_covariance_matrices[9] / 50.0, 1e-3)

    out["meta_T"] = np.array([T])
    np.savez_compressed(os.path.join(HERE, "pybeamformer_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
