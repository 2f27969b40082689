"""Golden vectors for the batch second-order-statistics beamformers from the REFERENCE's own Python arithmetic
(dev container only; same mechanism as gen_golden_pybeamformer.py -- nothing of the reference is written here).

Pinned (lib/pybeamformer.py): SubbandSOSBatchBeamformer.accu_stats_from_tfmask / accu_stats_from_label (:1043-1164),
SubbandBlindMVDRBeamformer.finalize_stats / calc_beamformer_weights (:1210-1263),
SubbandGEVBeamformer.finalize_stats / calc_beamformer_weights (:1266-1328, scipy.linalg.eigh),
and the beamformed frames of SubbandSOSBatchBeamformer.__iter__ (:1171-1186).

Run:  python tests/golden/gen_golden_pybeamformer_sos.py  -> tests/golden/pybeamformer_sos_golden.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden_pybeamformer import load_reference_module, NumpySnapshotSource, ROOT   # noqa: E402


def make(cls, X, M):
    bf = cls.__new__(cls)
    bf._array_source = NumpySnapshotSource(X)
    bf._chan_num, bf._fftlen, bf._fftlen2, bf._shiftlen = X.shape[1], M, M // 2, M // 2
    bf._isamp = 0
    bf._wqH = np.ones((M // 2 + 1, X.shape[1]), complex)
    bf.reset_stats()
    return bf


def main():
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    ref = load_reference_module()
    proto = np.load(os.path.join(HERE, "prototype_M256_m4_r1.npz"))
    pcm = np.load(os.path.join(HERE, "kinect_4ch_16k.npz"))["pcm"].astype(np.float32)
    M, fs, T = 256, 16000, 192
    X = np.stack([orc.analysis(proto["h"], M, 4, 1, 2, pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)
    rng = np.random.default_rng(11)
    # integer masks: the reference's per-bin counters are integer arrays (:1127-1128), so `count[m] += mask` truncates
    # fractional mask values while the covariance is still weighted by them; 0/1 (and an occasional 2) is the usable domain
    mask_t = ((rng.random((T, M // 2 + 1)) > 0.55) * (1 + (rng.random((T, M // 2 + 1)) > 0.9))).astype(np.int64)
    mask_j = ((rng.random((T, M // 2 + 1)) > 0.45) * (1 + (rng.random((T, M // 2 + 1)) > 0.9))).astype(np.int64)
    out = {"meta_T": np.array([T]), "mask_t": mask_t.astype(np.float32), "mask_j": mask_j.astype(np.float32)}

    # ---- blind MVDR from TF masks (confs/bmvdr_tfmask.json flow)
    bm = make(ref.SubbandBlindMVDRBeamformer, X, M)
    bm.accu_stats_from_tfmask(fs, mask_t, mask_j, energy_threshold=10)
    out["bm_cov_t_raw"], out["bm_cov_j_raw"] = bm._target_covariance_matrices.copy(), bm._noise_covariance_matrices.copy()
    out["bm_cnt_t"], out["bm_cnt_j"] = bm._target_frame_counts.copy(), bm._noise_frame_counts.copy()
    bm.finalize_stats(gamma=1e-6)
    out["bm_cov_t"], out["bm_cov_j"] = bm._target_covariance_matrices.copy(), bm._noise_covariance_matrices.copy()
    bm.calc_beamformer_weights(ref_micx=1, offset=0.0)
    out["bm_wqH"] = bm._wqH.copy()
    bm.reset()
    it = iter(bm)
    out["bm_Y"] = np.stack([np.array(next(it)) for _ in range(T)])[:, ::7].copy()

    # ---- GEV from a VAD label (confs/gev_vad.json flow)
    gv = make(ref.SubbandGEVBeamformer, X, M)
    gv.accu_stats_from_label(fs, target_labs=[(0.4, 1.1)], energy_threshold=10)
    out["gev_cnt_t"], out["gev_cnt_j"] = gv._target_frame_counts.copy(), gv._noise_frame_counts.copy()
    gv.finalize_stats(gamma=1e-6)
    out["gev_cov_t"], out["gev_cov_j"] = gv._target_covariance_matrices.copy(), gv._noise_covariance_matrices.copy()
    gv.calc_beamformer_weights()
    out["gev_wqH"] = gv._wqH.copy()
    gv.reset()
    it = iter(gv)
    out["gev_Y"] = np.stack([np.array(next(it)) for _ in range(T)])[:, ::7].copy()
    np.savez_compressed(os.path.join(HERE, "pybeamformer_sos_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})
    print("min counts", out["bm_cnt_t"].min(), out["bm_cnt_j"].min(), out["gev_cnt_t"][:3], out["gev_cnt_j"][:3])


if __name__ == "__main__":
    main()
