"""Golden vectors for the adaptive cancellers with Nc > 1 constraints, from the REFERENCE's own Python arithmetic (dev
container only; same in-memory translation as gen_golden_pybeamformer.py, nothing of the reference is written here).

Pinned: SubbandGSCLMSBeamformer.__iter__ with Nc = 2 and Nc = 3 (lib/pybeamformer.py:588-762: the blocking matrix keeps
the first N - Nc Gram-Schmidt columns, :309-341) -> tests/golden/pybeamformer_nc_golden.npz.

Run:  python tests/golden/gen_golden_pybeamformer_nc.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gen_golden_pybeamformer import NumpySnapshotSource, load_reference_module  # noqa: E402


def main():
    from oracle import oracle as orc
    ref = load_reference_module()
    proto = np.load(os.path.join(HERE, "prototype_M256_m4_r1.npz"))
    pcm = np.load(os.path.join(HERE, "kinect_4ch_16k.npz"))["pcm"].astype(np.float32)
    M, m, r, fs = 256, 4, 1, 16000
    T = 160
    X = np.stack([orc.analysis(proto["h"], M, m, r, 2, pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)
    mpos = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
    delays = ref.calc_la_delays(mpos, -1.306379)
    out = {"delays": delays}
    for Nc in (2, 3):
        cls = ref.SubbandGSCLMSBeamformer
        bf = cls.__new__(cls)
        bf._array_source = NumpySnapshotSource(X)
        bf._chan_num, bf._fftlen, bf._fftlen2, bf._shiftlen, bf._Nc = 4, M, M // 2, 128, Nc
        bf._wqH = np.ones((M // 2 + 1, 4), complex)
        bf._BmH = [np.zeros((4 - Nc, 4), complex) for _ in range(M // 2 + 1)]
        bf._beta, bf._init_gamma, bf._init_diagonal_load = 0.97, 0.05, 1.0e6
        bf._regularization_param, bf._energy_floor, bf._sil_thresh = 1.0e-4, 90, 1.0e8
        bf._max_wa_l2norm, bf._min_frames, bf._slowdown_after = 100.0, 16, 64
        bf._isamp = 0
        bf.reset_stats()
        bf._subband_no_printed = set([])
        bf.calc_beamformer_weights(fs, delays)
        it = iter(bf)
        Y = np.stack([np.array(next(it)) for _ in range(T)])
        tag = "nlms_nc%d" % Nc
        out[tag + "_Y"] = Y[:, : M // 2 + 1][:, ::5].copy()
        out[tag + "_waH"] = np.array(bf._waH)
        out[tag + "_subband_energy"] = np.array(bf._subband_energy)
        out[tag + "_blockmat_k40"] = np.array(bf._BmH[40]).T.copy()
    out["meta_T"] = np.array([T])
    np.savez_compressed(os.path.join(HERE, "pybeamformer_nc_golden.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
