"""Golden vectors for the RLS sidelobe canceller from the REFERENCE's own Python arithmetic (dev container only).

Same mechanism as gen_golden_pybeamformer.py (in-memory lib2to3 translation of the reference's
lib/pybeamformer.py, numpy snapshot source on the committed Kinect PCM fixture); nothing of the reference
is written to this repository, only its OUTPUTS -> tests/golden/pybeamformer_rls_golden.npz.

Pinned: SubbandGSCRLSBeamformer.__iter__ / calc_beamformer_weights / reset_stats (lib/pybeamformer.py:765-928)
with its default hyper-parameters (= confs/gscrls.json) and with a configuration that exercises the
quadratic constraint and the norm reset.

Round 3: the same recursion with Nc = 2 constraints (SubbandGSCRLSBeamformer(..., Nc), :784-797: the blocking matrix keeps
N - Nc columns) -> tests/golden/pybeamformer_rls_nc_golden.npz.

Run:  python tests/golden/gen_golden_pybeamformer_rls.py        (writes both files)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gen_golden_pybeamformer import load_reference_module, NumpySnapshotSource, ROOT   # noqa: E402

CASES = (
    ("rls_default", dict()),
    ("rls_constrained", dict(min_frames=8, gamma=0.5, alpha2=1.0e-3, max_wa_l2norm=4.0e-3, init_diagonal_load=1.0e3,
                             sil_thresh=1.0e2)),
    ("rls_quadonly", dict(min_frames=0, constraint_option=1, alpha2=1.0e-4, regularization_param=0.0)),
)


NC_CASES = (
    ("rlsnc2_default", dict(min_frames=16)),
    ("rlsnc2_constrained", dict(min_frames=8, gamma=0.5, alpha2=1.0e-3, max_wa_l2norm=4.0e-3, init_diagonal_load=1.0e3,
                                sil_thresh=1.0e2)),
)


def main(cases=CASES, Nc=1, outfile="pybeamformer_rls_golden.npz"):
    sys.path.insert(0, ROOT)
    from oracle import oracle as orc
    ref = load_reference_module()
    proto = np.load(os.path.join(HERE, "prototype_M256_m4_r1.npz"))
    pcm = np.load(os.path.join(HERE, "kinect_4ch_16k.npz"))["pcm"].astype(np.float32)
    M, fs, T = 256, 16000, 160
    X = np.stack([orc.analysis(proto["h"], M, 4, 1, 2, pcm[c][: (T + 8) * 128])[:T] for c in range(4)], axis=1)
    mpos = np.array([[-113.0, 0.0, 2.0], [36.0, 0.0, 2.0], [76.0, 0.0, 2.0], [113.0, 0.0, 2.0]])
    delays = ref.calc_la_delays(mpos, -1.306379)
    out = {"meta_T": np.array([T])}
    for tag, kw in cases:
        cls = ref.SubbandGSCRLSBeamformer
        bf = cls.__new__(cls)
        bf._array_source = NumpySnapshotSource(X)
        bf._chan_num, bf._fftlen, bf._fftlen2, bf._shiftlen, bf._Nc = 4, M, M // 2, 128, Nc
        bf._wqH = np.ones((M // 2 + 1, 4), complex)
        bf._BmH = [np.zeros((4 - Nc, 4), complex) for _ in range(M // 2 + 1)]
        p = dict(beta=0.97, gamma=0.04, mu=0.97, init_diagonal_load=1.0e6, regularization_param=1.0e-2, sil_thresh=1.0e8,
                 constraint_option=3, alpha2=10.0, max_wa_l2norm=100.0, min_frames=128, slowdown_after=4096)
        p.update(kw)
        bf._beta, bf._gamma, bf._mu = p["beta"], p["gamma"], p["mu"]
        bf._init_diagonal_load, bf._regularization_param = p["init_diagonal_load"], p["regularization_param"]
        bf._sil_thresh, bf._constraint_option, bf._alpha2 = p["sil_thresh"], p["constraint_option"], p["alpha2"]
        bf._max_wa_l2norm, bf._min_frames, bf._slowdown_after = p["max_wa_l2norm"], p["min_frames"], p["slowdown_after"]
        bf._isamp = 0
        bf.reset_stats()
        bf._subband_no_printed = set([])
        bf.calc_beamformer_weights(fs, delays)
        it = iter(bf)
        Y = np.stack([np.array(next(it)) for _ in range(T)])
        out[tag + "_Y"] = Y[:, : M // 2 + 1][:, ::5].copy()
        out[tag + "_Ymirror"] = Y[T - 1].copy()
        out[tag + "_waH"] = np.array(bf._waH)
        out[tag + "_Pz"] = np.array(bf._Pz)[::8].copy()
        out[tag + "_scal"] = np.array([bf._energy, bf._isamp, bf._ttl_updates], float)
        out[tag + "_params"] = np.array([p[k] for k in ("beta", "gamma", "mu", "init_diagonal_load", "regularization_param",
                                                         "sil_thresh", "constraint_option", "alpha2", "max_wa_l2norm",
                                                         "min_frames")], float)
        nrm = np.abs(np.sum(np.array(bf._waH) * np.conj(np.array(bf._waH)), axis=1))
        print(tag, "ttl_updates", bf._ttl_updates, "max|wa|^2", nrm.max(), "Pz resets",
              int(np.sum(np.abs(np.array(bf._Pz)[:, 0, 0] - 1.0 / p["init_diagonal_load"]) < 1e-30)))
    out["meta_Nc"] = np.array([Nc])
    np.savez_compressed(os.path.join(HERE, outfile), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
    main(NC_CASES, 2, "pybeamformer_rls_nc_golden.npz")
